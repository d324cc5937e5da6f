#!/bin/bash
# Memory-path PMC passes over the M > 1 tile kernel (run through gpurun from the repo root).
set -o pipefail
OUT=gpurun_out/pmc_tile_mem
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum" "TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout -k 10 70 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/bench_matvec.py --fmts ${FMT:-q4} --shapes ${SHAPE:-4096x4096} --M ${M:-32} --iters 64 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
  f=$(find $OUT/p$i -name "*counter_collection.csv" 2>/dev/null | head -1); test -n "$f" && cp "$f" $OUT/p${i}_counters.csv
done
ls $OUT/*.csv || true
