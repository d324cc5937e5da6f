#!/bin/bash
# PMC series over the M = 1 Q4_0 mat-vec ring (run through gpurun from the repo root): instruction counts, VALU
# activity, issue stalls, busy cycles, waves — one counter group per pass (separate runs, as the guide prescribes;
# the program sits directly after `--`). SHAPES: comma list, default the roofline shape and the gate/up-sized one.
set -e -o pipefail
TAG=${TAG:-r04}
OUT=gpurun_out/pmc_mv_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for SHAPE in ${SHAPES:-4096x4096 4096x11008}; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" "FETCH_SIZE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/${SHAPE}_p$i -- python3 tools/bench_matvec.py --fmts q4 --shapes $SHAPE --iters 128 > $OUT/${SHAPE}_p$i.log 2>&1
    f=$(find $OUT/${SHAPE}_p$i -name "*counter_collection.csv" | head -1); test -n "$f" && cp "$f" $OUT/${SHAPE}_p${i}_counters.csv
  done
done
ls $OUT/*.csv
