#!/bin/bash
# PMC passes over the M > 1 tile kernel (run through gpurun from the repo root): wave cycles, instruction
# mix, wait cycles, MFMA busy. One counter group per pass (separate runs, as the guide prescribes).
set -e -o pipefail
OUT=gpurun_out/pmc_tile
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/bench_matvec.py --fmts ${FMT:-q4} --shapes ${SHAPE:-4096x4096} --M ${M:-32} --iters 64 > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1); test -n "$f" && cp "$f" $OUT/p${i}_counters.csv
done
ls $OUT/*.csv
