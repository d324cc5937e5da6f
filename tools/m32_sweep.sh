#!/bin/bash
# xdl5 run-length sweep (through gpurun): time of the shared-A kernel against the number of K steps per workgroup
set -e -o pipefail
OUT=gpurun_out/r3
mkdir -p $OUT
: > $OUT/m32_sweep.log
for mr in ${RUNS:-2 4 8 16 32}; do
  echo "== MIN_RUN=$mr" >> $OUT/m32_sweep.log
  ZGML_QMM_XDL5_MIN_RUN=$mr python tools/bench_matvec.py --fmts q4 --shapes ${SHAPES:-4096x4096,4096x22016} --M 32 --iters 640 2>&1 | grep '"K"' | cut -c1-120 >> $OUT/m32_sweep.log
done
echo "== WGS_PER_CU=2" >> $OUT/m32_sweep.log
ZGML_QMM_XDL5_WGS_PER_CU=2 python tools/bench_matvec.py --fmts q4 --shapes ${SHAPES:-4096x4096,4096x22016} --M 32 --iters 640 2>&1 | grep '"K"' | cut -c1-120 >> $OUT/m32_sweep.log
cat $OUT/m32_sweep.log
