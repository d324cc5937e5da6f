#!/bin/bash
# A/B of the M <= 32 quantized tile kernels (run through gpurun from the repo root): the shared-A form against the per-workgroup-A form
set -e -o pipefail
OUT=gpurun_out/r3
mkdir -p $OUT
SHAPES=${SHAPES:-4096x4096,4096x12288,4096x11008,4096x22016,11008x4096,4096x32000,576x576,576x1536}
for v in 1 0; do
  echo "== ZGML_QMM_XDL5=$v" >> $OUT/m32_ab.log
  ZGML_QMM_XDL5=$v python tools/bench_matvec.py --fmts q4 --shapes $SHAPES --M ${MS:-32,16} --iters 640 >> $OUT/m32_ab.log 2>&1
done
