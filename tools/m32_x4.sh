#!/bin/bash
# narrow outputs at M = 32: form 3 (all of K per workgroup) against the K-split kernel of the M > 32 form at two m-tiles
set -e -o pipefail
OUT=gpurun_out/r3
mkdir -p $OUT
: > $OUT/m32_x4.log
SH=${SHAPES:-4096x4096,11008x4096,4096x8192,2048x2048}
run() { echo "== $*" >> $OUT/m32_x4.log; env "$@" python tools/bench_matvec.py --fmts q4 --shapes $SH --M ${MS:-32,16} --iters 640 2>&1 | grep '"K"' | cut -c1-125 >> $OUT/m32_x4.log; }
run ZGML_QMM_XDL4_M32=0
run ZGML_QMM_XDL4_M32=1
run ZGML_QMM_XDL4_M32=1 ZGML_QMM_XDL4_SK=4
run ZGML_QMM_XDL4_M32=1 ZGML_QMM_XDL4_SK=1
cat $OUT/m32_x4.log
