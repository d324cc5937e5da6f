#!/bin/bash
# xdl5 variants (through gpurun): pipelined (1 workgroup per CU) against LEAN (2 per CU)
set -e -o pipefail
OUT=gpurun_out/r3
mkdir -p $OUT
: > $OUT/m32_lean.log
SH=${SHAPES:-4096x4096,4096x12288,4096x22016,11008x4096}
run() { echo "== $*" >> $OUT/m32_lean.log; env "$@" python tools/bench_matvec.py --fmts q4 --shapes $SH --M 32 --iters 640 2>&1 | grep '"K"' | cut -c1-100 >> $OUT/m32_lean.log; }
run ZGML_QMM_XDL5=0
run ZGML_QMM_XDL5_LEAN=0
run ZGML_QMM_XDL5_LEAN=1 ZGML_QMM_XDL5_WGS_PER_CU=2
run ZGML_QMM_XDL5_LEAN=1 ZGML_QMM_XDL5_WGS_PER_CU=1
run ZGML_QMM_XDL5_LEAN=1 ZGML_QMM_XDL5_WGS_PER_CU=2 ZGML_QMM_XDL5_MIN_RUN=4
run ZGML_QMM_XDL5_LEAN=1 ZGML_QMM_XDL5_WGS_PER_CU=2 ZGML_QMM_XDL5_MIN_RUN=32
cat $OUT/m32_lean.log
