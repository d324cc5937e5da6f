#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/r3
mkdir -p $OUT
: > $OUT/m32_dbg.log
SH=${SHAPES:-4096x4096,4096x12288,4096x22016,11008x4096}
run() { echo "== $*" >> $OUT/m32_dbg.log; env "$@" python tools/bench_matvec.py --fmts q4 --shapes $SH --M 32 --iters 640 2>&1 | grep '"K"' | cut -c1-100 >> $OUT/m32_dbg.log; }
run ZGML_QMM_XDL5=1
run ZGML_QMM_XDL5_DEBUG=1
cat $OUT/m32_dbg.log
