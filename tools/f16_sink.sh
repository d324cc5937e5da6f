#!/bin/bash
# the dense f16 M = 32 tile kernel with its MFMAs replaced by a VALU sink (1) and with the A loads gone too (2): diagnostics library
for sk in 0 1 2; do for cg in 1 2; do echo "== SINK=$sk CG=$cg"; ZGML_HIP_LIB=zgml_amd/lib/libzgml_hip_trace.so ZGML_F16_SINK=$sk ZGML_F16_TILE2_CG=$cg python3 tools/bench_matvec.py --fmts f16 --M 32 --shapes 4096x4096,4096x22016,11008x4096 --iters 640 | grep -v copy | cut -c1-120; done; done
