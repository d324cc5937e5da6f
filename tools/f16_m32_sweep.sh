#!/bin/bash
# dense f16 M = 32 (BASELINE configs[4]) per shape, per column groups per workgroup (CG) of dense_f16_tile2_kernel
for cg in 1 2 4 0; do echo "== CG $cg (0 = the launcher's own choice)"; ZGML_F16_TILE2_CG=$cg python3 tools/bench_matvec.py --fmts f16 --M 32 --shapes 4096x4096,4096x12288,4096x22016,11008x4096,4096x32000 --iters 640; done
