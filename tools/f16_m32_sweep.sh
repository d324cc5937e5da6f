#!/bin/bash
# dense f16 M = 32 (BASELINE configs[4]) per shape: the shared-A K-split form (tile3, default) against the per-column-group form (ZGML_F16_TILE3=0)
for t3 in 0 1; do echo "== ZGML_F16_TILE3=$t3"; ZGML_F16_TILE3=$t3 python3 tools/bench_matvec.py --fmts f16 --M 32 --shapes 4096x4096,4096x12288,4096x22016,11008x4096,4096x32000 --iters 640; done
for sk in 1 2 4 8 16; do echo "== tile3 SK=$sk"; ZGML_F16_TILE3_SK=$sk python3 tools/bench_matvec.py --fmts f16 --M 32 --shapes 4096x4096,4096x22016,11008x4096 --iters 640; done
