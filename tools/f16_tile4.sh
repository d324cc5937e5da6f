#!/bin/bash
# the A-stationary f16 M = 32 kernel (tile4, default) against the per-column-group form (ZGML_F16_TILE4=0), and its slice count
for t4 in 0 1; do echo "== ZGML_F16_TILE4=$t4"; ZGML_F16_TILE4=$t4 python3 tools/bench_matvec.py --fmts f16 --M 32 --shapes 4096x4096,4096x12288,4096x22016,11008x4096,4096x32000 --iters 640 | grep -v copy | cut -c1-120; done
for sk in 2 4 6 16; do echo "== tile4 SK=$sk"; ZGML_F16_TILE4_SK=$sk python3 tools/bench_matvec.py --fmts f16 --M 32 --shapes 4096x4096,4096x22016,11008x4096 --iters 640 | grep -v copy | cut -c1-120; done
