#!/bin/bash
# the shared-A K-split form (diagnostics library) per slice count against the shipped form
echo "== shipped"; python3 tools/bench_matvec.py --fmts f16 --M 32 --shapes 4096x4096,4096x12288,4096x22016,11008x4096,4096x32000 --iters 640 | grep -v copy | cut -c1-120
for sk in 0 1 2 3 4 8; do echo "== tile3 SK=$sk (0 = the launcher's choice)"; ZGML_HIP_LIB=zgml_amd/lib/libzgml_hip_trace.so ZGML_F16_TILE3=1 ZGML_F16_TILE3_SK=$sk python3 tools/bench_matvec.py --fmts f16 --M 32 --shapes 4096x4096,4096x12288,4096x22016,11008x4096,4096x32000 --iters 640 | grep -v copy | cut -c1-120; done
