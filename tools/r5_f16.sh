#!/bin/bash
O=gpurun_out/r5; mkdir -p $O
export ZGML_HIP_LIB=zgml_amd/lib/libzgml_hip_trace.so
{
echo "== tile2 (shipped)"; python3 tools/bench_matvec.py --fmts f16 --M 32 --shapes 4096x4096,4096x12288,4096x22016,11008x4096,4096x32000 --iters 640
for d in 4 8; do echo "== tile4 depth $d"; ZGML_F16_TILE4=1 ZGML_F16_TILE4_DEPTH=$d python3 tools/bench_matvec.py --fmts f16 --M 32 --shapes 4096x4096,4096x12288,4096x22016,11008x4096,4096x32000 --iters 640; done
} 2>&1 | cut -c1-150 | tee $O/f16_tile4_depth.txt
