#!/bin/bash
O=gpurun_out/r5; mkdir -p $O
for d in 1; do
echo "== ZGML_KS_DEBUG_A=$d"
ZGML_KS_DEBUG_A=$d ZGML_HIP_LIB=$PWD/zgml_amd/lib/libzgml_hip_trace.so ZGML_HIP_KS_TRACE=1 python tools/decode_run.py smollm-135m 32 2>&1 | tail -7
done | tee $O/fuse_a_dbg.txt
