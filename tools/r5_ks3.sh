#!/bin/bash
O=gpurun_out/r5; mkdir -p $O
for v in "ZGML_KS_PROJ_WAVES=9" "ZGML_KS_PROJ_WAVES=12" "ZGML_KS_PROJ_WAVES=16" "ZGML_KS_MLP_WAVES=12" "ZGML_KS_MLP_WAVES=16" "ZGML_KS_PROJ_GP=1" "ZGML_KS_PROJ_GP=4"; do
  echo "== $v"; env ZGML_HIP_KSPLIT_FUSE_A=0 $v python tools/decode_run.py smollm-135m 256 2>&1 | tail -1
done | tee $O/waves_sweep.txt
ZGML_HIP_KSPLIT_FUSE_A=0 ZGML_KS_PROJ_WAVES=12 ZGML_HIP_LIB=$PWD/zgml_amd/lib/libzgml_hip_trace.so ZGML_HIP_KS_TRACE=1 python tools/decode_run.py smollm-135m 32 2>&1 | tail -8
