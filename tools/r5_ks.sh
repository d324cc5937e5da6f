#!/bin/bash
# round 5: K-split launches — parity tests, resident decode throughput with the pass on / off, in-kernel stamps (diagnostics build)
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_ksplit.py -x -q > $O/t_ksplit.txt 2>&1; echo "ksplit rc=$?"; tail -5 $O/t_ksplit.txt
for i in 1 2; do
ZGML_HIP_KSPLIT=1 python tools/decode_run.py smollm-135m 256 2>&1 | tail -1
ZGML_HIP_KSPLIT=0 python tools/decode_run.py smollm-135m 256 2>&1 | tail -1
done | tee $O/dec_ab.txt
ZGML_HIP_LIB=$PWD/zgml_amd/lib/libzgml_hip_trace.so ZGML_HIP_KSPLIT=1 ZGML_HIP_KS_TRACE=1 python tools/decode_run.py smollm-135m 32 > $O/ks_stamps.txt 2>&1; tail -30 $O/ks_stamps.txt
