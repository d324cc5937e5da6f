#!/bin/bash
# round 5 evidence for the K-split layer launches (zgml_amd/csrc/ksplit.hip): throughput of the three plans and the in-kernel stamps
# of workgroup 0 (diagnostics build) -> profiles/r05_ksplit_stamps.txt
O=gpurun_out/r5; mkdir -p $O
F=$O/r05_ksplit_stamps.txt
{
echo "SmolLM-135M Q4_0 resident greedy decode, positions 4..260, one MI355X (tools/decode_run.py; tools/r5_ks_evidence.sh)"
echo "plan A: four launches per layer (default)                         : $(ZGML_HIP_KSPLIT=0 python tools/decode_run.py smollm-135m 256 2>&1 | tail -1)"
echo "plan B: K-split, three launches (ks-proj | ks-attn-o | ks-mlp)    : $(ZGML_HIP_KSPLIT=1 ZGML_HIP_KSPLIT_FUSE_A=0 python tools/decode_run.py smollm-135m 256 2>&1 | tail -1)"
echo "plan C: K-split, two launches (ks-layer-a | ks-mlp)               : $(ZGML_HIP_KSPLIT=1 python tools/decode_run.py smollm-135m 256 2>&1 | tail -1)"
echo
echo "== plan B, in-kernel stamps (libzgml_hip_trace.so, ZGML_HIP_KS_TRACE=1)"
ZGML_HIP_LIB=$PWD/zgml_amd/lib/libzgml_hip_trace.so ZGML_HIP_KSPLIT=1 ZGML_HIP_KSPLIT_FUSE_A=0 ZGML_HIP_KS_TRACE=1 python tools/decode_run.py smollm-135m 32 2>&1 | grep -A19 "K-split launches"
echo
echo "== plan C, in-kernel stamps"
ZGML_HIP_LIB=$PWD/zgml_amd/lib/libzgml_hip_trace.so ZGML_HIP_KSPLIT=1 ZGML_HIP_KS_TRACE=1 python tools/decode_run.py smollm-135m 32 2>&1 | grep -A19 "K-split launches"
} > $F 2>&1
cat $F
