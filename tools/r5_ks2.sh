#!/bin/bash
O=gpurun_out/r5; mkdir -p $O
for v in "ZGML_HIP_KSPLIT_FUSE_A=0" "ZGML_HIP_HANDOFF_SLEEP=2" "ZGML_HIP_HANDOFF_SLEEP=20" "ZGML_HIP_HANDOFF_SLEEP=100" "ZGML_HIP_ATTN_SPLIT=1"; do
  echo "== $v"; env $v python tools/decode_run.py smollm-135m 256 2>&1 | tail -1
done | tee $O/fuse_a_sweep.txt
