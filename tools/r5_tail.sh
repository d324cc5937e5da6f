#!/bin/bash
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_hip_tail.py -x -q -m gpu 2>&1 | tail -2
{
for r in 1 2 3; do
ZGML_HIP_TAIL_FUSED=1 python tools/decode_run.py smollm-135m 512
ZGML_HIP_TAIL_FUSED=0 python tools/decode_run.py smollm-135m 512
done
for r in 1 2; do
ZGML_HIP_TAIL_FUSED=1 python tools/decode_run.py llama2-7b 128
ZGML_HIP_TAIL_FUSED=0 python tools/decode_run.py llama2-7b 128
done
} 2>&1 | tee $O/tail_ab2.txt
