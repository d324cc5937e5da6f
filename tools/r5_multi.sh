#!/bin/bash
O=gpurun_out/r5; mkdir -p $O
{
for r in 1 2; do
for g in 1 2 4 8; do
echo "tokens per graph $g"; ZGML_HIP_RESIDENT_TOKENS_PER_GRAPH=$g python tools/decode_run.py smollm-135m 512
done
done
for g in 1 4; do echo "7B tokens per graph $g"; ZGML_HIP_RESIDENT_TOKENS_PER_GRAPH=$g python tools/decode_run.py llama2-7b 128; done
} 2>&1 | tee $O/multi_token_graph.txt
