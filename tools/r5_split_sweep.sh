#!/bin/bash
O=gpurun_out/r5; mkdir -p $O
{
for mk in 128 64 32; do
  echo "== ZGML_HIP_ATTN_SPLIT_MIN_KEYS=$mk"
  ZGML_HIP_ATTN_SPLIT_MIN_KEYS=$mk python tools/decode_run.py llama2-7b 128 4
  ZGML_HIP_ATTN_SPLIT_MIN_KEYS=$mk python tools/decode_run.py llama2-7b 128 200
  ZGML_HIP_ATTN_SPLIT_MIN_KEYS=$mk python tools/decode_run.py llama2-7b 64 1900
  ZGML_HIP_ATTN_SPLIT_MIN_KEYS=$mk python tools/decode_run.py llama2-7b 128 4 32
  ZGML_HIP_ATTN_SPLIT_MIN_KEYS=$mk python tools/decode_run.py llama2-7b 64 1900 32
done
} 2>&1 | tee $O/split_sweep_7b.txt
