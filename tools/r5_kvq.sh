#!/bin/bash
# int8-KV decode attention, 16 dims per lane against the 4-dims-per-lane form (libzgml_hip_kvq4.so = the same tree built with
# -DZGML_ATTN_KVQ16=0, tools/build_variant.py) and against f32 KV, same box, alternating
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
{
for rep in 1 2; do
  for m in llama2-7b smollm-135m; do
    python3 tools/decode_run.py $m 64 1900 0
    python3 tools/decode_run.py $m 64 1900 32
    ZGML_HIP_LIB=zgml_amd/lib/libzgml_hip_kvq4.so python3 tools/decode_run.py $m 64 1900 32
    python3 tools/decode_run.py $m 128 4 32
    ZGML_HIP_LIB=zgml_amd/lib/libzgml_hip_kvq4.so python3 tools/decode_run.py $m 128 4 32
  done
done
} 2>&1 | tee $O/kvq16_ab.txt
