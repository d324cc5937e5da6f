#!/bin/bash
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_llama.py tests/test_hip_conformance.py tests/test_hip_kvq.py tests/test_hip_gguf_passthrough.py -x -q 2>&1 | tail -4
{
for r in 1 2 3; do
ZGML_HIP_HOST_PROF=1 python tools/decode_vtable_run.py smollm-135m 512 dyn
ZGML_HIP_HOST_PROF=1 ZGML_HIP_IO_GRAPH=0 python tools/decode_vtable_run.py smollm-135m 512 dyn
done
python tools/decode_run.py smollm-135m 512
python tools/decode_vtable_run.py llama2-7b 64 dyn
ZGML_HIP_IO_GRAPH=0 python tools/decode_vtable_run.py llama2-7b 64 dyn
} 2>&1 | tee $O/io_graph_ab.txt
