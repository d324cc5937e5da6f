#!/bin/bash
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_llama.py tests/test_hip_conformance.py tests/test_hip_longctx_filled.py -x -q 2>&1 | tail -4
{
python tools/decode_vtable_run.py smollm-135m 512 dyn
ZGML_HIP_SPIN_WAIT_US=0 python tools/decode_vtable_run.py smollm-135m 512 dyn
python tools/decode_vtable_run.py smollm-135m 512 dyn
ZGML_HIP_SPIN_WAIT_US=0 python tools/decode_vtable_run.py smollm-135m 512 dyn
python tools/decode_run.py smollm-135m 512
ZGML_HIP_TAIL_FUSED=0 python tools/decode_run.py smollm-135m 512
python tools/decode_run.py smollm-135m 512
ZGML_HIP_TAIL_FUSED=0 python tools/decode_run.py smollm-135m 512
python tools/decode_run.py llama2-7b 128
ZGML_HIP_TAIL_FUSED=0 python tools/decode_run.py llama2-7b 128
} 2>&1 | tee $O/vtable_prof4.txt
