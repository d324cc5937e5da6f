"""Greedy decode through execute_program per token (the vtable path: program graph only, no resident kernels)."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zgml_amd import Backend, llama  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "smollm-135m"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dyn = len(sys.argv) > 3 and sys.argv[3] == "dyn"  # the adapter's refresh: (slice_pos, seq_kv) through zgml_hip_refresh_dynamic
be = Backend(0)
m = llama.Model(llama.preset(name, 2048), llama.Q4_0, threads=16)
s = llama.Session(m, llama.hip_backend_fns(be))
if dyn:
    s.use_dynamic_refresh()
    s.pin_outputs(be, os.environ.get("ZGML_PIN_OUTPUTS", "1") != "0")
s.decode(1, 0, 4)
toks, secs = s.decode(1, 0, steps)
print(f"{name}: {steps / secs:.1f} tok/s through the vtable ({'dynamic' if dyn else 'full'} refresh); tokens {toks[:6].tolist()}")
s.close()
m.close()
be.close()
