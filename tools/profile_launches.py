"""Per-launch event timing of the first layers' plan entries (ZGML_HIP_OPT_PROFILE)."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["ZGML_HIP_DEBUG_PLAN"] = "2"
from zgml_amd import Backend, capi, llama  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "smollm-135m"
be = Backend(0)
m = llama.Model(llama.preset(name, 2048), llama.Q4_0, threads=16)
s = llama.Session(m, llama.hip_backend_fns(be))
for pos in range(40):
    s.step(1, pos, want_logits=False)
be.set_option(capi.OPT_PROFILE, 1)
for pos in range(40, 48):
    s.step(1, pos, want_logits=False)
