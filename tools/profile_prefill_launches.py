"""Per-launch event timing of the first launches of a prefill chunk (ZGML_HIP_OPT_PROFILE, eager plan)."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["ZGML_HIP_DEBUG_PLAN"] = "2"
from zgml_amd import Backend, capi, llama  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "llama2-7b"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
be = Backend(0)
m = llama.Model(llama.preset(name, 512), llama.Q4_0, threads=16, token_len=T)
s = llama.Session(m, llama.hip_backend_fns(be))
toks = [(7 * i + 3) % m.cfg.vocab_size for i in range(T)]
for _ in range(3):
    s.prefill(toks, 0)
be.set_option(capi.OPT_PROFILE, 1)
for _ in range(8):
    s.prefill(toks, 0)
s.close()
m.close()
be.close()
