"""Per-DeviceOp-kind time of one decode step (ZGML_HIP_OPT_PROFILE: hipEvents around every launch)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zgml_amd import Backend, capi, llama  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "smollm-135m"
be = Backend(0)
m = llama.Model(llama.preset(name, 2048), llama.Q4_0, threads=16)
s = llama.Session(m, llama.hip_backend_fns(be))
for pos in range(4):
    s.step(1, pos, want_logits=False)
be.set_option(capi.OPT_PROFILE, 1)
prof = be.getRuntimeProfile(s.handle)
base = list(prof.time_ns)
n = 4
for pos in range(4, 4 + n):
    s.step(1, pos, want_logits=False)
prof = be.getRuntimeProfile(s.handle)
tot = 0
for i, k in enumerate(capi.DOP_KINDS):
    us = (prof.time_ns[i] - base[i]) / n / 1e3
    tot += us
    print(f"{k:20s} {us:10.1f} us/token")
print(f"{'total':20s} {tot:10.1f} us/token (sum of per-launch event times, launches serialised)")
