"""Soak: many resident greedy-decode runs of one model over the whole context (graph replays of the fused launches: counters,
`seen` words and the hand-off time-out word carry over from run to run); every run must give the same tokens and no error."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zgml_amd import Backend, llama  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "smollm-135m"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
kvq = int(sys.argv[3]) if len(sys.argv) > 3 else 0
be = Backend(0)
cfg = llama.preset(name, 2048)
cfg.kv_quant_block = kvq
m = llama.Model(cfg, llama.Q4_0, threads=16)
s = llama.Session(m, llama.hip_backend_fns(be))
s.resident_setup(be)
first = None
t0 = time.perf_counter()
for r in range(runs):
    toks = s.resident_decode(1, 0, 2040)
    assert not be.last_error(), be.last_error()
    if first is None:
        first = toks.tolist()
    assert toks.tolist() == first, f"run {r} differs"
dt = time.perf_counter() - t0
print(f"SOAK_OK {name} kvq={kvq}: {runs} x 2040 tokens identical, {runs * 2040 / dt:.1f} tok/s over the whole context")
s.close()
m.close()
be.close()
