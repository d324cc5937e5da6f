"""Soak of the drop-in path: whole-context greedy decode through refresh_dynamic + execute_program (one graph launch per token,
outputs pinned as the adapter does), repeated; every run must reproduce the resident loop's tokens and raise no error."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zgml_amd import Backend, llama  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "smollm-135m"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kvq = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = 2040
be = Backend(0)
cfg = llama.preset(name, 2048)
cfg.kv_quant_block = kvq
m = llama.Model(cfg, llama.Q4_0, threads=16)
s = llama.Session(m, llama.hip_backend_fns(be))
s.resident_setup(be)
want = s.resident_decode(1, 0, n).tolist()
s.use_dynamic_refresh()
s.pin_outputs(be)
t0 = time.perf_counter()
for r in range(runs):
    toks, _ = s.decode(1, 0, n)
    assert not be.last_error(), be.last_error()
    got = toks.tolist()
    if got != want:
        k = next(i for i in range(n) if got[i] != want[i])
        raise SystemExit(f"run {r}: token {k} differs ({got[k]} vs {want[k]})")
dt = time.perf_counter() - t0
print(f"SOAK_VTABLE_OK {name} kvq={kvq}: {runs} x {n} tokens equal to the resident loop's, {runs * n / dt:.1f} tok/s over the whole context")
s.close()
m.close()
be.close()
