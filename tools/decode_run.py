"""Resident greedy decode of one model for N tokens (profiling driver)."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zgml_amd import Backend, llama  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "smollm-135m"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
start = int(sys.argv[3]) if len(sys.argv) > 3 else 4  # decode positions [start, start + steps): long contexts without a prefill
kvq = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # 32: int8 KV caches (kvq_store / attention_kvq extension ops)
be = Backend(0)
cfg = llama.preset(name, 2048)
cfg.kv_quant_block = kvq
m = llama.Model(cfg, llama.Q4_0, threads=16)
s = llama.Session(m, llama.hip_backend_fns(be))
s.resident_setup(be)
w = s.resident_decode(1, 0, 4)
t0 = time.perf_counter()
s.resident_decode(int(w[-1]), start, steps)
dt = time.perf_counter() - t0
tag = " int8-KV" if kvq else ""
print(f"{name}{tag} pos {start}..{start + steps}: {steps / dt:.1f} tok/s, {1e3 * dt / steps:.3f} ms/token")
s.close()
m.close()
be.close()
