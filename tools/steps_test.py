import sys, time
sys.path.insert(0, '.')
from zgml_amd import Backend, llama
be = Backend(0)
m = llama.Model(llama.preset("smollm-135m"), llama.Q4_0, include_dead_f32=False, threads=16)
s = llama.Session(m, llama.hip_backend_fns(be))
s.resident_setup(be)
w = s.resident_decode(1, 0, 8)
for K in (1, 5, 20, 50, 200, 20, 200):
    be.synchronize()
    t0 = time.perf_counter()
    s.resident_decode(int(w[-1]), 8, K)
    be.synchronize()
    dt = time.perf_counter() - t0
    print(f"K={K}: {K/dt:.1f} tok/s, {dt*1e3:.3f} ms total, {dt*1e3/K:.4f} ms/step")
