import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zgml_amd import Backend, llama, capi
be = Backend(0)
be.set_option(capi.OPT_F16_DENSE_WEIGHTS, 1)
m = llama.Model(llama.preset("llama2-7b", 2048), llama.F32_DENSE, threads=16)
s = llama.Session(m, llama.hip_backend_fns(be))
s.resident_setup(be)
w = s.resident_decode(1, 0, 4)
t0 = time.perf_counter()
steps = 64
s.resident_decode(int(w[-1]), 4, steps)
dt = time.perf_counter() - t0
print(f"llama2-7b dense f16 decode: {steps/dt:.1f} tok/s, {1e3*dt/steps:.3f} ms/token, weights 13.5 GB -> {13.48e9*steps/dt/1e12:.2f} TB/s")
s.close(); m.close(); be.close()
