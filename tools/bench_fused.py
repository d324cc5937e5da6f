"""Times fused mat-vec variants in isolation: chains of independent [prologue] qmatmul [epilogue]
groups over a ring of distinct 4096x4096 Q4_0 weights (graph replay, wall clock / launches)."""
import sys
import time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zgml_amd import Backend, DeviceOp, DeviceProgram, ProgramIO, QuantizedWeightUpload  # noqa: E402

K = N = 4096
RING = 24
rng = np.random.default_rng(0)
base_q = rng.integers(-8, 8, K * N).astype(np.int8)
scales = (rng.random(K * N // 32).astype(np.float16) * 0.01 + 0.001).astype(np.float32)
qws = [QuantizedWeightUpload(np.roll(base_q, i * 977), scales, K, N, 32) for i in range(RING)]
x = rng.standard_normal(K).astype(np.float32)
gamma = np.ones(K, np.float32)
ne, st = (K, 1, 1, 1), (1, K, K, K)


def variant(name):
    ops, sizes = [], [K, K]  # 0: x, 1: gamma
    def buf(n):
        sizes.append(n)
        return len(sizes) - 1
    for i in range(RING):
        if name == "plain":
            ops.append(DeviceOp.qmatmul(buf(N), 0, i, 1, N, K))
        elif name == "rmsnorm_pro":
            bare, rep, nx, y = buf(K), buf(K), buf(K), buf(N)
            ops += [DeviceOp.rmsnorm(bare, 0, 1, K, 1e-5), DeviceOp.repeat(rep, 1, K, ne, ne, st, st),
                    DeviceOp.elementwise("mul", nx, bare, rep, K), DeviceOp.qmatmul(y, nx, i, 1, N, K)]
        elif name == "add_epi":
            y, z = buf(N), buf(N)
            ops += [DeviceOp.qmatmul(y, 0, i, 1, N, K), DeviceOp.elementwise("add", z, 0, y, N)]
        elif name == "group3":
            if i % 3 == 0 and i + 2 < RING:
                ops += [DeviceOp.qmatmul(buf(N), 0, i + j, 1, N, K) for j in range(3)]
    return DeviceProgram(ops=ops, buffer_sizes=sizes, initial_uploads=[ProgramIO(0, x), ProgramIO(1, gamma)], qweights=qws)


be = Backend(0)
for name in ("plain", "rmsnorm_pro", "add_epi", "group3"):
    prog = variant(name)
    h = be.compileProgram(prog)
    n_mv = sum(1 for o in prog.ops if o.kind == "qmatmul")
    for _ in range(3):
        be._lib.zgml_hip_enqueue_program(be.ctx, h)
    be.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        be._lib.zgml_hip_enqueue_program(be.ctx, h)
    be.synchronize()
    dt = time.perf_counter() - t0
    prof = be.getRuntimeProfile(h)
    print(f"{name:12s} {1e6 * dt / (reps * n_mv):7.2f} us per mat-vec ({n_mv} mat-vecs, plan launches/replay: see debug)", flush=True)
    be.freeProgram(h)
