"""Parity of the HIP quantized mat-vec / matmul with the oracle's exact-dequant path
(src/quant.zig:475-578 == src/backend/reference.zig:530-565).

Tolerance: |delta| <= 2e-5 * sum_k |x_k * w_kn| — summation order only; the GPU does
not quantise activations."""
import numpy as np
import pytest

from zgml_amd import DeviceOp, DeviceProgram, ProgramIO, QuantizedWeightUpload
from tests.synth import q4_0_blocks_from_int8, synth_weight, synth_x

pytestmark = pytest.mark.gpu
f32 = np.float32
TOL = 2e-5


def bound(data, scales, x, M, N, K, bs):
    w = np.abs(data.astype(np.float64) * np.repeat(scales.astype(np.float64), bs)[: data.size]).reshape(K, N)
    return np.abs(x.reshape(M, K).astype(np.float64)) @ w


def run_both(hip_backend, oracle, data, scales, x, M, N, K, bs=32, in_off=0, in_rs=0, dst_off=0, dst_rs=0):
    xin = np.full(in_off + (M - 1) * (in_rs or K) + K + 3, 99, f32)
    for m in range(M):
        xin[in_off + m * (in_rs or K):][:K] = x.reshape(M, K)[m]
    dst_len = dst_off + (M - 1) * (dst_rs or N) + N + 2
    prog = DeviceProgram(ops=[DeviceOp.qmatmul(1, 0, 0, M, N, K, in_off, in_rs, dst_off, dst_rs)],
                         buffer_sizes=[xin.size, dst_len],
                         initial_uploads=[ProgramIO(0, xin), ProgramIO(1, np.full(dst_len, -7, f32))],
                         qweights=[QuantizedWeightUpload(data, scales, K, N, bs)])
    want = oracle.run_program(prog, 1, dst_len)
    got = oracle.run_program(prog, 1, dst_len, backend=hip_backend)
    return want, got


@pytest.mark.parametrize("K,N", [(576, 576), (576, 192), (1536, 576), (576, 1536), (4096, 4096), (100, 64), (33, 32),
                                 (4096, 11008)])
@pytest.mark.parametrize("kind", ["q4", "q8"])
def test_matvec_matches_oracle(hip_backend, oracle, K, N, kind):
    rng = np.random.default_rng(0xD3FA + K + N)
    x = rng.standard_normal(K).astype(f32)
    if kind == "q4":  # GGUF Q4_0-sourced: nibbles + f16 scales through the loader restatement
        data = rng.integers(-8, 8, K * N).astype(np.int8)
        scales = (rng.random(K * N // 32 + 1).astype(np.float16) * 0.05 + 0.001).astype(f32)[: (K * N + 31) // 32]
        raw = q4_0_blocks_from_int8(np.pad(data, (0, (-data.size) % 32)), scales)
        d2, s2 = oracle.gguf_to_int8(raw, data.size, "q4_0")
        assert np.array_equal(d2, data) and np.array_equal(s2, scales)
    else:  # session.quantize() path: fromSlice gives full-range int8 + arbitrary f32 scales
        w = rng.standard_normal(K * N).astype(f32) * 0.05
        data, scales = oracle.quantize_from_slice(w, K, N, 32)
    want, got = run_both(hip_backend, oracle, data, scales, x, 1, N, K)
    b = bound(data, scales, x, 1, N, K, 32).ravel()
    assert np.all(np.abs(got[:N] - want[:N]) <= TOL * b + 1e-30), np.max(np.abs(got[:N] - want[:N]) / (b + 1e-30))
    assert np.all(got[N:] == -7)  # nothing written past N


def test_matmul_rows_offsets_strides(hip_backend, oracle):
    rng = np.random.default_rng(7)
    M, K, N = 3, 192, 96
    x = rng.standard_normal(M * K).astype(f32)
    data = rng.integers(-8, 8, K * N).astype(np.int8)
    scales = (rng.random(K * N // 32).astype(np.float16) * 0.1).astype(f32)
    want, got = run_both(hip_backend, oracle, data, scales, x, M, N, K, in_off=5, in_rs=K + 7, dst_off=2, dst_rs=N + 3)
    np.testing.assert_allclose(got, want, atol=1e-4, rtol=1e-5)
    assert np.array_equal(got == -7, want == -7)  # sentinels untouched in the same places


@pytest.mark.parametrize("M", [2, 5, 16, 17, 32, 45])
@pytest.mark.parametrize("K,N,kind", [(576, 192, "q4"), (1536, 576, "q4"), (100, 64, "q4"), (4096, 512, "q4"),
                                      (11008, 64, "q4"), (576, 192, "q8"), (100, 64, "q8"), (2048, 96, "q8f16")])
def test_matmul_tile_kernel_matches_oracle(hip_backend, oracle, M, K, N, kind):
    """M > 1 goes to the tile kernels (prefill plans) — Q4_0 with f16-exact scales: split_a + qmatmul_xdl2
    (exact bf16 splits of x and of scale * q); Q8_0 / f32 scales: the A-side-scale form. Same tolerance as the
    mat-vec, with ragged M (partial 16-row tiles, three tiles at M = 45), K tails and row strides."""
    rng = np.random.default_rng(0xA11 + M * 7 + K + N)
    x = rng.standard_normal(M * K).astype(f32)
    if kind == "q4":
        data = rng.integers(-8, 8, K * N).astype(np.int8)
        scales = (rng.random((K * N + 31) // 32).astype(np.float16) * 0.05 + 0.001).astype(f32)
    elif kind == "q8f16":  # GGUF Q8_0-sourced: full-range int8 with f16-exact scales
        data = rng.integers(-127, 128, K * N).astype(np.int8)
        scales = (rng.random((K * N + 31) // 32).astype(np.float16) * 0.01 + 0.001).astype(f32)
    else:
        data, scales = oracle.quantize_from_slice(rng.standard_normal(K * N).astype(f32) * 0.05, K, N, 32)
    want, got = run_both(hip_backend, oracle, data, scales, x, M, N, K, in_off=4, in_rs=K + 4, dst_off=1, dst_rs=N + 5)
    b = np.zeros_like(want, dtype=np.float64)
    bb = bound(data, scales, x, M, N, K, 32)
    for m in range(M):
        b[1 + m * (N + 5):][:N] = bb[m]
    assert np.all(np.abs(got - want) <= TOL * b + 1e-30), np.max(np.abs(got - want) / (b + 1e-30))
    assert np.array_equal(got == -7, want == -7)  # sentinels between rows untouched


def _q4_case(rng, M, K, N):
    x = rng.standard_normal(M * K).astype(f32)
    data = rng.integers(-8, 8, K * N).astype(np.int8)
    scales = (rng.random((K * N + 31) // 32).astype(np.float16) * 0.05 + 0.001).astype(f32)
    return x, data, scales


@pytest.mark.parametrize("M", [5, 16, 32])
@pytest.mark.parametrize("K,N", [(576, 12288), (1100, 10240), (4096, 12288)])
def test_matmul_shared_a_kernel_matches_oracle(hip_backend, oracle, M, K, N):
    """M <= 32 with a WIDE output (>= 40 workgroup-columns of 256) takes qmatmul_xdl5_kernel: the A pieces staged once per
    workgroup in LDS, the (column, K step) pairs cut into even runs, partial tiles of a column summed by its last arriver in
    run order. K = 576: 5 steps and runs of 2 (every column is shared by 3 workgroups, runs straddle columns); K = 1100: the
    last step is mostly past the end; K = 4096 x 12288: the q/k/v shape of the 7B layer. Same bound as the other forms, and
    two executions of the same program agree bit for bit (the fan-in is ordered, not first come first served)."""
    rng = np.random.default_rng(0x5A + M + K + N)
    x, data, scales = _q4_case(rng, M, K, N)
    want, got = run_both(hip_backend, oracle, data, scales, x, M, N, K, in_off=4, in_rs=K + 4, dst_off=1, dst_rs=N + 5)
    b = np.zeros_like(want, dtype=np.float64)
    bb = bound(data, scales, x, M, N, K, 32)
    for m in range(M):
        b[1 + m * (N + 5):][:N] = bb[m]
    assert np.all(np.abs(got - want) <= TOL * b + 1e-30), np.max(np.abs(got - want) / (b + 1e-30))
    assert np.array_equal(got == -7, want == -7)
    _, again = run_both(hip_backend, oracle, data, scales, x, M, N, K, in_off=4, in_rs=K + 4, dst_off=1, dst_rs=N + 5)
    assert np.array_equal(got, again)


@pytest.mark.parametrize("M,K,N", [(45, 576, 12288), (64, 1100, 10240), (128, 4096, 12288), (200, 576, 12288), (100, 4096, 22016)])
def test_matmul_wide_outputs_above_32_rows_match_oracle(hip_backend, oracle, M, K, N):
    """33 <= M with a WIDE output: the K-split tile kernel at 4 or 8 m-tiles per workgroup (M = 45 / 64: four, 45 with padding
    rows; 128: eight, the prefill chunk of the 7B layer's q/k/v; 200: two tile groups, the second with padding rows; 100 x 22016:
    the gate / up width, 7 of 8 tiles live). Same bound as every other form (two-piece A operand: 2^-17 sum|x w|), bit-identical
    across two executions (ordered fan-in)."""
    rng = np.random.default_rng(0x7A + M + K + N)
    x, data, scales = _q4_case(rng, M, K, N)
    want, got = run_both(hip_backend, oracle, data, scales, x, M, N, K, in_off=4, in_rs=K + 4, dst_off=1, dst_rs=N + 5)
    b = np.zeros_like(want, dtype=np.float64)
    bb = bound(data, scales, x, M, N, K, 32)
    for m in range(M):
        b[1 + m * (N + 5):][:N] = bb[m]
    assert np.all(np.abs(got - want) <= TOL * b + 1e-30), np.max(np.abs(got - want) / (b + 1e-30))
    assert np.array_equal(got == -7, want == -7)
    _, again = run_both(hip_backend, oracle, data, scales, x, M, N, K, in_off=4, in_rs=K + 4, dst_off=1, dst_rs=N + 5)
    assert np.array_equal(got, again)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("knobs", [{}, {"ZGML_QMM_XDL7_MIN_COLS": "1", "ZGML_QMM_XDL7_MIN_RUN": "1"}, {"ZGML_QMM_XDL7_MIN_COLS": "1", "ZGML_QMM_XDL7_MIN_RUN": "1000"}])
def test_shared_a_experiment_kernel_above_32_rows(knobs):
    """The round-4 experiment kernel (qmatmul_xdl7_kernel: the M <= 32 form's shared-A work list at 4 / 8 m-tiles per workgroup;
    measured slower than the K-split kernel, so it exists in the diagnostics library only, ZGML_QMM_XDL7=1) stays parity green:
    the wide M > 32 cases above, then forced onto every M > 32 case of the tile tests and the 128-token prefill chunk (narrow and
    ragged outputs, grouped launches) with runs of one step and with whole columns only. A process per setting (switches are read once)."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    trace_lib = Path(__file__).resolve().parent.parent / "zgml_amd" / "lib" / "libzgml_hip_trace.so"
    assert trace_lib.exists(), "run __graft_entry__.build(): it builds the diagnostics library too"
    env = dict(os.environ, ZGML_HIP_LIB=str(trace_lib), ZGML_QMM_XDL7="1", **knobs)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_hip_qmatvec.py", "tests/test_hip_l7dims.py",
                        "-k", "tile_kernel or wide_outputs_above_32 or chunk128 or worst_case"], capture_output=True, text=True, timeout=880, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.timeout(900)
def test_shared_a_kernel_on_narrow_and_ragged_outputs():
    """The same kernel forced onto every M <= 32 case of this file and of the 7B-dimension file (ZGML_QMM_XDL5_MIN_COLS=1:
    outputs narrower than one workgroup-column, block-column counts that do not fill the last workgroup, grouped q/k/v and
    gate/up launches whose parts end mid-workgroup), in a process of its own because the switch is read once."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, ZGML_QMM_XDL5_MIN_COLS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_hip_qmatvec.py", "tests/test_hip_l7dims.py", "tests/test_hip_llama.py",
                        "-k", "tile_kernel or shared_a_kernel_matches or chunk32 or prefill"], capture_output=True, text=True, timeout=880, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("knobs", [{"ZGML_QMM_XDL5_MIN_RUN": "1", "ZGML_QMM_XDL5_WGS_PER_CU": "4"}, {"ZGML_QMM_XDL5_MIN_RUN": "3"},
                                   {"ZGML_QMM_XDL5_MIN_RUN": "7"}, {"ZGML_QMM_XDL5_MIN_RUN": "1000"}])
def test_shared_a_kernel_other_cuts_of_the_work_list(knobs):
    """The cut of the (column, K step) list into runs decides which workgroups share a column, where a run straddles two columns
    and how many partial tiles a column's last arriver sums: runs of one step (four workgroups per CU queued), of 3 and of 7 steps
    (boundaries that divide neither S = 5, 35 nor 128), and whole columns only (no fan-in at all) — every M <= 32 tile test of
    this file on the shared-A kernel, a process per cut because the switches are read once."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, ZGML_QMM_XDL5_MIN_COLS="1", **knobs)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_hip_qmatvec.py", "-k", "tile_kernel or shared_a_kernel_matches"],
                       capture_output=True, text=True, timeout=880, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_raw_layout_odd_shapes_bit_exact(hip_backend, oracle):
    """bs != 32 or N % 32 != 0 uses the k-sequential kernel: same loop order as the reference."""
    rng = np.random.default_rng(8)
    for (M, K, N, bs) in [(2, 3, 3, 4), (1, 50, 37, 32), (2, 64, 48, 16)]:
        x = rng.standard_normal(M * K).astype(f32)
        data = rng.integers(-127, 128, K * N).astype(np.int8)
        scales = rng.random((K * N + bs - 1) // bs).astype(f32)
        want, got = run_both(hip_backend, oracle, data, scales, x, M, N, K, bs=bs)
        assert np.array_equal(got, want)


def test_linearity_and_zero_input_full_size(hip_backend, oracle):
    """Size-independent properties at the BASELINE shape: y(a*x1 + x2) == a*y(x1) + y(x2) up to
    rounding, and x = 0 -> y = 0 exactly."""
    K = N = 4096
    data, scales = synth_weight(K, N, True, 3)
    rng = np.random.default_rng(9)
    x1, x2 = rng.standard_normal(K).astype(f32), rng.standard_normal(K).astype(f32)
    qw = QuantizedWeightUpload(data, scales, K, N, 32)
    prog = DeviceProgram(ops=[DeviceOp.qmatmul(1, 0, 0, 1, N, K)], buffer_sizes=[K, N], qweights=[qw])
    h = hip_backend.compileProgram(prog)

    def y(x):
        out = np.zeros(N, f32)
        hip_backend.executeProgram(h, [ProgramIO(0, np.ascontiguousarray(x, f32))], [ProgramIO(1, out)])
        return out
    y1, y2, y12, y0 = y(x1), y(x2), y(2.0 * x1 + x2), y(np.zeros(K, f32))
    hip_backend.freeProgram(h)
    assert np.all(y0 == 0)
    b = bound(data, scales, np.abs(2 * x1) + np.abs(x2), 1, N, K, 32).ravel()
    assert np.all(np.abs(y12 - (2 * y1 + y2)) <= 4 * TOL * b)


@pytest.mark.parametrize("q4", [1, 0])
def test_device_synth_generator_matches_host(hip_backend, oracle, q4):
    """The roofline ring's on-device generator produces the same matrices as tests/synth.py."""
    import ctypes as C
    K, N = 256, 128
    x = synth_x(K)
    y = np.zeros(N, f32)
    rc = hip_backend._lib.zgml_hip_qmatvec_synth(hip_backend.ctx, K, N, q4, 5, x.ctypes.data, y.ctypes.data)
    assert rc == 0
    data, scales = synth_weight(K, N, bool(q4), 5)
    want = oracle.qmatmul_exact(data, scales, x, 1, N, K)
    b = bound(data, scales, x, 1, N, K, 32).ravel()
    assert np.all(np.abs(y - want) <= TOL * b)


@pytest.mark.parametrize("kind", ["q4", "q8"])
def test_tile_kernel_rows_equal_matvec_full_size(hip_backend, oracle, kind):
    """Size-independent property at the BASELINE shape (4096 x 4096, prefill batch 32): every row of the
    M = 32 tile-kernel result equals the M = 1 mat-vec of that row (two different kernels, same packed
    weights) within the summation-order bound; zero rows give exact zeros."""
    K = N = 4096
    M = 32
    data, scales = synth_weight(K, N, kind == "q4", 7)
    rng = np.random.default_rng(21)
    x = rng.standard_normal((M, K)).astype(f32)
    x[5] = 0.0
    qw = QuantizedWeightUpload(data, scales, K, N, 32)
    prog = DeviceProgram(ops=[DeviceOp.qmatmul(1, 0, 0, M, N, K), DeviceOp.qmatmul(3, 2, 0, 1, N, K)],
                         buffer_sizes=[M * K, M * N, K, N], qweights=[qw])
    h = hip_backend.compileProgram(prog)
    Y, y = np.zeros(M * N, f32), np.zeros(N, f32)
    absw = np.abs(data.astype(np.float32) * np.repeat(scales, 32)).reshape(K, N)
    for m in (0, 5, 17, 31):
        hip_backend.executeProgram(h, [ProgramIO(0, x.ravel()), ProgramIO(2, x[m].copy())], [ProgramIO(1, Y), ProgramIO(3, y)])
        bound = np.abs(x[m]).astype(np.float64) @ absw.astype(np.float64)
        assert np.all(np.abs(Y.reshape(M, N)[m] - y) <= 2 * TOL * bound + 1e-30), m
    assert np.all(Y.reshape(M, N)[5] == 0)
    hip_backend.freeProgram(h)


def _q4_weight(rng, K, N):
    data = rng.integers(-8, 8, K * N).astype(np.int8)
    scales = (rng.random(K * N // 32).astype(np.float16) * 0.05 + 0.001).astype(f32)
    return QuantizedWeightUpload(data, scales, K, N, 32)


@pytest.mark.parametrize("K,N", [(576, 576), (4096, 1024)])
def test_matvec_prologues_and_epilogue_chains_store_every_intermediate(hip_backend, oracle, K, N):
    """The planner folds the ops around an M = 1 quantized mat-vec into its launch (prologue: rmsnorm -> mul gamma or a
    plain mul; epilogue: element-wise consumers) and the kernel still writes every intermediate buffer. Chains here:
    the residual add and the SiLU chain exactly as the LLaMA lowering emits them (straight-line fast paths), the SiLU
    chain with its steps regrouped (must NOT match the fast path's pattern), and a chain no fast path knows
    (relu, mul by a vector, swapped add, abs, neg) through the generic step interpreter. Every buffer against the oracle."""
    from zgml_amd import FusedEwStep
    rng = np.random.default_rng(K * 7 + N)
    x = rng.standard_normal(K).astype(f32)
    gamma = (rng.random(K).astype(f32) + 0.5)
    vec = rng.standard_normal(N).astype(f32)
    one = np.ones(N, f32)
    # buffers: 0 x, 1 gamma, 2 normed, 3 xg (mat-vec input), 4 vecN, 5 ones, 6.. outputs
    B = dict(x=0, gamma=1, normed=2, xg=3, vec=4, one=5, gate=6, up=7, exp_neg=8, silu=9, resid=10, y=11, t1=12, t2=13, t3=14, t4=15, t5=16,
             g2=17, e2=18, s2=19, s3=20, s4=21)
    sizes = [K, K, K, K, N, N] + [N] * 16
    ops = [
        DeviceOp.rmsnorm(B["normed"], B["x"], 1, K, 1e-5),
        DeviceOp.elementwise("mul", B["xg"], B["normed"], B["gamma"], K),
        # gate / up over the same input: a grouped launch; gate carries the LLaMA SiLU chain
        DeviceOp.qmatmul(B["gate"], B["xg"], 0, 1, N, K),
        DeviceOp.qmatmul(B["up"], B["xg"], 1, 1, N, K),
        DeviceOp.fused_elementwise([FusedEwStep("neg"), FusedEwStep("exp")], N, B["exp_neg"], B["gate"]),
        DeviceOp.fused_elementwise([FusedEwStep("add", False, B["one"], 0), FusedEwStep("recip"), FusedEwStep("mul", True, B["gate"], 0)], N,
                                   B["silu"], B["exp_neg"]),
        # a single-matrix launch with the residual add
        DeviceOp.qmatmul(B["y"], B["xg"], 2, 1, N, K),
        DeviceOp.elementwise("add", B["resid"], B["vec"], B["y"], N),
        # a chain no fast path knows
        DeviceOp.qmatmul(B["t1"], B["xg"], 3, 1, N, K),
        DeviceOp.elementwise("relu", B["t2"], B["t1"], B["t1"], N),
        DeviceOp.elementwise("mul", B["t3"], B["t2"], B["vec"], N),
        DeviceOp.elementwise("add", B["t4"], B["vec"], B["t3"], N),
        DeviceOp.fused_elementwise([FusedEwStep("abs"), FusedEwStep("neg")], N, B["t5"], B["t4"]),
        # the SiLU arithmetic grouped differently (neg alone, then exp + add, then recip, then mul): not the fast path's pattern
        DeviceOp.qmatmul(B["g2"], B["xg"], 4, 1, N, K),
        DeviceOp.elementwise("neg", B["e2"], B["g2"], B["g2"], N),
        DeviceOp.fused_elementwise([FusedEwStep("exp"), FusedEwStep("add", True, B["one"], 0)], N, B["s2"], B["e2"]),
        DeviceOp.elementwise("recip", B["s3"], B["s2"], B["s2"], N),
        DeviceOp.elementwise("mul", B["s4"], B["g2"], B["s3"], N),
    ]
    prog = DeviceProgram(ops=ops, buffer_sizes=sizes,
                         initial_uploads=[ProgramIO(B["x"], x), ProgramIO(B["gamma"], gamma), ProgramIO(B["vec"], vec), ProgramIO(B["one"], one)],
                         qweights=[_q4_weight(rng, K, N) for _ in range(5)])
    for name in ("normed", "xg", "gate", "up", "exp_neg", "silu", "y", "resid", "t1", "t2", "t3", "t4", "t5", "g2", "e2", "s2", "s3", "s4"):
        n = sizes[B[name]]
        want = oracle.run_program(prog, B[name], n)
        got = oracle.run_program(prog, B[name], n, backend=hip_backend)
        assert not hip_backend.last_error(), hip_backend.last_error()
        # summation order only — but exp() turns the mat-vec's absolute difference (~1e-4 at K = 4096: the oracle adds 4096
        # f32 products sequentially) into a relative one, hence 1e-3 here: the test is about structure, not the last bit
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-4 * scale, err_msg=name)


@pytest.mark.parametrize("K0,K,Ns", [(1536, 576, (576, 192, 192)), (576, 576, (1536, 1536)), (2048, 2048, (2048, 512, 512)), (4096, 4096, (4096, 4096, 4096))])
def test_prepared_norm_hand_over_stores_every_intermediate(hip_backend, oracle, K0, K, Ns):
    """The tail of one decoder half and the head of the next as the LLaMA lowering emits them (llama_transformer.zig:192-253):
    projection -> residual add -> rmsnorm -> mul(gamma) -> q/k/v (or gate/up) over the result. The planner folds this into TWO
    launches, and the first one PREPARES the second one's norm (h * gamma and the partial sums of h^2: QMV_PRO_PRENORM) so that
    the second streams one vector and scales its finished sums — for K-on-lanes weights (K > 2048) since round 3, for the
    x-direct n-on-lanes launches of short-K models (SmolLM-135M's 576 / 1536, and 2048) since round 4. Every buffer the
    unfused program would have written — the projection, the residual sum, the normalised vector, its product with gamma, each
    output — against the oracle; the plan's text says which form ran."""
    rng = np.random.default_rng(K0 + 3 * K + len(Ns))
    xin = rng.standard_normal(K0).astype(f32)
    resid = rng.standard_normal(K).astype(f32)
    gamma = (rng.random(K).astype(f32) + 0.5)
    B = dict(xin=0, resid=1, gamma=2, y=3, h=4, normed=5, xg=6)
    sizes = [K0, K, K, K, K, K, K] + list(Ns)
    ops = [
        DeviceOp.qmatmul(B["y"], B["xin"], 0, 1, K, K0),
        DeviceOp.elementwise("add", B["h"], B["resid"], B["y"], K),
        DeviceOp.rmsnorm(B["normed"], B["h"], 1, K, 1e-5),
        DeviceOp.elementwise("mul", B["xg"], B["normed"], B["gamma"], K),
    ] + [DeviceOp.qmatmul(7 + t, B["xg"], 1 + t, 1, n, K) for t, n in enumerate(Ns)]
    prog = DeviceProgram(ops=ops, buffer_sizes=sizes,
                         initial_uploads=[ProgramIO(B["xin"], xin), ProgramIO(B["resid"], resid), ProgramIO(B["gamma"], gamma)],
                         qweights=[_q4_weight(rng, K0, K)] + [_q4_weight(rng, K, n) for n in Ns])
    h = hip_backend.compileProgram(prog)
    assert h
    try:
        plan = hip_backend.planText(h)
        lines = plan.strip().splitlines()
        assert len(lines) == 2, plan
        import os
        if not any(os.environ.get(v) == "0" for v in ("ZGML_HIP_PRENORM", "ZGML_HIP_PRENORM_NOL", "ZGML_QMV_XDIRECT")):
            assert "prepares-next-norm" in lines[0] and "pro prenorm" in lines[1] and f"parts {len(Ns)}" in lines[1], plan
        names = ["y", "h", "normed", "xg"] + [f"out{t}" for t in range(len(Ns))]
        idx = [B["y"], B["h"], B["normed"], B["xg"]] + [7 + t for t in range(len(Ns))]
        outs = [np.zeros(sizes[i], f32) for i in idx]
        for _ in range(2):  # (twice: the second run replays the graph)
            hip_backend.executeProgram(h, [], [ProgramIO(i, o) for i, o in zip(idx, outs)])
        for name, i, got in zip(names, idx, outs):
            want = oracle.run_program(prog, i, sizes[i])
            scale = max(1.0, float(np.abs(want).max()))
            # summation order only (the oracle adds K f32 products sequentially: ~1e-4 absolute at K = 4096)
            np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-4 * scale, err_msg=name)
    finally:
        hip_backend.freeProgram(h)


@pytest.mark.parametrize("K,F", [(576, 1536), (2048, 2048), (4096, 11008)])
def test_gate_up_pair_launch_stores_every_intermediate(hip_backend, oracle, K, F):
    """The SwiGLU half as the LLaMA lowering emits it (llama_transformer.zig:129-133, silu = nn.zig:38-44): gate and up over the
    same input, the SiLU chain on gate, silu(gate) * up, the down projection over the product. The planner turns the first launch into
    a PAIR launch — one workgroup computes the same 16 columns of gate AND up and stores the product itself — and the down
    projection streams it with no prologue: K-on-lanes weights (K > 2048) since round 3, the x-direct n-on-lanes launches of short-K
    models since round 4. Every buffer the unfused program would have written against the oracle; the plan's text says which form ran."""
    from zgml_amd import FusedEwStep
    rng = np.random.default_rng(K * 3 + F)
    x = rng.standard_normal(K).astype(f32)
    one = np.ones(F, f32)
    B = dict(x=0, one=1, gate=2, up=3, exp_neg=4, silu=5, act=6, down=7)
    sizes = [K, F, F, F, F, F, F, K]
    ops = [
        DeviceOp.qmatmul(B["gate"], B["x"], 0, 1, F, K),
        DeviceOp.fused_elementwise([FusedEwStep("neg"), FusedEwStep("exp")], F, B["exp_neg"], B["gate"]),
        DeviceOp.fused_elementwise([FusedEwStep("add", False, B["one"], 0), FusedEwStep("recip"), FusedEwStep("mul", True, B["gate"], 0)], F,
                                   B["silu"], B["exp_neg"]),
        DeviceOp.qmatmul(B["up"], B["x"], 1, 1, F, K),
        DeviceOp.elementwise("mul", B["act"], B["silu"], B["up"], F),
        DeviceOp.qmatmul(B["down"], B["act"], 2, 1, K, F),
    ]
    prog = DeviceProgram(ops=ops, buffer_sizes=sizes, initial_uploads=[ProgramIO(B["x"], x), ProgramIO(B["one"], one)],
                         qweights=[_q4_weight(rng, K, F), _q4_weight(rng, K, F), _q4_weight(rng, F, K)])
    h = hip_backend.compileProgram(prog)
    assert h
    try:
        plan = hip_backend.planText(h)
        lines = plan.strip().splitlines()
        assert len(lines) == 2, plan
        import os
        if not any(os.environ.get(v) == "0" for v in ("ZGML_HIP_PAIR", "ZGML_HIP_PAIR_NOL", "ZGML_QMV_XDIRECT", "ZGML_QMV_EPI_SILU")):
            assert " pair" in lines[0] and "pro none" in lines[1], plan
        names = ["gate", "up", "exp_neg", "silu", "act", "down"]
        outs = [np.zeros(sizes[B[n]], f32) for n in names]
        for _ in range(2):  # (twice: the second run replays the graph)
            hip_backend.executeProgram(h, [], [ProgramIO(B[n], o) for n, o in zip(names, outs)])
        for name, got in zip(names, outs):
            want = oracle.run_program(prog, B[name], sizes[B[name]])
            scale = max(1.0, float(np.abs(want).max()))
            # summation order only — but exp() turns the mat-vec's absolute difference into a relative one (see the chain test above)
            np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-4 * scale, err_msg=name)
    finally:
        hip_backend.freeProgram(h)


@pytest.mark.parametrize("K0,K,F", [(1536, 576, 1536), (2048, 2048, 1024), (4096, 4096, 11008)])
def test_swiglu_half_with_prepared_norm_and_pair_stores_every_intermediate(hip_backend, oracle, K0, K, F):
    """Both hand-overs in one chain, as a decoder layer has them: projection -> residual add -> rmsnorm -> mul(gamma) -> gate / up
    -> SiLU chain -> silu * up -> down projection. Three launches: the first prepares the second's norm, the second is a gate / up
    PAIR launch that consumes it (every workgroup also stores a slice of the normalised vector and of its product with gamma), the
    third streams the product. 2048 x 1024: fewer pair workgroups than 16-element slices of the input — the n-on-lanes pair is not
    armed there (its workgroups could not cover the slices), the prepared norm still is. Every buffer against the oracle."""
    from zgml_amd import FusedEwStep
    rng = np.random.default_rng(K0 + 5 * K + F)
    xin = rng.standard_normal(K0).astype(f32)
    resid = rng.standard_normal(K).astype(f32)
    gamma = (rng.random(K).astype(f32) + 0.5)
    one = np.ones(F, f32)
    B = dict(xin=0, resid=1, gamma=2, one=3, y=4, h=5, normed=6, xg=7, gate=8, up=9, exp_neg=10, silu=11, act=12, down=13)
    sizes = [K0, K, K, F, K, K, K, K, F, F, F, F, F, K]
    ops = [
        DeviceOp.qmatmul(B["y"], B["xin"], 0, 1, K, K0),
        DeviceOp.elementwise("add", B["h"], B["resid"], B["y"], K),
        DeviceOp.rmsnorm(B["normed"], B["h"], 1, K, 1e-5),
        DeviceOp.elementwise("mul", B["xg"], B["normed"], B["gamma"], K),
        DeviceOp.qmatmul(B["gate"], B["xg"], 1, 1, F, K),
        DeviceOp.fused_elementwise([FusedEwStep("neg"), FusedEwStep("exp")], F, B["exp_neg"], B["gate"]),
        DeviceOp.fused_elementwise([FusedEwStep("add", False, B["one"], 0), FusedEwStep("recip"), FusedEwStep("mul", True, B["gate"], 0)], F,
                                   B["silu"], B["exp_neg"]),
        DeviceOp.qmatmul(B["up"], B["xg"], 2, 1, F, K),
        DeviceOp.elementwise("mul", B["act"], B["silu"], B["up"], F),
        DeviceOp.qmatmul(B["down"], B["act"], 3, 1, K, F),
    ]
    prog = DeviceProgram(ops=ops, buffer_sizes=sizes,
                         initial_uploads=[ProgramIO(B["xin"], xin), ProgramIO(B["resid"], resid), ProgramIO(B["gamma"], gamma), ProgramIO(B["one"], one)],
                         qweights=[_q4_weight(rng, K0, K), _q4_weight(rng, K, F), _q4_weight(rng, K, F), _q4_weight(rng, F, K)])
    h = hip_backend.compileProgram(prog)
    assert h
    try:
        plan = hip_backend.planText(h)
        lines = plan.strip().splitlines()
        assert len(lines) == 3, plan
        import os
        if not any(os.environ.get(v) == "0" for v in ("ZGML_HIP_PRENORM", "ZGML_HIP_PRENORM_NOL", "ZGML_HIP_PAIR", "ZGML_HIP_PAIR_NOL", "ZGML_QMV_XDIRECT",
                                                        "ZGML_QMV_EPI_SILU", "ZGML_HIP_QMV_KON")):
            assert "prepares-next-norm" in lines[0] and "pro prenorm" in lines[1], plan
            assert (" pair" in lines[1]) == (K > 2048 or F >= K), plan
        names = ["y", "h", "normed", "xg", "gate", "up", "exp_neg", "silu", "act", "down"]
        outs = [np.zeros(sizes[B[n]], f32) for n in names]
        for _ in range(2):
            hip_backend.executeProgram(h, [], [ProgramIO(B[n], o) for n, o in zip(names, outs)])
        for name, got in zip(names, outs):
            want = oracle.run_program(prog, B[name], sizes[B[name]])
            scale = max(1.0, float(np.abs(want).max()))
            np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4 * scale, err_msg=name)
    finally:
        hip_backend.freeProgram(h)


def test_refresh_to_a_matmul_over_a_matvec_only_weight_fails_loudly(hip_backend, oracle):
    """compile_program packs a Q4_0 weight K-on-lanes when every op that uses it has M == 1
    (K >= 2048). A later refresh_program that changes static fields is legal for the reference's CPU backend
    (src/backend/cpu.zig:128-131) — but an M > 1 qmatmul over that weight has no kernel for the layout. It used to print to
    stderr and skip the launch (stale destination, no error). Now the refresh is REFUSED with an error on the context and the
    program keeps its previous ops: the next execute still computes the M = 1 product."""
    K, N = 4096, 64
    data, scales = synth_weight(K, N, True)
    x = synth_x(2 * K)
    dst_len = 2 * N
    ops1 = [DeviceOp.qmatmul(1, 0, 0, 1, N, K)]
    prog = DeviceProgram(ops=ops1, buffer_sizes=[2 * K, dst_len], initial_uploads=[ProgramIO(0, x), ProgramIO(1, np.zeros(dst_len, f32))],
                         qweights=[QuantizedWeightUpload(data, scales, K, N, 32)])
    want = oracle.run_program(prog, 1, dst_len)
    h = hip_backend.compileProgram(prog)
    try:
        out = np.zeros(dst_len, f32)
        hip_backend.executeProgram(h, [], [ProgramIO(1, out)])
        assert not hip_backend.last_error(), hip_backend.last_error()
        b = bound(data, scales, x[:K], 1, N, K, 32)
        assert np.all(np.abs(out[:N] - want[:N]) <= TOL * b.ravel() + 1e-30)
        hip_backend.refreshProgram(h, [DeviceOp.qmatmul(1, 0, 0, 2, N, K)])
        err = hip_backend.last_error()
        if "K-on-lanes" not in err:  # (the weight kept the tile-kernel layout, e.g. ZGML_HIP_QMV_KON=0: the refresh is simply honoured)
            assert not err, err
            out2 = np.zeros(dst_len, f32)
            hip_backend.executeProgram(h, [], [ProgramIO(1, out2)])
            want2 = oracle.run_program(DeviceProgram(ops=[DeviceOp.qmatmul(1, 0, 0, 2, N, K)], buffer_sizes=prog.buffer_sizes,
                                                     initial_uploads=prog.initial_uploads, qweights=prog.qweights), 1, dst_len)
            b2 = bound(data, scales, x, 2, N, K, 32)
            assert np.all(np.abs(out2 - want2) <= TOL * b2.ravel() + 1e-30)
            return
        assert "refresh_program" in err and "recompile" in err, err
        import ctypes as C
        from zgml_amd import capi
        capi.load_hip().zgml_hip_clear_error(hip_backend.ctx)
        out3 = np.full(dst_len, 5, f32)
        hip_backend.executeProgram(h, [], [ProgramIO(1, out3)])  # the previous (M = 1) ops are still in force
        assert not hip_backend.last_error(), hip_backend.last_error()
        assert np.array_equal(out3[:N], out[:N]) and np.all(out3[N:] == 0)
    finally:
        hip_backend.freeProgram(h)


def _bf16_rne(v):
    u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)).view(np.float32)


@pytest.mark.parametrize("pattern", ["same_sign", "alternating"])
@pytest.mark.parametrize("M,K,N", [(32, 4096, 512), (32, 4096, 12288), (128, 4096, 512)])
def test_matmul_two_piece_operand_bound_is_exercised_by_a_worst_case_vector(hip_backend, oracle, M, K, N, pattern):
    """the M > 1 Q4_0 tile kernels carry x as TWO bf16 pieces by round-to-nearest-even (kernels.h: split_a_pieces),
    |x - (a1 + a2)| <= 2^-17 |x| per element, and compute all four piece products with the weight's exact two pieces, so
    |delta| <= 2^-17 sum|x w| ~ 7.6e-6 sum|x w| — inside the contract's 2e-5. Random data meets that bound
    with room to spare because the per-element residuals cancel; here they do NOT: every |x_k| is the SAME value, chosen
    (numpy model of the split) as the float whose two-piece residual is the largest of 2^16 candidates, against weights of one
    sign in every column (`same_sign`: the residuals of a column add up coherently) and with alternating signs of x
    (`alternating`: the sum cancels, the bound's right-hand side does not). Asserted: the contract, and that the measured error
    is what the model predicts — at least a quarter of the coherent residual sum in the same-sign case, i.e. the test really
    sits on the operand split and would see a third dropped piece product or a truncating split."""
    rng = np.random.default_rng(11)
    cand = (1.0 + rng.random(1 << 16)).astype(f32)
    a1 = _bf16_rne(cand)
    a2 = _bf16_rne(cand - a1)
    res = np.abs(cand.astype(np.float64) - a1.astype(np.float64) - a2.astype(np.float64))
    c = cand[int(np.argmax(res))]
    rel = float(res.max() / c)
    assert 2.0 ** -19 < rel <= 2.0 ** -17
    sign = np.where(np.arange(K) % 2 == 0, 1.0, -1.0) if pattern == "alternating" else np.ones(K)
    x = np.tile((c * sign).astype(f32), M)
    data = rng.integers(1, 8, K * N).astype(np.int8)  # one sign per column (all positive)
    scales = (rng.random((K * N + 31) // 32).astype(np.float16) * 0.05 + 0.001).astype(f32)
    want, got = run_both(hip_backend, oracle, data, scales, x, M, N, K)
    b = bound(data, scales, x, M, N, K, 32).ravel()
    err = np.abs(got.astype(np.float64)[:M * N] - want.astype(np.float64)[:M * N])
    assert np.all(err <= TOL * b + 1e-30), float(np.max(err / b))
    worst = float(np.max(err / b))
    print(f"{pattern} M={M} K={K} N={N}: worst |delta| / sum|x w| = {worst:.3e} (per-element residual {rel:.3e}, contract {TOL:.0e})")
    if pattern == "same_sign":
        assert worst >= 0.25 * rel, (worst, rel)  # the coherent residual really shows: the split is what is being measured
        assert worst <= 2.0 ** -17 + 4e-6, worst   # ... and nothing beyond it but f32 summation rounding over K same-sign terms (measured 8.3e-6 with the K split's fan-in at N = 12288)
