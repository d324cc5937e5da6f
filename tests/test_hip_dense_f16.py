"""Dense matmul with f16-promoted weights (BASELINE config 5's "fp16 dense (MFMA matmul)" variant):
HIP (opt-in ZGML_HIP_OPT_F16_DENSE_WEIGHTS) against the oracle's restatement of the reference's
promotion (src/backend/wgpu.zig:1071-1104) and f16 kernels (src/backend/metal.zig:680-760).

Tolerance: the f16 x f16 products are exact in f32 and both sides round A and B identically
(A only for M > 1), so only the summation order differs: |delta| <= 2e-5 * sum_k |a_k * b_kn|."""
import numpy as np
import pytest

from zgml_amd import DeviceOp, DeviceProgram, MatMulGeometry, ProgramIO, capi

pytestmark = pytest.mark.gpu
f32 = np.float32
TOL = 2e-5


@pytest.fixture()
def f16_mode(hip_backend, oracle):
    hip_backend.set_option(capi.OPT_F16_DENSE_WEIGHTS, 1)
    oracle.set_f16_dense(True)
    yield
    hip_backend.set_option(capi.OPT_F16_DENSE_WEIGHTS, 0)
    oracle.set_f16_dense(False)


def matmul_program(a, b, M, K, N, a_off=0, a_rs=None, b_off=0, b_rs=None, b_cs=1, dst_off=0, dst_rs=None, extra_ops=()):
    a_rs, b_rs, dst_rs = a_rs or K, b_rs or N, dst_rs or N
    g = MatMulGeometry(M=M, N=N, K=K, a_row_stride=a_rs, a_col_stride=1, b_row_stride=b_rs, b_col_stride=b_cs,
                       a_offset=a_off, b_offset=b_off, dst_offset=dst_off, dst_row_stride=dst_rs)
    dst_len = dst_off + (M - 1) * dst_rs + N + 1
    return DeviceProgram(ops=[DeviceOp.matmul(2, 0, 1, g), *extra_ops], buffer_sizes=[a.size, b.size, dst_len, b.size],
                         initial_uploads=[ProgramIO(0, a), ProgramIO(1, b), ProgramIO(2, np.full(dst_len, -7, f32))]), dst_len


@pytest.mark.parametrize("M,K,N", [(1, 576, 192), (1, 100, 64), (1, 4096, 512), (5, 576, 192), (16, 100, 64),
                                   (32, 1536, 576), (45, 512, 96), (33, 37, 48), (2, 3, 16)])
def test_promoted_matmul_matches_oracle(hip_backend, oracle, f16_mode, M, K, N):
    rng = np.random.default_rng(M * 131 + K + N)
    a_off, a_rs, b_off, dst_off, dst_rs = 4, K + 4, 7, 1, N + 3
    a = np.full(a_off + M * a_rs, 9, f32)
    for m in range(M):
        a[a_off + m * a_rs:][:K] = rng.standard_normal(K)
    b = np.concatenate([np.full(b_off, 5, f32), (rng.standard_normal(K * N) * 0.05).astype(f32)])
    prog, dst_len = matmul_program(a, b, M, K, N, a_off=a_off, a_rs=a_rs, b_off=b_off, dst_off=dst_off, dst_rs=dst_rs)
    want = oracle.run_program(prog, 2, dst_len)
    got = oracle.run_program(prog, 2, dst_len, backend=hip_backend)
    assert not hip_backend.last_error(), hip_backend.last_error()
    A = np.stack([a[a_off + m * a_rs:][:K] for m in range(M)]).astype(np.float64)
    bound = np.abs(A) @ np.abs(b[b_off:].reshape(K, N).astype(np.float64))
    for m in range(M):
        d = np.abs(got[dst_off + m * dst_rs:][:N] - want[dst_off + m * dst_rs:][:N])
        assert np.all(d <= TOL * bound[m] + 1e-30), (m, float(np.max(d / (bound[m] + 1e-30))))
    assert np.array_equal(got == -7, want == -7)
    # and the promotion is real: the f32 result differs from the f16 one
    oracle.set_f16_dense(False)
    exact = oracle.run_program(prog, 2, dst_len)
    oracle.set_f16_dense(True)
    if K >= 100:
        assert not np.allclose(exact, want, rtol=0, atol=1e-7)


def test_k_contiguous_b_is_promoted_too(hip_backend, oracle, f16_mode):
    """Tied LM head geometry: B stored [N][K] (b_row_stride = 1, b_col_stride = K)."""
    rng = np.random.default_rng(5)
    M, K, N = 1, 576, 512
    a = rng.standard_normal(K).astype(f32)
    b = (rng.standard_normal(N * K) * 0.05).astype(f32)
    prog, dst_len = matmul_program(a, b, M, K, N, b_rs=1, b_cs=K)
    want = oracle.run_program(prog, 2, dst_len)
    got = oracle.run_program(prog, 2, dst_len, backend=hip_backend)
    bound = np.abs(a.astype(np.float64)) @ np.abs(b.reshape(N, K).T.astype(np.float64))
    assert np.all(np.abs(got[:N] - want[:N]) <= TOL * bound)


def test_reference_known_answer(hip_backend, oracle, f16_mode):
    """src/backend/wgpu.zig:1276-1318 'f16 matmul (auto-promoted)': small integers are exact."""
    a = np.array([1, 2, 3, 4, 5, 6], f32)
    b = np.array([7, 8, 9, 10, 11, 12], f32)
    g = MatMulGeometry(M=2, N=2, K=3, a_row_stride=3, a_col_stride=1, b_row_stride=2, b_col_stride=1,
                       a_offset=0, b_offset=0, dst_offset=0, dst_row_stride=2)
    prog = DeviceProgram(ops=[DeviceOp.matmul(2, 0, 1, g)], buffer_sizes=[6, 6, 4],
                         initial_uploads=[ProgramIO(0, a), ProgramIO(1, b)])
    for be in (None, hip_backend):
        assert np.array_equal(oracle.run_program(prog, 2, 4, backend=be), np.array([58, 64, 139, 154], f32))


def test_weight_with_other_readers_is_not_promoted(hip_backend, oracle, f16_mode):
    """HIP keeps only the f16 image of a promoted buffer, so a B operand some other op also reads
    stays f32 (results then equal the unpromoted path bit for bit on the matmul's own order)."""
    rng = np.random.default_rng(6)
    M, K, N = 1, 256, 64
    a = rng.standard_normal(K).astype(f32)
    b = (rng.standard_normal(K * N) * 0.05).astype(f32)
    extra = [DeviceOp.elementwise("neg", 3, 1, 1, K * N)]
    prog, dst_len = matmul_program(a, b, M, K, N, extra_ops=extra)
    got = oracle.run_program(prog, 2, dst_len, backend=hip_backend)
    hip_backend.set_option(capi.OPT_F16_DENSE_WEIGHTS, 0)
    plain = oracle.run_program(prog, 2, dst_len, backend=hip_backend)
    hip_backend.set_option(capi.OPT_F16_DENSE_WEIGHTS, 1)
    assert np.array_equal(got, plain)


def test_io_on_promoted_buffer_fails_loudly(hip_backend, oracle, f16_mode):
    rng = np.random.default_rng(7)
    K, N = 64, 32
    a, b = rng.standard_normal(K).astype(f32), rng.standard_normal(K * N).astype(f32)
    prog, dst_len = matmul_program(a, b, 1, K, N)
    h = hip_backend.compileProgram(prog)
    out = np.zeros(K * N, f32)
    try:
        with pytest.raises(RuntimeError, match="buffer 1"):
            hip_backend.executeProgram(h, [], [ProgramIO(1, out)])
    finally:
        hip_backend._lib.zgml_hip_clear_error(hip_backend.ctx)
        hip_backend.freeProgram(h)


def test_f16_matmul_rows_equal_matvec_full_size(hip_backend, oracle, f16_mode):
    """Size-independent property at the BASELINE shape (4096 x 4096, batch 32): each row of the MFMA
    result equals the M = 1 kernel on that row up to the rounding of A to f16 (M > 1 only; the M = 1
    kernel keeps A in f32): |delta| <= 2^-10 * sum|a*b|. A zero row gives exact zeros."""
    K = N = 4096
    M = 32
    rng = np.random.default_rng(31)
    b = (rng.standard_normal(K * N) * 0.02).astype(f32)
    a = rng.standard_normal((M, K)).astype(f32)
    a[9] = 0.0
    g_m = MatMulGeometry(M=M, N=N, K=K, a_row_stride=K, a_col_stride=1, b_row_stride=N, b_col_stride=1,
                         a_offset=0, b_offset=0, dst_offset=0, dst_row_stride=N)
    g_1 = MatMulGeometry(M=1, N=N, K=K, a_row_stride=K, a_col_stride=1, b_row_stride=N, b_col_stride=1,
                         a_offset=0, b_offset=0, dst_offset=0, dst_row_stride=N)
    prog = DeviceProgram(ops=[DeviceOp.matmul(2, 0, 1, g_m), DeviceOp.matmul(4, 3, 1, g_1)],
                         buffer_sizes=[M * K, K * N, M * N, K, N], initial_uploads=[ProgramIO(1, b)])
    h = hip_backend.compileProgram(prog)
    Y, y = np.zeros(M * N, f32), np.zeros(N, f32)
    absb = np.abs(b.reshape(K, N)).astype(np.float64)
    for m in (0, 9, 31):
        hip_backend.executeProgram(h, [ProgramIO(0, a.ravel()), ProgramIO(3, a[m].copy())], [ProgramIO(2, Y), ProgramIO(4, y)])
        bound = np.abs(a[m]).astype(np.float64) @ absb
        assert np.all(np.abs(Y.reshape(M, N)[m] - y) <= 2.0 ** -10 * bound + 1e-30), m
    assert np.all(Y.reshape(M, N)[9] == 0)
    hip_backend.freeProgram(h)
