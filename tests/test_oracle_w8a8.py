"""Rows a4 / a5 / a6 / a7 of SURVEY §8 — the reference's W8A8 CPU plumbing (BASELINE configs[0]):
`prepareTransposed` (src/quant.zig:274-317), `quantizeInput` (:320-341), `gemvRange` (:358-440) and
`GemvPool.dispatch` (:24-198), restated in oracle/zgml_oracle.c. The reference pins these by tolerance
only (src/quant.zig:1133-1210: quantised matmul vs float 0.1, gemv vs matmul 0.15, gemv vs float 0.15 on the
data below); the structural properties (truncation rounding, zero block -> scale 1, blocks ascending,
int32-exact dot, N-split invariance) are pinned here as exact statements about the restatement.
CPU only (no GPU marker): this path never touches the device."""
import numpy as np
import pytest

# the reference's own fixtures (src/quant.zig:1165-1210, "gemv matches matmul for M=1")
W_4x8 = np.array([
    1.0, 0.5, -0.3, 0.8, -1.0, 0.2, 0.7, -0.4,
    -0.5, 1.0, 0.6, -0.9, 0.3, -0.7, 0.1, 0.5,
    0.25, -0.25, 1.0, 0.4, -0.6, 0.9, -0.2, 0.3,
    0.7, -0.8, 0.15, 1.0, 0.5, -0.3, 0.6, -0.1], dtype=np.float32)
X_4 = np.array([1.0, 2.0, -0.5, 0.3], dtype=np.float32)


def test_gemv_matches_matmul_reference_fixture(oracle):
    """src/quant.zig:1165-1210 verbatim: K=4, N=8, block 4; gemv ~ matmul (0.15) ~ float (0.15)."""
    K, N, bs = 4, 8, 4
    data, scales = oracle.quantize_from_slice(W_4x8, K, N, bs)
    t_data, t_scales = oracle.prepare_transposed(data, scales, K, N, bs)
    mm = oracle.qmatmul_exact(data, scales, X_4, 1, N, K, bs)
    gv = oracle.gemv(t_data, t_scales, X_4, N, K, bs)
    ref = X_4 @ W_4x8.reshape(K, N)
    assert np.abs(mm - gv).max() < 0.15
    assert np.abs(ref - gv).max() < 0.15
    # the restatement is in fact far inside the reference's bound on this data
    assert np.abs(mm - gv).max() < 0.03


def test_quantised_matmul_reference_fixture(oracle):
    """src/quant.zig:1133-1163: K=3, N=2, M=2, block 32 vs the float matmul, 0.1."""
    W = np.array([1.0, 0.5, -0.5, 1.0, 0.25, -0.25], dtype=np.float32)
    X = np.array([1, 2, 3, 4, 5, 6], dtype=np.float32)
    data, scales = oracle.quantize_from_slice(W, 3, 2, 32)
    out = oracle.qmatmul_exact(data, scales, X, 2, 2, 3, 32)
    ref = (X.reshape(2, 3) @ W.reshape(3, 2)).ravel()
    assert np.abs(out - ref).max() < 0.1


def test_quantize_input_truncates_toward_zero_and_zero_block_scale(oracle):
    """quant.zig:333-339 (@intFromFloat after the clamp) and :239-246: scale = max_abs/127, a zero block gets
    scale 1 / inv_scale 0, values are truncated (not rounded) toward zero."""
    x = np.zeros(64, np.float32)
    x[:32] = np.linspace(-1.0, 1.0, 32, dtype=np.float32)  # max_abs = 1 -> inv_scale = 127
    q, s = oracle.quantize_input(x, 32)
    assert s[0] == np.float32(1.0) / np.float32(127.0) and s[1] == 1.0
    assert (q[32:] == 0).all()
    want = np.trunc(np.clip(x[:32] * np.float32(127.0), -127, 127)).astype(np.int8)
    assert np.array_equal(q[:32], want)
    # truncation, not rounding: 0.996 * 127 = 126.49 -> 126, -0.999 * 127 = -126.87 -> -126
    y = np.array([1.0, 0.996, -0.999] + [0.0] * 29, np.float32)
    q2, _ = oracle.quantize_input(y, 32)
    assert q2[:3].tolist() == [127, 126, -126]


def test_prepare_transposed_layout_and_requantisation(oracle):
    """quant.zig:274-317: t_data is [N, K] row-major, blocks run along K per output column n, each block is
    re-quantised from the dequantised values (scale = max_abs/127, truncation), ragged last block included."""
    rng = np.random.default_rng(0xd3fa)
    K, N, bs = 37, 11, 32  # the shape of the reference's dequant test (quant.zig:1111-1131): ragged K
    W = ((rng.random(K * N, dtype=np.float32) - 0.5) * 2.0).astype(np.float32)
    data, scales = oracle.quantize_from_slice(W, K, N, bs)
    t_data, t_scales = oracle.prepare_transposed(data, scales, K, N, bs)
    deq = oracle.dequantize(data, scales, bs).reshape(K, N)
    bpr = (K + bs - 1) // bs
    assert t_data.size == N * K and t_scales.size == N * bpr
    for n in range(N):
        for b in range(bpr):
            blk = deq[b * bs:min(K, (b + 1) * bs), n]
            mx = np.abs(blk).max()
            sc = np.float32(mx) / np.float32(127.0) if mx > 0 else np.float32(1.0)
            inv = np.float32(127.0) / np.float32(mx) if mx > 0 else np.float32(0.0)
            assert t_scales[n * bpr + b] == sc
            want = np.trunc(np.clip(blk * inv, -127, 127)).astype(np.int8)
            assert np.array_equal(t_data[n * K + b * bs:n * K + b * bs + blk.size], want)


def _gemv_model(t_data, t_scales, q, s, N, K, bs):
    """gemvRange's arithmetic in numpy: int32-exact block dots, acc += f32(dot) * (s_x[b] * s_w[n,b]), blocks ascending."""
    bpr = (K + bs - 1) // bs
    out = np.zeros(N, np.float32)
    for n in range(N):
        acc = np.float32(0)
        for b in range(bpr):
            k0, k1 = b * bs, min(K, (b + 1) * bs)
            dot = int(np.dot(q[k0:k1].astype(np.int32), t_data[n * K + k0:n * K + k1].astype(np.int32)))
            acc = np.float32(acc + np.float32(dot) * np.float32(s[b] * t_scales[n * bpr + b]))
        out[n] = acc
    return out


@pytest.mark.parametrize("K,N", [(64, 32), (100, 24), (256, 96)])
def test_gemv_range_is_bit_exact_with_its_definition(oracle, K, N):
    rng = np.random.default_rng(K * 131 + N)
    W = rng.standard_normal(K * N).astype(np.float32)
    x = rng.standard_normal(K).astype(np.float32)
    data, scales = oracle.quantize_from_slice(W, K, N, 32)
    t_data, t_scales = oracle.prepare_transposed(data, scales, K, N, 32)
    q, s = oracle.quantize_input(x, 32)
    got = oracle.gemv(t_data, t_scales, x, N, K, 32)
    assert np.array_equal(got, _gemv_model(t_data, t_scales, q, s, N, K, 32))
    # and W8A8 stays within the reference's 0.15-style bound of the exact-dequant path, scaled to this data
    exact = oracle.qmatmul_exact(data, scales, x, 1, N, K, 32)
    assert np.abs(got - exact).max() < 0.02 * np.abs(x).max() * np.abs(W).max() * np.sqrt(K)


def test_gemv_pool_dispatch_is_n_split_invariant(oracle):
    """GemvPool.dispatch (quant.zig:139-198): outputs are split over min(workers, N*K / 2^20, 16) threads in chunks
    rounded up to 4; every split gives the single-thread result bit for bit. 4096 x 1024 is the smallest power-of-two
    shape that activates 4 workers (useful = N*K / 2^20)."""
    K, N = 1024, 4096
    rng = np.random.default_rng(7)
    data = rng.integers(-127, 128, K * N, dtype=np.int8)
    scales = (rng.random(K * N // 32, dtype=np.float32) * 0.01 + 0.001).astype(np.float32)
    x = rng.standard_normal(K).astype(np.float32)
    t_data, t_scales = oracle.prepare_transposed(data, scales, K, N, 32)
    one = oracle.gemv(t_data, t_scales, x, N, K, 32)
    for workers in (1, 3, 4, 16, 64):  # 64 is clamped to GemvPool.max_workers = 16
        assert np.array_equal(oracle.gemv(t_data, t_scales, x, N, K, 32, workers=workers), one), workers


def test_gemv_pool_small_problem_runs_inline(oracle):
    """N*K below min_work_per_thread (2^20, quant.zig:150) -> n_active = 1 -> the caller's thread (same numbers)."""
    K, N = 64, 48
    rng = np.random.default_rng(3)
    W = rng.standard_normal(K * N).astype(np.float32)
    x = rng.standard_normal(K).astype(np.float32)
    data, scales = oracle.quantize_from_slice(W, K, N, 32)
    t_data, t_scales = oracle.prepare_transposed(data, scales, K, N, 32)
    assert np.array_equal(oracle.gemv(t_data, t_scales, x, N, K, 32, workers=8), oracle.gemv(t_data, t_scales, x, N, K, 32))


@pytest.mark.parametrize("K,N", [(64, 8), (256, 40), (4096, 67), (96, 13)])
def test_gemv_range_vnni_arm_is_bit_identical_to_the_scalar_loop(oracle, K, N):
    """The x86 arm of quant.zig:358-440 (AVX-512 VNNI vpdpbusd with the sign moved from x to w, 8 rows per step, horizontal-add
    tree, the scalar loop's f32 combine): exact int32 dots and the same f32 order -> the same bits, including the row tail
    (N % 8) and extreme quantised values (+-127 on both sides). Skipped where the host has no AVX-512 VNNI."""
    from oracle import oracle as O
    O.set_vnni(1)
    if not O.gemv_uses_vnni():
        O.set_vnni(-1)
        pytest.skip("host has no AVX-512 VNNI")
    try:
        rng = np.random.default_rng(K * 7 + N)
        data = rng.integers(-127, 128, K * N, dtype=np.int8)
        data[:64] = 127
        data[64:128] = -127
        scales = (rng.random(K * N // 32, dtype=np.float32) * 0.01 + 0.001).astype(np.float32)
        x = rng.standard_normal(K).astype(np.float32) * 3
        x[:32] = np.abs(x[:32]).max()  # a block whose values all quantise to +127
        t_data, t_scales = oracle.prepare_transposed(data, scales, K, N, 32)
        fast = oracle.gemv(t_data, t_scales, x, N, K, 32)
        O.set_vnni(0)
        slow = oracle.gemv(t_data, t_scales, x, N, K, 32)
        assert not O.gemv_uses_vnni()
        assert np.array_equal(fast, slow)
        q, s = oracle.quantize_input(x, 32)
        assert np.array_equal(fast, _gemv_model(t_data, t_scales, q, s, N, K, 32))
    finally:
        O.set_vnni(-1)


def test_gemv_range_vnni_arm_with_minus_128_bytes(oracle):
    """zo_gemv_range is exported and takes arbitrary int8 bytes — raw GGUF Q8_0 blocks can hold w = -128, a caller's
    own quantiser x = -128. vpdpbusd multiplies unsigned by signed bytes: the arm now takes |w| as the UNSIGNED operand (128 fits)
    and x * sgn(w) as the signed one, and a block whose x holds -128 takes the scalar integers. Every combination of extreme
    bytes gives the scalar loop's bits (before the fix _mm256_sign_epi8(w, x) kept w = -128 at -128 for x < 0: sign flipped)."""
    import ctypes as C
    from oracle import oracle as O
    O.set_vnni(1)
    if not O.gemv_uses_vnni():
        O.set_vnni(-1)
        pytest.skip("host has no AVX-512 VNNI")
    try:
        K, N = 128, 16
        rng = np.random.default_rng(128)
        t_d = rng.integers(-128, 128, N * K, dtype=np.int8)
        t_d[:K] = -128                      # a whole row of -128
        t_d[K:K + 32] = 127
        q = rng.integers(-127, 128, K, dtype=np.int8)
        q[:16] = -127                       # x < 0 against w = -128: the case that flipped
        q[40] = -128                        # block 1 holds x = -128: scalar integers for that block
        t_s = (rng.random(N * (K // 32), dtype=np.float32) * 0.01 + 0.001).astype(np.float32)
        s = (rng.random(K // 32, dtype=np.float32) + 0.5).astype(np.float32)
        lib = O.load()

        def run():
            dst = np.zeros(N, np.float32)
            lib.zo_gemv_range(t_d.ctypes.data, t_s.ctypes.data, q.ctypes.data, s.ctypes.data, dst.ctypes.data, 0, N, K, 32)
            return dst
        fast = run()
        O.set_vnni(0)
        slow = run()
        assert np.array_equal(fast, slow)
        assert np.array_equal(fast, _gemv_model(t_d, t_s, q, s, N, K, 32))
        assert fast[0] > 0 or True  # (row 0: -128 x mostly negative x: the sign the old arm got wrong is pinned by the equality above)
    finally:
        O.set_vnni(-1)
