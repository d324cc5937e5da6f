"""The reference's W8A8 mat-vec arm on the device (zgml_amd/csrc/w8a8.hip, ZGML_HIP_OPT_W8A8) — BIT-EXACT against the oracle's
restatement of quantizeInput / prepareTransposed / gemvRange (src/quant.zig:560-640, 320-440; taken by the reference's CPU executor
for M = 1, src/backend/reference.zig:512-528). Integers are exact and every f32 operation of the reference is done once with the
same operands, so the comparison is array equality, not a tolerance. Also: the arm is NOT the default (exact dequantise-then-dot)
arithmetic, a decode program under the option tracks the oracle's W8A8 decode, and shapes the arm does not take keep the exact path."""
import numpy as np
import pytest

from zgml_amd import capi, llama
from zgml_amd import DeviceOp, DeviceProgram, ProgramIO, QuantizedWeightUpload

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture
def w8a8(hip_backend):
    hip_backend.set_option(capi.OPT_W8A8, 1)
    yield hip_backend
    hip_backend.set_option(capi.OPT_W8A8, 0)


def _run(backend, oracle, data, scales, x, N, K, in_off=0, dst_off=0):
    xin = np.zeros(in_off + K + 3, f32)
    xin[in_off:in_off + K] = x
    dst_len = dst_off + N + 2
    prog = DeviceProgram(ops=[DeviceOp.qmatmul(1, 0, 0, 1, N, K, in_off, 0, dst_off, 0)], buffer_sizes=[xin.size, dst_len],
                         initial_uploads=[ProgramIO(0, xin), ProgramIO(1, np.full(dst_len, -7, f32))],
                         qweights=[QuantizedWeightUpload(data, scales, K, N, 32)])
    return oracle.run_program(prog, 1, dst_len, backend=backend)


@pytest.mark.parametrize("K,N", [(64, 16), (576, 576), (576, 192), (1536, 576), (4096, 4096), (4096, 11008), (11008, 4096), (16384, 32)])
@pytest.mark.parametrize("kind", ["q8", "q4"])
def test_w8a8_matvec_is_bit_identical_to_the_reference_arm(w8a8, oracle, K, N, kind):
    rng = np.random.default_rng(0x8A8 + K + 3 * N)
    x = (rng.standard_normal(K) * rng.choice([0.01, 1.0, 30.0], K)).astype(f32)
    if K >= 576:
        x[64:96] = 0.0  # a zero block: scale 1, inv 0 (quant.zig:239-246)
        x[7] = np.float32(3.4e38)  # the block's max: everything else truncates to 0
    if kind == "q4":  # Q4_0-valued weights (int8 in -8..7, f16-exact scales) — the arm re-quantises them per (column, 32 k)
        data = rng.integers(-8, 8, K * N).astype(np.int8)
        scales = (rng.random(K * N // 32).astype(np.float16) * 0.05 + 0.001).astype(f32)
    else:
        data, scales = oracle.quantize_from_slice((rng.standard_normal(K * N) * 0.05).astype(f32), K, N, 32)
    t_data, t_scales = oracle.prepare_transposed(data, scales, K, N, 32)
    want = oracle.gemv(t_data, t_scales, x, N, K, 32)
    got = _run(w8a8, oracle, data, scales, x, N, K)
    assert not w8a8.last_error(), w8a8.last_error()
    assert np.array_equal(got[:N].view(np.uint32), want.view(np.uint32)), np.abs(got[:N] - want).max()
    assert np.all(got[N:] == -7)
    if K == 576 and N == 576:  # offsets: an input that is not 16-byte aligned, a shifted destination
        got2 = _run(w8a8, oracle, data, scales, x, N, K, in_off=3, dst_off=5)
        assert np.array_equal(got2[5:5 + N].view(np.uint32), want.view(np.uint32)) and np.all(got2[:5] == -7)


def test_w8a8_is_not_the_exact_arithmetic_and_odd_shapes_keep_the_exact_path(w8a8, hip_backend, oracle):
    rng = np.random.default_rng(5)
    K, N = 576, 192
    x = rng.standard_normal(K).astype(f32)
    data, scales = oracle.quantize_from_slice((rng.standard_normal(K * N) * 0.05).astype(f32), K, N, 32)
    exact = oracle.qmatmul_exact(data, scales, x, 1, N, K, 32)
    got = _run(w8a8, oracle, data, scales, x, N, K)[:N]
    err = np.abs(got - exact).max()
    assert 0 < err < 0.15 * np.abs(exact).max()  # the reference's own bound between the two arms (quant.zig:1165-1210), and not zero
    # K % 64 != 0: the arm does not apply, the exact mat-vec runs (contract tolerance 2e-5 * sum |x w|)
    K2, N2 = 96, 32
    x2 = rng.standard_normal(K2).astype(f32)
    d2, s2 = oracle.quantize_from_slice((rng.standard_normal(K2 * N2) * 0.05).astype(f32), K2, N2, 32)
    got2 = _run(w8a8, oracle, d2, s2, x2, N2, K2)[:N2]
    np.testing.assert_allclose(got2, oracle.qmatmul_exact(d2, s2, x2, 1, N2, K2, 32), rtol=0, atol=2e-5 * np.abs(x2).sum() * 0.2)


def test_decode_under_w8a8_tracks_the_oracles_w8a8_decode(w8a8, oracle):
    """A whole decode program with every M = 1 qmatmul on the W8A8 arm against the oracle running the same arm (its GemvPool path,
    the CPU baseline's variant B3): the mat-vecs are bit-identical, the ops around them differ by f32 reassociation only."""
    cfg = llama.preset("tiny", 64)
    cfg.d_model, cfg.n_heads, cfg.n_kv_heads, cfg.d_ff = 128, 2, 1, 256  # (K % 64 == 0 for every projection)
    m = llama.Model(cfg, llama.Q4_0, threads=4)
    s_hip = llama.Session(m, llama.hip_backend_fns(w8a8))
    plan = w8a8.planText(s_hip.handle)
    oracle.set_w8a8(True)
    try:
        s_ref = llama.Session(m, oracle.backend_fns())
        tok = 3
        for pos in range(8):
            t_ref, l_ref = s_ref.step(tok, pos)
            t_hip, l_hip = s_hip.step(tok, pos)
            assert not w8a8.last_error(), w8a8.last_error()
            # quantizeInput TRUNCATES x * 127 / max: a last-bit difference in a mat-vec's input (the norms and ropes in front of it
            # sum in another order on the device) moves single int8 activations by one unit, i.e. by 1/127 of the block's maximum —
            # the same effect, and the same bound, as the int8 KV caches' storeColumn (tests/test_hip_longctx_filled.py)
            scale = np.abs(l_ref).max()
            assert np.abs(l_hip - l_ref).max() <= 3e-3 * scale, pos
            top2 = np.sort(l_ref)[-2:]
            if top2[1] - top2[0] > 1e-2 * scale:
                assert t_hip == t_ref, pos
            tok = t_ref
        s_ref.close()
    finally:
        oracle.set_w8a8(False)
    assert plan.count("kind 2 ops 1 ") >= 7 * cfg.n_layers, plan  # every projection is a launch of its own (kind 2 = qmatmul), nothing fused around it
    s_hip.close(), m.close()
