"""Quantised KV cache as device ops (extension kinds kvq_store / attention_kvq):
HIP against the oracle's restatement of QuantizedKVCache.storeColumn and attentionQuantized
(src/quant.zig:645-1091).

Bars: the int8 cache contents and scales are BIT-EXACT (same quantizeInput arithmetic); attention
output within 1e-5 of the oracle (same dequantised operands, only the softmax streaming order differs)."""
import numpy as np
import pytest

from zgml_amd import DeviceOp, DeviceProgram, ProgramIO

pytestmark = pytest.mark.gpu
f32 = np.float32


def cache_elems(n_cols, dh, bs):
    return n_cols * dh // 4 + n_cols * (dh // bs)


def build(dh, bs, n_cols, n_store, seq_q, seq_kv, k_start, rng, masked=False):
    # buffers: 0 kcache, 1 vcache, 2 k source, 3 v source, 4 q, 5 dst, 6 mask
    ks = ((rng.random((n_store, dh)) - 0.5) * 2).astype(f32)
    vs = ((rng.random((n_store, dh)) - 0.5) * 2).astype(f32)
    ks[min(1, n_store - 1), :bs] = 0.0  # an all-zero block: scale 1, quants 0
    q = ((rng.random((seq_q, dh)) - 0.5) * 2).astype(f32)
    ops = []
    for c in range(n_store):
        ops.append(DeviceOp.kvq_store(0, 2, dh, bs, n_cols, c * dh, k_start, k_start + c, 1))
        ops.append(DeviceOp.kvq_store(1, 3, dh, bs, n_cols, c * dh, k_start, k_start + c, 1))
    mask = np.zeros((seq_q, seq_kv), f32)
    if masked:
        for j in range(seq_q):
            mask[j, max(1, seq_kv - seq_q + j + 1):] = -np.inf  # causal-ish
        mask[0, 0] = -np.inf
    ops.append(DeviceOp.attention_kvq(5, 4, 0, 1, 6, masked, dh, seq_q, seq_kv, float(1 / np.sqrt(dh)), bs, n_cols,
                                      k_start, k_start, 0, dh, 0, dh, 0, 1, seq_kv))
    ce = cache_elems(n_cols, dh, bs)
    prog = DeviceProgram(ops=ops, buffer_sizes=[ce, ce, n_store * dh, n_store * dh, seq_q * dh, seq_q * dh, seq_q * seq_kv],
                         initial_uploads=[ProgramIO(2, ks.ravel()), ProgramIO(3, vs.ravel()), ProgramIO(4, q.ravel()),
                                          ProgramIO(6, mask.ravel())])
    return prog, ce


@pytest.mark.parametrize("dh,bs", [(64, 32), (128, 32), (32, 16), (64, 64)])
@pytest.mark.parametrize("seq_q,seq_kv,masked", [(1, 8, False), (1, 37, True), (3, 21, True), (1, 1, False)])
def test_kvq_store_and_attention_match_oracle(hip_backend, oracle, dh, bs, seq_q, seq_kv, masked):
    rng = np.random.default_rng(dh * 3 + bs + seq_kv)
    n_cols, k_start = 48, 5
    prog, ce = build(dh, bs, n_cols, seq_kv, seq_q, seq_kv, k_start, rng, masked)
    assert hip_backend.supportsProgram(prog)

    def run(be):
        h = be.compileProgram(prog)
        out, kc, vc = np.zeros(seq_q * dh, f32), np.zeros(ce, f32), np.zeros(ce, f32)
        be.executeProgram(h, [], [ProgramIO(5, out), ProgramIO(0, kc), ProgramIO(1, vc)])
        be.freeProgram(h)
        return out, kc, vc
    want, kc_w, vc_w = run(oracle.OracleBackend())
    got, kc_g, vc_g = run(hip_backend)
    assert not hip_backend.last_error(), hip_backend.last_error()
    assert np.array_equal(kc_g.view(np.uint32), kc_w.view(np.uint32)) and np.array_equal(vc_g.view(np.uint32), vc_w.view(np.uint32))
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, atol=1e-5, rtol=0)


def test_kvq_dynamic_column_refresh_and_bounds(hip_backend, oracle):
    """The store column and seq_kv are dynamic fields (refresh_program); a column outside the cache is
    dropped instead of written (the reference would assert)."""
    dh, bs, n_cols = 64, 32, 8
    rng = np.random.default_rng(5)
    src = ((rng.random((n_cols + 1, dh)) - 0.5) * 2).astype(f32)
    ce = cache_elems(n_cols, dh, bs)
    q = ((rng.random(dh) - 0.5) * 2).astype(f32)

    def ops_for(pos):
        return [DeviceOp.kvq_store(0, 2, dh, bs, n_cols, 0, 0, pos, 1), DeviceOp.kvq_store(1, 2, dh, bs, n_cols, 0, 0, pos, 1),
                DeviceOp.attention_kvq(4, 3, 0, 1, 3, False, dh, 1, min(pos + 1, n_cols), 0.125, bs, n_cols, 0, 0, 0, dh, 0, dh)]
    prog = DeviceProgram(ops=ops_for(0), buffer_sizes=[ce, ce, dh, dh, dh], initial_uploads=[ProgramIO(3, q)])
    outs = []
    for be in (oracle.OracleBackend(), hip_backend):
        h = be.compileProgram(prog)
        res = []
        for pos in range(n_cols):
            be.refreshProgram(h, ops_for(pos))
            out = np.zeros(dh, f32)
            be.executeProgram(h, [ProgramIO(2, src[pos])], [ProgramIO(4, out)])
            res.append(out)
        kc = np.zeros(ce, f32)
        be.executeProgram(h, [ProgramIO(2, src[pos])], [ProgramIO(0, kc)])
        res.append(kc)
        if be is hip_backend:  # out-of-range column: nothing is written
            be.refreshProgram(h, ops_for(n_cols))
            kc2 = np.zeros(ce, f32)
            be.executeProgram(h, [ProgramIO(2, src[n_cols])], [ProgramIO(0, kc2)])
            assert np.array_equal(kc2.view(np.uint32), kc.view(np.uint32))
        be.freeProgram(h)
        outs.append(res)
    for a, b in zip(outs[0][:-1], outs[1][:-1]):
        np.testing.assert_allclose(b, a, atol=1e-5, rtol=0)
    assert np.array_equal(outs[0][-1].view(np.uint32), outs[1][-1].view(np.uint32))


@pytest.mark.parametrize("dh,n_pos", [(128, 560), (256, 300)])
def test_int8_kv_decode_attention_wide_heads_many_rounds(hip_backend, oracle, dh, n_pos):
    """The 16-dims-per-lane key loop of the fused decode attention (attention_decode.h, d_head >= 128) with ONE workgroup per head
    (split off): beyond 8 x 16 x 4 = 512 keys at d_head 128 (4 x 16 x 4 = 256 at d_head 256) the loop runs a second round on rows
    re-issued into the same registers. Against the oracle over a cache that fills from position 0 (1e-3 of the logit range, the int8
    caches' bound), the new column met in every position's last slot; then the same with the split on (several workgroups per head)."""
    from zgml_amd import capi, llama
    cfg = llama.preset("tiny", n_pos + 8)
    cfg.d_model, cfg.n_heads, cfg.n_kv_heads, cfg.d_ff, cfg.n_layers, cfg.vocab_size = 2 * dh, 2, 1, 256, 1, 256
    cfg.kv_quant_block = 32
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    s_ref = llama.Session(m, oracle.backend_fns())
    hip_backend.set_option(capi.OPT_ATTN_SPLIT_MIN_KEYS, 0)
    try:
        s_one = llama.Session(m, llama.hip_backend_fns(hip_backend))
        hip_backend.set_option(capi.OPT_ATTN_SPLIT_MIN_KEYS, 64)
        s_split = llama.Session(m, llama.hip_backend_fns(hip_backend))
        tok = 3
        for pos in range(n_pos):
            t_ref, l_ref = s_ref.step(tok, pos)
            t_a, l_a = s_one.step(tok, pos)
            t_b, l_b = s_split.step(tok, pos)
            assert not hip_backend.last_error(), hip_backend.last_error()
            scale = np.abs(l_ref).max()
            assert np.abs(l_a - l_ref).max() <= 1e-3 * scale, pos
            assert np.abs(l_b - l_ref).max() <= 1e-3 * scale, pos
            tok = t_ref
        s_one.close(), s_split.close()
    finally:
        hip_backend.set_option(capi.OPT_ATTN_SPLIT_MIN_KEYS, -1)
    s_ref.close(), m.close()
