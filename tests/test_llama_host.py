"""CPU tests of the host side: the decode DeviceProgram has the reference's op stream
(SURVEY Appendix A), per-step patching follows llama_smollm_bench.zig doStep, and the row-sharded
program reproduces the unsharded logits (single process here; 2-process gloo in
test_sharded_gloo.py)."""
import ctypes as C

import numpy as np
import pytest

from zgml_amd import capi, llama
from tests.oracle_executor import OracleExecutor
from zgml_amd.sharded import ShardedDecoder

KIND = capi.DOP


def op_kinds(model):
    prog = model.program
    return [prog.ops[i].kind for i in range(prog.n_ops)]


def test_smollm_program_shape():
    cfg = llama.preset("smollm-135m")
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    kinds = op_kinds(m)
    L, H, KV = cfg.n_layers, cfg.n_heads, cfg.n_kv_heads
    count = lambda k: sum(1 for x in kinds if x == KIND[k])
    assert count("qmatmul") == 7 * L
    assert count("matmul") == 1                      # tied LM head stays dense f32 (F10)
    assert count("rmsnorm") == 2 * L + 1
    assert count("repeat") == 3 * L + 1
    assert count("rope") == (H + KV) * L
    assert count("attention") == H * L
    assert count("slice_assign") == (2 * KV + H) * L
    assert count("fused_elementwise") == 2 * L
    assert len(kinds) == 1654                        # docs/perf-targets.md:83 reports 1,654 ops
    qb, n = m.quant_bytes()
    assert n == 106_168_320 and qb == n // 32 * 18   # 59.7 MB of Q4_0
    m.close()


def test_unfused_lowering_when_backend_lacks_fused_elementwise():
    cfg = llama.preset("tiny")
    m = llama.Model(cfg, fused_elementwise=False)
    kinds = op_kinds(m)
    assert KIND["fused_elementwise"] not in kinds
    assert kinds.count(KIND["elementwise"]) == (3 + 5 + 2) * cfg.n_layers + 1  # norm muls, silu chain(5)+act, adds
    m.close()


def test_step_patching_matches_bench_dostep(oracle):
    cfg = llama.preset("tiny")
    m = llama.Model(cfg)
    m.patch(7, 5)
    prog = m.program
    n = C.c_uint64()
    ins = m.lib.zh_model_step_inputs(m.ptr, C.byref(n))
    assert n.value == 2 + cfg.n_layers
    tok = np.ctypeslib.as_array(C.cast(ins[0].host_ptr, C.POINTER(C.c_float)), shape=(cfg.d_model,))
    mask = np.ctypeslib.as_array(C.cast(ins[1].host_ptr, C.POINTER(C.c_float)), shape=(cfg.max_seq_len,))
    rope = np.ctypeslib.as_array(C.cast(ins[2].host_ptr, C.POINTER(C.c_float)), shape=(2 * cfg.d_head,))
    assert np.array_equal(tok, m.token_embed()[7])
    assert np.all(mask[:6] == 0) and np.all(np.isneginf(mask[6:]))
    cos, sin = m.rope_tables()
    ocos, osin = oracle.rope_tables(cfg.d_head, cfg.max_seq_len, cfg.rope_base)
    np.testing.assert_allclose(cos, ocos, atol=1e-6)
    np.testing.assert_allclose(sin, osin, atol=1e-6)
    assert np.array_equal(rope[:cfg.d_head], cos[5]) and np.array_equal(rope[cfg.d_head:], sin[5])
    for i in range(prog.n_ops):
        op = prog.ops[i]
        if op.kind == KIND["slice_assign"]:
            sa = op.u.slice_assign
            assert sa.dst_offset == sa.dst_base_offset + 5 * sa.patch_stride or sa.patch_stride == 0
        if op.kind == KIND["attention"]:
            assert op.u.attention.seq_kv == 6
    m.close()


def run_decode(m, n_steps, first=3):
    s = llama.Session(m, __import__("oracle.oracle", fromlist=["x"]).backend_fns())
    tok, out = first, []
    for pos in range(n_steps):
        tok, logits = s.step(tok, pos)
        out.append(logits)
    s.close()
    return out


@pytest.mark.parametrize("kind", [llama.Q4_0, llama.Q8_0])
def test_sharded_program_reproduces_unsharded_logits(oracle, kind):
    """world_size 2 and 4 emulated in one process: every rank's program runs on the oracle and
    the all-gathers are done by hand between the op ranges."""
    cfg = llama.preset("tiny")
    cfg.tied_lm_head = 0
    cfg.n_kv_heads = 4
    full = llama.Model(cfg, kind)
    ref = run_decode(full, 4)
    full.close()
    for ws in (2, 4):
        models, exs = [], []
        for r in range(ws):
            c = llama.preset("tiny")
            c.tied_lm_head, c.n_kv_heads, c.shard_rank, c.shard_world = 0, 4, r, ws
            models.append(llama.Model(c, kind))
            exs.append(OracleExecutor(models[-1]))
        points = models[0].gather_points()
        assert len(points) == 4 * cfg.n_layers + 1
        tok = 3
        for pos in range(4):
            for m, ex in zip(models, exs):
                m.patch(tok, pos)
                ex.refresh()
                ex.upload()
            prev = 0
            for gp in points + [None]:
                end = gp.op_end if gp else exs[0].n_ops
                for ex in exs:
                    ex.run_ops(prev, end - prev)
                if gp:
                    n = gp.len_per_rank
                    views = [ex.tensor(gp.buf, gp.offset, ws * n) for ex in exs]
                    for r in range(ws):
                        for v in views:
                            v[r * n:(r + 1) * n] = views[r][r * n:(r + 1) * n].clone()
                prev = end
            logits = [ex.download_logits() for ex in exs]
            for lg in logits:
                np.testing.assert_allclose(lg, ref[pos], atol=2e-5 * np.abs(ref[pos]).max(), rtol=0)
            tok = int(np.argmax(ref[pos]))
        for ex, m in zip(exs, models):
            ex.close()
            m.close()


def test_invalid_shard_specs_rejected():
    c = llama.preset("tiny")
    c.shard_world = 2  # tied LM head cannot shard
    with pytest.raises(ValueError):
        llama.Model(c)
    c = llama.preset("smollm-135m")
    c.tied_lm_head, c.shard_world = 0, 2  # 3 kv heads do not split in two: replicas only
    with pytest.raises(ValueError):
        llama.Model(c)


def test_prefill_plan_equals_sequential_on_oracle():
    """token_len = N plan vs N single-token steps (src/llama_inference.zig:983-1034, tol 1e-4)."""
    from oracle import oracle as O
    cfg = llama.preset("tiny")
    toks = [3, 17, 200, 5, 9, 44, 1]
    m1 = llama.Model(cfg, llama.Q4_0)
    s1 = llama.Session(m1, O.backend_fns())
    for p, t in enumerate(toks):
        nxt, lg = s1.step(t, p)
    mN = llama.Model(cfg, llama.Q4_0, token_len=len(toks))
    assert mN.program.n_ops == m1.program.n_ops  # same op stream, wider shapes
    sN = llama.Session(mN, O.backend_fns())
    nxt2, lg2 = sN.prefill(toks, 0)
    assert nxt2 == nxt
    np.testing.assert_allclose(lg2, lg, atol=1e-4)
    s1.close(), sN.close(), m1.close(), mN.close()


def test_dense_model_is_the_dequantised_q4_model():
    """weight_kind F32_DENSE lowers every projection to a `matmul` over the dequantised [K, N]
    weights: same op count, logits equal to the Q4_0 model's up to f32 rounding; with the f16
    promotion (wgpu.zig:1071-1104) restated in the oracle they move by ~1e-4 of the range."""
    from oracle import oracle as O
    cfg = llama.preset("tiny")
    mq, md = llama.Model(cfg, llama.Q4_0), llama.Model(cfg, llama.F32_DENSE)
    assert mq.program.n_ops == md.program.n_ops
    assert md.program.n_qweights == 0
    sq, sd = llama.Session(mq, O.backend_fns()), llama.Session(md, O.backend_fns())
    tq, lq = sq.step(3, 0)
    td, ld = sd.step(3, 0)
    assert tq == td
    np.testing.assert_allclose(ld, lq, atol=1e-5)
    O.set_f16_dense(True)
    try:
        sh = llama.Session(md, O.backend_fns())
        th, lh = sh.step(3, 0)
    finally:
        O.set_f16_dense(False)
    assert th == td and 0 < np.abs(lh - ld).max() < 2e-3
    for s in (sq, sd, sh):
        s.close()
    mq.close(), md.close()


def test_quantised_kv_program_tracks_the_f32_cache_program():
    """kv_quant_block = 32 swaps slice_assign / attention for kvq_store / attention_kvq (same op count per layer, one
    int8 cache buffer per kv head) and, on the oracle, decodes within the int8 cache's accuracy of the f32-cache program
    (the reference bounds attentionQuantized vs float attention at 0.05-0.1, src/quant.zig:1133-1210)."""
    from oracle import oracle as O
    cfg = llama.preset("tiny")
    m_f = llama.Model(cfg, llama.Q4_0)
    cq = llama.preset("tiny")
    cq.kv_quant_block = 32
    m_q = llama.Model(cq, llama.Q4_0)
    kinds_f, kinds_q = op_kinds(m_f), op_kinds(m_q)
    assert len(kinds_f) == len(kinds_q)
    assert kinds_q.count(KIND["kvq_store"]) == 2 * cfg.n_kv_heads * cfg.n_layers
    assert kinds_q.count(KIND["attention_kvq"]) == cfg.n_heads * cfg.n_layers
    assert KIND["attention"] not in kinds_q
    s_f, s_q = llama.Session(m_f, O.backend_fns()), llama.Session(m_q, O.backend_fns())
    tok = 3
    for pos in range(8):
        t_f, l_f = s_f.step(tok, pos)
        t_q, l_q = s_q.step(tok, pos)
        assert np.abs(l_q - l_f).max() <= 0.05 * np.abs(l_f).max(), pos
        tok = t_f
    s_f.close(), s_q.close(), m_f.close(), m_q.close()


def test_greedy_pick_is_the_reference_loops_index():
    """The session's argmax (two vector passes where the host has AVX2) against the reference's compare-and-keep loop
    (src/nn.zig:122-138: first maximum wins, a NaN never wins, a NaN in front keeps index 0): ties, maxima in every lane and in
    the scalar tail, -inf rows, NaNs, +inf, short vectors."""
    lib = llama.load_host()
    rng = np.random.default_rng(5)

    def both(v):
        v = np.ascontiguousarray(v, np.float32)
        return lib.zh_argmax(v.ctypes.data, v.size), lib.zh_argmax_reference(v.ctypes.data, v.size)

    for n in (1, 7, 63, 64, 65, 127, 1000, 49152, 32000):
        for _ in range(6):
            v = rng.standard_normal(n).astype(np.float32)
            a, b = both(v)
            assert a == b == int(np.argmax(v)), n
            k = int(rng.integers(0, n))
            v[k] = v.max()  # a tie: the first one wins
            a, b = both(v)
            assert a == b, n
        for pos in {0, n // 2, n - 1, max(0, n - 3)}:  # the maximum in the head, the body and the scalar tail
            v = np.full(n, -1.0, np.float32)
            v[pos] = 2.0
            assert both(v) == (pos, pos)
        assert both(np.full(n, -np.inf, np.float32)) == (0, 0)
        v = rng.standard_normal(n).astype(np.float32)
        v[n // 3] = np.inf
        assert both(v) == (n // 3, n // 3)
        for pos in {0, n // 2, n - 1}:  # NaNs: never the winner; in front they freeze index 0 (every later compare is false)
            v = rng.standard_normal(n).astype(np.float32)
            v[pos] = np.nan
            a, b = both(v)
            assert a == b, (n, pos)
