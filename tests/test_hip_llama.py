"""Decode parity on the MI355X: the same DeviceProgram (DeviceInference lowering of
forwardCachedMasked) stepped through the HIP C ABI and through the oracle.

Tolerance: logits within 2e-4 of the logit range (f32 reassociation across ~50 ops per layer; the
reference's own integration tests use 1e-4..1e-5 on much smaller models,
src/llama_inference.zig:854-1034); greedy tokens must be identical."""
import ctypes as C

import numpy as np
import pytest

from zgml_amd import capi, llama

pytestmark = pytest.mark.gpu


def decode_both(hip_backend, oracle, cfg, kind, n_steps, fused=True, dead=False, first=3):
    m = llama.Model(cfg, kind, fused_elementwise=fused, include_dead_f32=dead, threads=8)
    s_ref = llama.Session(m, oracle.backend_fns())
    s_hip = llama.Session(m, llama.hip_backend_fns(hip_backend))
    tok, worst = first, 0.0
    for pos in range(n_steps):
        t_ref, l_ref = s_ref.step(tok, pos)
        t_hip, l_hip = s_hip.step(tok, pos)
        err = hip_backend.last_error()
        assert not err, err
        assert np.isfinite(l_hip).all()
        scale = np.abs(l_ref).max()
        worst = max(worst, float(np.abs(l_hip - l_ref).max() / scale))
        assert t_hip == t_ref, (pos, t_hip, t_ref)
        tok = t_ref
    s_ref.close()
    s_hip.close()
    m.close()
    return worst


@pytest.mark.parametrize("kind", [llama.Q4_0, llama.Q8_0])
@pytest.mark.parametrize("fused", [True, False])
def test_tiny_decode_matches_oracle(hip_backend, oracle, kind, fused):
    worst = decode_both(hip_backend, oracle, llama.preset("tiny"), kind, 12, fused=fused, dead=True)
    assert worst < 2e-4, worst


def test_tiny_untied_head_and_graph_off(hip_backend, oracle):
    cfg = llama.preset("tiny")
    cfg.tied_lm_head = 0
    hip_backend.set_option(capi.OPT_GRAPH, 0)
    try:
        assert decode_both(hip_backend, oracle, cfg, llama.Q4_0, 6) < 2e-4
    finally:
        hip_backend.set_option(capi.OPT_GRAPH, 1)


def test_smollm_135m_decode_matches_oracle(hip_backend, oracle):
    """BASELINE config 2 at full size, a few tokens (the CPU oracle needs ~0.1 s per token)."""
    oracle.set_threads(8)
    worst = decode_both(hip_backend, oracle, llama.preset("smollm-135m"), llama.Q4_0, 4, dead=False)
    assert worst < 2e-4, worst


def test_kv_cache_position_independence(hip_backend, oracle):
    """Size-independent property: replaying the same token stream after a reset gives the same
    logits (LlamaInferenceSession.reset replay test, src/llama_inference.zig:854-909)."""
    m = llama.Model(llama.preset("tiny"), llama.Q4_0)
    s = llama.Session(m, llama.hip_backend_fns(hip_backend))
    toks = [3, 17, 200, 5, 9]
    a = [s.step(t, p)[1] for p, t in enumerate(toks)]
    b = [s.step(t, p)[1] for p, t in enumerate(toks)]
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    s.close()
    m.close()


@pytest.mark.parametrize("name,kind", [("tiny", llama.Q4_0), ("tiny", llama.Q8_0), ("smollm-135m", llama.Q4_0)])
def test_resident_decode_equals_vtable_stepping(hip_backend, name, kind):
    """The device-resident loop (on-device embedding/mask/rope/argmax) must produce exactly the
    tokens of doStep through execute_program, and leave the program usable by the vtable path."""
    m = llama.Model(llama.preset(name), kind, threads=8)
    s = llama.Session(m, llama.hip_backend_fns(hip_backend))
    n = 24 if name == "tiny" else 12
    s.resident_setup(hip_backend)
    got = s.resident_decode(5, 0, n)  # first execution of the program is the resident loop
    want, _ = s.decode(5, 0, n)
    assert got.tolist() == want.tolist()
    again, _ = s.decode(5, 0, n)  # vtable path after a resident run
    assert again.tolist() == want.tolist()
    half = s.resident_decode(int(want[n // 2 - 1]), n // 2, n - n // 2)  # resume mid-stream on the warm KV cache
    assert half.tolist() == want[n // 2:].tolist()
    s.close()
    m.close()


@pytest.mark.parametrize("kvq", [0, 32])
def test_dynamic_refresh_equals_full_refresh(hip_backend, kvq):
    """The adapter's per-token refresh — (slice_pos, seq_kv) through zgml_hip_refresh_dynamic, as the reference's wgpu backend
    reduces refresh_program (src/backend/program.zig:7452-7490, src/backend/wgpu.zig:1162-1169) — against the whole op list through
    zgml_hip_refresh_program: the same logits, bit for bit, at every position, also after jumping back in the context (the KV
    store offsets and seq_kv of ALL dynamic ops move) and with the quantised-KV extension ops (kvq_store.col)."""
    cfg = llama.preset("tiny", 128)
    cfg.kv_quant_block = kvq
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    s_full = llama.Session(m, llama.hip_backend_fns(hip_backend))
    s_dyn = llama.Session(m, llama.hip_backend_fns(hip_backend))
    s_dyn.use_dynamic_refresh()
    toks = [3, 17, 200, 5, 9, 11, 400, 2, 77, 31]
    for rep in range(2):  # the second pass restarts at position 0 over a warm cache
        for pos, t in enumerate(toks):
            t_f, l_f = s_full.step(t, pos)
            t_d, l_d = s_dyn.step(t, pos)
            assert not hip_backend.last_error(), hip_backend.last_error()
            assert t_f == t_d and np.array_equal(l_f, l_d), (rep, pos)
    s_full.close(), s_dyn.close(), m.close()


def test_full_refresh_still_detects_a_static_change(hip_backend, oracle):
    """zgml_hip_refresh_program compares the static part of every op (without copying them: ~4 us per 1654-op SmolLM token) and
    rebuilds the launch list when one changed — legal for the reference's cpu backend, which re-reads the ops on every execute
    (src/backend/cpu.zig:128-131). Here the rmsnorm eps of a two-op program changes between executes."""
    from zgml_amd import DeviceOp, DeviceProgram, ProgramIO
    rng = np.random.default_rng(5)
    rows, cols = 2, 320
    x = rng.standard_normal(rows * cols).astype(np.float32)
    g = rng.standard_normal(rows * cols).astype(np.float32)

    def ops(eps):
        return [DeviceOp.rmsnorm(1, 0, rows, cols, eps), DeviceOp.elementwise("mul", 2, 1, 3, rows * cols)]
    prog = DeviceProgram(ops=ops(1e-5), buffer_sizes=[rows * cols] * 4, initial_uploads=[ProgramIO(0, x), ProgramIO(3, g)])
    h = hip_backend.compileProgram(prog)
    out = np.zeros(rows * cols, np.float32)
    try:
        for eps in (1e-5, 0.5, 1e-5):
            hip_backend.refreshProgram(h, ops(eps))
            hip_backend.executeProgram(h, [], [ProgramIO(2, out)])
            assert not hip_backend.last_error(), hip_backend.last_error()
            want = oracle.run_program(DeviceProgram(ops=ops(eps), buffer_sizes=[rows * cols] * 4, initial_uploads=[ProgramIO(0, x), ProgramIO(3, g)]), 2, rows * cols)
            np.testing.assert_allclose(out, want, atol=1e-5, rtol=1e-5, err_msg=f"eps {eps}")
    finally:
        hip_backend.freeProgram(h)


def test_resident_decode_survives_a_plan_rebuild(hip_backend):
    """The resident graph bakes the plan's kernel nodes and parameter arrays: a plan rebuild (here forced by
    set_barriers, and by a small-seq_kv refresh in between) must re-capture it, and the rebuilt plan must keep the
    compile-time attention bound (KV store and its attention never on one level). Checked against vtable stepping."""
    import ctypes as C
    hip = capi.load_hip()
    m = llama.Model(llama.preset("tiny"), llama.Q4_0, threads=8)
    s = llama.Session(m, llama.hip_backend_fns(hip_backend))
    n = 16
    s.resident_setup(hip_backend)
    first = s.resident_decode(5, 0, n)
    want, _ = s.decode(5, 0, n)  # the vtable path refreshes seq_kv down to 1, 2, ... (small-seq_kv refreshes)
    assert first.tolist() == want.tolist()
    bars = (C.c_uint64 * 1)(m.program.n_ops // 2)
    assert hip.zgml_hip_program_set_barriers(hip_backend.ctx, s.handle, bars, 1) == 0  # plan_dirty: parameter arrays are freed
    again = s.resident_decode(5, 0, n)
    assert not hip_backend.last_error(), hip_backend.last_error()
    assert again.tolist() == want.tolist()
    assert hip.zgml_hip_program_set_barriers(hip_backend.ctx, s.handle, None, 0) == 0  # and back
    s.step(5, 0)  # a refresh with seq_kv = 1 right before the rebuild
    third = s.resident_decode(5, 0, n)
    assert third.tolist() == want.tolist()
    s.close()
    m.close()


@pytest.mark.parametrize("kind", [llama.Q4_0, llama.Q8_0])
@pytest.mark.parametrize("T", [5, 32])
def test_prefill_plan_matches_oracle_and_sequential(hip_backend, oracle, kind, T):
    """token_len = T plan (LlamaInferencePlan with token_len N; the reference's own check is
    'prefill == sequential decode', tol 1e-4, src/llama_inference.zig:983-1034): the HIP prefill
    matches the oracle's prefill of the same program, and HIP sequential stepping."""
    cfg = llama.preset("tiny")
    toks = [(7 * i + 3) % cfg.vocab_size for i in range(T)]
    mN = llama.Model(cfg, kind, token_len=T)
    s_ref = llama.Session(mN, oracle.backend_fns())
    s_hip = llama.Session(mN, llama.hip_backend_fns(hip_backend))
    t_ref, l_ref = s_ref.prefill(toks, 0)
    t_hip, l_hip = s_hip.prefill(toks, 0)
    assert not hip_backend.last_error(), hip_backend.last_error()
    scale = np.abs(l_ref).max()
    assert np.abs(l_hip - l_ref).max() / scale < 2e-4
    assert t_hip == t_ref
    # a second chunk at pos = T continues the same cache
    toks2 = [(5 * i + 11) % cfg.vocab_size for i in range(T)]
    t_ref2, l_ref2 = s_ref.prefill(toks2, T)
    t_hip2, l_hip2 = s_hip.prefill(toks2, T)
    assert np.abs(l_hip2 - l_ref2).max() / np.abs(l_ref2).max() < 2e-4 and t_hip2 == t_ref2
    s_ref.close(), s_hip.close(), mN.close()
    m1 = llama.Model(cfg, kind)
    s1 = llama.Session(m1, llama.hip_backend_fns(hip_backend))
    for p, t in enumerate(toks + toks2):
        t_seq, l_seq = s1.step(t, p)
    assert np.abs(l_seq - l_hip2).max() / np.abs(l_seq).max() < 2e-4 and t_seq == t_hip2
    s1.close(), m1.close()


@pytest.mark.parametrize("T", [1, 32])
@pytest.mark.parametrize("promote", [False, True])
def test_dense_model_f32_and_f16_promoted(hip_backend, oracle, T, promote):
    """BASELINE config 5's dense variant: the same model with dequantised f32 weights as `matmul`
    ops. promote=False: exact-f32 path (2e-4 of the logit range, as the quantized tests);
    promote=True: weights promoted to f16 on both sides (wgpu.zig:1071-1104) — A is rounded to f16
    for T > 1, and a 1-ulp f32 difference in A can flip that rounding (2^-11 relative on one term),
    so the bound is 2e-3 of the logit range; greedy tokens must agree."""
    cfg = llama.preset("tiny")
    toks = [(7 * i + 3) % cfg.vocab_size for i in range(T)]
    hip_backend.set_option(capi.OPT_F16_DENSE_WEIGHTS, int(promote))
    oracle.set_f16_dense(promote)
    try:
        m = llama.Model(cfg, llama.F32_DENSE, token_len=T)
        s_ref = llama.Session(m, oracle.backend_fns())
        s_hip = llama.Session(m, llama.hip_backend_fns(hip_backend))
        worst = 0.0
        for chunk in range(2):
            t_ref, l_ref = s_ref.prefill(toks, chunk * T)
            t_hip, l_hip = s_hip.prefill(toks, chunk * T)
            assert not hip_backend.last_error(), hip_backend.last_error()
            worst = max(worst, float(np.abs(l_hip - l_ref).max() / np.abs(l_ref).max()))
            assert t_hip == t_ref
        assert worst < (2e-3 if promote else 2e-4), worst
        s_ref.close(), s_hip.close(), m.close()
    finally:
        hip_backend.set_option(capi.OPT_F16_DENSE_WEIGHTS, 0)
        oracle.set_f16_dense(False)


def test_gguf_block_model_equals_expanded_model(hip_backend, oracle):
    """the model handed over as raw Q4_0 file blocks (packed pass-through) decodes
    bit-identically to the same model uploaded in the reference's expanded int8 + f32-scale form."""
    cfg = llama.preset("tiny")
    ma, mb = llama.Model(cfg, llama.Q4_0), llama.Model(cfg, llama.Q4_0_GGUF)
    sa, sb = llama.Session(ma, llama.hip_backend_fns(hip_backend)), llama.Session(mb, llama.hip_backend_fns(hip_backend))
    so = llama.Session(mb, oracle.backend_fns())  # the oracle expands the blocks with the loader restatement
    tok = 3
    for pos in range(6):
        ta, la = sa.step(tok, pos)
        tb, lb = sb.step(tok, pos)
        to, lo = so.step(tok, pos)
        assert np.array_equal(la, lb) and ta == tb == to
        assert np.abs(lb - lo).max() / np.abs(lo).max() < 2e-4
        tok = ta
    for s in (sa, sb, so):
        s.close()
    ma.close(), mb.close()


def test_prefill_plan_hands_kv_to_decode_plan(hip_backend, oracle):
    """The reference's flow (llama_inference.prefill, then step): a token_len = 8 plan fills the KV
    caches, the decode plan continues from them. The two compiled programs own separate buffers, so
    the caches are handed over on the device (zgml_hip_copy_program_buffer); the continuation must
    equal decoding all tokens one by one."""
    cfg = llama.preset("tiny")
    T, n_dec = 8, 5
    toks = [(5 * i + 2) % cfg.vocab_size for i in range(T)]
    fns = llama.hip_backend_fns(hip_backend)
    m_pre, m_dec, m_seq = llama.Model(cfg, llama.Q4_0, token_len=T), llama.Model(cfg, llama.Q4_0), llama.Model(cfg, llama.Q4_0)
    s_pre, s_dec, s_seq = llama.Session(m_pre, fns), llama.Session(m_dec, fns), llama.Session(m_seq, fns)
    nxt, logits_pre = s_pre.prefill(toks, 0)
    S, dh, kv = cfg.max_seq_len, cfg.d_model // cfg.n_heads, cfg.n_kv_heads
    for layer in range(cfg.n_layers):
        for which in ("k_cache", "v_cache"):
            rc = hip_backend._lib.zgml_hip_copy_program_buffer(hip_backend.ctx, s_dec.handle, m_dec.buf(which, layer), 0,
                                                              s_pre.handle, m_pre.buf(which, layer), 0, dh * S * kv)
            assert rc == 0, hip_backend.last_error()
    # sequential reference: all T tokens one by one, then the same continuation
    for p, t in enumerate(toks):
        t_seq, l_seq = s_seq.step(t, p)
    assert t_seq == nxt and np.abs(l_seq - logits_pre).max() / np.abs(l_seq).max() < 2e-4
    tok_a = tok_b = nxt
    for i in range(n_dec):
        ta, la = s_dec.step(tok_a, T + i)
        tb, lb = s_seq.step(tok_b, T + i)
        assert ta == tb and np.abs(la - lb).max() / np.abs(lb).max() < 2e-4
        tok_a, tok_b = ta, tb
    for s in (s_pre, s_dec, s_seq):
        s.close()
    for m in (m_pre, m_dec, m_seq):
        m.close()


@pytest.mark.parametrize("name,steps", [("tiny", 10), ("smollm-135m", 4)])
def test_quantised_kv_cache_decode(hip_backend, oracle, name, steps):
    """the decode program with int8 KV caches (kvq_store + attention_kvq extension ops,
    block 32): HIP equals the oracle's restatement (identical greedy tokens, logits within 2e-4 of the
    range), through the vtable and through the device-resident loop."""
    cfg = llama.preset(name)
    cfg.kv_quant_block = 32
    oracle.set_threads(8)
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    s_ref, s_hip = llama.Session(m, oracle.backend_fns()), llama.Session(m, llama.hip_backend_fns(hip_backend))
    tok, toks = 3, []
    for pos in range(steps):
        t_ref, l_ref = s_ref.step(tok, pos)
        t_hip, l_hip = s_hip.step(tok, pos)
        assert not hip_backend.last_error(), hip_backend.last_error()
        assert np.abs(l_hip - l_ref).max() / np.abs(l_ref).max() < 2e-4 and t_hip == t_ref
        tok = t_ref
        toks.append(tok)
    s_hip.close()
    s_res = llama.Session(m, llama.hip_backend_fns(hip_backend))
    s_res.resident_setup(hip_backend)
    got = s_res.resident_decode(3, 0, steps)
    assert [int(t) for t in got] == toks
    s_res.close(), s_ref.close(), m.close()


@pytest.mark.parametrize("kvq", [0, 32])
def test_long_context_attention_split(hip_backend, oracle, kvq):
    """Flash-decoding split of the fused decode attention (ZGML_HIP_OPT_ATTN_SPLIT_MIN_KEYS): with the
    threshold lowered to 32 keys a 256-position tiny model splits every head over up to 8 workgroups
    from position 63 on. Logits stay within 2e-4 of the oracle's over 140 positions (the merge order is
    fixed, only the summation order differs), the unsplit build of the same program agrees, and the
    graph-replayed resident loop (arrival counters re-armed in-kernel) gives the same tokens twice."""
    cfg = llama.preset("tiny", 256)
    cfg.kv_quant_block = kvq  # 32: the int8 KV cache ops (kvq_store / attention_kvq) take the same split
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    n = 140
    s_ref = llama.Session(m, oracle.backend_fns())
    hip_backend.set_option(capi.OPT_ATTN_SPLIT_MIN_KEYS, 32)
    try:
        s_split = llama.Session(m, llama.hip_backend_fns(hip_backend))
        hip_backend.set_option(capi.OPT_ATTN_SPLIT_MIN_KEYS, 0)
        s_plain = llama.Session(m, llama.hip_backend_fns(hip_backend))
        hip_backend.set_option(capi.OPT_ATTN_SPLIT_MIN_KEYS, 32)
        tok, toks = 3, []
        for pos in range(n):
            t_ref, l_ref = s_ref.step(tok, pos)
            t_a, l_a = s_split.step(tok, pos)
            t_b, l_b = s_plain.step(tok, pos)
            assert not hip_backend.last_error(), hip_backend.last_error()
            scale = np.abs(l_ref).max()
            assert np.abs(l_a - l_ref).max() / scale < 2e-4, pos
            assert np.abs(l_a - l_b).max() / scale < 2e-5, pos
            if pos < 63:
                assert np.array_equal(l_a, l_b), pos  # below the threshold the split launch is the plain one
            top2 = np.sort(l_ref)[-2:]
            if top2[1] - top2[0] > 1e-3 * scale:
                assert t_a == t_ref, pos
            tok = t_ref
            toks.append(tok)
        s_split.close(), s_plain.close()
        s_res = llama.Session(m, llama.hip_backend_fns(hip_backend))
        s_res.resident_setup(hip_backend)
        first = s_res.resident_decode(3, 0, n)
        second = s_res.resident_decode(3, 0, n)
        assert first.tolist() == second.tolist()
        s_res.close()
    finally:
        hip_backend.set_option(capi.OPT_ATTN_SPLIT_MIN_KEYS, -1)
    s_ref.close(), m.close()


@pytest.mark.parametrize("kvq", [0, 32])
@pytest.mark.parametrize("token_len", [1, 5])
def test_fused_plan_equals_serial_plan(hip_backend, kvq, token_len):
    """Every launch fusion of the runtime (mat-vec prologue / epilogue / grouping, fused decode attention incl. the
    quantised-KV form, row chains, elementwise chains, grouped tile launches, streaming prefill attention) against the
    one-launch-per-DeviceOp plan of the same program (ZGML_HIP_OPT_FUSION = 0): logits within 1e-5 of the logit range
    (the fused kernels keep each op's arithmetic; only summation orders inside mat-vec / attention kernels differ) and the
    same greedy token wherever the top-2 margin exceeds that. With int8 KV caches a last-bit difference in a projection can
    flip one quantised value, so the bound there is the oracle-parity one (3e-4)."""
    cfg = llama.preset("tiny", 128)
    cfg.kv_quant_block = kvq
    m = llama.Model(cfg, llama.Q4_0, threads=8, token_len=token_len)
    s_f = llama.Session(m, llama.hip_backend_fns(hip_backend))
    hip_backend.set_option(capi.OPT_FUSION, 0)
    try:
        s_s = llama.Session(m, llama.hip_backend_fns(hip_backend))
    finally:
        hip_backend.set_option(capi.OPT_FUSION, 1)
    tol = 3e-4 if kvq else 1e-5

    def check(t_f, l_f, t_s, l_s, where):
        assert not hip_backend.last_error(), hip_backend.last_error()
        scale = np.abs(l_s).max()
        assert np.abs(l_f - l_s).max() <= tol * scale, where
        top2 = np.sort(l_s)[-2:]
        if top2[1] - top2[0] > 4 * tol * scale:
            assert t_f == t_s, where

    if token_len == 1:
        tok = 3
        for pos in range(40):
            t_f, l_f = s_f.step(tok, pos)
            t_s, l_s = s_s.step(tok, pos)
            check(t_f, l_f, t_s, l_s, pos)
            tok = t_s
    else:
        for chunk in range(4):
            toks = [(11 * (chunk * token_len + i) + 5) % cfg.vocab_size for i in range(token_len)]
            t_f, l_f = s_f.prefill(toks, chunk * token_len)
            t_s, l_s = s_s.prefill(toks, chunk * token_len)
            check(t_f, l_f, t_s, l_s, chunk)
    s_f.close(), s_s.close(), m.close()


def _download(hip_backend, handle, buf, n):
    """Raw f32 image of a program buffer (zgml_hip_download_outputs: the read half of execute_program)."""
    import ctypes as C
    out = np.zeros(n, np.float32)
    io = (capi.ProgramIOC * 1)(capi.ProgramIOC(buf, 0, 0, out.ctypes.data, 4 * n, 0))
    capi.load_hip().zgml_hip_download_outputs(hip_backend.ctx, handle, io, 1)
    return out


@pytest.mark.parametrize("kind", [llama.Q4_0, llama.Q8_0])
def test_prefill_chunk_of_128_tokens(hip_backend, oracle, kind):
    """The reference's prefill chunk size (LlamaInferenceSession.prefill, src/llama_inference.zig:474: chunks of 128):
    a token_len = 128 plan (8 m-tiles per matmul, 128 query columns per attention) against the oracle's execution of the
    same program and against 128 sequential decode steps ('prefill == sequential', src/llama_inference.zig:983-1034)."""
    cfg = llama.preset("tiny", 256)
    T = 128
    toks = [(7 * i + 3) % cfg.vocab_size for i in range(T)]
    oracle.set_threads(8)
    mN = llama.Model(cfg, kind, token_len=T)
    s_ref = llama.Session(mN, oracle.backend_fns())
    s_hip = llama.Session(mN, llama.hip_backend_fns(hip_backend))
    t_ref, l_ref = s_ref.prefill(toks, 0)
    t_hip, l_hip = s_hip.prefill(toks, 0)
    assert not hip_backend.last_error(), hip_backend.last_error()
    assert np.abs(l_hip - l_ref).max() / np.abs(l_ref).max() < 2e-4 and t_hip == t_ref
    toks2 = [(5 * i + 11) % cfg.vocab_size for i in range(T)]  # second chunk: queries 128..255 over 256 cached keys
    t_ref2, l_ref2 = s_ref.prefill(toks2, T)
    t_hip2, l_hip2 = s_hip.prefill(toks2, T)
    assert np.abs(l_hip2 - l_ref2).max() / np.abs(l_ref2).max() < 2e-4 and t_hip2 == t_ref2
    s_ref.close(), s_hip.close(), mN.close()
    m1 = llama.Model(cfg, kind)
    s1 = llama.Session(m1, llama.hip_backend_fns(hip_backend))
    for p, t in enumerate(toks + toks2):
        t_seq, l_seq = s1.step(t, p, want_logits=(p == 2 * T - 1))
    assert np.abs(l_seq - l_hip2).max() / np.abs(l_seq).max() < 2e-4 and t_seq == t_hip2
    s1.close(), m1.close()


@pytest.mark.parametrize("name,T", [("tiny", 5), ("tiny", 32), ("smollm-135m", 128)])
def test_resident_prefill_equals_vtable_prefill(hip_backend, name, T):
    """zgml_hip_resident_prefill (only the T token ids cross PCIe; embedding gather, causal-mask columns, RoPE rows, KV
    positions and the argmax on the device) leaves the same KV caches and returns the same token as execute_program on
    host-patched inputs, for two consecutive chunks; and the program stays usable through the vtable afterwards."""
    cfg = llama.preset(name, 512 if name != "tiny" else 0)
    m = llama.Model(cfg, llama.Q4_0, threads=16, token_len=T)
    a = llama.Session(m, llama.hip_backend_fns(hip_backend))
    b = llama.Session(m, llama.hip_backend_fns(hip_backend))
    b.resident_setup(hip_backend)
    n_cache = cfg.max_seq_len * cfg.d_head * cfg.n_kv_heads
    for chunk in range(2):
        toks = [(7 * (i + chunk * T) + 3) % cfg.vocab_size for i in range(T)]
        t_a, _ = a.prefill(toks, chunk * T)
        t_b = b.resident_prefill(toks, chunk * T)
        assert not hip_backend.last_error(), hip_backend.last_error()
        assert t_a == t_b, (chunk, t_a, t_b)
        for which in ("k_cache", "v_cache"):
            ca = _download(hip_backend, a.handle, m.buf(which, cfg.n_layers - 1), n_cache)
            cb = _download(hip_backend, b.handle, m.buf(which, cfg.n_layers - 1), n_cache)
            assert np.array_equal(ca, cb), (chunk, which)
    toks = [(3 * i + 1) % cfg.vocab_size for i in range(T)]  # the resident program through the vtable again (position 0)
    assert b.prefill(toks, 0)[0] == a.prefill(toks, 0)[0]
    a.close(), b.close(), m.close()


def test_pinned_outputs_equal_the_staged_path(hip_backend):
    """zgml_hip_program_pin_outputs: after three steps into the same host buffer the logits are written straight into it by the
    step's own graph; the numbers are those of the staging path, a different buffer falls back (and still gets its data), and
    switching the promise off returns to staging."""
    hip = capi.load_hip()
    m = llama.Model(llama.preset("tiny", 64), llama.Q4_0, threads=4)
    s_a = llama.Session(m, llama.hip_backend_fns(hip_backend))
    s_b = llama.Session(m, llama.hip_backend_fns(hip_backend))
    s_a.use_dynamic_refresh(), s_b.use_dynamic_refresh()
    s_b.pin_outputs(hip_backend)
    tok = 3
    for pos in range(10):  # (zh_session_step copies the session's own logits buffer out: same host address every step)
        t_a, l_a = s_a.step(tok, pos)
        t_b, l_b = s_b.step(tok, pos)
        assert not hip_backend.last_error(), hip_backend.last_error()
        assert t_a == t_b and np.array_equal(l_a, l_b), pos
        tok = t_a
    # the raw entry point with caller buffers: the same address three times, then another one, then the promise withdrawn
    n_in, n_out = C.c_uint64(), C.c_uint64()
    ins = s_b.lib.zh_model_step_inputs(m.ptr, C.byref(n_in))
    outs = s_b.lib.zh_model_step_outputs(m.ptr, C.byref(n_out))
    assert n_out.value == 1
    vocab = m.cfg.vocab_size
    want = l_a.copy()  # position 9's logits: re-executing the same step gives them again
    buf1, buf2 = np.zeros(vocab, np.float32), np.zeros(vocab, np.float32)

    def run(buf):
        buf[:] = -7.0
        io = (capi.ProgramIOC * 1)(capi.ProgramIOC(outs[0].buf_idx, 0, outs[0].offset, buf.ctypes.data, outs[0].size, 0))
        hip.zgml_hip_execute_program(hip_backend.ctx, s_b.handle, ins, n_in.value, io, 1)
        assert not hip_backend.last_error(), hip_backend.last_error()
        return buf

    for _ in range(5):
        assert np.array_equal(run(buf1), want)
    assert np.array_equal(run(buf2), want)  # moved: unregistered, staged
    assert np.array_equal(run(buf1), want)
    hip.zgml_hip_program_pin_outputs(hip_backend.ctx, s_b.handle, 0)
    assert np.array_equal(run(buf1), want)
    s_a.close(), s_b.close(), m.close()
