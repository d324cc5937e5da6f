"""The K-split layer launches of short-K models (zgml_amd/csrc/ksplit.hip, runtime.hip: fuse_ksplit) against the oracle.

A decoder layer of such a model runs as three launches — [deferred vector -> rmsnorm -> gamma -> q / k / v], [attention of each
head -> that head's partial of the O projection], [O partials + residual -> rmsnorm -> gamma -> gate / up -> SiLU chain,
product -> partial of the down projection] — and the O / down outputs and their residual sums are stored late, by the launch
that sums the partials. So what is checked here is (a) EVERY buffer of the program after a step, not just the logits: the
projection outputs and residual sums stored by the consuming launch, the norm's outputs, the SiLU chain's intermediates, the KV
caches; (b) that two executions give identical bits (fixed summation orders, no atomics on data); (c) that the plan really is the
three-launch form; (d) the long-context split of the attention feeding the partial projection; (e) that a caller who enqueues op
ranges gets a plan without deferred vectors.

Tolerances: the mat-vecs differ from the reference's k-sequential sum in order only — 2e-5 * sum|x w| per output,
which over a 2-layer model shows as ~1e-5 of a buffer's range; buffers downstream of exp() (the SiLU chain) and of the softmax get
the relative form. Op stream: src/models/llama_transformer.zig:192-253."""
import ctypes as C

import numpy as np
import pytest

from zgml_amd import capi, llama

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _ksplit_on(hip_backend):
    """The pass is opt-in (measured slower than the four-launch layer on MI355X: DESIGN.md section 4, round 5)."""
    hip_backend.set_option(capi.OPT_KSPLIT, 1)
    yield
    hip_backend.set_option(capi.OPT_KSPLIT, 0)


def _cfg(d_head, n_heads, n_kv, d_ff, n_layers=2, vocab=512, max_seq=64, kvq=0):
    c = llama.preset("tiny", max_seq)
    c.d_model, c.n_heads, c.n_kv_heads, c.d_ff, c.n_layers, c.vocab_size = d_head * n_heads, n_heads, n_kv, d_ff, n_layers, vocab
    c.kv_quant_block = kvq
    return c


def _download(hip_backend, handle, buf, n):
    out = np.zeros(n, np.float32)
    io = (capi.ProgramIOC * 1)(capi.ProgramIOC(buf, 0, 0, out.ctypes.data, 4 * n, 0))
    capi.load_hip().zgml_hip_download_outputs(hip_backend.ctx, handle, io, 1)
    return out


def _plan(hip_backend, handle):
    return hip_backend.planText(handle)


def _all_buffers(hip_backend, oracle, m, s_hip, s_ref):
    """(index, hip image, oracle image) of every buffer that has an f32 image on the device."""
    ob = oracle.OracleBackend()
    hip = capi.load_hip()
    prog = m.program
    out = []
    for b in range(prog.n_buffers):
        n = int(prog.buffer_sizes[b])
        if n == 0 or n > (1 << 22):
            continue
        got = _download(hip_backend, s_hip.handle, b, n)
        err = hip_backend.last_error()
        if err:  # elided at compile time (a weight no op reads as f32): nothing to compare
            assert "no f32 image" in err, err
            hip.zgml_hip_clear_error(hip_backend.ctx)
            continue
        want = np.array(ob.buffer(s_ref.handle, b)[:n])
        out.append((b, got, want))
    return out


@pytest.mark.parametrize("dh,nh,nkv,dff", [(64, 3, 1, 320), (64, 9, 3, 1536), (128, 2, 2, 512)])
def test_ksplit_layer_stores_every_buffer(hip_backend, oracle, dh, nh, nkv, dff):
    cfg = _cfg(dh, nh, nkv, dff)
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    s_ref = llama.Session(m, oracle.backend_fns())
    s_hip = llama.Session(m, llama.hip_backend_fns(hip_backend))
    tok = 3
    for pos in range(6):
        t_ref, l_ref = s_ref.step(tok, pos)
        t_hip, l_hip = s_hip.step(tok, pos)
        assert not hip_backend.last_error(), hip_backend.last_error()
        assert np.abs(l_hip - l_ref).max() <= 2e-4 * np.abs(l_ref).max(), pos
        assert t_hip == t_ref, pos
        if pos in (0, 5):
            for b, got, want in _all_buffers(hip_backend, oracle, m, s_hip, s_ref):
                fin = np.isfinite(want)
                assert np.array_equal(np.isfinite(got), fin), b  # (-inf mask entries)
                scale = max(1e-6, float(np.abs(want[fin]).max())) if fin.any() else 1.0
                np.testing.assert_allclose(got[fin], want[fin], rtol=2e-3, atol=2e-4 * scale, err_msg=f"buffer {b} at position {pos}")
        tok = t_ref
    plan = _plan(hip_backend, s_hip.handle)
    # (q / k / v + attention + partial O as ONE launch where the grid is resident, else as two)
    assert plan.count("ks-layer-a") + plan.count("ks-proj") == cfg.n_layers and plan.count("ks-layer-a") + plan.count("ks-attention") == cfg.n_layers, plan
    assert plan.count("ks-mlp") == cfg.n_layers, plan
    assert plan.count("sums the deferred vector") == 2 * cfg.n_layers, plan  # every ks-mlp, every ks-proj but the first, the final norm
    assert "ks-materialise" not in plan, plan
    s_ref.close(), s_hip.close(), m.close()


def test_ksplit_two_executions_are_bit_identical(hip_backend):
    cfg = _cfg(64, 9, 3, 1536, n_layers=3)
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    s = llama.Session(m, llama.hip_backend_fns(hip_backend))
    toks = [3, 17, 200, 5, 9, 11, 400, 2]
    a = [s.step(t, p)[1] for p, t in enumerate(toks)]
    b = [s.step(t, p)[1] for p, t in enumerate(toks)]
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    s.close(), m.close()


def test_ksplit_equals_the_four_launch_plan(hip_backend, monkeypatch):
    """The same program with the K-split pass off (the default plan: four launches per layer): logits within 1e-5 of their range,
    same greedy token where the top-2 margin allows."""
    cfg = _cfg(64, 9, 3, 1536, n_layers=3, max_seq=128)
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    s_f = llama.Session(m, llama.hip_backend_fns(hip_backend))
    assert "ks-mlp" in _plan(hip_backend, s_f.handle)  # (the plan is built here, under the option)
    hip_backend.set_option(capi.OPT_KSPLIT, 0)
    try:
        s_s = llama.Session(m, llama.hip_backend_fns(hip_backend))
        assert "ks-" not in _plan(hip_backend, s_s.handle)
    finally:
        hip_backend.set_option(capi.OPT_KSPLIT, 1)
    tok = 3
    for pos in range(48):
        t_f, l_f = s_f.step(tok, pos)
        t_s, l_s = s_s.step(tok, pos)
        assert not hip_backend.last_error(), hip_backend.last_error()
        scale = np.abs(l_s).max()
        assert np.abs(l_f - l_s).max() <= 1e-5 * scale, pos
        top2 = np.sort(l_s)[-2:]
        if top2[1] - top2[0] > 4e-5 * scale:
            assert t_f == t_s, pos
        tok = t_s
    assert "ks-mlp" in _plan(hip_backend, s_f.handle)
    s_f.close(), s_s.close(), m.close()


@pytest.mark.parametrize("kvq", [0, 32])
def test_ksplit_attention_with_the_long_context_split(hip_backend, oracle, kvq):
    """Several workgroups per head (flash-decoding split, forced by a small minimum of keys per split): the LAST arriver of a head
    merges the chunks and computes the head's partial of the O projection. Against the oracle over a filling cache."""
    cfg = _cfg(64, 3, 1, 320, max_seq=256, kvq=kvq)
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    s_ref = llama.Session(m, oracle.backend_fns())
    hip_backend.set_option(capi.OPT_ATTN_SPLIT_MIN_KEYS, 16)
    try:
        s_hip = llama.Session(m, llama.hip_backend_fns(hip_backend))
        tok = 3
        for pos in range(120):
            t_ref, l_ref = s_ref.step(tok, pos)
            t_hip, l_hip = s_hip.step(tok, pos)
            assert not hip_backend.last_error(), hip_backend.last_error()
            scale = np.abs(l_ref).max()
            assert np.abs(l_hip - l_ref).max() <= (1e-3 if kvq else 2e-4) * scale, pos
            top2 = np.sort(l_ref)[-2:]
            if top2[1] - top2[0] > 4e-3 * scale:
                assert t_hip == t_ref, pos
            tok = t_ref
        plan = _plan(hip_backend, s_hip.handle)
        assert "ks-attention" in plan or "ks-layer-a" in plan, plan
        s_hip.close()
    finally:
        hip_backend.set_option(capi.OPT_ATTN_SPLIT_MIN_KEYS, -1)
    s_ref.close(), m.close()


def test_op_ranges_switch_the_deferred_vectors_off(hip_backend, oracle):
    """zgml_hip_enqueue_ops (the multi-GPU harness's way of running a program piecewise) must leave every buffer of its range
    written: the first partial range rebuilds the plan without deferred vectors, and the results stay those of the oracle."""
    hip = capi.load_hip()
    cfg = _cfg(64, 3, 1, 320)
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    s_ref = llama.Session(m, oracle.backend_fns())
    s_hip = llama.Session(m, llama.hip_backend_fns(hip_backend))
    t_ref, l_ref = s_ref.step(3, 0)
    t_hip, _ = s_hip.step(3, 0)
    assert t_hip == t_ref and "ks-mlp" in _plan(hip_backend, s_hip.handle)
    n_ops = m.program.n_ops
    hip.zgml_hip_enqueue_ops.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
    hip.zgml_hip_enqueue_ops(hip_backend.ctx, s_hip.handle, 0, n_ops // 2)
    hip.zgml_hip_synchronize(hip_backend.ctx)
    err = hip_backend.last_error()
    if err:  # (a range that cuts a batched launch is refused loudly: also fine, but the plan must have been rebuilt)
        hip.zgml_hip_clear_error(hip_backend.ctx)
    plan = _plan(hip_backend, s_hip.handle)
    assert "ks-" not in plan, plan
    t2, l2 = s_hip.step(3, 0)
    assert not hip_backend.last_error(), hip_backend.last_error()
    assert t2 == t_ref and np.abs(l2 - l_ref).max() <= 2e-4 * np.abs(l_ref).max()
    s_ref.close(), s_hip.close(), m.close()
