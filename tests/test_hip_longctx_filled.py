"""Long-context decode over a FILLED KV cache (the position-1900 fixtures of round 3 step over a
cache that is 99.8 % zero columns, so a wrong chunk base, stride or merge order of the 8-16-way context split could not
show). Three layers of evidence, all at seq_kv ~ 1900 with every cache column holding data:

 1. `*_filled` cases of tests/golden/longctx.json (tools/gen_golden_longctx.py --filled): the reference's own flow —
    `LlamaInferenceSession.prefill` in chunks of 128 (src/llama_inference.zig:474), the caches handed to the decode plan,
    4 greedy steps (`step`, :405-466) — computed by the ORACLE; HIP fills its cache through zgml_hip_resident_prefill
    (last chunk through the vtable for its logits), hands the caches over on the device and decodes through the fused
    q/k/v + attention launch and through the stand-alone decode attention. f32 KV 2e-4 of the logit range, int8 KV 1e-3.
 2. oracle and HIP decode over the SAME cache image (random, uploaded to both): whatever the projections do to
    storeColumn's truncation, 1900 of 1901 columns are identical, so the int8 path must meet the f32 bound here. This is
    tools/dbg_kvq_long.py's finding as a test: the attention is exact on a given cache.
 3. the int8 cache contents themselves, oracle against HIP after the same token stream: every differing byte differs by
    ONE unit (a projection's last bit straddling the truncation toward zero of quant.zig:239-246), at most 5 % of a
    head's stored values, and every block scale by <= 2e-4 of its column's largest.
Semantics: src/backend/reference.zig:568-672, src/quant.zig:687-699, :925-1091."""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np
import pytest

from zgml_amd import capi, llama

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


@pytest.mark.parametrize("attn", ["fused", "standalone"])
@pytest.mark.parametrize("case", ["smollm_f32kv_filled", "smollm_int8kv_filled", "l7dims_f32kv_filled", "l7dims_int8kv_filled"])
def test_prefilled_cache_decode_matches_the_oracle_fixture(hip_backend, case, attn):
    import bench
    assert "filled" in bench.verify_longctx_filled(hip_backend, llama, case, fused_attention=(attn == "fused"))
    assert not hip_backend.last_error(), hip_backend.last_error()


def _upload(be, handle, buf, arr):
    io = (capi.ProgramIOC * 1)(capi.ProgramIOC(buf, 0, 0, arr.ctypes.data, arr.nbytes, 0))
    capi.load_hip().zgml_hip_upload_inputs(be.ctx, handle, io, 1)


def _download(be, handle, buf, n):
    out = np.zeros(n, np.float32)
    io = (capi.ProgramIOC * 1)(capi.ProgramIOC(buf, 0, 0, out.ctypes.data, 4 * n, 0))
    capi.load_hip().zgml_hip_download_outputs(be.ctx, handle, io, 1)
    return out


def _cfg(name, kvq):
    cfg = llama.preset("smollm-135m" if name == "smollm" else "llama2-7b", 2048)
    cfg.n_layers, cfg.vocab_size, cfg.kv_quant_block = 2, 2048, kvq  # (a small vocabulary keeps the CPU oracle quick)
    return cfg


def _random_cache(rng, cfg, elems):
    """A cache image with every column filled. f32: values in (-1, 1). int8 (QuantizedKVCache, quant.zig:645-700: q_data then
    scales): random quants in [-127, 127] and block scales in (0.4, 1.2) / 127, i.e. dequantised values in (-1.2, 1.2)."""
    if not cfg.kv_quant_block:
        return ((rng.random(elems) - 0.5) * 2).astype(np.float32)
    S, dh = cfg.max_seq_len, cfg.d_head
    nq = S * dh // 4
    img = np.zeros(elems, np.float32)
    img[:nq] = rng.integers(-127, 128, S * dh, dtype=np.int8).view(np.float32)
    img[nq:] = ((0.4 + 0.8 * rng.random(elems - nq)) / 127).astype(np.float32)
    return img


@pytest.mark.parametrize("attn", ["fused", "standalone"])
@pytest.mark.parametrize("kvq", [0, 32])
@pytest.mark.parametrize("name", ["smollm", "l7"])
def test_decode_over_a_common_filled_cache(hip_backend, oracle, name, kvq, attn):
    """4 greedy steps at positions 1900..1903 with oracle and HIP reading the SAME fully populated cache (1900 random columns
    uploaded to both, the steps' own columns appended by each side): logits within 2e-4 of the oracle's range for f32 AND for
    int8 caches, identical greedy tokens. seq_kv 1901+ puts every head on the maximum context split (8-16 workgroups per
    head), inside the fused q/k/v + attention launch (`fused`) and in the stand-alone decode attention (`standalone`)."""
    cfg = _cfg(name, kvq)
    oracle.set_threads(8)
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    s_ref = llama.Session(m, oracle.backend_fns())
    hip_backend.set_option(capi.OPT_FUSE_RESIDENT_WGS, -1 if attn == "fused" else 0)
    try:
        s_hip = llama.Session(m, llama.hip_backend_fns(hip_backend))
    finally:
        hip_backend.set_option(capi.OPT_FUSE_RESIDENT_WGS, -1)
    rng = np.random.default_rng(0xd3fa + kvq)
    ob = oracle.OracleBackend()
    for buf, elems in m.kv_buffers():
        img = _random_cache(rng, cfg, elems)
        ob.buffer(s_ref.handle, buf)[:elems] = img
        _upload(hip_backend, s_hip.handle, buf, img)
    # the heads' concatenated attention outputs (the `sliceAssignRows` destination, llama_transformer.zig:230-238) per layer:
    # compared directly, because the logits dilute an attention error behind the residual stream (in the oracle, dropping one
    # split's worth of columns moves the logits by 1.5e-4 .. 0.18 of their range depending on the case, but these rows by ~10 %)
    prog = m.program
    attn_bufs = []
    for i in range(prog.n_ops):
        op = prog.ops[i]
        if op.kind == capi.DOP["slice_assign"] and op.u.slice_assign.patch_stride == 0 and op.u.slice_assign.dst not in attn_bufs:
            attn_bufs.append(int(op.u.slice_assign.dst))
    assert len(attn_bufs) == cfg.n_layers
    tok = 5
    for pos in range(1900, 1904):
        if kvq and pos > 1900:  # the previous steps' own columns too: HIP continues from the ORACLE's image, so exactly one column
            for buf, elems in m.kv_buffers():  # (this step's) can differ by storeColumn units between the two sides
                _upload(hip_backend, s_hip.handle, buf, np.ascontiguousarray(ob.buffer(s_ref.handle, buf)[:elems]))
        t_ref, l_ref = s_ref.step(tok, pos)
        t_hip, l_hip = s_hip.step(tok, pos)
        assert not hip_backend.last_error(), hip_backend.last_error()
        for layer, ab in enumerate(attn_bufs):
            a_ref = ob.buffer(s_ref.handle, ab)[:cfg.d_model]
            a_hip = _download(hip_backend, s_hip.handle, ab, cfg.d_model)
            err = np.abs(a_hip - a_ref).max() / np.abs(a_ref).max()
            # f32 caches: the rows of layer 0 agree to 2e-5 of their largest element (same q up to the mat-vec's summation order, same
            # keys). int8 caches: the step's OWN column(s) are quantised by each side from its own projection and may differ by single
            # units (next test) — one of ~1900 columns, but a real key that takes more of the softmax than a random one: measured
            # 2.5-3.0e-5, bar 1e-4 — identical in the fused and the stand-alone
            # launch, i.e. a property of the cache, not of the kernel. Layer 1 reads activations that already differ at the 1e-6 level.
            bar = (1e-4 if kvq else 2e-5) * (1 if layer == 0 else 10)
            assert err < bar, (pos, layer, err)
        scale = np.abs(l_ref).max()
        assert np.isfinite(l_hip).all() and np.abs(l_hip - l_ref).max() / scale < 2e-4, (pos, np.abs(l_hip - l_ref).max() / scale)
        top2 = np.sort(l_ref)[-2:]
        if top2[1] - top2[0] > 1e-3 * scale:
            assert t_hip == t_ref, pos
        tok = t_ref
    # an untouched column of the uploaded image is still bit-identical on the device (nothing wrote outside its column)
    buf, elems = m.kv_buffers()[0]
    img_dev, img_ref = _download(hip_backend, s_hip.handle, buf, elems), ob.buffer(s_ref.handle, buf)[:elems]
    dh = cfg.d_head
    if kvq:
        assert np.array_equal(img_dev[:1900 * dh // 4].view(np.uint32), img_ref[:1900 * dh // 4].view(np.uint32))
    else:
        assert np.array_equal(img_dev[:1900 * dh].view(np.uint32), img_ref[:1900 * dh].view(np.uint32))
    s_ref.close(), s_hip.close(), m.close()


@pytest.mark.parametrize("name", ["smollm", "l7"])
def test_int8_cache_differs_from_the_oracles_by_single_units(hip_backend, oracle, name):
    """The int8 caches after the same 48-token stream, oracle against HIP, byte by byte. storeColumn truncates
    `x * 127 / max|x|` toward zero (quant.zig:239-246), so a projection that differs in its last bit (the mat-vec's summation
    order, bound 2e-5 * sum|x w|) can land on the other side of an integer: such bytes differ by exactly ONE unit. Bars: no byte
    differs by more than one unit; at most 5 % of any head's stored values differ (measured at Llama-2-7B dimensions with the
    grid-like synthetic weights: 0.8-3.8 %; SmolLM: < 1 %); a block scale differs by at most 2e-4 of its column's largest scale (layer 0, where the inputs are common). With these met, the logits bound
    of the int8-KV model tests (1e-3 of the range) is the propagation of single cache units, not of the attention kernel —
    which test_decode_over_a_common_filled_cache pins at the f32 bound on a common cache."""
    cfg = _cfg(name, 32)
    cfg.max_seq_len = 64
    oracle.set_threads(8)
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    s_ref, s_hip = llama.Session(m, oracle.backend_fns()), llama.Session(m, llama.hip_backend_fns(hip_backend))
    n = 48
    for pos in range(n):
        tok = (7 * pos + 3) % cfg.vocab_size
        s_ref.step(tok, pos, want_logits=False)
        s_hip.step(tok, pos, want_logits=False)
    assert not hip_backend.last_error(), hip_backend.last_error()
    ob = oracle.OracleBackend()
    S, dh = cfg.max_seq_len, cfg.d_head
    nq = S * dh // 4
    worst_frac = worst_scale = 0.0
    # layer 0 only: its caches are embedding -> rmsnorm -> k / v projection (-> rope) -> storeColumn, i.e. exactly the effect being
    # bounded; deeper layers read activations that already went through an attention over differing units
    for i, (buf, elems) in enumerate(m.kv_buffers()[:2 * cfg.n_kv_heads]):
        ref, got = ob.buffer(s_ref.handle, buf)[:elems].copy(), _download(hip_backend, s_hip.handle, buf, elems)
        qr, qg = ref[:nq].view(np.int8).astype(np.int32)[:n * dh], got[:nq].view(np.int8).astype(np.int32)[:n * dh]
        assert np.abs(qr - qg).max() <= 1, (i, int(np.abs(qr - qg).max()))
        frac = float((qr != qg).mean())
        worst_frac = max(worst_frac, frac)
        assert frac <= 0.05, (i, frac)
        # block scales (max|x| / 127 of 32 projected values): relative to the column's largest scale, because a block whose
        # values nearly cancel carries the mat-vec's absolute error (2e-5 * sum|x w|) on a small maximum — measured 6.5e-5 at
        # Llama-2-7B dimensions (1.5e-4 of the block's own scale); bar 2e-4 of the column's largest
        nb = dh // 32
        sr, sg = ref[nq:nq + n * nb].reshape(n, nb), got[nq:nq + n * nb].reshape(n, nb)
        rel = np.abs(sr - sg).max(axis=1) / sr.max(axis=1)
        worst_scale = max(worst_scale, float(rel.max()))
        assert rel.max() <= 2e-4, (i, float(rel.max()))
    print(f"{name}: worst fraction of differing int8 values per head {worst_frac:.4f}, worst block-scale difference {worst_scale:.2e} of the column's largest")
    s_ref.close(), s_hip.close(), m.close()


def test_full_32_layer_llama2_7b_matches_the_oracle_fixture(hip_backend):
    """tests/golden/l7full.json (tools/gen_golden_l7.py --full: the ORACLE on the whole 32-layer Llama-2-7B Q4_0 program, max_seq
    2048, 4 greedy positions with logit probes): the program bench.py times, through the vtable (logits 2e-4 of the range, tokens)
    and through the device-resident loop (tokens). BASELINE configs[2]; VERDICT r03 missing #5."""
    import bench
    full = json.loads((ROOT / "tests" / "golden" / "l7full.json").read_text())
    cfg = llama.preset("llama2-7b", full["max_seq"])
    assert cfg.n_layers == full["n_layers"] == 32
    m = llama.Model(cfg, llama.Q4_0, threads=16)
    s = llama.Session(m, llama.hip_backend_fns(hip_backend))
    try:
        tok = full["first_token"]
        for pos, (want_tok, probe) in enumerate(zip(full["tokens"], full["probes"])):
            tok, logits = s.step(tok, pos)
            assert not hip_backend.last_error(), hip_backend.last_error()
            bench._check_probe(logits, probe, 2e-4, f"32-layer Llama-2-7B decode, position {pos}")
            assert tok == want_tok, pos
        s.close()
        s = llama.Session(m, llama.hip_backend_fns(hip_backend))  # fresh caches
        s.resident_setup(hip_backend)
        assert s.resident_decode(full["first_token"], 0, len(full["tokens"])).tolist() == full["tokens"]
    finally:
        s.close()
        m.close()
