"""The launch that carries the q / k / v projection AND the decode attention waiting for it (qkv_attn_kernel, chosen by
fuse_qkv_attention for K <= 2048 and d_head 64 / 128) against the oracle, in the regimes the other decode tests do not
reach: the long-context split ACTIVE inside the fused launch (several consumer workgroups per head, several consumers per
kv-head counter, last-arriver merge in the launch) and the d_head = 128 instances; plus the residency guard and the loud
failure + two-launch fallback when a hand-off wait gives up.
Semantics: src/backend/reference.zig:568-672 (attention), :530-565 (qmatmul); tolerance as tests/test_hip_llama.py."""
import ctypes as C

import numpy as np
import pytest

from zgml_amd import capi, llama

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(__import__("os").environ.get("ZGML_HIP_FUSE_QKV_ATTN") == "0",
                                                  reason="the diagnostic switch under test turns the fused launch off")]


def _cfg(name, max_seq, kvq):
    """2 layers of a model that really takes the fused launch; a small vocabulary keeps the CPU oracle quick."""
    if name == "smollm":  # SmolLM-135M's layer: d 576, 9 heads x 64, 3 kv heads, d_ff 1536 -> qkv_attn_kernel<..., 16, ...>
        cfg = llama.preset("smollm-135m", max_seq)
    elif name == "l7":  # Llama-2-7B's layer: d 4096, 32 x 128, K-on-lanes weights -> qkv_attn_kon_kernel<32, ...> (256-thread workgroups)
        cfg = llama.preset("llama2-7b", max_seq)
    else:  # d_head 128 with d_model <= 2048: 2048 wide, 16 x 128, 4 kv heads -> qkv_attn_kernel<..., 32, ...>
        cfg = llama.preset("smollm-135m", max_seq)
        cfg.d_model, cfg.n_heads, cfg.n_kv_heads, cfg.d_ff = 2048, 16, 4, 2048
    cfg.n_layers, cfg.vocab_size, cfg.kv_quant_block = 2, 2048, kvq
    return cfg


def _dispatches(hip_backend, s):
    return capi.load_hip().zgml_hip_get_runtime_profile(hip_backend.ctx, s.handle).contents.backend_dispatch_count


@pytest.mark.parametrize("kvq", [0, 32])
@pytest.mark.parametrize("name", ["smollm", "dh128", "l7"])
def test_fused_qkv_attention_with_the_split_active(hip_backend, oracle, name, kvq):
    """148 positions with the split threshold at 32 keys: from position 63 on every head's context is spread over several
    consumer workgroups INSIDE the fused launch. Logits within 2e-4 of the oracle's range (1e-3 with int8 KV: a 1-ulp
    difference in a projected key can straddle storeColumn's truncation), greedy tokens equal wherever the oracle's top-2
    margin is not a tie, the resident loop (graph replays: counters and `seen` words carry over) gives the vtable's tokens,
    and the same program with the fusion refused (resident-workgroup capacity 0) takes one launch more per layer and
    agrees to 2e-5."""
    cfg = _cfg(name, 256, kvq)
    oracle.set_threads(8)
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    n = 148
    s_ref = llama.Session(m, oracle.backend_fns())
    hip_backend.set_option(capi.OPT_ATTN_SPLIT_MIN_KEYS, 32)
    try:
        s_f = llama.Session(m, llama.hip_backend_fns(hip_backend))
        tok, toks, logits = 3, [], []
        # (int8 KV at 7B dimensions: the synthetic projections sit on storeColumn's truncation boundaries, a 1-ulp difference flips
        # 5-24 cache units per head — up to 4.2e-3 of the range over these positions, bit for bit the same with the fusion on
        # and off: the launch form is not what is being bounded there; tests/test_hip_longctx.py documents the effect)
        tol = (6e-3 if name == "l7" else 1e-3) if kvq else 2e-4
        for pos in range(n):
            t_ref, l_ref = s_ref.step(tok, pos)
            t_f, l_f = s_f.step(tok, pos)
            assert not hip_backend.last_error(), hip_backend.last_error()
            scale = np.abs(l_ref).max()
            assert np.isfinite(l_f).all() and np.abs(l_f - l_ref).max() / scale < tol, pos
            top2 = np.sort(l_ref)[-2:]
            if top2[1] - top2[0] > 5 * tol * scale:
                assert t_f == t_ref, pos
            tok = t_ref
            toks.append(tok)
            logits.append(l_f)
        fused_launches = _dispatches(hip_backend, s_f) // n
        # the resident loop on the same program: the tokens it feeds itself, position by position, against the vtable's
        s_f.resident_setup(hip_backend)
        got = s_f.resident_decode(3, 0, n)
        want, _ = s_f.decode(3, 0, n)
        assert got.tolist() == want.tolist()
        s_f.close()
        # the same program with the fusion refused: one launch more per layer, same numbers
        hip_backend.set_option(capi.OPT_FUSE_RESIDENT_WGS, 0)
        try:
            s_u = llama.Session(m, llama.hip_backend_fns(hip_backend))
            tok = 3
            for pos in range(n):
                t_u, l_u = s_u.step(tok, pos)
                scale = np.abs(logits[pos]).max()
                assert np.abs(l_u - logits[pos]).max() / scale < (tol if kvq else 2e-5), pos
                tok = toks[pos]
            # (K-on-lanes models: the first layer's q / k / v launch computes its rmsnorm itself and stays apart; and the diagnostic
            # switches that take away the layout, the prepared norm or this fusion leave nothing to refuse)
            import os
            kon_off = any(os.environ.get(v) == "0" for v in ("ZGML_HIP_FUSE_QKV_ATTN_KON", "ZGML_HIP_QMV_KON", "ZGML_HIP_PRENORM"))
            more = (0 if kon_off else cfg.n_layers - 1) if name == "l7" else cfg.n_layers
            assert _dispatches(hip_backend, s_u) // n == fused_launches + more
            s_u.close()
        finally:
            hip_backend.set_option(capi.OPT_FUSE_RESIDENT_WGS, -1)
    finally:
        hip_backend.set_option(capi.OPT_ATTN_SPLIT_MIN_KEYS, -1)
    s_ref.close(), m.close()


def test_residency_guard_shrinks_the_split_and_refuses(hip_backend):
    """Capacity = projection workgroups + one per head: the fusion is built with ONE split per head; one workgroup less
    and it is refused (two launches). Tokens never change. SmolLM layer: 60 projection workgroups + 9 heads."""
    cfg = _cfg("smollm", 256, 0)
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    n, n_mv = 80, (576 + 192 + 192) // 16
    hip_backend.set_option(capi.OPT_ATTN_SPLIT_MIN_KEYS, 32)
    try:
        counts, tokens = {}, {}
        for cap in (-1, n_mv + 9, n_mv + 8):
            hip_backend.set_option(capi.OPT_FUSE_RESIDENT_WGS, cap)
            s = llama.Session(m, llama.hip_backend_fns(hip_backend))
            tokens[cap], _ = s.decode(3, 0, n)
            assert not hip_backend.last_error(), hip_backend.last_error()
            counts[cap] = _dispatches(hip_backend, s) // n
            s.close()
        assert counts[n_mv + 9] == counts[-1] and counts[n_mv + 8] == counts[-1] + cfg.n_layers
        assert tokens[n_mv + 9].tolist() == tokens[-1].tolist() == tokens[n_mv + 8].tolist()
    finally:
        hip_backend.set_option(capi.OPT_FUSE_RESIDENT_WGS, -1)
        hip_backend.set_option(capi.OPT_ATTN_SPLIT_MIN_KEYS, -1)
    m.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("model", ["smollm", "l7"])
def test_handoff_timeout_is_loud_and_falls_back(model):
    """One producer never signals (debug switch): the bounded wait gives up, execute_program's synchronisation reports it,
    the context switches the fusion off, the rebuilt plan (one launch more per layer) gives the two-launch plan's tokens."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    worker = Path(__file__).parent / "handoff_timeout_worker.py"
    if model == "l7" and any(os.environ.get(v) == "0" for v in ("ZGML_HIP_FUSE_QKV_ATTN_KON", "ZGML_HIP_QMV_KON", "ZGML_HIP_PRENORM")):
        pytest.skip("the diagnostic switch under test turns the fused K-on-lanes launch off")
    # the switch that suppresses one publish exists only in the diagnostics library (-DZGML_TRACE): the product build cannot drop it
    trace_lib = Path(__file__).resolve().parent.parent / "zgml_amd" / "lib" / "libzgml_hip_trace.so"
    assert trace_lib.exists(), "run __graft_entry__.build(): it builds the diagnostics library too"
    env = dict(os.environ, ZGML_HIP_DEBUG_DROP_PUBLISH="1", ZGML_TEST_HANDOFF_MODEL=model, ZGML_HIP_LIB=str(trace_lib))
    r = subprocess.run([sys.executable, str(worker)], capture_output=True, text=True, timeout=580, env=env)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("HANDOFF_RESULT ")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads(line[0][len("HANDOFF_RESULT "):])
    assert not out["plain_err"]
    assert "hand-off" in out["first_err"] and "execute_program" in out["first_err"], out
    assert not out["second_err"] and not out["third_err"], out
    assert out["launches_after"] == out["plain_launches"], out
    assert out["tokens_equal"] and out["resident_equal"], out
    assert out["kv_equal_after_rerun"], out  # re-running the failed step at the same position rewrites its KV column (zgml_hip.h)


@pytest.mark.timeout(600)
def test_attention_plus_o_projection_launch():
    """The decode attention + O projection in one launch of 256-thread workgroups (attn_o_kon_kernel; measured slower than two
    launches and therefore off by default): parity at Llama-2-7B dimensions against the oracle fixtures, short and long context."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    if os.environ.get("ZGML_HIP_QMV_KON") == "0":
        pytest.skip("the launch is built from the K-on-lanes mat-vec, which the diagnostic switch under test turns off")
    worker = Path(__file__).parent / "attn_o_worker.py"
    trace_lib = Path(__file__).resolve().parent.parent / "zgml_amd" / "lib" / "libzgml_hip_trace.so"  # (the kernel is not in the product library)
    assert trace_lib.exists(), "run __graft_entry__.build(): it builds the diagnostics library too"
    r = subprocess.run([sys.executable, str(worker)], capture_output=True, text=True, timeout=580,
                       env=dict(os.environ, ZGML_HIP_FUSE_ATTN_O="1", ZGML_HIP_LIB=str(trace_lib)))
    assert r.returncode == 0 and "ATTN_O_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
