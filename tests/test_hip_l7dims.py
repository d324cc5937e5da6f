"""Parity at Llama-2-7B dimensions (BASELINE configs[2] and configs[4]: d_model 4096, 32 heads x 128, d_ff 11008,
vocab 32000; op stream src/models/llama_transformer.zig:192-253) with a reduced layer count, so the kernel
instances the 7B numbers come from — the LDS-staged rmsnorm prologue at K = 4096, grouped q/k/v and gate/up
launches, the K = 11008 down projection with its residual epilogue, `attention_decode_kernel<32>` (f32 and int8
KV, one position past the attention-split threshold), the 32000-column head and the M = 32 tile kernels — are
compared with the oracle at THEIR shapes.

Two references: the oracle run here on the same program (full logits, 2e-4 of the logit range — the bar of
tests/test_hip_llama.py) and the committed fixture tests/golden/l7dims.json (tools/gen_golden_l7.py: the oracle's
greedy tokens and logit probes), which is also what bench.py checks before it times the 7B legs."""
import json
from pathlib import Path

import numpy as np
import pytest

from zgml_amd import capi, llama
from tests.test_hip_qmatvec import TOL, bound, run_both

pytestmark = pytest.mark.gpu
GOLD = json.loads((Path(__file__).parent / "golden" / "l7dims.json").read_text())


def l7cfg(n_layers, max_seq=512, kvq=0):
    c = llama.preset("llama2-7b", max_seq)
    c.n_layers = n_layers
    c.kv_quant_block = kvq
    return c


def check_probe(logits, want, tol=2e-4):
    """Logits against a fixture probe: the oracle's top-8 values at the oracle's indices, the extrema and the sums,
    within `tol` of the logit range (sums: relative), and the same greedy token."""
    rng = want["max"] - want["min"]
    got_vals = logits[np.array(want["top8"])]
    assert np.abs(got_vals - np.array(want["top8_vals"], np.float32)).max() <= tol * rng
    assert abs(float(logits.max()) - want["max"]) <= tol * rng and abs(float(logits.min()) - want["min"]) <= tol * rng
    assert abs(float(np.sum(np.abs(logits), dtype=np.float64)) - want["abs_sum"]) <= tol * want["abs_sum"]
    assert int(np.argmax(logits)) == want["top8"][0]


@pytest.mark.parametrize("kvq", [0, 32])
def test_decode_two_layers_matches_oracle_and_fixture(hip_backend, oracle, kvq):
    """f32 KV: 2e-4 of the logit range. int8 KV: storeColumn TRUNCATES x * 127 / max to an integer, so a 1-ulp f32
    difference in a projected key / value (summation order of the mat-vec) that straddles an integer moves the cached
    value by a whole quantisation step (1/127 of its block's maximum); over 2 x 32 heads x 128 dims x 12 positions a
    few such flips happen, hence 1e-3 there (measured 2.1e-4; the f16-promoted path has the same kind of bound)."""
    tol = 1e-3 if kvq else 2e-4
    gold = GOLD["decode_int8kv" if kvq else "decode_f32kv"]
    oracle.set_threads(16)
    m = llama.Model(l7cfg(2, kvq=kvq), llama.Q4_0, threads=16)
    s_hip = llama.Session(m, llama.hip_backend_fns(hip_backend))
    s_ref = llama.Session(m, oracle.backend_fns())
    tok = gold["first_token"]
    for pos in range(len(gold["tokens"])):
        t_hip, l_hip = s_hip.step(tok, pos)
        assert not hip_backend.last_error(), hip_backend.last_error()
        check_probe(l_hip, gold["probes"][pos], tol)
        assert t_hip == gold["tokens"][pos]
        if pos < 4:  # full logits against the oracle run here
            t_ref, l_ref = s_ref.step(tok, pos)
            assert np.abs(l_hip - l_ref).max() <= tol * np.abs(l_ref).max() and t_hip == t_ref
        tok = t_hip
    # the device-resident loop (what bench.py times) produces the same tokens at these shapes
    s_hip.resident_setup(hip_backend)
    got = s_hip.resident_decode(gold["first_token"], 0, len(gold["tokens"]))
    assert got.tolist() == gold["tokens"]
    s_hip.close(), s_ref.close(), m.close()


def test_decode_past_the_attention_split_threshold(hip_backend):
    """One layer, a fixed token stream to position 300: seq_kv >= 2 x 128 keys, so every head's context is split
    over several workgroups (flash-decoding partials + last-arriver merge). Against the fixture, and against the
    same program with the split switched off."""
    gold = GOLD["decode_long"]
    n_pos, tail = gold["n_pos"], gold["tail"]

    def run(split_min_keys):
        hip_backend.set_option(capi.OPT_ATTN_SPLIT_MIN_KEYS, split_min_keys)
        try:
            m = llama.Model(l7cfg(1), llama.Q4_0, threads=16)
            s = llama.Session(m, llama.hip_backend_fns(hip_backend))
            out = []
            for pos in range(n_pos):
                want = pos >= n_pos - tail
                t, logits = s.step((7 * pos + 3) % m.cfg.vocab_size, pos, want_logits=want)
                if want:
                    out.append((t, logits))
            s.close(), m.close()
            return out
        finally:
            hip_backend.set_option(capi.OPT_ATTN_SPLIT_MIN_KEYS, -1)

    split = run(128)
    for (t, logits), tok, probe in zip(split, gold["tokens"], gold["probes"]):
        check_probe(logits, probe)
        assert t == tok
    plain = run(0)
    for (t, logits), (t2, logits2) in zip(split, plain):
        assert t == t2 and np.abs(logits - logits2).max() <= 1e-5 * np.abs(logits2).max()


def test_prefill_chunk128_q4_0_matches_fixture(hip_backend):
    """The reference's prefill chunk (128 tokens, llama_inference.zig:474) at 7B shapes: the M > 32 tile kernel
    (qmatmul_xdl4_kernel: 8 m-tiles per workgroup, K split with a last-arriver fan-in on the 4096-wide projections)
    and 128-query attention, two chunks (the second reads 256 cached keys), against the oracle's fixture; the
    device-resident chunk returns the same tokens."""
    gold = GOLD["prefill128_q4_0"]
    T = gold["T"]
    m = llama.Model(l7cfg(2), llama.Q4_0, threads=16, token_len=T)
    s = llama.Session(m, llama.hip_backend_fns(hip_backend))
    r = llama.Session(m, llama.hip_backend_fns(hip_backend))
    r.resident_setup(hip_backend)
    for ci, ch in enumerate(gold["chunks"]):
        toks = [(7 * (i + ci * T) + 3) % m.cfg.vocab_size for i in range(T)]
        t_hip, l_hip = s.prefill(toks, ch["pos"])
        assert not hip_backend.last_error(), hip_backend.last_error()
        check_probe(l_hip, ch["probe"])
        assert t_hip == ch["token"]
        assert r.resident_prefill(toks, ch["pos"]) == ch["token"]
    s.close(), r.close(), m.close()


@pytest.mark.parametrize("kind", ["q4_0", "f16"])
def test_prefill_chunk32_matches_oracle_and_fixture(hip_backend, oracle, kind):
    """BASELINE configs[4] at its shapes: one token_len = 32 chunk at position 0 and a second at 32 through the tile
    kernels (Q4_0: split_a + qmatmul_xdl2; f16: pack_a + dense_f16_tile2). f16 promotion rounds A and B to f16 on
    both sides; a 1-ulp f32 difference in A can flip its rounding, hence 2e-3 there (tests/test_hip_llama.py)."""
    gold = GOLD["prefill32_" + kind]
    dense = kind == "f16"
    tol = 2e-3 if dense else 2e-4
    T = gold["T"]
    hip_backend.set_option(capi.OPT_F16_DENSE_WEIGHTS, int(dense))
    oracle.set_f16_dense(dense)
    oracle.set_threads(16)
    try:
        m = llama.Model(l7cfg(2), llama.F32_DENSE if dense else llama.Q4_0, threads=16, token_len=T)
        s_hip = llama.Session(m, llama.hip_backend_fns(hip_backend))
        s_ref = llama.Session(m, oracle.backend_fns())
        for ci, ch in enumerate(gold["chunks"]):
            toks = [(7 * (i + ci * T) + 3) % m.cfg.vocab_size for i in range(T)]
            t_hip, l_hip = s_hip.prefill(toks, ch["pos"])
            assert not hip_backend.last_error(), hip_backend.last_error()
            check_probe(l_hip, ch["probe"], tol)
            assert t_hip == ch["token"]
            if ci == 0:
                t_ref, l_ref = s_ref.prefill(toks, 0)
                assert np.abs(l_hip - l_ref).max() <= tol * np.abs(l_ref).max() and t_hip == t_ref
        s_hip.close(), s_ref.close(), m.close()
    finally:
        hip_backend.set_option(capi.OPT_F16_DENSE_WEIGHTS, 0)
        oracle.set_f16_dense(False)


@pytest.mark.parametrize("K,N", [(11008, 4096), (4096, 32000)])
def test_matvec_q4_0_at_the_remaining_7b_shapes(hip_backend, oracle, K, N):
    """M = 1 against the oracle at the down-projection and LM-head shapes (4096 x 4096 and 4096 x 11008 are in
    tests/test_hip_qmatvec.py)."""
    rng = np.random.default_rng(0x7B + K + N)
    x = rng.standard_normal(K).astype(np.float32)
    data = rng.integers(-8, 8, K * N).astype(np.int8)
    scales = (rng.random(K * N // 32).astype(np.float16) * 0.05 + 0.001).astype(np.float32)
    oracle.set_threads(16)
    want, got = run_both(hip_backend, oracle, data, scales, x, 1, N, K)
    b = bound(data, scales, x, 1, N, K, 32).ravel()
    assert np.all(np.abs(got[:N] - want[:N]) <= TOL * b + 1e-30), np.max(np.abs(got[:N] - want[:N]) / (b + 1e-30))
    assert np.all(got[N:] == -7)
