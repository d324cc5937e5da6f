#!/usr/bin/env python3
"""Generate tests/golden/l7dims.json: greedy tokens + logit probes of the ORACLE on Llama-2-7B-dimension programs
with a reduced layer count (d_model 4096, 32 heads x 128, d_ff 11008, vocab 32000; the synthetic weight generator
of SURVEY §8d, so nothing but this script and the oracle is needed to regenerate it).

    python tools/gen_golden_l7.py            # ~2-4 minutes on 8 cores

The fixture is what `bench.py`'s Llama-2-7B legs and tests/test_hip_l7dims.py check the HIP path against at these
shapes (the GPU box has no time for the oracle at 4096 x 32000 per token inside every test; it still re-runs the
oracle for the short cases). Cases:
  decode_f32kv / decode_int8kv : n_layers = 2, max_seq 512, tokens fed greedily from token 1, positions 0..11
  decode_long                  : n_layers = 1, max_seq 512, fixed token stream to position 300 (past the attention
                                 split threshold of 2 x 128 keys), greedy tokens of the last 4 positions
  prefill32_q4_0 / prefill32_f16: n_layers = 2, one token_len = 32 chunk at position 0, then a second at 32
"""
from __future__ import annotations

import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import oracle as O  # noqa: E402
from zgml_amd import llama  # noqa: E402


def l7cfg(n_layers, max_seq=512, kvq=0):
    c = llama.preset("llama2-7b", max_seq)
    c.n_layers = n_layers
    c.kv_quant_block = kvq
    return c


def probe(logits):
    """A few numbers that pin the logits without storing 32000 floats: top-8 indices, their values, range, sum."""
    top = np.argsort(-logits, kind="stable")[:8]
    return {"top8": [int(i) for i in top], "top8_vals": [float(logits[i]) for i in top], "min": float(logits.min()),
            "max": float(logits.max()), "sum": float(np.sum(logits, dtype=np.float64)),
            "abs_sum": float(np.sum(np.abs(logits), dtype=np.float64))}


def decode_case(cfg, n):
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    s = llama.Session(m, O.backend_fns())
    tok, toks, probes = 1, [], []
    for pos in range(n):
        tok, logits = s.step(tok, pos)
        toks.append(tok)
        probes.append(probe(logits))
    s.close()
    m.close()
    return {"first_token": 1, "tokens": toks, "probes": probes}


def long_case(cfg, n_pos, tail):
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    s = llama.Session(m, O.backend_fns())
    out, probes = [], []
    for pos in range(n_pos):
        tok_in = (7 * pos + 3) % cfg.vocab_size
        want = pos >= n_pos - tail
        t, logits = s.step(tok_in, pos, want_logits=want)
        if want:
            out.append(t)
            probes.append(probe(logits))
    s.close()
    m.close()
    return {"stream": "token(pos) = (7 pos + 3) mod vocab", "n_pos": n_pos, "tail": tail, "tokens": out, "probes": probes}


def prefill_case(cfg, kind, T):
    dense = kind == "f16"
    O.set_f16_dense(dense)
    try:
        m = llama.Model(cfg, llama.F32_DENSE if dense else llama.Q4_0, threads=8, token_len=T)
        s = llama.Session(m, O.backend_fns())
        res = []
        for chunk in range(2):
            toks = [(7 * (i + chunk * T) + 3) % cfg.vocab_size for i in range(T)]
            t, logits = s.prefill(toks, chunk * T)
            res.append({"pos": chunk * T, "token": t, "probe": probe(logits)})
        s.close()
        m.close()
    finally:
        O.set_f16_dense(False)
    return {"T": T, "stream": "token(i) = (7 i + 3) mod vocab", "chunks": res}


def full_model():
    """--full: the whole 32-layer Llama-2-7B Q4_0 program (max_seq 2048, as bench.py builds it), 4 greedy positions
    from token 1 -> tests/golden/l7full.json (bench.py checks the first tokens of its timed model against it)."""
    O.set_threads(8)
    t0 = time.time()
    cfg = llama.preset("llama2-7b", 2048)
    res = decode_case(cfg, 4)
    out = {"generator": "tools/gen_golden_l7.py --full", "n_layers": int(cfg.n_layers), "max_seq": 2048, **res}
    dst = ROOT / "tests" / "golden" / "l7full.json"
    dst.write_text(json.dumps(out, indent=1))
    print("wrote", dst, round(time.time() - t0, 1), "s", res["tokens"])


def add_prefill128():
    """--prefill128: add the token_len = 128 chunk case (the reference's prefill chunk size) to l7dims.json."""
    O.set_threads(8)
    dst = ROOT / "tests" / "golden" / "l7dims.json"
    out = json.loads(dst.read_text())
    t0 = time.time()
    out["prefill128_q4_0"] = dict(n_layers=2, **prefill_case(l7cfg(2), "q4_0", 128))
    dst.write_text(json.dumps(out, indent=1))
    print("prefill128_q4_0", round(time.time() - t0, 1), "s; wrote", dst)


def main():
    if "--full" in sys.argv:
        return full_model()
    if "--prefill128" in sys.argv:
        return add_prefill128()
    O.set_threads(8)
    out = {"generator": "tools/gen_golden_l7.py", "dims": {"d_model": 4096, "n_heads": 32, "n_kv_heads": 32, "d_ff": 11008,
                                                            "vocab": 32000, "max_seq": 512}}
    t0 = time.time()
    out["decode_f32kv"] = dict(n_layers=2, **decode_case(l7cfg(2), 12))
    print("decode_f32kv", round(time.time() - t0, 1), "s", out["decode_f32kv"]["tokens"], flush=True)
    out["decode_int8kv"] = dict(n_layers=2, **decode_case(l7cfg(2, kvq=32), 12))
    print("decode_int8kv", round(time.time() - t0, 1), "s", out["decode_int8kv"]["tokens"], flush=True)
    out["decode_long"] = dict(n_layers=1, **long_case(l7cfg(1), 301, 4))
    print("decode_long", round(time.time() - t0, 1), "s", out["decode_long"]["tokens"], flush=True)
    out["prefill32_q4_0"] = dict(n_layers=2, **prefill_case(l7cfg(2), "q4_0", 32))
    print("prefill32_q4_0", round(time.time() - t0, 1), "s", flush=True)
    out["prefill32_f16"] = dict(n_layers=2, **prefill_case(l7cfg(2), "f16", 32))
    print("prefill32_f16", round(time.time() - t0, 1), "s", flush=True)
    dst = ROOT / "tests" / "golden" / "l7dims.json"
    dst.write_text(json.dumps(out, indent=1))
    print("wrote", dst)


if __name__ == "__main__":
    main()
