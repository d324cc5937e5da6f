#!/usr/bin/env python3
"""Generate tests/golden/longctx.json: what the ORACLE computes at position ~1900 of 2048 — the regime bench.py's
`long_context_pos1900_*` legs time (every head's context split over several workgroups; for SmolLM inside the fused
q/k/v + attention launch). Deterministic synthetic weights: this script + the oracle regenerate it.

    python tools/gen_golden_longctx.py          # ~1-2 minutes on 8 cores

Procedure of every case (bench.py and tests/test_hip_longctx.py replay it on the HIP path): a FRESH session (KV caches
zero, as OwnedBufferTable.init leaves them, reference.zig:81-97); 4 greedy steps from token 1 at positions 0..3; then 4
greedy steps at positions 1900..1903 starting from the last token (the attention reads 1901+ cache columns, all but the
first four and the new ones zero). Recorded: the 8 greedy tokens and a logits probe of each of the 4 long-context steps.
Cases: SmolLM-135M (all 30 layers) with f32 and int8 KV caches; 2 layers at Llama-2-7B dimensions, f32 and int8 KV.

`*_filled` cases (round 4; `--filled` adds them to the existing file, ~6 minutes on 8 cores): the same four programs with a
FILLED cache — the reference's own flow, `LlamaInferenceSession.prefill` in chunks of 128 (src/llama_inference.zig:474) and then
`step`: 15 chunks of a token_len = 128 plan over the stream token(i) = (7 i + 3) mod vocab (positions 0..1919, every cache
column a real key / value), the caches handed to the decode plan, then 4 greedy steps at 1920..1923 from the last chunk's
greedy token. Recorded: the greedy token of every chunk, the logits probe of the last chunk and of each decode step."""
from __future__ import annotations

import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import oracle as O  # noqa: E402
from tools.gen_golden_l7 import probe  # noqa: E402
from zgml_amd import llama  # noqa: E402

POS = 1900


def case(cfg):
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    s = llama.Session(m, O.backend_fns())
    tok, toks, probes = 1, [], []
    for pos in list(range(4)) + list(range(POS, POS + 4)):
        tok, logits = s.step(tok, pos)
        toks.append(tok)
        if pos >= POS:
            probes.append(probe(logits))
    s.close()
    m.close()
    return {"first_token": 1, "positions": list(range(4)) + list(range(POS, POS + 4)), "tokens": toks, "probes": probes}


FILL_T, FILL_CHUNKS = 128, 15


def fill_stream(i, vocab):
    return (7 * i + 3) % vocab


def filled_case(cfg):
    m_pre, m_dec = llama.Model(cfg, llama.Q4_0, threads=8, token_len=FILL_T), llama.Model(cfg, llama.Q4_0, threads=8)
    s_pre, s_dec = llama.Session(m_pre, O.backend_fns()), llama.Session(m_dec, O.backend_fns())
    be = O.OracleBackend()
    chunk_tokens = []
    for c in range(FILL_CHUNKS):
        toks = [fill_stream(c * FILL_T + i, cfg.vocab_size) for i in range(FILL_T)]
        tok, logits = s_pre.prefill(toks, c * FILL_T, want_logits=(c == FILL_CHUNKS - 1))
        chunk_tokens.append(tok)
    pre_probe = probe(logits)
    for (bp, n), (bd, n2) in zip(m_pre.kv_buffers(), m_dec.kv_buffers()):  # the hand-off: the two plans own separate buffers
        assert n == n2
        be.buffer(s_dec.handle, bd)[:n] = be.buffer(s_pre.handle, bp)[:n]
    pos0 = FILL_T * FILL_CHUNKS
    toks, probes = [], []
    for pos in range(pos0, pos0 + 4):
        tok, logits = s_dec.step(tok, pos)
        toks.append(tok)
        probes.append(probe(logits))
    for x in (s_pre, s_dec, m_pre, m_dec):
        x.close()
    return {"stream": "token(i) = (7 i + 3) mod vocab", "chunk": FILL_T, "n_chunks": FILL_CHUNKS, "chunk_tokens": chunk_tokens,
            "prefill_probe": pre_probe, "positions": list(range(pos0, pos0 + 4)), "tokens": toks, "probes": probes}


def add_filled():
    O.set_threads(8)
    dst = ROOT / "tests" / "golden" / "longctx.json"
    out = json.loads(dst.read_text())
    t0 = time.time()
    for name, model, nl, kvq in (("smollm_f32kv_filled", "smollm-135m", 0, 0), ("smollm_int8kv_filled", "smollm-135m", 0, 32),
                                 ("l7dims_f32kv_filled", "llama2-7b", 2, 0), ("l7dims_int8kv_filled", "llama2-7b", 2, 32)):
        cfg = llama.preset(model, 2048)
        if nl:
            cfg.n_layers = nl
        cfg.kv_quant_block = kvq
        out[name] = dict(model=model + (" dimensions" if nl else ""), n_layers=int(cfg.n_layers), kv_quant_block=kvq, **filled_case(cfg))
        print(name, round(time.time() - t0, 1), "s", out[name]["chunk_tokens"][-3:], out[name]["tokens"], flush=True)
        dst.write_text(json.dumps(out, indent=1))
    print("wrote", dst)


def main():
    if "--filled" in sys.argv:
        return add_filled()
    O.set_threads(8)
    out = {"generator": "tools/gen_golden_longctx.py", "pos": POS}
    t0 = time.time()
    for name, kvq in (("smollm_f32kv", 0), ("smollm_int8kv", 32)):
        cfg = llama.preset("smollm-135m", 2048)
        cfg.kv_quant_block = kvq
        out[name] = dict(model="smollm-135m", n_layers=int(cfg.n_layers), kv_quant_block=kvq, **case(cfg))
        print(name, round(time.time() - t0, 1), "s", out[name]["tokens"], flush=True)
    for name, kvq in (("l7dims_f32kv", 0), ("l7dims_int8kv", 32)):
        cfg = llama.preset("llama2-7b", 2048)
        cfg.n_layers, cfg.kv_quant_block = 2, kvq
        out[name] = dict(model="llama2-7b dimensions", n_layers=2, kv_quant_block=kvq, **case(cfg))
        print(name, round(time.time() - t0, 1), "s", out[name]["tokens"], flush=True)
    dst = ROOT / "tests" / "golden" / "longctx.json"
    dst.write_text(json.dumps(out, indent=1))
    print("wrote", dst)


if __name__ == "__main__":
    main()
