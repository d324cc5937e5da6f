#!/usr/bin/env python3
"""Generate tests/golden/longctx.json: what the ORACLE computes at position ~1900 of 2048 — the regime bench.py's
`long_context_pos1900_*` legs time (every head's context split over several workgroups; for SmolLM inside the fused
q/k/v + attention launch). Deterministic synthetic weights: this script + the oracle regenerate it.

    python tools/gen_golden_longctx.py          # ~1-2 minutes on 8 cores

Procedure of every case (bench.py and tests/test_hip_longctx.py replay it on the HIP path): a FRESH session (KV caches
zero, as OwnedBufferTable.init leaves them, reference.zig:81-97); 4 greedy steps from token 1 at positions 0..3; then 4
greedy steps at positions 1900..1903 starting from the last token (the attention reads 1901+ cache columns, all but the
first four and the new ones zero). Recorded: the 8 greedy tokens and a logits probe of each of the 4 long-context steps.
Cases: SmolLM-135M (all 30 layers) with f32 and int8 KV caches; 2 layers at Llama-2-7B dimensions, f32 and int8 KV."""
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


def main():
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
