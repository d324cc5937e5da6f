"""Prefill of one model at a fixed chunk width (BASELINE config 5: Llama-2-7B, batch/token_len 32,
Q4_0 vs dense f16): builds the token_len = T plan, runs it once through the vtable (with host I/O)
and then times resident replays of the same chunk (inputs already in HBM)."""
import argparse
import json
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
from zgml_amd import Backend, capi, llama  # noqa: E402


def run(be, name, kind, T, reps, max_seq, threads=16, pos=0):
    dense = kind == "f16"
    be.set_option(capi.OPT_F16_DENSE_WEIGHTS, int(dense))
    t0 = time.perf_counter()
    m = llama.Model(llama.preset(name, max_seq), llama.F32_DENSE if dense else llama.Q4_0, threads=threads, token_len=T)
    t1 = time.perf_counter()
    s = llama.Session(m, llama.hip_backend_fns(be))
    t2 = time.perf_counter()
    toks = [(7 * i + 3) % m.cfg.vocab_size for i in range(T)]
    nxt, logits = s.prefill(toks, pos)  # pos > 0: a later chunk of a long prompt (the earlier KV rows are whatever the cache holds)
    assert np.isfinite(logits).all()
    t3 = time.perf_counter()
    s.prefill(toks, pos)
    t4 = time.perf_counter()
    lib = be._lib
    lib.zgml_hip_enqueue_program(be.ctx, s.handle)
    be.synchronize()
    t5 = time.perf_counter()
    for _ in range(reps):
        lib.zgml_hip_enqueue_program(be.ctx, s.handle)
    be.synchronize()
    dt = (time.perf_counter() - t5) / reps
    err = be.last_error()
    assert not err, err
    out = {"model": name, "weights": kind, "token_len": T, "pos": pos, "ms_per_chunk": round(dt * 1e3, 3),
           "prefill_tok_s": round(T / dt, 1), "vtable_ms_per_chunk": round((t4 - t3) * 1e3, 3),
           "first_token": int(nxt), "build_s": round(t1 - t0, 1), "compile_s": round(t2 - t1, 1), "n_ops": m.program.n_ops}
    s.close()
    m.close()
    be.set_option(capi.OPT_F16_DENSE_WEIGHTS, 0)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("model", nargs="?", default="llama2-7b")
    ap.add_argument("--T", type=int, default=32)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--max-seq", type=int, default=512)
    ap.add_argument("--kinds", default="q4,f16")
    ap.add_argument("--pos", type=int, default=0, help="position of the chunk's first token (a later chunk of a long prompt)")
    a = ap.parse_args()
    be = Backend(0)
    for kind in a.kinds.split(","):
        print(json.dumps(run(be, a.model, kind, a.T, a.reps, a.max_seq, pos=a.pos)), flush=True)
    be.close()
