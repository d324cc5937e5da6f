"""Worker for tests/test_hip_sharded.py: the row-shard step behind the C ABI at WORLD SIZE 2 on ONE GPU, gathers by peer stores
(zgml_amd/csrc/shard_peer.hip), the greedy token as a (max, index) pair per rank.

  mode "threads" (default): two contexts of this process act as ranks 0 / 1 (the peers' blocks are plain device pointers),
      stepped from two host threads — each rank's gather kernel waits (bounded) for the other's push, so the two streams must
      be in flight together;
  mode "procs": rank 0 = this process, rank 1 = a child process on the same GPU; the blocks cross the process boundary as
      hipIpc handles over a pipe — the path `bench.py --gpus N` takes with ZGML_SHARD_GATHER=peer.
Checked: the tokens of both ranks equal the unsharded program's, step by step, in graph and eager form; a rank whose peer never
shows up gives up loudly (ZGML_SHARD_PEER_WAIT_MS) instead of hanging. Semantics of the sharding: SURVEY section 8e; argmax
src/nn.zig:122-138."""
import os
import subprocess
import sys
import threading
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

N_STEPS = 10


def config(llama, name):
    if name == "l7":  # 2 layers at Llama-2-7B dimensions: the real message sizes (2 KB .. 5.5 KB per rank at world 2 it is 8 / 22 KB)
        cfg = llama.preset("llama2-7b", 64)
        cfg.n_layers, cfg.vocab_size = 2, 4096
    else:
        cfg = llama.preset("tiny")
        cfg.tied_lm_head, cfg.n_kv_heads = 0, 4
    return cfg


def reference_tokens(be, llama, cfg):
    m = llama.Model(cfg, llama.Q4_0, threads=8)
    s = llama.Session(m, llama.hip_backend_fns(be))
    toks, tok = [], 3
    for pos in range(N_STEPS):
        tok, _ = s.step(tok, pos)
        toks.append(tok)
    s.close()
    m.close()
    return toks


def rank_model(llama, cfg, rank, world):
    import copy
    c = copy.copy(cfg)
    c.shard_rank, c.shard_world = rank, world
    return llama.Model(c, llama.Q4_0, threads=8)


def run_threads(name):
    from zgml_amd import Backend, capi, llama
    from zgml_amd.sharded import NativeShardedDecoder
    be = Backend(0)
    cfg = config(llama, name)
    want = reference_tokens(be, llama, cfg)
    be.close()
    for graph in (1, 0):
        bes = [Backend(0), Backend(0)]
        for b in bes:
            b.set_option(capi.OPT_GRAPH, graph)
        models = [rank_model(llama, cfg, r, 2) for r in range(2)]
        decs = [NativeShardedDecoder(bes[r], models[r], r, 2, gather="peer") for r in range(2)]
        handles = [d.export_handle() for d in decs]
        for d in decs:
            for r, h in enumerate(handles):
                d.import_handle(r, h)
        got = [[], []]
        errs = []

        def loop(r):
            try:
                tok = 3
                for pos in range(N_STEPS):
                    tok = decs[r].step(tok, pos)
                    got[r].append(tok)
            except Exception as e:  # noqa: BLE001
                errs.append((r, repr(e)))
        th = [threading.Thread(target=loop, args=(r,)) for r in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        assert got[0] == want and got[1] == want, (graph, got, want)
        assert decs[0].mode == ("graph" if graph else "eager")
        n_pts = decs[0].n_points
        for d in decs:
            d.close()
        for m in models:
            m.close()
        for b in bes:
            assert not b.last_error(), b.last_error()
            b.close()
    print(f"PEER_THREADS_OK model={name} points={n_pts}")


def run_lonely():
    """A rank whose peer never steps: the wait gives up after ZGML_SHARD_PEER_WAIT_MS and the step fails loudly."""
    os.environ["ZGML_SHARD_PEER_WAIT_MS"] = "200"
    from zgml_amd import Backend, llama
    from zgml_amd.sharded import NativeShardedDecoder
    cfg = config(llama, "tiny")
    bes = [Backend(0), Backend(0)]
    models = [rank_model(llama, cfg, r, 2) for r in range(2)]
    decs = [NativeShardedDecoder(bes[r], models[r], r, 2, gather="peer") for r in range(2)]
    handles = [d.export_handle() for d in decs]
    for d in decs:
        for r, h in enumerate(handles):
            d.import_handle(r, h)
    import time
    try:
        decs[0].step(3, 0)
        raise SystemExit("a lonely rank's step succeeded")
    except RuntimeError as e:
        assert "peer gather" in str(e) and "timed out" in str(e), str(e)  # (the PEER cause: the q/k/v + attention fusion of the context stays on)
    # ... and the rank that was merely late must not get a token out of the failed rank's stale slices: rank 0's
    # give-up poisoned every block, so rank 1's step fails too — at once, not after its own bound
    t0 = time.perf_counter()
    try:
        tok = decs[1].step(3, 0)
        raise SystemExit(f"the late rank's step returned token {tok} although its peer had failed the step")
    except RuntimeError as e:
        assert "peer gather" in str(e), str(e)
    assert time.perf_counter() - t0 < 0.15, "the late rank waited its own bound instead of seeing the poison"
    print("PEER_LONELY_OK")


def run_procs(name, rank):
    """rank 0 spawns rank 1; handles travel over pipes (stdin / stdout of the child, hex lines)."""
    from zgml_amd import Backend, llama
    from zgml_amd.sharded import NativeShardedDecoder
    be = Backend(0)
    cfg = config(llama, name)
    want = reference_tokens(be, llama, cfg) if rank == 0 else None
    m = rank_model(llama, cfg, rank, 2)
    dec = NativeShardedDecoder(be, m, rank, 2, gather="peer")
    mine = dec.export_handle()
    if rank == 0:
        child = subprocess.Popen([sys.executable, __file__, "procs", name, "1"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
        child.stdin.write(mine.hex() + "\n")
        child.stdin.flush()
        theirs = bytes.fromhex(child.stdout.readline().strip())
        handles = [mine, theirs]
    else:
        theirs = bytes.fromhex(sys.stdin.readline().strip())
        sys.stdout.write(mine.hex() + "\n")
        sys.stdout.flush()
        handles = [theirs, mine]
    for r, h in enumerate(handles):
        dec.import_handle(r, h)
    tok, got = 3, []
    for pos in range(N_STEPS):
        tok = dec.step(tok, pos)
        got.append(tok)
    assert not be.last_error(), be.last_error()
    if rank == 0:
        out = child.stdout.read()
        child.wait(timeout=120)
        assert child.returncode == 0 and "RANK1_TOKENS" in out, out[-2000:]
        other = [int(t) for t in out.split("RANK1_TOKENS")[1].split()[:N_STEPS]]
        assert got == want and other == want, (got, other, want)
        print(f"PEER_PROCS_OK model={name} mode={dec.mode}")
    else:
        print("RANK1_TOKENS " + " ".join(map(str, got)), flush=True)
    dec.close()
    m.close()
    be.close()


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "threads"
    if mode == "threads":
        run_threads(sys.argv[2] if len(sys.argv) > 2 else "tiny")
    elif mode == "lonely":
        run_lonely()
    else:
        run_procs(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0)
