"""The N>1 path on CPU: 2 processes, gloo backend, each rank runs its row-shard of the decode
program on the oracle and `ShardedDecoder` all-gathers between op ranges — the same driver code
that runs over RCCL on the GPUs."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cfg(llama, rank, ws):
    c = llama.preset("tiny")
    c.tied_lm_head, c.n_kv_heads, c.shard_rank, c.shard_world = 0, 4, rank, ws
    return c


def _worker(rank, ws, port, out_path):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from zgml_amd import llama
    from zgml_amd.sharded import ShardedDecoder
    from tests.oracle_executor import OracleExecutor
    m = llama.Model(_cfg(llama, rank, ws), llama.Q4_0, threads=1)
    ex = OracleExecutor(m)
    dec = ShardedDecoder(m, ex, inplace=False)
    tok, outs = 3, []
    for pos in range(3):
        logits = dec.step(tok, pos)
        outs.append(logits.copy())
        tok = int(np.argmax(logits))
    # every rank must hold identical replicated logits
    t = torch.from_numpy(np.stack(outs))
    gathered = [torch.empty_like(t) for _ in range(ws)]
    dist.all_gather(gathered, t)
    assert all(torch.equal(g, gathered[0]) for g in gathered)
    if rank == 0:
        np.save(out_path, np.stack(outs))
    ex.close()
    m.close()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_unsharded(tmp_path):
    from zgml_amd import llama
    from oracle import oracle as O
    out = tmp_path / "sharded.npy"
    mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    got = np.load(out)
    full = llama.Model(_cfg(llama, 0, 1), llama.Q4_0, threads=1)
    s = llama.Session(full, O.backend_fns())
    tok = 3
    for pos in range(3):
        tok, logits = s.step(tok, pos)
        np.testing.assert_allclose(got[pos], logits, atol=2e-5 * np.abs(logits).max(), rtol=0)
    s.close()
    full.close()
