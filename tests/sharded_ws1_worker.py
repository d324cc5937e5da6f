"""Worker for tests/test_hip_sharded.py: run in its own process because torch must initialise HIP
BEFORE libzgml_hip.so is loaded (torch bundles its own libamdhip64; loaded second it finds no GPU,
loaded first the backend library simply binds to it)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    torch.cuda.init()
    from zgml_amd import Backend, capi, llama
    from zgml_amd.sharded import HipExecutor, ShardedDecoder

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    be = Backend(0)
    cfg = llama.preset("tiny")
    cfg.tied_lm_head, cfg.n_kv_heads = 0, 4
    m = llama.Model(cfg, llama.Q4_0)
    ref = llama.Session(m, llama.hip_backend_fns(be))
    want, tok = [], 3
    for pos in range(5):
        tok, lg = ref.step(tok, pos)
        want.append((tok, lg))
    ref.close()

    be.set_option(capi.OPT_GRAPH, 0)
    ex = HipExecutor(be, m, 0)
    dec = ShardedDecoder(m, ex)
    tok = 3
    for pos in range(5):
        m.patch(tok, pos)
        ex.refresh()
        ex.upload()
        prev = 0
        with ex.collective_stream():
            for gp in dec.points:  # run the gathers even at world size 1 to cover the RCCL call path
                ex.run_ops(prev, gp.op_end - prev)
                t = ex.tensor(gp.buf, gp.offset, gp.len_per_rank)
                dist.all_gather_into_tensor(t, t)
                prev = gp.op_end
            ex.run_ops(prev, ex.n_ops - prev)
        logits = ex.download_logits()
        assert not be.last_error(), be.last_error()
        np.testing.assert_allclose(logits, want[pos][1], atol=1e-6 * np.abs(want[pos][1]).max(), rtol=0)
        nxt = be.argmax(ex.handle, m.buf("logits"), 0, cfg.vocab_size)
        assert nxt == want[pos][0], (nxt, want[pos][0])
        tok = nxt
    # and the packaged step() helper
    dec2_tok = 3
    for pos in range(3):
        lg = dec.step(dec2_tok, pos)
        np.testing.assert_allclose(lg, want[pos][1], atol=1e-6 * np.abs(want[pos][1]).max(), rtol=0)
        dec2_tok = want[pos][0]
    # one graph per token: ops + (forced) collectives + argmax recorded once, replayed per token
    os.environ["ZGML_SHARD_FORCE_GATHER"] = "1"
    dec3 = ShardedDecoder(m, ex)
    dec3.step(3, 0, download=False)  # plan + transfer tables exist before the capture
    dec3.capture(m.buf("logits"))
    tok = 3
    for pos in range(5):
        tok = dec3.step_graph(tok, pos)
        assert not be.last_error(), be.last_error()
        assert tok == want[pos][0], (pos, tok, want[pos][0])
    ex.close()
    m.close()
    be.close()
    dist.destroy_process_group()
    print("SHARDED_WS1_OK")


if __name__ == "__main__":
    main()
