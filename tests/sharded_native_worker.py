"""Worker for tests/test_hip_sharded.py::test_native_shard_path_world_size_one: the row-shard step behind the C ABI
(zgml_hip_shard_*: RCCL communicator in the library, in-place ncclAllGather between op ranges, one graph per token) at
world size 1 — every call the N > 1 bench makes — against plain vtable stepping of the same program. Own process: the
RCCL communicator and its threads end with it."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    from zgml_amd import Backend, llama
    from zgml_amd.sharded import NativeShardedDecoder

    be = Backend(0)
    cfg = llama.preset("tiny")
    cfg.tied_lm_head, cfg.n_kv_heads = 0, 4  # the shard builder needs an untied head and kv heads divisible by the world
    m = llama.Model(cfg, llama.Q4_0)
    assert len(m.gather_points()) == 0  # world 1: the builder marks no gathers ...
    ref = llama.Session(m, llama.hip_backend_fns(be))
    want, tok = [], 3
    for pos in range(8):
        tok, _ = ref.step(tok, pos)
        want.append(tok)
    ref.close()
    m.close()
    # ... so force the all-gathers by building rank 0 of a 1-rank world through the sharded builder path: every gather
    # point is then a 1-rank in-place ncclAllGather (the RCCL call path, captured into the step graph)
    os.environ["ZGML_HOST_SHARD_POINTS_WORLD1"] = "1"
    m = llama.Model(cfg, llama.Q4_0)
    n_pts = len(m.gather_points())
    from zgml_amd import capi
    for graph in ("1", "0"):
        be2 = Backend(0)
        be2.set_option(capi.OPT_GRAPH, int(graph))  # 0: the step is issued eagerly (ops and collectives in stream order)
        dec = NativeShardedDecoder(be2, m, 0, 1)
        tok, got = 3, []
        for pos in range(8):
            tok = dec.step(tok, pos)
            got.append(tok)
        assert not be2.last_error(), be2.last_error()
        assert got == want, (graph, got, want)
        assert dec.mode == ("graph" if graph == "1" else "eager"), dec.mode
        dec.close()
        be2.close()
    m.close()
    be.close()
    print(f"SHARDED_NATIVE_OK points={n_pts}")


if __name__ == "__main__":
    main()
