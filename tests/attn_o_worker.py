"""Worker for tests/test_hip_fused_qkv.py::test_attention_plus_o_projection_launch — own process: the switch that builds the
(measured slower, off by default) attention + O projection launch, ZGML_HIP_FUSE_ATTN_O=1, is read once per process. Runs bench.py's
own gates at Llama-2-7B dimensions (2 layers: oracle fixture, vtable logits + resident tokens, short and long context)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from zgml_amd import Backend, capi, llama  # noqa: E402

be = Backend(0)
bench.verify_l7_shapes(be, llama)
bench.verify_longctx(be, llama, "l7dims_f32kv")
# the fused launch was really built: one launch fewer per layer than with the fusion refused
cfg = llama.preset("llama2-7b", 512)
cfg.n_layers = 2
m = llama.Model(cfg, llama.Q4_0, threads=8)
hip = capi.load_hip()
counts = []
for cap in (-1, 0):
    be.set_option(capi.OPT_FUSE_RESIDENT_WGS, cap)
    s = llama.Session(m, llama.hip_backend_fns(be))
    s.decode(1, 0, 4)
    counts.append(hip.zgml_hip_get_runtime_profile(be.ctx, s.handle).contents.backend_dispatch_count // 4)
    s.close()
assert counts[1] == counts[0] + cfg.n_layers, counts
assert not be.last_error(), be.last_error()
print("ATTN_O_OK", counts)
