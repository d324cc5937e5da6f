"""Worker for tests/test_hip_fused_qkv.py::test_handoff_timeout_is_loud_and_falls_back — its own process because the switch
that suppresses one publish (ZGML_HIP_DEBUG_DROP_PUBLISH=1, set by the test) is read once per process.
A fused q/k/v + attention launch whose consumers never see one column group's counter must: finish (bounded wait), report the
time-out at the next host synchronisation of WHATEVER entry point ran it, switch the fusion off for the context, and give the
tokens of the two-launch plan from then on."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zgml_amd import Backend, capi, llama  # noqa: E402


def _download(be, hip, handle, buf, n):
    import numpy as np
    out = np.zeros(n, np.float32)
    io = (capi.ProgramIOC * 1)(capi.ProgramIOC(buf, 0, 0, out.ctypes.data, 4 * n, 0))
    hip.zgml_hip_download_outputs(be.ctx, handle, io, 1)
    return out


def main():
    be = Backend(0)
    hip = capi.load_hip()
    import os
    if os.environ.get("ZGML_TEST_HANDOFF_MODEL") == "l7":  # the 256-thread form of the launch (K-on-lanes weights): layer 1 of a 7B-dimension model
        cfg = llama.preset("llama2-7b", 64)
        cfg.n_layers, cfg.vocab_size = 2, 1024
    else:
        cfg = llama.preset("smollm-135m", 64)
        cfg.n_layers, cfg.vocab_size = 1, 1024
    m = llama.Model(cfg, llama.Q4_0, threads=4)
    out = {}
    # reference tokens: the fusion refused from the start (the dropped publish only exists inside the fused launch)
    be.set_option(capi.OPT_FUSE_RESIDENT_WGS, 0)
    s = llama.Session(m, llama.hip_backend_fns(be))
    want, _ = s.decode(3, 0, 6)
    out["plain_err"] = be.last_error()
    kv_want = [_download(be, hip, s.handle, buf, n) for buf, n in m.kv_buffers()]
    plain_launches = hip.zgml_hip_get_runtime_profile(be.ctx, s.handle).contents.backend_dispatch_count // 6
    s.close()
    be.set_option(capi.OPT_FUSE_RESIDENT_WGS, -1)
    # the fused plan with one publish suppressed, through the vtable entry point (execute_program)
    s = llama.Session(m, llama.hip_backend_fns(be))
    s.step(3, 0)
    out["first_err"] = be.last_error()
    hip.zgml_hip_clear_error(be.ctx)
    before = hip.zgml_hip_get_runtime_profile(be.ctx, s.handle).contents.backend_dispatch_count
    got, _ = s.decode(3, 0, 6)  # the plan is rebuilt without the fusion
    out["second_err"] = be.last_error()
    out["launches_after"] = (hip.zgml_hip_get_runtime_profile(be.ctx, s.handle).contents.backend_dispatch_count - before) // 6
    out["plain_launches"] = plain_launches
    out["tokens_equal"] = got.tolist() == want.tolist()
    # the contract of include/zgml_hip.h (ZGML_HIP_OPT_FUSE_RESIDENT_WGS): the failed execution's KV column is wrong, re-running the
    # step at the same position rewrites it — after the re-run above the caches hold exactly what the two-launch plan writes
    kv_got = [_download(be, hip, s.handle, buf, n) for buf, n in m.kv_buffers()]
    out["kv_equal_after_rerun"] = all((a.view("uint32") == b.view("uint32")).all() for a, b in zip(kv_got, kv_want))
    s.close()
    # a program compiled afterwards in the same context never fuses again
    s = llama.Session(m, llama.hip_backend_fns(be))
    s.resident_setup(be)
    res = s.resident_decode(3, 0, 6)
    out["resident_equal"] = res.tolist() == want.tolist()
    out["third_err"] = be.last_error()
    s.close()
    m.close()
    be.close()
    print("HANDOFF_RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
