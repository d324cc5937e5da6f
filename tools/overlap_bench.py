"""Independent mat-vec launches: graph branches (one graph forked over n branches) against explicit streams (one graph per
stream, all in flight), and the copy-kernel calibration variants."""
import ctypes as C
import json
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zgml_amd import Backend  # noqa: E402

be = Backend(0)
for (k, n) in ((4096, 4096), (4096, 11008), (11008, 4096), (4096, 32000)):
    for ns in (1, 2, 4, 8):
        nb = C.c_uint64()
        a = be._lib.zgml_hip_qmatvec_overlap_bench(be.ctx, k, n, 1, 64, ns, 1024, C.byref(nb))
        b = be._lib.zgml_hip_qmatvec_streams_bench(be.ctx, k, n, 1, 64, ns, 1024, C.byref(nb))
        print(json.dumps({"K": k, "N": n, "n": ns, "graph_branches_us": round(a, 3), "explicit_streams_us": round(b, 3),
                          "streams_GBps": round(nb.value / b / 1e3, 1), "streams_frac": round(nb.value / b / 1e3 / 8000, 4)}), flush=True)
be.close()
