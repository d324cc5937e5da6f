"""Dependent chain of square Q4_0 mat-vecs: launch i + 1 consumes y_i (the true-dependency form of the roofline ring)."""
import ctypes as C
import json
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zgml_amd import Backend  # noqa: E402

be = Backend(0)
for K in [int(k) for k in (sys.argv[1] if len(sys.argv) > 1 else "4096").split(",")]:
    ring = 64 if K <= 4096 else 16
    nb = C.c_uint64()
    us = be._lib.zgml_hip_qmatvec_chain_bench(be.ctx, K, 1, ring, 16, int(sys.argv[2]) if len(sys.argv) > 2 else 2048, C.byref(nb))
    print(json.dumps({"K": K, "ring": ring, "us": round(us, 3), "GBps": round(nb.value / us / 1e3, 1),
                      "frac": round(nb.value / us / 1e3 / 8000, 4), "err": be.last_error()}), flush=True)
be.close()
