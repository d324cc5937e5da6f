"""Mat-vec roofline micro-benchmark: ring of distinct matrices to defeat the 256 MiB
Infinity Cache; HIP-event time per launch; prints GB/s and fraction of 8 TB/s nominal."""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zgml_amd import Backend  # noqa: E402

PEAK = 8000.0  # GB/s, MI355X HBM3E nominal


def run(be, K, N, q4, ring, warmup, iters, M=1):
    nbytes = C.c_uint64()
    if q4 == 2:
        us = be._lib.zgml_hip_dense_f16_bench(be.ctx, M, K, N, ring, warmup, iters, C.byref(nbytes))
    else:
        us = be._lib.zgml_hip_qmatmul_bench(be.ctx, M, K, N, q4, ring, warmup, iters, C.byref(nbytes))
    if us <= 0:
        raise RuntimeError(be.last_error())
    gbs = nbytes.value / us / 1e3
    r = {"K": K, "N": N, "fmt": ("q8_0", "q4_0", "f16")[q4], "ring": ring, "us": round(us, 3),
         "bytes": nbytes.value, "GBps": round(gbs, 1), "frac": round(gbs / PEAK, 4)}
    if M > 1:
        r["M"] = M
        r["TFLOPs"] = round(2.0 * M * K * N / us / 1e6, 2)
    return r


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--ring", type=int, default=64)
    ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096,4096x32000,576x576,576x1536,1536x576")
    ap.add_argument("--fmts", default="q4,q8", help="comma list of q4,q8,f16")
    ap.add_argument("--M", default="1", help="comma list of input row counts (M > 1: tile kernel)")
    a = ap.parse_args()
    be = Backend(0)
    cp = be._lib.zgml_hip_copy_bench(be.ctx, 1 << 30, 3, 20)
    print(json.dumps({"copy_1GiB_us": round(cp, 2), "copy_GBps_rw": round(2 * (1 << 30) / cp / 1e3, 1)}))
    for shp in a.shapes.split(","):
        K, N = map(int, shp.split("x"))
        for q4 in [{"q4": 1, "q8": 0, "f16": 2}[f] for f in a.fmts.split(",")]:
            ring = max(2, min(a.ring, int((20e9 if q4 == 2 else 40e9) // (K * N))))
            for M in map(int, a.M.split(",")):
                print(json.dumps(run(be, K, N, q4, ring, 20, max(64, a.iters // M) if M > 1 else a.iters, M)), flush=True)
    be.close()
