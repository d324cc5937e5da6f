import sys, ctypes as C
sys.path.insert(0, "/root/repo")
from zgml_amd import Backend
be = Backend(0)
for (K, N) in ((4096, 4096), (4096, 11008)):
    for ns in (1, 2, 4, 8):
        nb = C.c_uint64()
        us = be._lib.zgml_hip_qmatvec_overlap_bench(be.ctx, K, N, 1, 64, ns, 1024, C.byref(nb))
        print(K, N, "streams", ns, "us", round(us, 3), "GB/s", round(nb.value / us / 1e3, 1), be.last_error())
be.close()
