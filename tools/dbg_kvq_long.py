import sys, ctypes as C
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zgml_amd import Backend, capi, llama
from oracle import oracle as O
O.set_threads(16)
be = Backend(0)
hip = capi.load_hip(); ol = O.load()
hiprt = C.CDLL("libamdhip64.so")
FUS = int(sys.argv[1]) if len(sys.argv) > 1 else 0
be.set_option(capi.OPT_FUSION, FUS)
cfg = llama.preset("llama2-7b", 2048)
cfg.n_layers, cfg.kv_quant_block = 1, 32
m = llama.Model(cfg, llama.Q4_0, threads=16)
sr, sh = llama.Session(m, O.backend_fns()), llama.Session(m, llama.hip_backend_fns(be))
tok = 1
for pos in list(range(4)) + [1900]:
    tr, lr = sr.step(tok, pos); th, lh = sh.step(tok, pos); tok = tr
be.synchronize()
prog = m.program
def obuf(b):
    n = C.c_uint64(); p = ol.zo_program_buffer(sr.handle, b, C.byref(n))
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n.value,)).copy()
def hbuf(b, n):
    dev = hip.zgml_hip_program_buffer_ptr(sh.handle, b); got = np.zeros(n, np.float32)
    hiprt.hipMemcpy(got.ctypes.data_as(C.c_void_p), C.c_void_p(dev), C.c_size_t(n * 4), 2); return got
for i in range(prog.n_ops):
    op = prog.ops[i]
    if op.kind != 14: continue
    a = op.u.attention_kvq
    dh, nb = a.d_head, a.d_head // a.block_size
    q_o = obuf(a.q)[a.q_off:a.q_off + dh].astype(np.float64)
    q_h = hbuf(a.q, len(obuf(a.q)))[a.q_off:a.q_off + dh].astype(np.float64)
    kc, vc = obuf(a.k), obuf(a.v)
    ncols = a.n_cols
    nq = ncols * dh // 4
    kq = kc[:nq].view(np.int8).reshape(ncols, dh).astype(np.float64); ks = kc[nq:nq + ncols * nb].reshape(ncols, nb).astype(np.float64)
    vq = vc[:nq].view(np.int8).reshape(ncols, dh).astype(np.float64); vs = vc[nq:nq + ncols * nb].reshape(ncols, nb).astype(np.float64)
    K = kq * np.repeat(ks, a.block_size, axis=1); V = vq * np.repeat(vs, a.block_size, axis=1)
    n = a.seq_kv
    sc = (K[a.k_col_start:a.k_col_start + n] @ q_o) * a.scale
    w = np.exp(sc - sc.max()); out64 = (w[:, None] * V[a.v_col_start:a.v_col_start + n]).sum(0) / w.sum()
    d_o = obuf(a.dst)[a.dst_off:a.dst_off + dh]; d_h = hbuf(a.dst, len(obuf(a.dst)))[a.dst_off:a.dst_off + dh]
    s = np.abs(out64).max()
    kch, vch = hbuf(a.k, len(kc)), hbuf(a.v, len(vc))
    nmis = int((kch[:nq].view(np.int8) != kc[:nq].view(np.int8)).sum()), int((vch[:nq].view(np.int8) != vc[:nq].view(np.int8)).sum())
    smis = int((kch[nq:] != kc[nq:]).sum()), int((vch[nq:] != vc[nq:]).sum())
    kqh = kch[:nq].view(np.int8).reshape(ncols, dh).astype(np.float64); ksh = kch[nq:nq + ncols * nb].reshape(ncols, nb).astype(np.float64)
    vqh = vch[:nq].view(np.int8).reshape(ncols, dh).astype(np.float64); vsh = vch[nq:nq + ncols * nb].reshape(ncols, nb).astype(np.float64)
    Kh = kqh * np.repeat(ksh, a.block_size, axis=1); Vh = vqh * np.repeat(vsh, a.block_size, axis=1)
    sch = (Kh[a.k_col_start:a.k_col_start + n] @ q_h) * a.scale
    wh = np.exp(sch - sch.max()); outh = (wh[:, None] * Vh[a.v_col_start:a.v_col_start + n]).sum(0) / wh.sum()
    print(f"   fusion {FUS}: cache int8 mismatches K {nmis[0]} V {nmis[1]}, scale mismatches {smis}; hip dst vs f64 of HIP's own q/caches: {np.abs(d_h-outh).max()/np.abs(outh).max():.2e}")
    print(f"op {i}: seq_kv {n} q diff {np.abs(q_o-q_h).max():.1e} | oracle vs f64 {np.abs(d_o-out64).max()/s:.2e} | hip vs f64 {np.abs(d_h-out64).max()/s:.2e} | scores max {sc.max():.3f} real {sc[[0,1,2,3,n-1]].round(2)}")
    if i > 200: break
