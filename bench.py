#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X backend for zgml's forward-inference path.

    python bench.py --gpus N --steps K --warmup W

A "step" is one greedy-decode token through the compiled DeviceProgram.

N = 1 : BASELINE.json configs[1] — SmolLM-135M, Q4_0 weights, batch 1. `value` is decode tokens/s
        of the device-resident loop (token embedding, causal-mask column, RoPE row, KV position and
        the argmax all stay in HBM; one hipGraph launch per token). The same JSON line carries
          roofline     : the 4096x4096 Q4_0 mat-vec (the metric's second half) — algorithmic bytes
                         per launch / average launch time, HIP events on the launch stream over a
                         ring of 64 distinct matrices (defeats the 256 MiB Infinity Cache);
          cpu_baseline : the oracle (a C restatement of the reference's CPU path) decoding the same
                         model on the host cores, bounded sample;
          extra        : the vtable-faithful (host I/O every token) rate, Llama-2-7B Q4_0 @ 1 GPU,
                         further mat-vec shapes.
N > 1 : BASELINE.json configs[3] — Llama-2-7B Q4_0, every weight split along N across the ranks
        (one process per GPU, torch.distributed "nccl" = RCCL over xGMI), all-gather of the
        replicated activations between op ranges. One token stream for the whole job => "strong"
        scaling; compare with extra.llama2_7b.tok_s of the N = 1 line.

Weights are synthetic (deterministic generator of SURVEY §8d); there is no network for GGUF files.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E nominal (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def _pmc_traffic(K, N, q4):
    """HBM bytes per launch of the roofline kernel from the committed PMC pass (FETCH_SIZE x 2 on gfx950, calibrated on
    the 1 GiB copy kernel of the same run), newest round first; stamped with the commit the pass was collected at."""
    if (K, N, q4) != (4096, 4096, 1):
        return None, None
    for name in ("r05_qmatvec_pmc.json", "r04_qmatvec_pmc.json", "r03_qmatvec_pmc.json", "r02_qmatvec_pmc.json", "r01_qmatvec_pmc.json"):
        f = ROOT / "profiles" / name
        if f.exists():
            d = json.loads(f.read_text())
            e = d.get("4096x4096_q4_0", {})
            return e.get("traffic_bytes_per_launch"), {"file": "profiles/" + name, "source_commit": d.get("source_commit", "round 1 (unstamped)")}
    return None, None


def _rocprof_reconcile():
    """The committed reconciliation of the three clocks on the roofline kernel (profiles/r04_qmatvec_chain_reconcile.txt / .json,
    tools/reconcile_chain.sh: ONE rocprofv3 --kernel-trace pass of the chain, per-dispatch begin / end): the profiler's mean kernel
    duration and what it implies for `frac`, next to the event figure this run measures live. None when no table is committed."""
    f = next((ROOT / "profiles" / n for n in ("r05_qmatvec_chain_reconcile.json", "r04_qmatvec_chain_reconcile.json") if (ROOT / "profiles" / n).exists()), None)
    if f is None:
        return None
    d = json.loads(f.read_text())
    k = d["rocprof_mean_duration_us"]
    return {"file": "profiles/" + f.name.replace(".json", ".txt"), "rocprof_mean_kernel_duration_us": round(k, 3),
            "frac_kernel_rocprof": round(d["bytes_per_launch"] / k / 1e3 / HBM_PEAK_GBPS, 4),
            "events_us_in_that_pass_unprofiled": d["events_unprofiled_us"], "events_us_while_profiled": d["events_profiled_us"],
            "in_kernel_stamp_span_us": d.get("stamps_span_us"),
            "frac_in_kernel": round(d["bytes_per_launch"] / d["stamps_span_us"] / 1e3 / HBM_PEAK_GBPS, 4) if d.get("stamps_span_us") else None,
            "why_they_differ": "events time launch-to-launch periods of the un-profiled chain (boundary included); the profiler's duration runs from "
                               "the command processor picking the packet up to its completion signal while every dispatch is serialised and "
                               "timestamped (the same program's event period under the profiler is in events_us_while_profiled); the in-kernel "
                               "stamps run from the first workgroup's first instruction to the last workgroup's last store"}


def matvec_roofline(be, K=4096, N=4096, q4=1, ring=64, iters=2048, chain=False):
    """chain=True: launch i + 1 consumes y_i (ping-pong vectors, an epilogue multiply keeps magnitudes bounded) — a
    true data dependency, the shape of the decode path; chain=False: the ring on one x / y, ordered by the stream only."""
    nbytes = C.c_uint64()
    if chain:
        us = be._lib.zgml_hip_qmatvec_chain_bench(be.ctx, K, q4, ring, 64, iters, C.byref(nbytes))
    else:
        us = be._lib.zgml_hip_qmatvec_bench(be.ctx, K, N, q4, ring, 64, iters, C.byref(nbytes))
    if us <= 0:
        raise RuntimeError("qmatvec_bench: " + be.last_error())
    gbps = nbytes.value / us / 1e3
    traffic, src = _pmc_traffic(K, N, q4)
    rec = _rocprof_reconcile() if chain and (K, N, q4) == (4096, 4096, 1) else None
    # `frac`: the figure a reader can reproduce from profiles/ — algorithmic bytes / the profiler's mean kernel duration of the committed
    # rocprofv3 pass of this very chain — when such a pass is committed; the live HIP-event figure of THIS run is
    # `frac_events` (the un-profiled launch-to-launch period: boundary included, the profiler's ~1.1 us per dispatch not), and the span
    # of the in-kernel stamps gives `frac_in_kernel`
    frac_ev = round(gbps / HBM_PEAK_GBPS, 4)
    return {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": rec["frac_kernel_rocprof"] if rec else frac_ev,
            "frac_is": ("frac_kernel_rocprof: algorithmic bytes / the mean kernel duration of the committed rocprofv3 pass (" + rec["file"] + "); `achieved` / "
                        "`us_per_launch` / `frac_events` are THIS run's HIP events") if rec else "frac_events: algorithmic bytes / the HIP-event time per launch measured in THIS run",
            "frac_events": frac_ev, "frac_kernel_rocprof": rec["frac_kernel_rocprof"] if rec else None,
            "frac_in_kernel": rec.get("frac_in_kernel") if rec else None, "reconcile": rec,
            "traffic": traffic, "traffic_source": src,
            "kernel": f"qmatvec_kon_kernel (K-on-lanes Q4_0 mat-vec) {K}x{N}, single launch per mat-vec" if q4 else f"qmatvec_kernel<q8_0> {K}x{N}",
            "dependency": "data (y_i is x_{i+1}; epilogue multiply by a constant vector)" if chain else "stream order (one x / y)",
            "bytes_per_launch": nbytes.value, "us_per_launch": round(us, 3), "ring_matrices": ring,
            "launches": iters, "timing": "HIP events on the launch stream; ring captured in a hipGraph"}


def calibration_streams():
    """What THIS box's HBM delivers to plain streams of 1 GiB (beyond the Infinity Cache), in child processes because the kernel form
    is selected once per process (ZGML_COPY_VARIANT): the textbook copy with one float4 per thread (the guide's 6.29 TB/s form), the
    grid-stride copy loop, and READ-ONLY streams shaped like the mat-vec's (256-thread workgroups, 4 x 16 B in flight per lane, every
    workgroup its own contiguous region: 256 / 688 / 2048 workgroups). The mat-vec is a read stream: these are what to hold it against."""
    import subprocess
    out = {}
    code = ("import sys; sys.path.insert(0, %r)\nfrom zgml_amd import Backend\nbe = Backend(0)\n"
            "print(be._lib.zgml_hip_copy_bench(be.ctx, 1 << 30, 3, 20))" % str(ROOT))
    for name, variant, rw in (("copy_one_float4_per_thread_rw", 1, 2), ("copy_grid_stride_loop_rw", 0, 2), ("read_only_8_in_flight_4096_wgs", 2, 1),
                              ("read_only_regions_256_wgs", 3 + 256 * 256, 1), ("read_only_regions_688_wgs", 3 + 688 * 256, 1),
                              ("read_only_regions_2048_wgs", 3 + 2048 * 256, 1)):
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, ZGML_COPY_VARIANT=str(variant)))
            us = float(r.stdout.strip().splitlines()[-1])
            out[name + "_GBps"] = round(rw * (1 << 30) / us / 1e3, 1)
        except Exception as e:  # noqa: BLE001
            out[name + "_GBps"] = "error: " + str(e)[:80]
    out["note"] = "1 GiB, 20 launches each, HIP events; _rw counts read + written bytes, read_only_* the bytes read"
    return out


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(llama, cfg, kind, budget_s=7.0, max_tokens=64):
    """The oracle decoding the same model on the host cores, bounded sample, three variants:
    B1 exact-dequant f32 path on ONE thread (what the reference runs for GGUF weights), B2 the same
    arithmetic with the output columns split over all host cores, B3 the W8A8 path (quantizeInput + gemvRange over
    GemvPool's <= 16 workers). `value` is the FASTEST of the three; all three are reported."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    m = llama.Model(cfg, kind, threads=min(cores, 16))
    variants = {}

    def run(label, threads, w8a8, what):
        O.set_threads(threads)
        O.set_w8a8(w8a8)
        try:
            s = llama.Session(m, O.backend_fns())
        finally:
            O.set_w8a8(False)
        s.step(1, 0, want_logits=False)  # warm-up step (llama_smollm_bench.zig:147-170)
        t0 = time.perf_counter()
        n, tok = 0, 1
        while n < max_tokens and (time.perf_counter() - t0) < budget_s:
            tok, _ = s.step(tok, 1 + n, want_logits=False)
            n += 1
        dt = time.perf_counter() - t0
        s.close()
        variants[label] = {"value": round(n / dt, 2), "threads": threads, "tokens": n, "what": what}

    run("B1_exact_1thread", 1, False, "exact-dequant f32 path (quant.zig:475-578), sequential")
    # (round 5: the 64-thread split is gone — on 576-wide layers the pool's per-op wake-up of 63 threads made it SLOWER than one
    # thread (14.95 against 38.5 tok/s, BENCH_r04): noise, not a baseline; 8 and 16 threads are what the split can use here)
    for t in sorted({min(cores, 8), min(cores, 16)}):
        if t > 1:
            run(f"B2_exact_{t}_threads", t, False, f"same arithmetic, output columns split over {t} threads")
    arm = "AVX-512 VNNI vpdpbusd arm (bit-identical to the scalar loop)" if O.gemv_uses_vnni() else "scalar / auto-vectorised loop (no AVX-512 VNNI on this host)"
    run("B3_w8a8_gemvpool", min(cores, 16), True, "quantizeInput + gemvRange over GemvPool (<= 16 workers, quant.zig:24-198,320-440); gemvRange: " + arm)
    O.set_threads(1)
    m.close()
    best = max(variants, key=lambda k: variants[k]["value"])
    return {"value": variants[best]["value"], "unit": "tokens/s", "cores": variants[best]["threads"], "kind": "port",
            "fastest": best, "variants": variants, "host_cores": cores, "cpu_model": _cpu_model(),
            "sample": f"greedy tokens of the same SmolLM-135M Q4_0 program through the C oracle, <= {max_tokens} tokens or "
                      f"{budget_s:.0f} s per variant; `value` = the fastest variant ({best})"}


def cpu_config1_q8_matvec(iters=200):
    """BASELINE configs[0]: one 4096 x 4096 Q8_0 mat-vec through the W8A8 CPU path (prepareTransposed once,
    then quantizeInput + GemvPool.dispatch per call) on min(cores, 16) workers. Plumbing, no GPU."""
    import numpy as np
    from oracle import oracle as O
    K = N = 4096
    cores = os.cpu_count() or 1
    workers = min(cores, 16)
    i = np.arange(K * N, dtype=np.int64)
    data = (((i * 13) % 255) - 127).astype(np.int8)  # SURVEY §8d generators
    b = np.arange(K * N // 32, dtype=np.int64)
    scales = (2.0 ** -6 * (1.0 + (b % 7) / 8.0)).astype(np.float16).astype(np.float32)
    x = (((np.arange(K) % 17) - 8) * 0.03125).astype(np.float32)
    t_data, t_scales = O.prepare_transposed(data, scales, K, N, 32)
    lib = O.load()
    O.set_threads(workers)
    q = np.zeros(K, np.int8)
    s = np.zeros(K // 32, np.float32)
    dst = np.zeros(N, np.float32)
    args_q = (x.ctypes.data, K, 32, q.ctypes.data, s.ctypes.data)
    args_g = (t_data.ctypes.data, t_scales.ctypes.data, q.ctypes.data, s.ctypes.data, dst.ctypes.data, N, K, 32, workers)
    for _ in range(8):
        lib.zo_quantize_input(*args_q)
        lib.zo_gemv_pool_dispatch(*args_g)
    t0 = time.perf_counter()
    for _ in range(iters):
        lib.zo_quantize_input(*args_q)
        lib.zo_gemv_pool_dispatch(*args_g)
    us = (time.perf_counter() - t0) / iters * 1e6
    O.set_threads(1)
    exact = O.qmatmul_exact(data, scales, x, 1, N, K, 32)
    nbytes = K * N // 32 * 36 + 4 * K + 4 * N  # int8 [N, K] + one f32 scale per 32 along K, as the transposed copy sits in memory
    return {"workload": "4096x4096 Q8_0 mat-vec, W8A8 (quantizeInput + gemvRange) on the CPU thread pool (BASELINE configs[0])",
            "us_per_matvec": round(us, 1), "GBps": round(nbytes / us / 1e3, 2), "bytes_per_matvec": nbytes, "workers": workers,
            "iters": iters, "host_cores": cores, "cpu_model": _cpu_model(), "gemv_arm": "avx512-vnni" if O.gemv_uses_vnni() else "scalar",
            "max_abs_diff_vs_exact_dequant": float(np.abs(dst - exact).max()), "out_abs_max": float(np.abs(exact).max())}


def w8a8_device_leg(be, llama, cpu_us=None, ring=16, chain=64, reps=6):
    """BASELINE configs[0]'s workload on the device through the reference's own W8A8 arithmetic (ZGML_HIP_OPT_W8A8, w8a8.hip): a
    data-dependent chain of 4096 x 4096 Q8_0 mat-vecs (y_i is x_{i+1}) over a ring of `ring` distinct weights (300 MB: beyond
    the Infinity Cache), one program of `chain` qmatmul ops replayed as a graph; checked BIT-EXACT against the oracle's
    prepareTransposed + quantizeInput + gemvRange on the first link. Then SmolLM-135M decode with every M = 1 qmatmul on that arm
    (unfused launches: the arm is there for like-for-like arithmetic with the CPU baseline's variant B3, not for speed)."""
    import numpy as np
    from oracle import oracle as O
    from zgml_amd import DeviceOp, DeviceProgram, ProgramIO, QuantizedWeightUpload, capi
    K = N = 4096
    out = {"workload": "4096x4096 Q8_0 mat-vec through the W8A8 arm (quantizeInput + prepareTransposed'd weight + gemvRange) ON THE DEVICE, data-dependent chain"}
    rng = np.random.default_rng(11)
    ws = []
    for r in range(ring):
        data = rng.integers(-127, 128, K * N, dtype=np.int8)
        scales = np.full(K * N // 32, 1.0 / (127.0 * np.sqrt(K / 3.0)), np.float32) * (1.0 + (np.arange(K * N // 32) % 5) / 16.0).astype(np.float32)
        ws.append((data, scales))
    x = (((np.arange(K) % 17) - 8) * 0.03125).astype(np.float32)
    bufs = [K] * (chain + 1)
    ops = [DeviceOp.qmatmul(i + 1, i, i % ring, 1, N, K, 0, 0, 0, 0) for i in range(chain)]
    prog = DeviceProgram(ops=ops, buffer_sizes=bufs, initial_uploads=[ProgramIO(0, x)],
                         qweights=[QuantizedWeightUpload(d, sc, K, N, 32) for d, sc in ws])
    be.set_option(capi.OPT_W8A8, 1)
    try:
        h = be.compileProgram(prog)
        if not h:
            raise RuntimeError(be.last_error())
        y1 = np.zeros(N, np.float32)
        ylast = np.zeros(N, np.float32)
        ios_in = [ProgramIO(0, x)]
        ios_out = [ProgramIO(1, y1), ProgramIO(chain, ylast)]
        be.executeProgram(h, ios_in, ios_out)
        t_d, t_s = O.prepare_transposed(ws[0][0], ws[0][1], K, N, 32)
        want = O.gemv(t_d, t_s, x, N, K, 32)
        if not np.array_equal(y1.view(np.uint32), want.view(np.uint32)):
            raise ParityError(f"W8A8 device mat-vec differs from the oracle's gemvRange (max {np.abs(y1 - want).max()})")
        out["verified_bit_exact_against_oracle"] = True
        for _ in range(2):
            be.executeProgram(h, ios_in, ios_out)
        t0 = time.perf_counter()
        for _ in range(reps):
            be.executeProgram(h, ios_in, ios_out)
        us = (time.perf_counter() - t0) / (reps * chain) * 1e6
        nbytes = K * N // 32 * 36 + 4 * K + 4 * N
        out.update({"us_per_matvec": round(us, 3), "GBps": round(nbytes / us / 1e3, 1), "frac_of_hbm_peak": round(nbytes / us / 1e3 / HBM_PEAK_GBPS, 4),
                    "bytes_per_matvec": nbytes, "timing": f"wall clock around {reps} executions of a {chain}-op program (graph replay; includes the step's input / output kernels)",
                    "out_finite": bool(np.isfinite(ylast).all())})
        if cpu_us:
            out["x_cpu_config1"] = round(cpu_us / us, 1)
        be.freeProgram(h)
        # SmolLM-135M decode on the arm
        cfg = llama.preset("smollm-135m")
        m = llama.Model(cfg, llama.Q4_0, include_dead_f32=False, threads=16)
        s = llama.Session(m, llama.hip_backend_fns(be))
        s.resident_setup(be)
        w = s.resident_decode(1, 0, 8)
        be.synchronize()
        t0 = time.perf_counter()
        s.resident_decode(int(w[-1]), 8, 64)
        be.synchronize()
        out["smollm135m_decode_tok_s"] = round(64 / (time.perf_counter() - t0), 1)
        prof = be.getRuntimeProfile(s.handle)
        out["smollm135m_note"] = "every M = 1 qmatmul on the W8A8 arm, launched unfused (no norm prologue / residual epilogue / grouping)"
        s.close()
        m.close()
    finally:
        be.set_option(capi.OPT_W8A8, 0)
    return out


def bench_single(args):
    from zgml_amd import Backend, llama
    be = Backend(0)
    cfg = llama.preset("smollm-135m")
    K = min(args.steps, cfg.max_seq_len - args.warmup - 1)
    t0 = time.perf_counter()
    model = llama.Model(cfg, llama.Q4_0, include_dead_f32=False, threads=16)
    sess = llama.Session(model, llama.hip_backend_fns(be))
    sess.resident_setup(be)
    log(f"[bench] SmolLM-135M built+compiled in {time.perf_counter() - t0:.1f}s, {model.program.n_ops} DeviceOps")

    # warm-up: W untimed tokens, then EXACTLY K timed tokens continuing the same stream
    warm = sess.resident_decode(1, 0, max(1, args.warmup))
    be.synchronize()
    t0 = time.perf_counter()
    toks = sess.resident_decode(int(warm[-1]), max(1, args.warmup), K)
    be.synchronize()
    dt = time.perf_counter() - t0
    value = K / dt

    extra = {}
    lc_parity_failed = False
    # long context: the same stream continued at position ~1900 of 2048 (attention reads 1900 KV rows per head;
    # the fused decode attention splits each head over several workgroups there)
    if cfg.max_seq_len >= 2048:
        try:
            extra["long_context_verified_against_oracle"] = verify_longctx_filled(be, llama, "smollm_f32kv_filled")  # no number without parity
            n_lc = min(64, K)
            sess.resident_decode(int(toks[-1]), 1896, 4)  # (untimed: a plan rebuild or graph capture pending on the session happens here)
            be.synchronize()
            t0 = time.perf_counter()
            sess.resident_decode(int(toks[-1]), 1900, n_lc)
            be.synchronize()
            extra["long_context_pos1900_tok_s"] = round(n_lc / (time.perf_counter() - t0), 1)
        except ParityError as e:
            extra["long_context_pos1900_tok_s"] = None
            extra["long_context_error"] = "PARITY FAILURE: " + str(e)[:300]
            lc_parity_failed = True
            log("[bench] " + extra["long_context_error"])
    # vtable-faithful path: host patches + 32 uploads + logits download + host argmax per token
    n_vt = min(K, 128)
    # the drop-in path: host patches + 32 uploads + logits download + host argmax per token. Its refresh is the adapter's
    # (zig/backend_hip.zig: refreshProgram — (slice_pos, seq_kv) through zgml_hip_refresh_dynamic, as the reference's wgpu backend
    # reduces it); the cpu backend's contract (the whole op list, static changes detected) is timed beside it
    sess.decode(1, 0, 4)
    _, secs_full = sess.decode(1, 0, n_vt)
    sess.use_dynamic_refresh()
    sess.pin_outputs(be)  # (the adapter's: DeviceInference's logits slice is session-owned — zgml_hip_program_pin_outputs)
    sess.decode(1, 0, 4)
    _, secs = sess.decode(1, 0, n_vt)
    extra["vtable_path_tok_s"] = round(n_vt / secs, 1)
    extra["vtable_path_full_refresh_tok_s"] = round(n_vt / secs_full, 1)
    extra["vtable_path_note"] = "execute_program per token incl. PCIe uploads/logits download (never `value`)"
    prof = be.getRuntimeProfile(sess.handle)
    extra["launches_per_token"] = int(prof.backend_dispatch_count // max(1, prof.call_count))
    extra["device_ops_per_token"] = int(model.program.n_ops)
    qb, nw = model.quant_bytes()
    extra["weight_bytes_per_token"] = qb + cfg.vocab_size * cfg.d_model * 4
    sess.close()
    model.close()

    roof = matvec_roofline(be, chain=True)
    log(f"[bench] 4096x4096 Q4_0 mat-vec chain (data-dependent launches): {roof['us_per_launch']} us, {roof['achieved']} GB/s")
    ring0 = matvec_roofline(be)
    extra["matvec_q4_0_4096_stream_ordered_ring"] = {"us": ring0["us_per_launch"], "GBps": ring0["achieved"], "frac": ring0["frac"],
                                                     "note": "the round-1 figure: same x / y for every launch, ordered by the stream only"}
    shapes = {}
    for (k, n) in ((4096, 11008), (11008, 4096), (4096, 32000)):
        r = matvec_roofline(be, k, n, 1, 32, 512)
        shapes[f"{k}x{n}"] = {"us": r["us_per_launch"], "GBps": r["achieved"], "frac": r["frac"]}
    r8 = matvec_roofline(be, 4096, 4096, 0, 64, 1024)
    shapes["4096x4096_q8_0"] = {"us": r8["us_per_launch"], "GBps": r8["achieved"], "frac": r8["frac"]}
    extra["matvec_q4_0_other_shapes"] = shapes
    # the same Q4_0 rings as INDEPENDENT launches (graph forked over 4 branches): what the kernel streams when
    # launches may overlap — not the decode path, where every mat-vec waits for its predecessor (`roofline`)
    indep = {}
    for (k, n, ns) in ((4096, 4096, 8), (4096, 11008, 8), (4096, 32000, 2)):
        nb = C.c_uint64()
        us_st = be._lib.zgml_hip_qmatvec_streams_bench(be.ctx, k, n, 1, 64, ns, 1024, C.byref(nb))
        us_br = be._lib.zgml_hip_qmatvec_overlap_bench(be.ctx, k, n, 1, 64, 4, 1024, C.byref(nb))
        if us_st > 0:
            indep[f"{k}x{n}"] = {"us_per_launch": round(us_st, 3), "GBps": round(nb.value / us_st / 1e3, 1),
                                 "frac": round(nb.value / us_st / 1e3 / HBM_PEAK_GBPS, 4), "streams": ns,
                                 "graph_branches_us_per_launch": round(us_br, 3) if us_br > 0 else None}
    indep["note"] = ("64 mat-vecs on EXPLICIT streams (one graph per stream, all in flight): what the kernel streams when launch ramps and tails "
                     "overlap — not the dependent-launch figure of `roofline` (every mat-vec of the decode path waits for its predecessor); "
                     "graph_branches: the same ring forked over 4 branches of ONE graph (placement is the runtime's: branches on one queue serialise)")
    extra["matvec_q4_0_independent_launches"] = indep
    cp = be._lib.zgml_hip_copy_bench(be.ctx, 1 << 30, 3, 20)
    extra["copy_kernel_GBps_read_plus_write"] = round(2 * (1 << 30) / cp / 1e3, 1)
    extra["copy_kernel_note"] = "1 GiB device copy, 8 x 16 B in flight per lane, non-temporal, grid-stride loop; see calibration_streams for the forms that reach more"
    extra["calibration_streams"] = calibration_streams()

    # the int8 KV cache variant of the same program (extension ops kvq_store / attention_kvq)
    try:
        cq = llama.preset("smollm-135m")
        cq.kv_quant_block = 32
        mq = llama.Model(cq, llama.Q4_0, include_dead_f32=False, threads=16)
        sq = llama.Session(mq, llama.hip_backend_fns(be))
        sq.resident_setup(be)
        wq = sq.resident_decode(1, 0, 8)
        be.synchronize()
        t0 = time.perf_counter()
        wq = sq.resident_decode(int(wq[-1]), 8, 128)
        be.synchronize()
        short = 128 / (time.perf_counter() - t0)
        ver_q = verify_longctx_filled(be, llama, "smollm_int8kv_filled")
        sq.resident_decode(int(wq[-1]), 1896, 4)  # (untimed)
        be.synchronize()
        t0 = time.perf_counter()
        sq.resident_decode(int(wq[-1]), 1900, 64)
        be.synchronize()
        extra["quantised_kv_int8"] = {"tok_s": round(short, 1), "long_context_pos1900_tok_s": round(64 / (time.perf_counter() - t0), 1),
                                      "long_context_verified_against_oracle": ver_q,
                                      "workload": "SmolLM-135M Q4_0 decode with int8 KV caches (block 32)"}
        sq.close()
        mq.close()
    except ParityError as e:
        extra["quantised_kv_int8"] = {"error": "PARITY FAILURE: " + str(e)[:300]}
        lc_parity_failed = True
    except Exception as e:
        extra["quantised_kv_int8"] = {"error": str(e)[:200]}

    parity_failed = lc_parity_failed
    if not args.skip_llama7b:
        try:
            extra["llama2_7b"] = bench_llama7b_single(be, llama, args)
            parity_failed = parity_failed or any("PARITY" in str(v.get("error", "")) for v in extra["llama2_7b"].get("prefill_batch32", {}).values() if isinstance(v, dict))
        except ParityError as e:  # no Llama-2-7B number without parity: the leg fails loudly (exit code 3 below)
            extra["llama2_7b"] = {"error": "PARITY FAILURE: " + str(e)[:300]}
            parity_failed = True
            log("[bench] " + extra["llama2_7b"]["error"])
        except Exception as e:  # never lose the headline line to the optional leg
            extra["llama2_7b"] = {"error": str(e)[:200]}

    cpu = None
    if not args.skip_cpu:
        cpu = cpu_baseline(llama, cfg, llama.Q4_0)
        log(f"[bench] cpu baseline: {cpu['value']} tok/s ({cpu['fastest']}); " +
            ", ".join(f"{k}={v['value']}" for k, v in cpu["variants"].items()))
        try:
            extra["cpu_config1_q8_0_matvec"] = cpu_config1_q8_matvec()
        except Exception as e:
            extra["cpu_config1_q8_0_matvec"] = {"error": str(e)[:200]}
    try:  # the same workload (and SmolLM decode) through the W8A8 arm on the device: like-for-like with the CPU's arithmetic
        extra["w8a8_device"] = w8a8_device_leg(be, llama, (extra.get("cpu_config1_q8_0_matvec") or {}).get("us_per_matvec"))
        log(f"[bench] W8A8 arm on the device: {extra['w8a8_device'].get('us_per_matvec')} us per 4096x4096 mat-vec, SmolLM {extra['w8a8_device'].get('smollm135m_decode_tok_s')} tok/s")
    except ParityError as e:
        extra["w8a8_device"] = {"error": "PARITY FAILURE: " + str(e)[:300]}
        parity_failed = True
        log("[bench] " + extra["w8a8_device"]["error"])
    except Exception as e:
        extra["w8a8_device"] = {"error": str(e)[:200]}
    be.close()

    def stream_roofline(bytes_per_token, tok_s, what):  # a decode step as a weight stream: bytes every token must read / token time / HBM peak
        gb = bytes_per_token * tok_s / 1e9
        return {"bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gb / HBM_PEAK_GBPS, 4),
                "bytes_per_token": int(bytes_per_token), "tok_s": round(tok_s, 1), "what": what}
    l7 = extra.get("llama2_7b", {})
    out = {
        "metric": "decode_tokens_per_sec", "value": round(value, 1), "unit": "tokens/s", "n_gpus": 1,
        "steps": K, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / K, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "SmolLM-135M Q4_0 greedy decode, batch 1, 1xMI355X (BASELINE configs[1])",
                   "weights": "synthetic Q4_0 (int4 + f16 block scale / 32), tied f32 LM head",
                   "max_seq": cfg.max_seq_len, "loop": "device-resident (inputs in HBM)", "parallelism": "none"},
        "roofline": roof, "cpu_baseline": cpu,
        # (round 5, VERDICT r04 #7: the decode steps' own rooflines and the drop-in path's rate in `parsed`, not only under `extra`)
        "vtable_tok_s": extra.get("vtable_path_tok_s"),
        "x_cpu_baseline": round(value / cpu["value"], 2) if cpu and cpu.get("value") else None,
        "roofline_smollm_decode": stream_roofline(extra["weight_bytes_per_token"], value,
                                                  "SmolLM-135M token: Q4_0 weights (59.7 MB) + tied f32 LM head (113.2 MB); the whole set fits the 256 MB Infinity Cache, "
                                                  "so this is a latency figure against an HBM peak, not HBM traffic"),
        "roofline_llama7b_decode": stream_roofline(l7["q4_0_weight_bytes"], l7["tok_s"], "Llama-2-7B token: 3.716 GB of Q4_0 weights from HBM") if l7.get("tok_s") else None,
        "extra": extra,
    }
    print(json.dumps(out), flush=True)
    if parity_failed:
        sys.exit(3)


def verify_l7_prefill(be, llama, kind, T=32):
    """The tile kernels at 7B shapes (2 layers, one token_len = 32 chunk at position 0 and one at 32) against the
    oracle's fixture; f16 promotion is compared at 2e-3 (a 1-ulp f32 difference in A can flip its f16 rounding)."""
    gold = json.loads((ROOT / "tests" / "golden" / "l7dims.json").read_text())[f"prefill{T}_" + kind]
    dense = kind == "f16"
    cfg = llama.preset("llama2-7b", 512)
    cfg.n_layers = gold["n_layers"]
    m = llama.Model(cfg, llama.F32_DENSE if dense else llama.Q4_0, threads=16, token_len=T)
    s = llama.Session(m, llama.hip_backend_fns(be))
    try:
        for ci, ch in enumerate(gold["chunks"]):
            toks = [(7 * (i + ci * T) + 3) % cfg.vocab_size for i in range(T)]
            t, logits = s.prefill(toks, ch["pos"])
            _check_probe(logits, ch["probe"], 2e-3 if dense else 2e-4, f"2-layer 7B-shape prefill ({kind}), chunk at {ch['pos']}")
            if t != ch["token"]:
                raise ParityError(f"2-layer 7B-shape prefill ({kind}): token {t} != oracle {ch['token']}")
    finally:
        s.close()
        m.close()


def prefill_leg(be, llama, kind, T=32, reps=8, max_seq=512):
    """BASELINE configs[4]: Llama-2-7B prefill of one token_len = 32 chunk, Q4_0 (bf16-XDL tile kernel over the
    packed 4-bit weights: exact bf16 splits of x and of scale * q, f32 accumulation) or dense f16 (weights
    promoted to f16, v_mfma_f32_16x16x32_f16). Timed region:
    resident replays of the chunk's program (inputs already in HBM), HIP work only."""
    import numpy as np
    from zgml_amd import capi
    dense = kind == "f16"
    be.set_option(capi.OPT_F16_DENSE_WEIGHTS, int(dense))
    try:
        has_fixture = f"prefill{T}_{kind}" in json.loads((ROOT / "tests" / "golden" / "l7dims.json").read_text())
        if has_fixture:
            verify_l7_prefill(be, llama, kind, T)  # ParityError: no number is reported for this leg
        t0 = time.perf_counter()
        m = llama.Model(llama.preset("llama2-7b", max_seq), llama.F32_DENSE if dense else llama.Q4_0, threads=16, token_len=T)
        s = llama.Session(m, llama.hip_backend_fns(be))
        build_s = time.perf_counter() - t0
        toks = [(7 * i + 3) % m.cfg.vocab_size for i in range(T)]
        nxt, logits = s.prefill(toks, 0)  # through the vtable once (uploads + logits download)
        finite = bool(np.isfinite(logits).all())
        be._lib.zgml_hip_enqueue_program(be.ctx, s.handle)
        be.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            be._lib.zgml_hip_enqueue_program(be.ctx, s.handle)
        be.synchronize()
        dt = (time.perf_counter() - t0) / reps
        # the device-resident chunk (only the T token ids cross PCIe; embedding gather, mask, RoPE rows, argmax on the
        # device): consecutive chunks of one prompt, blocking per chunk
        resident = None
        if not dense:
            s.resident_setup(be)
            nxt_r = s.resident_prefill(toks, 0)
            if nxt_r != nxt:
                raise ParityError(f"resident prefill token {nxt_r} != vtable prefill token {nxt}")
            n_chunks = min(reps, max_seq // T)
            be.synchronize()
            t0 = time.perf_counter()
            for c in range(n_chunks):
                s.resident_prefill(toks, c * T)
            resident = round(T * n_chunks / (time.perf_counter() - t0), 1)
        dec = None
        if dense:  # the same f16-promoted weights at token_len 1, for the "fp16 dense vs Q4_0" comparison at batch 1
            s.close()
            m.close()
            m = llama.Model(llama.preset("llama2-7b", max_seq), llama.F32_DENSE, threads=16)
            s = llama.Session(m, llama.hip_backend_fns(be))
            s.resident_setup(be)
            w = s.resident_decode(1, 0, 4)
            be.synchronize()
            t0 = time.perf_counter()
            s.resident_decode(int(w[-1]), 4, 48)
            be.synchronize()
            dec = round(48 / (time.perf_counter() - t0), 1)
        s.close()
        m.close()
        return {"verified_against_oracle": f"tests/golden/l7dims.json prefill{T}_{kind} (2 layers at 7B shapes, both chunks)" if has_fixture else None,
                "verified": has_fixture, "resident_prefill_tok_s": resident, "decode_tok_s_batch1": dec, "prefill_tok_s": round(T / dt, 1), "ms_per_chunk": round(dt * 1e3, 3), "token_len": T, "first_token": int(nxt),
                "logits_finite": finite, "build_s": round(build_s, 1)}
    finally:
        be.set_option(capi.OPT_F16_DENSE_WEIGHTS, 0)


class ParityError(RuntimeError):
    pass


def _check_probe(logits, want, tol, what):
    """Logits against a fixture probe of the oracle (tests/golden/*.json, tools/gen_golden_l7.py): its top-8 values at
    its indices, the extrema and the absolute sum within `tol` of the logit range, and the same greedy token."""
    import numpy as np
    rng = want["max"] - want["min"]
    err = float(np.abs(logits[np.array(want["top8"])] - np.array(want["top8_vals"], np.float32)).max())
    ok = (err <= tol * rng and abs(float(logits.max()) - want["max"]) <= tol * rng and abs(float(logits.min()) - want["min"]) <= tol * rng
          and abs(float(np.sum(np.abs(logits), dtype=np.float64)) - want["abs_sum"]) <= tol * want["abs_sum"]
          and int(np.argmax(logits)) == want["top8"][0])
    if not ok:
        raise ParityError(f"{what}: logits differ from the oracle fixture (top-8 error {err:.3e} of range {rng:.3e}, "
                          f"argmax {int(np.argmax(logits))} vs {want['top8'][0]}, max {float(logits.max()):.6g} vs {want['max']:.6g}, "
                          f"min {float(logits.min()):.6g} vs {want['min']:.6g}, abs sum {float(np.sum(np.abs(logits), dtype=np.float64)):.8g} vs {want['abs_sum']:.8g})")


def verify_longctx(be, llama, case):
    """Before a `long_context_pos1900_*` number is reported: the HIP path against the oracle's committed fixture of the same
    regime (tests/golden/longctx.json, tools/gen_golden_longctx.py): a fresh session, 4 greedy steps at positions 0..3, 4 at
    1900..1903 — vtable steps with logits probes, then the device-resident loop's tokens. Raises ParityError."""
    gold = json.loads((ROOT / "tests" / "golden" / "longctx.json").read_text())
    g, pos0 = gold[case], gold["pos"]
    cfg = llama.preset("smollm-135m" if case.startswith("smollm") else "llama2-7b", 2048)
    cfg.n_layers, cfg.kv_quant_block = g["n_layers"], g["kv_quant_block"]
    # int8 KV: storeColumn TRUNCATES x * 127 / max toward zero; with the synthetic weights many of those products sit exactly on an
    # integer in one summation order and one ulp below it in another, so a few of the 640 stored int8 values per kv head come out
    # one unit apart between oracle and HIP (tools/dbg_kvq_long.py counts them: 5-24 per head, scales 18-20 ulp-level differences;
    # the HIP attention itself agrees with an f64 evaluation of ITS OWN cache to 2e-7). At position 1900 the attention output is
    # small (1896 zero columns share the softmax), so those units show: 1.4e-3 of the logit range at Llama-2-7B dimensions.
    tol = 3e-3 if g["kv_quant_block"] else 2e-4
    m = llama.Model(cfg, llama.Q4_0, include_dead_f32=False, threads=16)
    try:
        s = llama.Session(m, llama.hip_backend_fns(be))
        tok = g["first_token"]
        for i, pos in enumerate(g["positions"]):
            tok, logits = s.step(tok, pos)
            if pos >= pos0:
                _check_probe(logits, g["probes"][i - 4], tol, f"{case}: position {pos}")
            if tok != g["tokens"][i]:
                raise ParityError(f"{case}: token {tok} != oracle {g['tokens'][i]} at position {pos}")
        s.close()
        s = llama.Session(m, llama.hip_backend_fns(be))  # fresh caches for the resident loop
        s.resident_setup(be)
        got = s.resident_decode(g["first_token"], 0, 4).tolist()
        got += s.resident_decode(got[-1], pos0, 4).tolist()
        if got != g["tokens"]:
            raise ParityError(f"{case}: resident tokens {got} != oracle {g['tokens']}")
        s.close()
    finally:
        m.close()
    return f"tests/golden/longctx.json {case} (oracle; vtable logits probes + resident tokens at positions {pos0}..{pos0 + 3})"


def verify_longctx_filled(be, llama, case, fused_attention=True):
    """The HIP path against a `*_filled` case of tests/golden/longctx.json (tools/gen_golden_longctx.py --filled; the ORACLE): the
    reference's flow — prefill in chunks of 128 (src/llama_inference.zig:474) over 1920 real tokens, so every cache column the
    decode attention reads holds a key / value, then 4 greedy decode steps. HIP: the chunks through zgml_hip_resident_prefill
    (greedy token of every chunk against the fixture; the last chunk through the vtable for its logits probe), the caches handed
    to the decode plan on the device (zgml_hip_copy_program_buffer), the decode steps through the vtable with logits probes, then
    the same positions again through the device-resident loop (tokens). f32 KV: 2e-4 of the logit range; int8 KV: 1e-3 (single
    cache units on the truncation boundary of storeColumn, tests/test_hip_longctx_filled.py). fused_attention=False refuses the
    q/k/v + attention fusion, so the stand-alone decode attention (and its context split) is what runs. Raises ParityError."""
    import ctypes as C
    from zgml_amd import capi
    g = json.loads((ROOT / "tests" / "golden" / "longctx.json").read_text())[case]
    cfg = llama.preset("smollm-135m" if case.startswith("smollm") else "llama2-7b", 2048)
    cfg.n_layers, cfg.kv_quant_block = g["n_layers"], g["kv_quant_block"]
    tol = 1e-3 if g["kv_quant_block"] else 2e-4
    T, n_chunks = g["chunk"], g["n_chunks"]
    m_pre, m_dec = llama.Model(cfg, llama.Q4_0, threads=16, token_len=T), llama.Model(cfg, llama.Q4_0, threads=16)
    s_pre = s_dec = None
    try:
        s_pre = llama.Session(m_pre, llama.hip_backend_fns(be))
        if fused_attention:  # (the option is left alone: setting it makes every program of the context rebuild its plan at its next run)
            s_dec = llama.Session(m_dec, llama.hip_backend_fns(be))
        else:
            be.set_option(capi.OPT_FUSE_RESIDENT_WGS, 0)
            try:
                s_dec = llama.Session(m_dec, llama.hip_backend_fns(be))
            finally:
                be.set_option(capi.OPT_FUSE_RESIDENT_WGS, -1)
        s_pre.resident_setup(be)
        for c in range(n_chunks):
            toks = [(7 * (c * T + i) + 3) % cfg.vocab_size for i in range(T)]
            if c < n_chunks - 1:
                tok = s_pre.resident_prefill(toks, c * T)
            else:
                tok, logits = s_pre.prefill(toks, c * T)
                _check_probe(logits, g["prefill_probe"], tol, f"{case}: last prefill chunk")
            if tok != g["chunk_tokens"][c]:
                raise ParityError(f"{case}: prefill chunk {c} gives token {tok}, oracle {g['chunk_tokens'][c]}")
        hip = capi.load_hip()

        def hand_over():
            for (bp, n), (bd, n2) in zip(m_pre.kv_buffers(), m_dec.kv_buffers()):
                if n != n2 or hip.zgml_hip_copy_program_buffer(be.ctx, s_dec.handle, bd, 0, s_pre.handle, bp, 0, n) != 0:
                    raise ParityError(f"{case}: KV hand-off failed: " + be.last_error())
        hand_over()
        first = tok
        for i, pos in enumerate(g["positions"]):
            tok, logits = s_dec.step(tok, pos)
            _check_probe(logits, g["probes"][i], tol, f"{case}: position {pos}")
            if tok != g["tokens"][i]:
                raise ParityError(f"{case}: token {tok} != oracle {g['tokens'][i]} at position {pos}")
        hand_over()  # the prefilled image again (the steps above appended their own columns), now the resident loop
        s_dec.resident_setup(be)
        got = s_dec.resident_decode(first, g["positions"][0], len(g["positions"])).tolist()
        if got != g["tokens"]:
            raise ParityError(f"{case}: resident tokens {got} != oracle {g['tokens']}")
    finally:
        for x in (s_pre, s_dec, m_pre, m_dec):
            if x is not None:
                x.close()
    return (f"tests/golden/longctx.json {case} (oracle; 1920-token prefilled cache, prefill tokens + last-chunk probe, vtable logits "
            f"probes + resident tokens at positions {g['positions'][0]}..{g['positions'][-1]}, "
            f"{'fused q/k/v + attention launch' if fused_attention else 'stand-alone decode attention'})")


def verify_l7_shapes(be, llama):
    """Before any Llama-2-7B number is reported: the HIP path at 7B shapes against the oracle's committed fixture
    (tests/golden/l7dims.json: 2 layers, d 4096 / 32 x 128 heads / d_ff 11008 / vocab 32000) — vtable steps with
    logits, then the resident loop's tokens. Raises ParityError on any difference."""
    gold = json.loads((ROOT / "tests" / "golden" / "l7dims.json").read_text())["decode_f32kv"]
    cfg = llama.preset("llama2-7b", 512)
    cfg.n_layers = gold["n_layers"]
    m = llama.Model(cfg, llama.Q4_0, threads=16)
    s = llama.Session(m, llama.hip_backend_fns(be))
    try:
        tok = gold["first_token"]
        for pos in range(6):
            tok, logits = s.step(tok, pos)
            _check_probe(logits, gold["probes"][pos], 2e-4, f"2-layer 7B-shape decode, position {pos}")
            if tok != gold["tokens"][pos]:
                raise ParityError(f"2-layer 7B-shape decode: token {tok} != oracle {gold['tokens'][pos]} at position {pos}")
        s.resident_setup(be)
        got = s.resident_decode(gold["first_token"], 0, len(gold["tokens"])).tolist()
        if got != gold["tokens"]:
            raise ParityError(f"2-layer 7B-shape resident decode: tokens {got} != oracle {gold['tokens']}")
    finally:
        s.close()
        m.close()
    return {"fixture": "tests/golden/l7dims.json (oracle, 2 layers at Llama-2-7B dimensions)", "positions_with_logits": 6,
            "resident_tokens": len(gold["tokens"])}


def bench_llama7b_single(be, llama, args):
    verified = {"reduced_layers": verify_l7_shapes(be, llama)}
    cfg = llama.preset("llama2-7b", 2048)
    t0 = time.perf_counter()
    model = llama.Model(cfg, llama.Q4_0, threads=16)
    sess = llama.Session(model, llama.hip_backend_fns(be))
    # the timed 32-layer model itself against the oracle's fixture of the SAME program (tests/golden/l7full.json,
    # `tools/gen_golden_l7.py --full`): 4 greedy positions with logits through the vtable
    full = json.loads((ROOT / "tests" / "golden" / "l7full.json").read_text())
    tok = full["first_token"]
    for pos, (want_tok, probe) in enumerate(zip(full["tokens"], full["probes"])):
        tok, logits = sess.step(tok, pos)
        _check_probe(logits, probe, 2e-4, f"32-layer Llama-2-7B decode, position {pos}")
        if tok != want_tok:
            raise ParityError(f"32-layer Llama-2-7B decode: token {tok} != oracle {want_tok} at position {pos}")
    verified["full_model"] = {"fixture": "tests/golden/l7full.json (oracle, all 32 layers)", "positions_with_logits": len(full["tokens"])}
    sess.resident_setup(be)
    build_s = time.perf_counter() - t0
    W, K = 4, min(args.steps, 128)
    warm = sess.resident_decode(1, 0, W)
    if warm.tolist() != full["tokens"][:W]:
        raise ParityError(f"32-layer resident decode: tokens {warm.tolist()} != oracle {full['tokens'][:W]}")
    be.synchronize()
    t0 = time.perf_counter()
    sess.resident_decode(int(warm[-1]), W, K)
    be.synchronize()
    dt = time.perf_counter() - t0
    qb, nw = model.quant_bytes()
    verified["long_context"] = verify_longctx_filled(be, llama, "l7dims_f32kv_filled")  # (2 layers at these dimensions; ParityError: no 7B numbers)
    verified["long_context_int8_kv"] = verify_longctx_filled(be, llama, "l7dims_int8kv_filled")
    sess.resident_decode(1, 1896, 4)  # (untimed: a plan rebuild or graph capture pending on the session happens here)
    be.synchronize()
    t0 = time.perf_counter()
    sess.resident_decode(1, 1900, 32)  # long context (see bench_single)
    be.synchronize()
    long_ctx = round(32 / (time.perf_counter() - t0), 1)
    sess.close()
    model.close()
    kvq_long = None
    try:  # int8 KV caches at the same long context (3.6x fewer KV bytes per token)
        cq = llama.preset("llama2-7b", 2048)
        cq.kv_quant_block = 32
        mq = llama.Model(cq, llama.Q4_0, threads=16)
        sq = llama.Session(mq, llama.hip_backend_fns(be))
        sq.resident_setup(be)
        sq.resident_decode(1, 0, 4)
        be.synchronize()
        t0 = time.perf_counter()
        sq.resident_decode(1, 1900, 32)
        be.synchronize()
        kvq_long = round(32 / (time.perf_counter() - t0), 1)
        sq.close()
        mq.close()
    except Exception as e:
        kvq_long = str(e)[:120]
    tok_s = K / dt
    prefill = {}
    if not args.skip_prefill:
        for kind in ("q4_0", "f16"):
            try:
                prefill[kind] = prefill_leg(be, llama, kind)
            except ParityError as e:
                prefill[kind] = {"error": "PARITY FAILURE: " + str(e)[:300]}
                log("[bench] " + prefill[kind]["error"])
            except Exception as e:
                prefill[kind] = {"error": str(e)[:200]}
        prefill["workload"] = "Llama-2-7B prefill, one chunk of 32 tokens, Q4_0 vs dense f16 (BASELINE configs[4])"
        try:  # the reference's own chunk size (llama_inference.zig:474): 128 tokens per execution, M > 32 tile kernel
            prefill["chunk128_q4_0"] = prefill_leg(be, llama, "q4_0", T=128, reps=4)
        except ParityError as e:
            prefill["chunk128_q4_0"] = {"error": "PARITY FAILURE: " + str(e)[:300]}
        except Exception as e:
            prefill["chunk128_q4_0"] = {"error": str(e)[:200]}
    return {"verified_against_oracle": verified, "prefill_batch32": prefill, "tok_s": round(tok_s, 1), "long_context_pos1900_tok_s": long_ctx, "long_context_pos1900_int8_kv_tok_s": kvq_long, "ms_per_token": round(1e3 * dt / K, 3), "steps": K, "build_s": round(build_s, 1),
            "q4_0_weight_bytes": qb, "weight_stream_GBps": round(qb * tok_s / 1e9, 1),
            "frac_of_hbm_peak": round(qb * tok_s / 1e9 / HBM_PEAK_GBPS, 4),
            "workload": "Llama-2-7B Q4_0 greedy decode, batch 1, 1xMI355X (BASELINE configs[2])"}


def bench_sharded(args):
    """N > 1: one process per GPU; the data path (op ranges + RCCL all-gathers + argmax, one graph per token) runs
    behind the C ABI (zgml_hip_shard_*); torch.distributed (gloo) is only the control plane here: it carries the
    128-byte communicator id, the barriers around the timed region and the max over ranks."""
    import torch
    import torch.distributed as dist
    from zgml_amd import Backend, llama
    from zgml_amd.sharded import NativeShardedDecoder

    # RCCL prints a version banner on stdout at init; the contract is ONE JSON line there, so
    # everything until the final print goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    rank, ws = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", rank))
    if os.environ.get("ZGML_BENCH_ONE_GPU"):  # rehearsal on a one-GPU box: every rank on device 0 (only the peer gather can run there: RCCL refuses two ranks on one device)
        local = 0
    if ws != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ws}")
    if "MASTER_ADDR" not in os.environ:  # single-process rehearsal (ZGML_BENCH_FORCE_SHARDED)
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    be = Backend(local)
    # one diagnosable log for the first real multi-device run: per rank, the device it sits on and which peers it can
    # address (hipDeviceCanAccessPeer: what the peer-store gather needs); gather mode and microseconds per gather point follow below
    n_dev = be._lib.zgml_hip_device_count()
    peer_row = [be._lib.zgml_hip_device_can_access_peer(local, d) for d in range(n_dev)]
    log(f"[bench] rank {rank}/{ws}: device ordinal {local} of {n_dev} visible; can access peers {peer_row}; "
        f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}; ZGML_SHARD_GATHER={os.environ.get('ZGML_SHARD_GATHER', 'auto')}")

    def exchange_id(mine: bytes) -> bytes:
        t = torch.tensor(list(mine), dtype=torch.uint8)
        dist.broadcast(t, 0)
        return bytes(t.tolist())

    def exchange_handles(mine: bytes):
        out = [None] * ws
        dist.all_gather_object(out, mine)
        return out

    cfg = llama.preset("llama2-7b", 2048)
    cfg.shard_rank, cfg.shard_world = rank, ws
    model = llama.Model(cfg, llama.Q4_0, threads=max(2, 16 // max(1, ws // 2)))
    K = min(args.steps, 256)
    W = max(2, args.warmup)
    gold = json.loads((ROOT / "tests" / "golden" / "l7full.json").read_text())["tokens"]

    def all_ok(flag: bool) -> bool:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def bring_up(backend, gather):
        """A decoder in gather mode `gather`, warmed up over W greedy steps; (decoder, tokens, last token) when EVERY rank produced
        the oracle's tokens (tests/golden/l7full.json: the unsharded 32-layer program) and the ranks agree, else (None, why, None) on
        every rank alike (the decision is an all-reduce over the control plane)."""
        dec, why, toks, tok = None, "", [], 1
        try:
            dec = NativeShardedDecoder(backend, model, rank, ws, exchange_id if ws > 1 else None, gather=gather,
                                       exchange_handles=exchange_handles if gather == "peer" else None)
        except Exception as e:  # noqa: BLE001
            why = f"rank {rank}: {type(e).__name__}: {str(e)[:200]}"
        if not all_ok(dec is not None):
            if dec is not None:
                dec.close()
            return None, why or "another rank failed to set the gather up", None
        try:
            for pos in range(W):
                tok = dec.step(tok, pos)
                toks.append(tok)
        except Exception as e:  # noqa: BLE001
            why = f"rank {rank}: {type(e).__name__}: {str(e)[:200]}"
        n_chk = min(len(gold), len(toks))
        good = not why and len(toks) == W and toks[:n_chk] == gold[:n_chk]
        if not why and not good:
            why = f"rank {rank}: tokens {toks[:n_chk]} != oracle {gold[:n_chk]}"
        if not all_ok(good):
            dec.close()
            return None, why or "another rank failed the token check", None
        t = torch.tensor(toks, dtype=torch.int64)
        gathered = [torch.empty_like(t) for _ in range(ws)]
        dist.all_gather(gathered, t)
        if not all(torch.equal(g, gathered[0]) for g in gathered):
            dec.close()
            return None, "ranks disagree on the greedy tokens", None
        return dec, toks, tok

    # Gather mode: peer stores first (zgml_amd/csrc/shard_peer.hip: one small kernel per gather point instead of an ncclAllGather),
    # RCCL when that cannot be set up or does not reproduce the oracle's tokens on every rank — each attempt on a context of its own.
    # ZGML_SHARD_GATHER=rccl / peer pins one of them.
    want = os.environ.get("ZGML_SHARD_GATHER", "auto")
    modes = ["rccl"] if want == "rccl" or ws == 1 else (["peer"] if want == "peer" else ["peer", "rccl"])
    dec, gather_mode, fallback_reason = None, None, None
    for mode in modes:
        if dec is None and mode != modes[0]:
            be.close()
            be = Backend(local)  # a fresh context: the failed attempt may have left an error (or a timed-out step) on the old one
        dec, toks, tok = bring_up(be, mode)
        if dec is not None:
            gather_mode = mode
            break
        fallback_reason = f"{mode}: {toks}"
        log(f"[bench] rank {rank}: gather mode {mode} not usable ({toks})")
    if dec is None:
        raise SystemExit(f"PARITY / SETUP FAILURE: no gather mode produced the oracle's tokens ({fallback_reason})")
    n_chk = min(len(gold), len(toks))
    dist.barrier()
    be.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        tok = dec.step(tok, W + i)
    be.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    qb, _ = model.quant_bytes()
    # where a step's device time goes: a few EAGER steps with HIP events around every all-gather (collective: all ranks)
    prof = {"step_us": [], "gather_us": []}
    ptok = tok
    for i in range(4):
        ptok, su, gu = dec.profile_step(ptok, W + K + i)
        if i:  # (the first eager step after graph replays pays one-off costs)
            prof["step_us"].append(su)
            prof["gather_us"].append(gu)
    point_us = dec.last_point_us()
    log(f"[bench] rank {rank}: gather mode {gather_mode}; step mode {dec.mode}; eager step {prof['step_us'][-1] if prof['step_us'] else None} us, "
        f"of which gathers {prof['gather_us'][-1] if prof['gather_us'] else None} us; per gather point (first layer's four, then the last two): "
        f"{point_us[:4]} ... {point_us[-2:]}")
    n_gather = max(1, dec.n_points)
    gather_us = sum(prof["gather_us"]) / max(1, len(prof["gather_us"]))
    eager_us = sum(prof["step_us"]) / max(1, len(prof["step_us"]))
    step_mode = dec.mode
    dec.close()
    # rank 0: the unsharded model on the same GPU (what the N > 1 value is to be compared with) and the CPU baseline
    single, cpu = None, None
    if rank == 0:
        try:
            m1 = llama.Model(llama.preset("llama2-7b", 2048), llama.Q4_0, threads=16)
            s1 = llama.Session(m1, llama.hip_backend_fns(be))
            s1.resident_setup(be)
            w1 = s1.resident_decode(1, 0, 4)
            be.synchronize()
            t1 = time.perf_counter()
            s1.resident_decode(int(w1[-1]), 4, min(K, 128))
            be.synchronize()
            single = round(min(K, 128) / (time.perf_counter() - t1), 1)
            s1.close()
            m1.close()
        except Exception as e:
            single = str(e)[:200]
        if not args.skip_cpu:
            try:
                cpu = cpu_baseline(llama, llama.preset("smollm-135m"), llama.Q4_0)
                cpu["note"] = "the N = 1 line's baseline (SmolLM-135M Q4_0 through the C oracle on this host), repeated here so the line stands alone"
            except Exception as e:
                cpu = {"error": str(e)[:200]}
    dist.barrier()
    if rank == 0:
        tok_s = K / dt
        rank_gbps = qb * tok_s / 1e9  # this rank's weight bytes per token x tokens/s
        roof = {"bound": "hbm", "achieved": round(rank_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(rank_gbps / HBM_PEAK_GBPS, 4),
                "traffic": None, "kernel": "per-rank weight stream of one sharded decode token (all quantized mat-vecs of the rank)",
                "bytes_per_token_per_rank": qb, "all_gathers_per_token": dec.n_points if ws > 1 else 0,
                "us_per_all_gather": round(gather_us / n_gather, 2), "gather_us_per_token": round(gather_us, 1),
                "us_per_gather_point_first_layer": point_us[:4], "peer_access_row_rank0": peer_row, "device_ordinal_rank0": local,
                "eager_step_us": round(eager_us, 1), "gather_share_of_eager_step": round(gather_us / eager_us, 3) if eager_us else None,
                "timing": "value: host clock around K graph-replayed steps, max over ranks; gather figures: HIP events around every gather point "
                          "of 3 eager steps on rank 0 (zgml_hip_shard_profile_step)"}
        out = {
            "metric": "decode_tokens_per_sec", "value": round(K / dt, 1), "unit": "tokens/s", "n_gpus": ws,
            "steps": K, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / K, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Llama-2-7B Q4_0 greedy decode, batch 1, weight-row shard over {ws}xMI355X "
                                   "(BASELINE configs[3])",
                       "weights": "synthetic Q4_0", "max_seq": cfg.max_seq_len,
                       "parallelism": f"tp{ws} (N-split weights, replicated activations, head-sharded KV)",
                       "collectives_per_token": dec.n_points if ws > 1 else 0, "rccl_ranks": ws, "step_mode": step_mode,
                       "gather": gather_mode + (" (fallback: " + fallback_reason + ")" if fallback_reason else ""),
                       "data_path": "zgml_hip_shard_step (C ABI): op ranges + " + ("peer-store gather kernels (xGMI stores into the peers' fine-grained staging, "
                                    "arrival counters, bounded waits)" if gather_mode == "peer" else "in-place ncclAllGather") +
                                    " + (max, index) pair gather of the greedy token, one graph per token",
                       "verified": f"first {n_chk} greedy tokens equal the oracle's for the unsharded 32-layer program; all ranks agree",
                       "compare_with": "extra.llama2_7b.tok_s of the --gpus 1 line (same model, unsharded)"},
            "roofline": roof, "cpu_baseline": cpu,
            "extra": {"q4_0_weight_bytes_per_rank": qb, "unsharded_llama2_7b_tok_s_on_rank0_gpu": single,
                      "speedup_vs_unsharded": round(tok_s / single, 3) if isinstance(single, float) and single else None},
        }
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    model.close()
    be.close()
    dist.destroy_process_group()


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as FRESH child processes
    (torch.distributed.run, one per GPU, rendezvous on 127.0.0.1) before this process has made any GPU call, relay
    rank 0's JSON line, return the launcher's exit code. Nothing here touches HIP (no exec of a GPU process)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    log("[bench] launching", " ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line:
        print(line, flush=True)
    elif r.returncode == 0:
        log("[bench] ranks exited cleanly but printed no JSON line")
        return 1
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--skip-cpu", action="store_true", help="omit the cpu_baseline leg")
    ap.add_argument("--skip-llama7b", action="store_true", help="omit the Llama-2-7B @ 1 GPU extra leg")
    ap.add_argument("--skip-prefill", action="store_true", help="omit the Llama-2-7B prefill (configs[4]) leg")
    args = ap.parse_args()
    if not (ROOT / "zgml_amd" / "lib" / "libzgml_hip.so").exists():
        import __graft_entry__ as g
        g.build()
    if args.gpus <= 1 and not os.environ.get("ZGML_BENCH_FORCE_SHARDED"):
        bench_single(args)
    elif args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))  # no launcher around us: become one (before any GPU call in this process)
    else:
        bench_sharded(args)


if __name__ == "__main__":
    main()
