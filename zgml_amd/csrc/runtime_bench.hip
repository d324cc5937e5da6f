// runtime_bench.hip — the measurement entry points of include/zgml_hip.h (synthetic weights, mat-vec / matmul / dense-f16 timing
// loops on explicit streams, rings and hipGraphs, the device copy calibration): what bench.py and tools/ time. Split out of
// runtime.hip in round 4; nothing here is on the product's decode or prefill path.
#include "runtime_internal.h"

extern "C" {

static bool make_synth_weight(zgml_hip_ctx* ctx, uint32_t K, uint32_t N, int q4, uint32_t id, QWeightDev* w, uint32_t M = 1) {
    static const bool kon_on = !(getenv("ZGML_HIP_QMV_KON") && atoi(getenv("ZGML_HIP_QMV_KON")) == 0);
    w->format = q4 ? (M == 1 && kon_on ? QW_Q4K : QW_Q4) : QW_Q8; // (what compile_program picks for a weight that only feeds mat-vecs)
    w->K = K, w->N = N, w->bs = 32;
    w->KC = (K + 31) / 32;
    w->scale_f16 = 1;
    packed_bytes(w->format, 1, K, N, &w->qs_bytes, &w->sc_bytes);
    if (!CTX_CHECK(ctx, hipMalloc(&w->qs, w->qs_bytes)) || !CTX_CHECK(ctx, hipMalloc(&w->sc, w->sc_bytes))) return false;
    launch_synth_packed(ctx->stream, *w, id);
    return true;
}

double zgml_hip_dense_f16_bench(zgml_hip_ctx* ctx, uint32_t M, uint32_t K, uint32_t N, uint32_t n_matrices, uint32_t warmup,
                                uint32_t iters, uint64_t* bytes_per_launch) {
    if (!ctx || !f16_packable(K, N) || !n_matrices || !iters || !M) return -1.0;
    hipSetDevice(ctx->device);
    std::vector<void*> ring(n_matrices, nullptr);
    float *b32 = nullptr, *x = nullptr, *y = nullptr;
    void* a_scratch = nullptr;
    std::vector<float> bh((size_t)K * N), xh((size_t)M * K);
    for (size_t i = 0; i < bh.size(); i++) bh[i] = ((int)((i * 7 + (i >> 5) * 3) % 31) - 15) * 0.00390625f;
    for (size_t i = 0; i < xh.size(); i++) xh[i] = ((int)(i % 17) - 8) * 0.03125f;
    bool ok = CTX_CHECK(ctx, hipMalloc((void**)&b32, bh.size() * 4)) && CTX_CHECK(ctx, hipMalloc((void**)&x, xh.size() * 4)) &&
              CTX_CHECK(ctx, hipMalloc((void**)&y, (size_t)M * N * 4)) &&
              CTX_CHECK(ctx, hipMemcpy(b32, bh.data(), bh.size() * 4, hipMemcpyHostToDevice)) &&
              CTX_CHECK(ctx, hipMemcpy(x, xh.data(), xh.size() * 4, hipMemcpyHostToDevice));
    for (uint32_t i = 0; ok && i < n_matrices; i++) {
        ok = CTX_CHECK(ctx, hipMalloc(&ring[i], f16_packed_bytes(K, N)));
        if (ok) launch_pack_f16(ctx->stream, b32, N, 1, K, N, ring[i]);
    }
    double us = -1.0;
    if (ok) {
        DenseF16Params fp{y, x, nullptr, M, N, K, K, N, f16_packed_bytes(K, N) * n_matrices >= (192ull << 20) ? 1u : 0u};
        if (const uint64_t sb = dense_f16_scratch_bytes(M, K)) { // (kernels.h: the head in front of the block holds the K-split counters)
            ok = CTX_CHECK(ctx, hipMalloc(&a_scratch, kQmmScratchHead + sb)) && CTX_CHECK(ctx, hipMemsetAsync(a_scratch, 0, kQmmScratchHead, ctx->stream));
        }
        fp.scratch = a_scratch ? (char*)a_scratch + kQmmScratchHead : nullptr;
        for (uint32_t i = 0; ok && i < warmup; i++) {
            fp.bp = ring[i % n_matrices];
            launch_dense_f16(ctx->stream, fp);
        }
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        ok = CTX_CHECK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        if (ok) {
            for (uint32_t i = 0; i < n_matrices; i++) {
                fp.bp = ring[i];
                launch_dense_f16(ctx->stream, fp);
            }
            ok = CTX_CHECK(ctx, hipStreamEndCapture(ctx->stream, &g)) &&
                 CTX_CHECK(ctx, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        if (ok) {
            const uint32_t reps = (iters + n_matrices - 1) / n_matrices;
            hipGraphLaunch(ge, ctx->stream);
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, ctx->stream);
            for (uint32_t r = 0; r < reps; r++) hipGraphLaunch(ge, ctx->stream);
            hipEventRecord(e1, ctx->stream);
            if (CTX_CHECK(ctx, hipEventSynchronize(e1))) {
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                us = (double)ms * 1000.0 / ((double)reps * n_matrices);
            }
            hipEventDestroy(e0);
            hipEventDestroy(e1);
        }
        if (ge) hipGraphExecDestroy(ge);
        if (g) hipGraphDestroy(g);
    }
    if (bytes_per_launch) *bytes_per_launch = 2ull * K * N + 4ull * M * K + 4ull * M * N;
    for (void* r : ring) hipFree(r);
    hipFree(a_scratch);
    hipFree(b32);
    hipFree(x);
    hipFree(y);
    return us;
}

double zgml_hip_qmatvec_bench(zgml_hip_ctx* ctx, uint32_t K, uint32_t N, int q4, uint32_t n_matrices, uint32_t warmup,
                              uint32_t iters, uint64_t* bytes_per_launch) {
    return zgml_hip_qmatmul_bench(ctx, 1, K, N, q4, n_matrices, warmup, iters, bytes_per_launch);
}

double zgml_hip_qmatmul_bench(zgml_hip_ctx* ctx, uint32_t M, uint32_t K, uint32_t N, int q4, uint32_t n_matrices,
                              uint32_t warmup, uint32_t iters, uint64_t* bytes_per_launch) {
    if (!ctx || N % 32 || !n_matrices || !iters || !M) return -1.0;
    hipSetDevice(ctx->device);
    std::vector<QWeightDev> ring(n_matrices);
    bool ok = true;
    for (uint32_t i = 0; ok && i < n_matrices; i++) ok = make_synth_weight(ctx, K, N, q4, i, &ring[i], M);
    if (ok && (ring[0].qs_bytes + ring[0].sc_bytes) * n_matrices >= (192ull << 20)) // the ring stands for a model beyond the cache
        for (auto& w : ring) w.stream_nt = 1;
    float *x = nullptr, *y = nullptr, *scratch = nullptr;
    char* scratch_base = nullptr;
    std::vector<float> xh((size_t)M * K);
    for (size_t i = 0; i < xh.size(); i++) xh[i] = ((int)(i % 17) - 8) * 0.03125f;
    uint64_t sb = ok ? qmatmul_scratch_bytes(ring[0], M) : 0;
    ok = ok && CTX_CHECK(ctx, hipMalloc((void**)&x, xh.size() * 4)) && CTX_CHECK(ctx, hipMalloc((void**)&y, (size_t)M * N * 4)) &&
         (!sb || (CTX_CHECK(ctx, hipMalloc((void**)&scratch_base, kQmmScratchHead + sb)) && CTX_CHECK(ctx, hipMemsetAsync(scratch_base, 0, kQmmScratchHead, ctx->stream)))) &&
         CTX_CHECK(ctx, hipMemcpy(x, xh.data(), xh.size() * 4, hipMemcpyHostToDevice));
    if (scratch_base) scratch = (float*)(scratch_base + kQmmScratchHead);
    double us = -1.0;
    if (ok) {
        QMatmulParams qp{y, x, M, N, K, K, N};
        for (uint32_t i = 0; i < warmup; i++) launch_qmatmul(ctx->stream, ring[i % n_matrices], qp, scratch);
        // One pass over the ring is captured into a hipGraph and replayed: back-to-back eager
        // launches are host-bound below ~3 us per kernel on this platform, a graph is not.
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        ok = CTX_CHECK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        if (ok) {
            for (uint32_t i = 0; i < n_matrices; i++) launch_qmatmul(ctx->stream, ring[i], qp, scratch);
            ok = CTX_CHECK(ctx, hipStreamEndCapture(ctx->stream, &g));
            if (ok) dump_graph(g, "ring");
            ok = ok && CTX_CHECK(ctx, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        if (ok) {
            const uint32_t reps = (iters + n_matrices - 1) / n_matrices;
            hipGraphLaunch(ge, ctx->stream); // warm replay
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, ctx->stream);
            for (uint32_t r = 0; r < reps; r++) hipGraphLaunch(ge, ctx->stream);
            hipEventRecord(e1, ctx->stream);
            if (CTX_CHECK(ctx, hipEventSynchronize(e1))) {
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                us = (double)ms * 1000.0 / ((double)reps * n_matrices);
            }
            hipEventDestroy(e0);
            hipEventDestroy(e1);
        }
        if (ge) hipGraphExecDestroy(ge);
        if (g) hipGraphDestroy(g);
    }
    if (bytes_per_launch) *bytes_per_launch = (uint64_t)K * N / 32 * (q4 ? 18 : 34) + 4ull * M * K + 4ull * M * N;
    for (auto& w : ring) {
        hipFree(w.qs);
        hipFree(w.sc);
    }
    hipFree(x);
    hipFree(y);
    hipFree(scratch_base);
    return us;
}

// ── dependent CHAIN of square mat-vecs (true data dependency) ───────────────────────────────────────────────────
// Launch i computes y_i = (x_i^T W_i) * c and launch i + 1 consumes y_i as its x (ping-pong vectors; the epilogue
// multiply by a constant vector keeps the magnitudes bounded), so consecutive launches are ordered by DATA, not only by
// the stream. (Round-2 experiments on top of this chain — an Infinity-Cache warmer kernel on a second graph branch,
// per launch and as one progress-paced kernel per pass — made it slower or stalled it)
double zgml_hip_qmatvec_chain_bench(zgml_hip_ctx* ctx, uint32_t K, int q4, uint32_t n_matrices, uint32_t warmup, uint32_t iters,
                                    uint64_t* bytes_per_launch) {
    const uint32_t N = K;
    if (!ctx || N % 32 || n_matrices < 2 || n_matrices % 2 || !iters) return -1.0;
    hipSetDevice(ctx->device);
    std::vector<QWeightDev> ring(n_matrices);
    bool ok = true;
    for (uint32_t i = 0; ok && i < n_matrices; i++) ok = make_synth_weight(ctx, K, N, q4, i, &ring[i]);
    if (ok && (ring[0].qs_bytes + ring[0].sc_bytes) * n_matrices >= (192ull << 20))
        for (auto& w : ring) w.stream_nt = 1;
    float *v0 = nullptr, *v1 = nullptr, *cvec = nullptr;
    std::vector<float> xh(K), ch(N, 0.2f);
    for (size_t i = 0; i < xh.size(); i++) xh[i] = ((int)(i % 17) - 8) * 0.03125f;
    ok = ok && CTX_CHECK(ctx, hipMalloc((void**)&v0, K * 4)) && CTX_CHECK(ctx, hipMalloc((void**)&v1, K * 4)) &&
         CTX_CHECK(ctx, hipMalloc((void**)&cvec, N * 4)) && CTX_CHECK(ctx, hipMemcpy(v0, xh.data(), K * 4, hipMemcpyHostToDevice)) &&
         CTX_CHECK(ctx, hipMemcpy(cvec, ch.data(), N * 4, hipMemcpyHostToDevice));
    double us = -1.0;
    if (ok) {
        // ZGML_HIP_QMV_TRACE=1 (trace build of the library): in-kernel stamps of ONE launch in the middle of the chain, printed below
        unsigned long long* trace = nullptr;
        if (getenv("ZGML_HIP_QMV_TRACE") && atoi(getenv("ZGML_HIP_QMV_TRACE")) &&
            hipHostMalloc((void**)&trace, 16 * sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess)
            memset(trace, 0, 16 * sizeof(unsigned long long));
        auto one = [&](uint32_t i) { // launch i: x = (i even ? v0 : v1), y = the other (even ring: the wrap keeps the ping-pong)
            QmvLaunch L;
            L.trace = i == n_matrices / 2 ? trace : nullptr;
            L.n_parts = 1, L.K = K;
            L.parts[0].w = ring[i];
            L.parts[0].dst = (i & 1) ? v0 : v1;
            L.parts[0].n_epi = 1;
            L.parts[0].epi[0] = QmvEpiStep{ZGML_OP_MUL, 0, cvec, (i & 1) ? v0 : v1};
            L.pro.kind = QMV_PRO_NONE, L.pro.a = (i & 1) ? v1 : v0;
            launch_qmatvec_fused(ctx->stream, L);
        };
        for (uint32_t i = 0; i < warmup; i++) one(i % n_matrices);
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        ok = CTX_CHECK(ctx, hipStreamSynchronize(ctx->stream)) && CTX_CHECK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        if (ok) {
            for (uint32_t i = 0; i < n_matrices; i++) one(i);
            ok = CTX_CHECK(ctx, hipStreamEndCapture(ctx->stream, &g)) && CTX_CHECK(ctx, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        if (ok) {
            const uint32_t reps = (iters + n_matrices - 1) / n_matrices;
            hipGraphLaunch(ge, ctx->stream);
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, ctx->stream);
            for (uint32_t r = 0; r < reps; r++) hipGraphLaunch(ge, ctx->stream);
            hipEventRecord(e1, ctx->stream);
            if (CTX_CHECK(ctx, hipEventSynchronize(e1))) {
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                us = (double)ms * 1000.0 / ((double)reps * n_matrices);
            }
            hipEventDestroy(e0);
            hipEventDestroy(e1);
        }
        if (ge) hipGraphExecDestroy(ge);
        if (g) hipGraphDestroy(g);
        if (trace) { // 100 MHz wall clock -> ns; workgroup 0, and (K-on-lanes body) the last workgroup
            const unsigned long long* t = trace;
            fprintf(stderr, "[zgml_hip] chain mat-vec stamps, launch %u of the ring (ns): workgroup 0: start->loads issued %lld | ->streamed %lld | ->reduced+epilogue+stored %lld",
                    n_matrices / 2, (long long)(t[1] - t[0]) * 10, (long long)(t[4] - t[1]) * 10, (long long)(t[5] - t[4]) * 10);
            if (t[8])
                fprintf(stderr, " || last workgroup: starts %lld after workgroup 0 | ->loads issued %lld | ->streamed %lld | ->end %lld || first start -> last end %lld",
                        (long long)(t[8] - t[0]) * 10, (long long)(t[9] - t[8]) * 10, (long long)(t[12] - t[9]) * 10, (long long)(t[13] - t[12]) * 10,
                        (long long)(std::max(t[13], t[5]) - t[0]) * 10);
            fprintf(stderr, "\n");
            hipHostFree(trace);
        }
    }
    if (bytes_per_launch) *bytes_per_launch = (uint64_t)K * N / 32 * (q4 ? 18 : 34) + 4ull * K + 4ull * N;
    for (auto& w : ring) {
        hipFree(w.qs);
        hipFree(w.sc);
    }
    hipFree(v0);
    hipFree(v1);
    hipFree(cvec);
    return us;
}

// The same ring of M = 1 mat-vecs, but as INDEPENDENT launches: the captured graph forks the ring over
// `n_streams` branches (distinct outputs per branch), so consecutive launches may overlap on the device.
// Not the decode path (there every mat-vec waits for its predecessor) — it separates what the kernel can
// stream from what a dependent launch of this size costs. Returns microseconds per launch.
double zgml_hip_qmatvec_overlap_bench(zgml_hip_ctx* ctx, uint32_t K, uint32_t N, int q4, uint32_t n_matrices, uint32_t n_streams,
                                      uint32_t iters, uint64_t* bytes_per_launch) {
    if (!ctx || N % 32 || !n_matrices || !iters || !n_streams || n_streams > 16) return -1.0;
    hipSetDevice(ctx->device);
    std::vector<QWeightDev> ring(n_matrices);
    bool ok = true;
    for (uint32_t i = 0; ok && i < n_matrices; i++) ok = make_synth_weight(ctx, K, N, q4, i, &ring[i]);
    if (ok && (ring[0].qs_bytes + ring[0].sc_bytes) * n_matrices >= (192ull << 20))
        for (auto& w : ring) w.stream_nt = 1;
    float *x = nullptr, *y = nullptr;
    std::vector<float> xh(K);
    for (size_t i = 0; i < xh.size(); i++) xh[i] = ((int)(i % 17) - 8) * 0.03125f;
    ok = ok && CTX_CHECK(ctx, hipMalloc((void**)&x, xh.size() * 4)) && CTX_CHECK(ctx, hipMalloc((void**)&y, (size_t)n_streams * N * 4)) &&
         CTX_CHECK(ctx, hipMemcpy(x, xh.data(), xh.size() * 4, hipMemcpyHostToDevice));
    std::vector<hipStream_t> side(n_streams, nullptr);
    std::vector<hipEvent_t> joined(n_streams, nullptr);
    hipEvent_t fork = nullptr;
    for (uint32_t t = 1; ok && t < n_streams; t++)
        ok = CTX_CHECK(ctx, hipStreamCreateWithFlags(&side[t], hipStreamNonBlocking)) && CTX_CHECK(ctx, hipEventCreateWithFlags(&joined[t], hipEventDisableTiming));
    ok = ok && CTX_CHECK(ctx, hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    side[0] = ctx->stream;
    double us = -1.0;
    if (ok) {
        for (uint32_t i = 0; i < n_matrices; i++) { // warm (and resolve the kernels) outside the capture
            QMatmulParams qp{y + (size_t)(i % n_streams) * N, x, 1, N, K, K, N};
            launch_qmatmul(ctx->stream, ring[i], qp, nullptr);
        }
        hipStreamSynchronize(ctx->stream);
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        ok = CTX_CHECK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        if (ok) {
            hipEventRecord(fork, ctx->stream);
            for (uint32_t t = 1; t < n_streams; t++) hipStreamWaitEvent(side[t], fork, 0);
            for (uint32_t i = 0; i < n_matrices; i++) {
                QMatmulParams qp{y + (size_t)(i % n_streams) * N, x, 1, N, K, K, N};
                launch_qmatmul(side[i % n_streams], ring[i], qp, nullptr);
            }
            for (uint32_t t = 1; t < n_streams; t++) {
                hipEventRecord(joined[t], side[t]);
                hipStreamWaitEvent(ctx->stream, joined[t], 0);
            }
            ok = CTX_CHECK(ctx, hipStreamEndCapture(ctx->stream, &g)) && CTX_CHECK(ctx, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        if (ok) {
            const uint32_t reps = (iters + n_matrices - 1) / n_matrices;
            hipGraphLaunch(ge, ctx->stream);
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, ctx->stream);
            for (uint32_t r = 0; r < reps; r++) hipGraphLaunch(ge, ctx->stream);
            hipEventRecord(e1, ctx->stream);
            if (CTX_CHECK(ctx, hipEventSynchronize(e1))) {
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                us = (double)ms * 1000.0 / ((double)reps * n_matrices);
            }
            hipEventDestroy(e0);
            hipEventDestroy(e1);
        }
        if (ge) hipGraphExecDestroy(ge);
        if (g) hipGraphDestroy(g);
    }
    if (bytes_per_launch) *bytes_per_launch = (uint64_t)K * N / 32 * (q4 ? 18 : 34) + 4ull * K + 4ull * N;
    for (uint32_t t = 1; t < n_streams; t++) {
        if (side[t]) hipStreamDestroy(side[t]);
        if (joined[t]) hipEventDestroy(joined[t]);
    }
    if (fork) hipEventDestroy(fork);
    for (auto& w : ring) {
        hipFree(w.qs);
        hipFree(w.sc);
    }
    hipFree(x);
    hipFree(y);
    return us;
}

// The same ring as INDEPENDENT launches on EXPLICIT streams: stream t replays its own graph of the matrices i = t (mod n_streams),
// the n_streams replays run concurrently on their own hardware queues (the fork / join form above puts the branches of ONE graph
// wherever the runtime likes — on one queue they serialise). Throughput of overlapping launches, not the decode path's figure.
double zgml_hip_qmatvec_streams_bench(zgml_hip_ctx* ctx, uint32_t K, uint32_t N, int q4, uint32_t n_matrices, uint32_t n_streams, uint32_t iters,
                                      uint64_t* bytes_per_launch) {
    if (!ctx || N % 32 || !n_matrices || !iters || !n_streams || n_streams > 16 || n_matrices % n_streams) return -1.0;
    hipSetDevice(ctx->device);
    std::vector<QWeightDev> ring(n_matrices);
    bool ok = true;
    for (uint32_t i = 0; ok && i < n_matrices; i++) ok = make_synth_weight(ctx, K, N, q4, i, &ring[i]);
    if (ok && (ring[0].qs_bytes + ring[0].sc_bytes) * n_matrices >= (192ull << 20))
        for (auto& w : ring) w.stream_nt = 1;
    float *x = nullptr, *y = nullptr;
    std::vector<float> xh(K);
    for (size_t i = 0; i < xh.size(); i++) xh[i] = ((int)(i % 17) - 8) * 0.03125f;
    ok = ok && CTX_CHECK(ctx, hipMalloc((void**)&x, xh.size() * 4)) && CTX_CHECK(ctx, hipMalloc((void**)&y, (size_t)n_streams * N * 4)) &&
         CTX_CHECK(ctx, hipMemcpy(x, xh.data(), xh.size() * 4, hipMemcpyHostToDevice));
    std::vector<hipStream_t> st(n_streams, nullptr);
    std::vector<hipGraph_t> g(n_streams, nullptr);
    std::vector<hipGraphExec_t> ge(n_streams, nullptr);
    std::vector<hipEvent_t> done(n_streams, nullptr);
    for (uint32_t t = 0; ok && t < n_streams; t++)
        ok = CTX_CHECK(ctx, hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking)) && CTX_CHECK(ctx, hipEventCreateWithFlags(&done[t], hipEventDisableTiming));
    double us = -1.0;
    if (ok) {
        for (uint32_t i = 0; i < n_matrices; i++) { // warm (and resolve the kernels) outside the captures
            QMatmulParams qp{y + (size_t)(i % n_streams) * N, x, 1, N, K, K, N};
            launch_qmatmul(ctx->stream, ring[i], qp, nullptr);
        }
        hipStreamSynchronize(ctx->stream);
        for (uint32_t t = 0; ok && t < n_streams; t++) {
            ok = CTX_CHECK(ctx, hipStreamBeginCapture(st[t], hipStreamCaptureModeThreadLocal));
            for (uint32_t i = t; ok && i < n_matrices; i += n_streams) {
                QMatmulParams qp{y + (size_t)t * N, x, 1, N, K, K, N};
                launch_qmatmul(st[t], ring[i], qp, nullptr);
            }
            ok = ok && CTX_CHECK(ctx, hipStreamEndCapture(st[t], &g[t])) && CTX_CHECK(ctx, hipGraphInstantiate(&ge[t], g[t], nullptr, nullptr, 0));
        }
    }
    if (ok) {
        const uint32_t reps = (iters + n_matrices - 1) / n_matrices;
        for (uint32_t t = 0; t < n_streams; t++) hipGraphLaunch(ge[t], st[t]);
        for (uint32_t t = 0; t < n_streams; t++) hipStreamSynchronize(st[t]);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, ctx->stream); // the timed region starts on the context stream; every side stream waits for it
        for (uint32_t t = 0; t < n_streams; t++) hipStreamWaitEvent(st[t], e0, 0);
        for (uint32_t r = 0; r < reps; r++)
            for (uint32_t t = 0; t < n_streams; t++) hipGraphLaunch(ge[t], st[t]);
        for (uint32_t t = 0; t < n_streams; t++) {
            hipEventRecord(done[t], st[t]);
            hipStreamWaitEvent(ctx->stream, done[t], 0);
        }
        hipEventRecord(e1, ctx->stream);
        if (CTX_CHECK(ctx, hipEventSynchronize(e1))) {
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            us = (double)ms * 1000.0 / ((double)reps * n_matrices);
        }
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
    if (bytes_per_launch) *bytes_per_launch = (uint64_t)K * N / 32 * (q4 ? 18 : 34) + 4ull * K + 4ull * N;
    for (uint32_t t = 0; t < n_streams; t++) {
        if (ge[t]) hipGraphExecDestroy(ge[t]);
        if (g[t]) hipGraphDestroy(g[t]);
        if (st[t]) hipStreamDestroy(st[t]);
        if (done[t]) hipEventDestroy(done[t]);
    }
    for (auto& w : ring) {
        hipFree(w.qs);
        hipFree(w.sc);
    }
    hipFree(x);
    hipFree(y);
    return us;
}

int zgml_hip_qmatvec_synth(zgml_hip_ctx* ctx, uint32_t K, uint32_t N, int q4, uint32_t matrix_id, const float* x_host,
                           float* y_host) {
    if (!ctx || N % 32 || !x_host || !y_host) return -1;
    hipSetDevice(ctx->device);
    QWeightDev w{};
    float *x = nullptr, *y = nullptr, *scratch = nullptr;
    char* scratch_base = nullptr;
    bool ok = make_synth_weight(ctx, K, N, q4, matrix_id, &w);
    uint64_t sb = ok ? qmatmul_scratch_bytes(w, 1) : 0;
    ok = ok && CTX_CHECK(ctx, hipMalloc((void**)&x, K * 4)) && CTX_CHECK(ctx, hipMalloc((void**)&y, N * 4)) &&
         (!sb || (CTX_CHECK(ctx, hipMalloc((void**)&scratch_base, kQmmScratchHead + sb)) && CTX_CHECK(ctx, hipMemsetAsync(scratch_base, 0, kQmmScratchHead, ctx->stream)))) &&
         CTX_CHECK(ctx, hipMemcpyAsync(x, x_host, K * 4, hipMemcpyHostToDevice, ctx->stream));
    if (scratch_base) scratch = (float*)(scratch_base + kQmmScratchHead);
    if (ok) {
        QMatmulParams qp{y, x, 1, N, K, K, N};
        launch_qmatmul(ctx->stream, w, qp, scratch);
        ok = CTX_CHECK(ctx, hipMemcpyAsync(y_host, y, N * 4, hipMemcpyDeviceToHost, ctx->stream)) &&
             CTX_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    hipFree(w.qs);
    hipFree(w.sc);
    hipFree(x);
    hipFree(y);
    hipFree(scratch_base);
    return ok ? 0 : -1;
}

double zgml_hip_copy_bench(zgml_hip_ctx* ctx, uint64_t bytes, uint32_t warmup, uint32_t iters) {
    if (!ctx || !iters || bytes < 16) return -1.0;
    hipSetDevice(ctx->device);
    void *a = nullptr, *b = nullptr;
    if (!CTX_CHECK(ctx, hipMalloc(&a, bytes)) || !CTX_CHECK(ctx, hipMalloc(&b, bytes))) {
        hipFree(a);
        return -1.0;
    }
    hipMemsetAsync(a, 1, bytes, ctx->stream);
    for (uint32_t i = 0; i < warmup; i++) launch_copy_f4(ctx->stream, b, a, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, ctx->stream);
    for (uint32_t i = 0; i < iters; i++) launch_copy_f4(ctx->stream, b, a, bytes);
    hipEventRecord(e1, ctx->stream);
    double us = -1.0;
    if (CTX_CHECK(ctx, hipEventSynchronize(e1))) {
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        us = (double)ms * 1000.0 / iters;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(a);
    hipFree(b);
    return us;
}

} // extern "C"
