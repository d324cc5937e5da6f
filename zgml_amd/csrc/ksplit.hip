// ksplit.hip — the decoder layer of a short-K model as launches that end at a K-SPLIT (kernels.h: KsVec and friends).
//
// Why: a SmolLM-135M layer as four launches (q/k/v + attention | O | gate / up | down) is four all-to-all seams of ~4.3 us each,
// and every launch keeps a few dozen workgroups busy for ~1.5 us of its own latency chain (profiles/r04_smollm135m_decode_by_grid.txt).
// The O and down projections are where the seams can go: their K dimension is exactly what ONE producer workgroup holds —
// a head's attention output (d_head values) / a gate-up workgroup's silu(gate) * up (32 values) — so the producer multiplies
// those by ITS rows of the weight and stores a partial output vector (the weight's scale blocks run along n, so a slice of K
// rows is a set of whole blocks). The partials are summed, in a fixed order, by the launch that reads the vector next:
//     launch A1  [sum of the down partials + residual -> rmsnorm -> * gamma -> q / k / v]
//     launch A2  [rope, KV stores, attention of each head -> that head's partial of the O projection]
//     launch B   [sum of the O partials + residual -> rmsnorm -> * gamma -> gate / up -> SiLU chain, product -> partial of down]
// Every buffer the op-by-op plan writes is still written (by workgroup 0 of the consuming launch for the deferred vector).
// Results: the mat-vec sums differ from the reference's k-sequential loop in summation order only (2e-5 sum|x w|),
// two executions give identical bits (no atomics on data, every order fixed).
//
// Weights: the n-on-lanes packed Q4_0 layout of qmatvec.hip with f16 scales (QW_Q4):
//     qs: uint4[N/16][KC][16]   item (g, c, i) = the 32 nibbles of column 16 g + i for k = 32 c .. 32 c + 31
//     sc: half2[N/32][KC][16]   entry i = { scale(32 c + i, j), scale(32 c + 16 + i, j) }, j = g / 2
// A DPP row of 16 lanes computes one item ("unit": 16 columns x 32 k) with the row_newbcast FMAs of qmv_common.h; the rows of a
// workgroup divide units, not the lanes of a row.
#include "qmv_common.h"
#include "attention_decode.h"

#include <hip/hip_fp16.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

namespace zgml {

namespace {

struct KsUnit {
    uint4 wq;
    uint32_t s2; // half2 { scale(k = 32 c + i), scale(k = 32 c + 16 + i) }
};
__device__ __forceinline__ KsUnit ks_load(const uint4* qs, const uint32_t* sc, uint32_t KC, uint32_t g, uint32_t c, uint32_t i) {
    KsUnit u;
    u.wq = qs[((uint64_t)g * KC + c) * 16 + i];
    u.s2 = sc[((uint64_t)(g >> 1) * KC + c) * 16 + i];
    return u;
}
// acc += the unit's 32 k x this lane's column; xa / xb = x[32 c + i], x[32 c + 16 + i] (every lane of the row runs this: the DPP
// broadcasts read the row's lanes); !ok: the unit is a clamped duplicate and contributes 0 (its weights are finite)
__device__ __forceinline__ void ks_fma(const KsUnit& u, float xa, float xb, bool ok, float& a0, float& a1, float& a2, float& a3) {
    const __half2 h = *(const __half2*)&u.s2;
    float tA = ok ? (__low2float(h) * 16.0f) * xa : 0.f; // (scale * x first, as the reference rounds it — reference.zig:552; the 16 undoes v_cvt_off_f32_i4's 1/16)
    float tB = ok ? (__high2float(h) * 16.0f) * xb : 0.f;
    dpp_fence(tA, tB);
    q4_dword<0>(a0, a1, u.wq.x, tA);
    q4_dword<8>(a2, a3, u.wq.y, tA);
    q4_dword<0>(a0, a1, u.wq.z, tB);
    q4_dword<8>(a2, a3, u.wq.w, tB);
}

// A barrier for exchanges through LDS only: __syncthreads() is also a workgroup-scope fence over GLOBAL memory, i.e. s_waitcnt
// vmcnt(0) — every barrier of a launch would wait for the weight loads requested at its start and, in the workgroup that stores the
// absorbed ops' buffers, for those stores (in-kernel stamps: 0.5-1 us per barrier there). Nothing in these kernels hands global
// data from one wave to another inside a launch.
__device__ __forceinline__ void ks_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// block_total (qmv_common.h) without the trailing barrier: `red` is used once per kernel
__device__ __forceinline__ float ks_block_total(float v, float* red, uint32_t bdim) {
    v = rows_sum4(row16_sum(v));
    const uint32_t n_waves = bdim >> 6, slot = threadIdx.x & 15;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    ks_barrier();
    const float t = red[min(slot, n_waves - 1)];
    return row16_sum(slot < n_waves ? t : 0.f);
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// LDS carve-up of the launches with a prologue (floats): xs [KC * 32] | sl [4 * S * Q] (S * Q <= T) | red [16] | red2 [T] | su [32]
struct KsLds {
    float* xs;
    float4* sl;
    float* red;
    float* red2;
    float* su;
};
__device__ __forceinline__ KsLds ks_lds(float* smem, uint32_t KC, uint32_t T) {
    KsLds l;
    l.xs = smem;
    l.sl = (float4*)(smem + KC * 32);
    l.red = smem + KC * 32 + 4 * T;
    l.red2 = l.red + 16;
    l.su = l.red2 + T;
    return l;
}
inline size_t ks_lds_bytes(uint32_t KC, uint32_t T) { return ((size_t)KC * 32 + 4 * (size_t)T + 16 + T + 32) * sizeof(float); }

// The prologue every workgroup of a consuming launch runs: x = base + sum of the parts (KsVec), h = x, then (gamma != nullptr)
// mid = h / sqrt(mean(h^2) + eps) (reference.zig:365), xs = mid * gamma -> LDS, zero-padded to whole 32-k chunks. T threads, all of
// them call this (barriers inside); K % 4 == 0, K / 4 <= T, vectors 16-byte aligned (host-checked). `store`: this workgroup also
// writes the absorbed ops' outputs (sum_dst, add_dst, store_mid, store_x).
// Summation order (fixed by K, T and n_parts alone): S = min(T / (K / 4), n_parts) slices; slice s adds parts s, s + S, s + 2 S, ...
// in that order; the slices are then added in slice order; the residual base comes last (h = sum + base: exactly the residual
// add over the stored projection output).
// diagnostics build: stamps of workgroup (0, 0), 100 MHz wall clock, kept in registers and written out by KS_FLUSH at the very
// end (a store to the host-mapped trace block in mid-kernel sits in front of every later vmcnt wait of the stamping wave)
struct KsStamps {
    unsigned long long t[12];
    bool on;
    uint32_t base; // 0: thread 0's stamps, 16: those of lane 0 of the workgroup's LAST wave
};
#ifdef ZGML_TRACE
#define KS_STAMPS(ts, trace, first) KsStamps ts; ts.on = (trace) && (first) && (threadIdx.x == 0 || threadIdx.x == KS_T - 64); ts.base = threadIdx.x ? 16 : 0; for (int z_ = 0; z_ < 12; z_++) ts.t[z_] = 0
#define KS_STAMP(ts, i) do { if ((ts).on) (ts).t[i] = wall_clock64(); } while (0)
#define KS_FLUSH(ts, trace) do { if ((ts).on) for (int z_ = 0; z_ < 12; z_++) (trace)[(ts).base + z_] = (ts).t[z_]; } while (0)
#else
#define KS_STAMPS(ts, trace, first) KsStamps ts; ts.on = false; ts.base = 0
#define KS_STAMP(ts, i) do { } while (0)
#define KS_FLUSH(ts, trace) do { } while (0)
#endif
// The rmsnorm is DEFERRED (the mat-vec is linear in x): the rows multiply by h * gamma, and the finished sums are scaled by
// inv = 1 / sqrt(mean(h^2) + eps) (reference.zig:365) — the same reassociation as the PRENORM launches of qmatvec.hip, inside the
// 2e-5 sum|x w| contract. The prologue leaves one sum of h^2 per wave in red[0 .. waves); the barrier that publishes xs publishes
// them too, so ANY thread can finish the factor later (ks_inv) without another barrier, off the path to the first FMA.
// `store` (workgroup 0): the absorbed ops' buffers are stored as soon as their values exist — the barriers of these kernels do not
// wait for vector memory (ks_barrier), so the stores complete in the background instead of lengthening the kernel's tail.
__device__ __forceinline__ float ks_inv(const KsNormIn& in, const float* red, uint32_t T) { // the same value in every thread (fixed order: lanes, rows, waves)
    const uint32_t n_waves = T >> 6, slot = threadIdx.x & 15;
    const float t = red[min(slot, n_waves - 1)];
    const float total = row16_sum(slot < n_waves ? t : 0.f);
    return 1.0f / sqrtf(total / (float)in.K + in.eps); // reference.zig:365
}
// The prologue every workgroup of a consuming launch runs: h = base + sum of the parts (KsVec), xs = h * gamma -> LDS (zero-padded
// to whole 32-k chunks), per-wave sums of h^2 -> red. T threads, all of them call this (barriers inside); K % 4 == 0, K / 4 <= T,
// vectors 16-byte aligned (host-checked). `after_issue()` requests the caller's weights (in front of the prologue's own loads).
// Summation order (fixed by K, T and n_parts alone): up to 12 parts — every quad's thread adds them in part order; more — S =
// min(T / (K / 4), n_parts) slices, slice s adds parts s, s + S, s + 2 S, ... in that order and the slices are added in slice
// order; the residual base comes last (h = sum + base: exactly the residual add over the stored projection output).
template <typename F>
__device__ __forceinline__ void ks_prologue(const KsNormIn& in, const KsLds& l, uint32_t KC, uint32_t T, bool store, KsStamps& tr, F after_issue) {
    const uint32_t K = in.K, Q = K >> 2, tid = threadIdx.x, P = in.x.n_parts;
    const bool mine = tid < Q;
    const uint32_t qc = mine ? tid : 0; // clamped: the loads below are unconditional
    after_issue(); // (measured: the weights FIRST — with the parts in front of them the first FMA waits for the weights instead, -2 % tok/s)
    const float4 base4 = ldg4(in.x.base + 4 * qc);
    const float4 g4 = in.gamma ? ldg4(in.gamma + 4 * qc) : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 h = base4, sum = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr uint32_t kFly = 12; // parts one thread keeps in flight (clamped, unconditional loads)
    if (P && P <= kFly) { // (uniform) few parts (the heads' O partials): every quad's thread sums them itself, in part order — no exchange
        float4 v[kFly];
#pragma unroll
        for (uint32_t j = 0; j < kFly; j++) v[j] = ldg4(in.x.parts + (uint64_t)(j < P ? j : 0) * K + 4 * qc);
        sum = v[0];
#pragma unroll
        for (uint32_t j = 1; j < kFly; j++)
            if (j < P) sum = add4(sum, v[j]);
        KS_STAMP(tr, 2);
        h = add4(sum, base4);
    } else if (P) { // (uniform) many parts (the gate / up workgroups' down partials): S slices of the parts per quad, exchanged through LDS
        uint32_t S = T / Q;
        S = S > P ? P : S;
        const uint32_t q = tid % Q, s = tid / Q;
        const bool act = s < S;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        { // the slice's first kFly parts: requested by every thread (clamped), in front of the caller's weights
            float4 v[kFly];
#pragma unroll
            for (uint32_t j = 0; j < kFly; j++) {
                const uint32_t p = s + j * S;
                v[j] = ldg4(in.x.parts + (uint64_t)(act && p < P ? p : 0) * K + 4 * q);
            }
#pragma unroll
            for (uint32_t j = 0; j < kFly; j++)
                if (act && s + j * S < P) a = add4(a, v[j]);
        }
        for (uint32_t p0 = s + kFly * S; act && p0 < P; p0 += kFly * S) { // (more than 12 parts per slice: n_parts > 12 T / (K / 4))
            float4 v[kFly];
#pragma unroll
            for (uint32_t j = 0; j < kFly; j++) {
                const uint32_t p = p0 + j * S;
                v[j] = ldg4(in.x.parts + (uint64_t)(p < P ? p : p0) * K + 4 * q);
            }
#pragma unroll
            for (uint32_t j = 0; j < kFly; j++)
                if (p0 + j * S < P) a = add4(a, v[j]);
        }
        if (act) l.sl[s * Q + q] = a;
        KS_STAMP(tr, 2); // this thread's parts arrived and are summed
        ks_barrier();
        float4 sv[16]; // (S <= 16: T <= 1024, Q >= 64 for K >= 256; smaller K: the loop below)
        sum = l.sl[qc];
        if (S <= 16) {
#pragma unroll
            for (uint32_t s2 = 1; s2 < 16; s2++) sv[s2] = l.sl[(s2 < S ? s2 : 0) * Q + qc]; // independent reads, issued together
#pragma unroll
            for (uint32_t s2 = 1; s2 < 16; s2++)
                if (s2 < S) sum = add4(sum, sv[s2]);
        } else {
            for (uint32_t s2 = 1; s2 < S; s2++) sum = add4(sum, l.sl[s2 * Q + qc]);
        }
        h = add4(sum, base4);
    }
    if (store && mine && P) {
        if (in.x.sum_dst) stg4(in.x.sum_dst + 4 * tid, sum);
        if (in.x.add_dst) stg4(in.x.add_dst + 4 * tid, h);
    }
    KS_STAMP(tr, 3); // the deferred vector is whole
    if (mine) *(float4*)(l.xs + 4 * tid) = make_float4(h.x * g4.x, h.y * g4.y, h.z * g4.z, h.w * g4.w);
    for (uint32_t z = Q + tid; z < KC * 8; z += T) *(float4*)(l.xs + 4 * z) = make_float4(0.f, 0.f, 0.f, 0.f); // k >= K of the last chunk
    if (in.gamma) { // (uniform) this wave's sum of h^2
        const float ss = rows_sum4(row16_sum(mine ? (h.x * h.x + h.y * h.y) + (h.z * h.z + h.w * h.w) : 0.f));
        if ((tid & 63) == 0) l.red[tid >> 6] = ss;
    }
    ks_barrier();
    KS_STAMP(tr, 4); // x staged
    if (store && in.gamma) { // (uniform) mid = h * inv, x = mid * gamma: the reference's order for the buffers a host can read
        const float inv = ks_inv(in, l.red, T);
        if (mine) {
            const float4 mid = make_float4(h.x * inv, h.y * inv, h.z * inv, h.w * inv);
            if (in.store_mid) stg4(in.store_mid + 4 * tid, mid);
            if (in.store_x) stg4(in.store_x + 4 * tid, make_float4(mid.x * g4.x, mid.y * g4.y, mid.z * g4.z, mid.w * g4.w));
        }
    }
}

// the units of one DPP row: column group g of (qs, sc), chunks slot, slot + n_slots, ... (U of them at most; clamped duplicates beyond KC)
template <int U>
__device__ __forceinline__ void ks_row_load(KsUnit (&u)[U], const uint4* qs, const uint32_t* sc, uint32_t KC, uint32_t g, uint32_t slot, uint32_t n_slots, uint32_t i) {
#pragma unroll
    for (int j = 0; j < U; j++) {
        const uint32_t c = slot + j * n_slots;
        u[j] = ks_load(qs, sc, KC, g, c < KC ? c : KC - 1, i);
    }
}
template <int U>
__device__ __forceinline__ float ks_row_fma(const KsUnit (&u)[U], const float* xs, uint32_t KC, uint32_t slot, uint32_t n_slots, uint32_t i) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int j = 0; j < U; j++) {
        const uint32_t c = slot + j * n_slots;
        const bool ok = c < KC;
        const uint32_t cc = ok ? c : KC - 1;
        ks_fma(u[j], xs[cc * 32 + i], xs[cc * 32 + 16 + i], ok, a0, a1, a2, a3);
    }
    return (a0 + a1) + (a2 + a3);
}

// ── launch A1: [deferred vector -> rmsnorm -> gamma] -> up to three mat-vecs (q / k / v) ─────────────────────────────────────
// Workgroup b owns GP consecutive column groups of one matrix and all of K; row (w, r) takes column group r % GP and the chunks
// of slot (4 w + r) / GP; rows of a wave that share a column group are folded by the row swaps, the waves through LDS by wave 0.
struct KsProjArgs {
    KsNormIn in;
    const uint4* qs[3];
    const uint32_t* sc[3];
    float* out[3];
    uint32_t wg_begin[3]; // first workgroup of each matrix
    uint32_t n_parts, KC;
    unsigned long long* trace;
    uint32_t n_threads; // fused launch: the threads of the workgroup that project (whole waves; the rest retire at once), else 0 = blockDim.x
};
// The hand-off of a launch that also holds the attention's workgroups (ks_layer_a_kernel): the owning lanes store the outputs
// write-through (agent scope), the wave drains, lane 0 bumps the counter of the head slice the workgroup's columns belong to
// (the protocol of qmatvec.hip's QmvPublish / attention_decode.h's DecodeHandoff: monotonic counters, consumers keep private
// `seen` words, every wait bounded).
struct KsPublish {
    uint32_t* cnt;         // [n_heads | n_kv | n_kv] counters, 32 words apart
    uint32_t base[3];      // first counter of q / k / v
    uint32_t groups_shift; // log2(d_head / 16): column group -> head slice
};
template <int GP, int U>
__device__ __forceinline__ void ks_proj_body(const KsProjArgs& a, float* smem, const KsPublish* pub) {
    const uint32_t T = a.n_threads ? a.n_threads : blockDim.x, W = T >> 6, b = blockIdx.x;
    if (threadIdx.x >= T) return; // (whole waves: a retired wave no longer counts at s_barrier)
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane >> 4, i = lane & 15;
#define KS_T T
    KS_STAMPS(ts, a.trace, blockIdx.x == 0);
#undef KS_T
    KS_STAMP(ts, 0);
    uint32_t pi = 0;
    if (a.n_parts > 1 && b >= a.wg_begin[1]) pi = 1;
    if (a.n_parts > 2 && b >= a.wg_begin[2]) pi = 2;
    const uint4* const qs = pi == 2 ? a.qs[2] : pi == 1 ? a.qs[1] : a.qs[0];
    const uint32_t* const sc = pi == 2 ? a.sc[2] : pi == 1 ? a.sc[1] : a.sc[0];
    float* const out = pi == 2 ? a.out[2] : pi == 1 ? a.out[1] : a.out[0];
    const uint32_t g0 = (b - (pi == 2 ? a.wg_begin[2] : pi == 1 ? a.wg_begin[1] : 0u)) * GP;
    const uint32_t rho = 4 * w + r, combo = r % GP, slot = rho / GP, n_slots = 4 * W / GP, g = g0 + combo;
    KsUnit u[U];
    const KsLds l = ks_lds(smem, a.KC, T);
    ks_prologue(a.in, l, a.KC, T, b == 0, ts, [&]() {
        ks_row_load<U>(u, qs, sc, a.KC, g, slot, n_slots, i); // the weights do not depend on the vector: requested with it
        KS_STAMP(ts, 1);
    });
    float v = ks_row_fma<U>(u, l.xs, a.KC, slot, n_slots, i);
    KS_STAMP(ts, 5);
    if (GP == 1) v = rows_sum4(v);
    if (GP == 2) v = xor32_sum(v);
    if (r < (uint32_t)GP) l.red2[(w * GP + r) * 16 + i] = v;
    ks_barrier();
    KS_STAMP(ts, 6);
    if (w == 0) {
      const float inv = ks_inv(a.in, l.red, T); // (every lane of wave 0: the row fold inside needs whole rows)
      if (r < (uint32_t)GP) {
        float pw[16], t = 0.f; // (the waves' sums: independent LDS reads issued together, added in wave order)
#pragma unroll
        for (uint32_t ww = 0; ww < 16; ww++) pw[ww] = l.red2[((ww < W ? ww : 0) * GP + r) * 16 + i];
#pragma unroll
        for (uint32_t ww = 0; ww < 16; ww++)
            if (ww < W) t += pw[ww];
        if (pub)
            __hip_atomic_store((__attribute__((address_space(1))) float*)out + (g0 + r) * 16 + i, t * inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            out[(g0 + r) * 16 + i] = t * inv;
      }
      if (pub) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the write-through stores of this wave have left
#endif
        if (lane == 0)
            __hip_atomic_fetch_add((__attribute__((address_space(1))) uint32_t*)pub->cnt + 32 * (pub->base[pi] + (g0 >> pub->groups_shift)), 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    KS_STAMP(ts, 7);
    KS_FLUSH(ts, a.trace);
}
template <int GP, int U>
__global__ void __launch_bounds__(1024) ks_proj_kernel(KsProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[]; // (16-byte aligned: behind a kernel's static LDS the default is 4, and every b128 access of the carve-up would split)
    ks_proj_body<GP, U>(a, smem, nullptr);
}

// ── launch B: [deferred vector -> rmsnorm -> gamma] -> gate / up for 32 columns -> SiLU chain, product -> partial of down ────
// Workgroup b owns columns [32 b, 32 b + 32) of gate and up (two column groups each: the four "combos" of a wave's four rows)
// and rows [32 b, 32 b + 32) = chunk b of the down projection, whose every column group is one unit for one row.
struct KsMlpArgs {
    KsNormIn in;
    const uint4 *qs_g, *qs_u, *qs_d;
    const uint32_t *sc_g, *sc_u, *sc_d;
    uint32_t KC;    // chunks of the gate / up K
    uint32_t NB2_d; // column groups of the down projection (N / 16)
    uint32_t KC_d;  // chunks of its K (= gate.N / 32 = gridDim.x)
    float *gate_out, *up_out, *exp_out, *silu_out, *prod_out;
    const float* ones;
    float* parts_out; // [gridDim.x][16 * NB2_d]
    unsigned long long* trace;
};
template <int UG, int UD>
__global__ void __launch_bounds__(1024) ks_mlp_kernel(KsMlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[]; // (16-byte aligned: behind a kernel's static LDS the default is 4, and every b128 access of the carve-up would split)
    const uint32_t T = blockDim.x, W = T >> 6, R = 4 * W, b = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane >> 4, i = lane & 15, rho = 4 * w + r;
#define KS_T blockDim.x
    KS_STAMPS(ts, a.trace, blockIdx.x == 0);
    KS_STAMP(ts, 0);
    // row r of every wave: matrix r >> 1 (gate, up), column group 2 b + (r & 1); chunks w, w + W, ...
    const uint4* const qs = (r >> 1) ? a.qs_u : a.qs_g;
    const uint32_t* const sc = (r >> 1) ? a.sc_u : a.sc_g;
    KsUnit ug[UG], ud[UD];
    const uint32_t n = 32 * b + (lane & 31);
    float ones = 0.f;
    const KsLds l = ks_lds(smem, a.KC, T);
    ks_prologue(a.in, l, a.KC, T, b == 0, ts, [&]() {
        ks_row_load<UG>(ug, qs, sc, a.KC, 2 * b + (r & 1), w, W, i);
#pragma unroll
        for (int j = 0; j < UD; j++) { // the down projection's units of this row: column groups rho, rho + R, ... of chunk b
            const uint32_t gd = rho + j * R;
            ud[j] = ks_load(a.qs_d, a.sc_d, a.KC_d, gd < a.NB2_d ? gd : a.NB2_d - 1, b, i);
        }
        ones = a.ones[n]; // (the SiLU chain's constant operand: requested with the weights; used by wave 0 only)
        KS_STAMP(ts, 1);
    });
    const float acc = ks_row_fma<UG>(ug, l.xs, a.KC, w, W, i);
    KS_STAMP(ts, 5);
    l.red2[(w * 4 + r) * 16 + i] = acc;
    ks_barrier();
    KS_STAMP(ts, 6);
    if (w == 0) { // lanes 0-31: the gate's 32 columns, lanes 32-63: the up projection's
        const float inv = ks_inv(a.in, l.red, T);
        float pw[16], v = 0.f; // (the waves' sums: independent LDS reads issued together, added in wave order)
#pragma unroll
        for (uint32_t ww = 0; ww < 16; ww++) pw[ww] = l.red2[((ww < W ? ww : 0) * 4 + r) * 16 + i];
#pragma unroll
        for (uint32_t ww = 0; ww < 16; ww++)
            if (ww < W) v += pw[ww];
        v *= inv;
        const float other = __shfl_xor(v, 32, 64);
        if (lane < 32) {
            const float gate = v, up = other;
            a.gate_out[n] = gate;
            a.up_out[n] = up;
            const float e = expf(-gate); // NEG, EXP [store], ADD ones, RECIP, MUL by the gate [store] (nn.zig:38-44), then MUL by up [store]
            a.exp_out[n] = e;
            const float sl = (1.0f / (e + ones)) * gate;
            a.silu_out[n] = sl;
            const float pr = sl * up;
            a.prod_out[n] = pr;
            l.su[lane] = pr;
        }
        KS_STAMP(ts, 7); // gate / up folded, SiLU chain and product stored
    }
    ks_barrier();
    KS_STAMP(ts, 8);
    const float xa = l.su[i], xb = l.su[16 + i];
#pragma unroll
    for (int j = 0; j < UD; j++) {
        const uint32_t gd = rho + j * R;
        const bool ok = gd < a.NB2_d;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        ks_fma(ud[j], xa, xb, ok, a0, a1, a2, a3);
        if (ok) a.parts_out[(uint64_t)b * (16 * a.NB2_d) + gd * 16 + i] = (a0 + a1) + (a2 + a3);
    }
    KS_STAMP(ts, 9);
    KS_FLUSH(ts, a.trace);
}

// ── one workgroup: materialise a deferred vector (and its rmsnorm -> gamma, if any) — the tail of the last layer ─────────────
__global__ void __launch_bounds__(1024) ks_norm_kernel(KsNormIn in, uint32_t KC) {
    extern __shared__ __attribute__((aligned(16))) float smem[]; // (16-byte aligned: behind a kernel's static LDS the default is 4, and every b128 access of the carve-up would split)
    const KsLds l = ks_lds(smem, KC, blockDim.x);
    KS_STAMPS(ts, (unsigned long long*)nullptr, false);
    ks_prologue(in, l, KC, blockDim.x, true, ts, []() {});
}

// ── launch A2: the decode attention of one head (attention_decode.h's body, with every wave kept alive to the end) and that
// head's partial of the O projection ───────────────────────────────────────────────────────────────────────────────────────
// The stand-alone attention retires the waves a short context does not need; here they wait at the barriers instead, because
// the workgroup's 12 waves share the partial projection behind the attention: d_head rows x N columns of W_o = (d_head / 32)
// chunks of N / 16 column groups; row rho takes column groups rho, rho + 48, ... with all of the head's chunks.
struct KsAttnArgs {
    const AttnDecodeParams* params;
    float* split_buf;
    uint32_t* split_cnt;
    uint32_t split_min_keys, n_sp;
    const uint4* qs_o;
    const uint32_t* sc_o;
    uint32_t KC_o, NB2_o; // chunks of the O projection's K (n_heads * d_head / 32), its column groups
    float* parts_out;     // [n_heads][16 * NB2_o]
    unsigned long long* trace;
    uint8_t head_of[64];  // record -> head (= its row store's static offset / d_head)
};
constexpr int kKsAttnBlock = 768; // 12 waves: 168 registers per lane hold the attention's double-buffered rows AND the head's W_o units (16 waves: 128, spills)
template <int LPK, bool KVQ, int UO>
__device__ __forceinline__ void ks_attn_o_body(const KsAttnArgs& f, const uint32_t hx, const uint32_t sp, const DecodeHandoff* ho) {
    constexpr int DH = 4 * LPK, KPW = 64 / LPK, U = kAttnUnroll, HALF = DH / 2, CH = DH / 32, BLOCK = kKsAttnBlock;
    static_assert(DH % 32 == 0, "the head's rows of W_o are whole 32-k chunks");
    using Row = AttnRow<KVQ>;
    const uint32_t n_sp = f.n_sp;
    KS_STAMPS(ts, f.trace, hx == 0 && sp == 0);
    KS_STAMP(ts, 0);
    // ---- this head's rows of W_o: they depend on nothing but the kernel's arguments and are requested FIRST (loads return in
    // order: by the time the attention waits for its own rows these have long arrived)
    const uint32_t head = f.head_of[hx], c_o = head * CH;
    const uint32_t r16 = (threadIdx.x & 63) >> 4, i16 = threadIdx.x & 15, rho = 4 * (threadIdx.x >> 6) + r16;
    KsUnit uo[UO][CH];
#pragma unroll
    for (int j = 0; j < UO; j++) {
        const uint32_t gd = rho + j * (BLOCK / 16);
        const uint32_t gc = gd < f.NB2_o ? gd : f.NB2_o - 1;
#pragma unroll
        for (int c = 0; c < CH; c++) uo[j][c] = ks_load(f.qs_o, f.sc_o, f.KC_o, gc, c_o + c, i16);
    }
    static_assert(sizeof(AttnDecodeParams) == 248, "the bulk fetch below covers exactly 62 dwords");
    AttnDecodeParams P;
    { // the head's record in one scalar round trip (attention_decode.h)
        typedef uint32_t u32x16_t __attribute__((ext_vector_type(16)));
        typedef uint32_t u32x8_t __attribute__((ext_vector_type(8)));
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        u32x16_t ra, rb, rc;
        u32x8_t rd;
        u32x4_t re;
        u32x2_t rf;
#if defined(__HIP_DEVICE_COMPILE__)
        const AttnDecodeParams* const rec = f.params + hx;
        asm volatile("s_load_dwordx16 %0, %6, 0x0\n\ts_load_dwordx16 %1, %6, 0x40\n\ts_load_dwordx16 %2, %6, 0x80\n\t"
                     "s_load_dwordx8 %3, %6, 0xc0\n\ts_load_dwordx4 %4, %6, 0xe0\n\ts_load_dwordx2 %5, %6, 0xf0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(ra), "=&s"(rb), "=&s"(rc), "=&s"(rd), "=&s"(re), "=&s"(rf)
                     : "s"(rec)
                     : "memory");
#else
        ra = rb = rc = u32x16_t{}, rd = u32x8_t{}, re = u32x4_t{}, rf = u32x2_t{};
#endif
        char* const dst = (char*)&P;
        __builtin_memcpy(dst, &ra, 64), __builtin_memcpy(dst + 64, &rb, 64), __builtin_memcpy(dst + 128, &rc, 64);
        __builtin_memcpy(dst + 192, &rd, 32), __builtin_memcpy(dst + 224, &re, 16), __builtin_memcpy(dst + 240, &rf, 8);
    }
    const AttentionParams& p = P.att;
    KS_STAMP(ts, 1); // record fetched
    __shared__ __attribute__((aligned(8))) float part_ml[2 * (BLOCK / 64)];
    __shared__ float4 part_acc[(BLOCK / 64) * LPK];
    __shared__ __attribute__((aligned(16))) float o_lds[DH];
    __shared__ uint32_t last_flag;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t slot = lane / LPK, li = lane % LPK, d0 = 4 * li, pair = d0 & (HALF - 1);
    const bool is_hi = d0 >= (uint32_t)HALF;
    // ---- phase A: dynamic words (scalar) and everything the ropes need (vector), one round
    const uint32_t seq_kv = KVQ ? min(ldgu(p.dyn_seq_kv), P.kvq_cols) : ldgu(p.dyn_seq_kv), dk = ldgu(P.dyn_k_off), dv = ldgu(P.dyn_v_off);
    const uint32_t d2_off = head * DH; // (the head's row store: a static offset, checked by the planner)
    float4 q_own, q_par, k_own, k_par, v_new;
    if (!ho) { // (a fused launch reads the projections' outputs behind the hand-off below; `ho` is a literal nullptr in the stand-alone kernel)
        q_own = ldg4(P.q_src + d0), q_par = ldg4(P.q_src + (d0 ^ HALF));
        k_own = ldg4(P.k_src + d0), k_par = ldg4(P.k_src + (d0 ^ HALF));
        v_new = ldg4(P.v_src + d0);
    }
    const float4 q_c = ldg4(P.q_cs + pair), q_s = ldg4(P.q_cs + HALF + pair);
    const float4 k_c = ldg4(P.k_cs + pair), k_s = ldg4(P.k_cs + HALF + pair);
    // this workgroup's keys [k_begin, k_end)
    uint32_t n_active = 1, k_begin = 0, k_end = seq_kv;
    if (n_sp > 1) {
        n_active = seq_kv / f.split_min_keys;
        n_active = n_active < 1 ? 1 : (n_active > n_sp ? n_sp : n_active);
        if (sp >= n_active) { // (the whole workgroup: uniform) — in a fused launch an idle split still moves its private `seen` words with the counters
            if (ho && tid == 0) {
                uint32_t* const seen = ho->seen + (hx * n_sp + sp) * 3;
                seen[0] += ho->need, seen[1] += ho->need, seen[2] += ho->need;
            }
            return;
        }
        const uint32_t chunk = (seq_kv + n_active - 1) / n_active;
        k_begin = min(sp * chunk, seq_kv), k_end = min(k_begin + chunk, seq_kv);
    }
    const uint32_t n_keys = k_end - k_begin;
    uint32_t NW = (n_keys + KPW * U - 1) / (KPW * U); // waves the chunk needs: one step of a wave covers KPW * U keys
    NW = NW < 1 ? 1 : (NW > (uint32_t)(BLOCK / 64) ? (uint32_t)(BLOCK / 64) : NW);
    const uint32_t keys_per_iter = KPW * NW, step_keys = keys_per_iter * U, last = k_end ? k_end - 1 : 0;
    constexpr uint32_t BPC = DH / 32; // quantised form: 32-dim blocks per column
    const int8_t* const kq = (const int8_t*)P.k_cache + d0;
    const int8_t* const vq = (const int8_t*)P.v_cache + d0;
    const float* const ksc = P.k_cache + (uint64_t)P.kvq_cols * DH / 4 + d0 / 32;
    const float* const vsc = P.v_cache + (uint64_t)P.kvq_cols * DH / 4 + d0 / 32;
    auto load_k = [&](uint32_t s) -> Row {
        if constexpr (KVQ)
            return Row{ldgu((const uint32_t*)(kq + (uint64_t)s * DH)), ldg1(ksc + (uint64_t)s * BPC)};
        else
            return Row{ldg4(p.k + (uint64_t)s * p.k_cs + d0)};
    };
    auto load_v = [&](uint32_t s) -> Row {
        if constexpr (KVQ)
            return Row{ldgu((const uint32_t*)(vq + (uint64_t)s * DH)), ldg1(vsc + (uint64_t)s * BPC)};
        else
            return Row{ldg4(p.v + (uint64_t)s * p.v_cs + d0)};
    };
    // ---- phase B: first step's K / V / mask rows of the waves the chunk needs (clamped to live rows)
    Row kv[U], vv[U];
    float mk[U];
    if (w < NW) { // (wave-uniform)
#pragma unroll
        for (int j = 0; j < U; j++) {
            const uint32_t s = min(k_begin + j * keys_per_iter + w * KPW + slot, last);
            kv[j] = load_k(s);
            vv[j] = load_v(s);
            mk[j] = ldg1(p.mask + (uint64_t)s * p.mask_rs); // host passes a zero word with stride 0 when there is no mask
        }
    }
    KS_STAMP(ts, 2); // every load issued
    uint32_t target[3] = {0, 0, 0}; // fused launch: the counter values this execution waits for (written back at the end)
    if (ho) { // ---- hand-off: the q / k / v column groups of this head have been stored (attention_decode.h: DecodeHandoff)
        const uint32_t ci[3] = {ho->idx[3 * hx], ho->idx[3 * hx + 1], ho->idx[3 * hx + 2]};
        const uint32_t* const seen = ho->seen + (hx * n_sp + sp) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) target[c] = seen[c] + ho->need;
        if (w == 0) { // one wave polls (the counters are 128 bytes apart; every poll is a trip to memory), the others wait at the barrier
            uint32_t spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int c = 0; c < 3; c++) // (wrap-safe comparison)
                    ok = ok && (int32_t)(__hip_atomic_load((gu32*)ho->cnt + 32 * ci[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target[c]) >= 0;
                if (ok) break;
                if (++spins > 400000u) { // bounded: never hang the device; the caller sees the flag
                    if (tid == 0) __hip_atomic_store(ho->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                for (uint32_t z = 0; z < ho->poll_sleep; z++) __builtin_amdgcn_s_sleep(1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // the agent-scope loads of q / k / v below stay behind the poll (compiler ordering; they bypass L1 themselves)
        ks_barrier();
        auto ld_agent4 = [](const float* p) { // agent scope: never a stale line of this XCD's L2
            float4 v;
            split_get(p, v.x, v.y);
            split_get(p + 2, v.z, v.w);
            return v;
        };
        q_own = ld_agent4(P.q_src + d0), q_par = ld_agent4(P.q_src + (d0 ^ HALF));
        k_own = ld_agent4(P.k_src + d0), k_par = ld_agent4(P.k_src + (d0 ^ HALF));
        v_new = ld_agent4(P.v_src + d0);
    }
    SoftState st{-INFINITY, 0.f, make_float4(0.f, 0.f, 0.f, 0.f)};
    if (w < NW) { // (wave-uniform) the attention proper, as attention_decode_body
        // ---- ropes; side outputs and the cache stores (one lane group writes each value)
        const float4 qv = rope4(q_own, q_par, q_c, q_s, is_hi);
        const float4 k_new = rope4(k_own, k_par, k_c, k_s, is_hi);
        uint32_t col_k, col_v;
        Row k_col, v_col; // the new column as the attention sees it (quantised KV: what storeColumn writes, dequantised on use)
        if constexpr (KVQ) {
            col_k = dk, col_v = dv;
            k_col = quantise_block32(k_new), v_col = quantise_block32(v_new);
        } else {
            col_k = (dk - P.k_off) / p.k_cs, col_v = (dv - P.v_off) / p.v_cs;
            k_col = Row{k_new}, v_col = Row{v_new};
        }
        if (w == 0 && slot == 0 && sp == 0) {
            stg4(P.q_rot + d0, qv);
            if (P.owner) {
                stg4(P.k_rot + d0, k_new);
                if constexpr (KVQ) { // never outside the cache (kvq_store_kernel's guard)
                    if (col_k < P.kvq_cols) {
                        stgu((uint32_t*)((int8_t*)P.k_cache + (uint64_t)col_k * DH + d0), k_col.w);
                        if ((li & 7) == 0) stg1(P.k_cache + (uint64_t)P.kvq_cols * DH / 4 + (uint64_t)col_k * BPC + d0 / 32, k_col.sc);
                    }
                    if (col_v < P.kvq_cols) {
                        stgu((uint32_t*)((int8_t*)P.v_cache + (uint64_t)col_v * DH + d0), v_col.w);
                        if ((li & 7) == 0) stg1(P.v_cache + (uint64_t)P.kvq_cols * DH / 4 + (uint64_t)col_v * BPC + d0 / 32, v_col.sc);
                    }
                } else {
                    stg4(P.k_cache + dk + d0, k_new);
                    stg4(P.v_cache + dv + d0, v_new);
                }
            }
        }
        auto step = [&](uint32_t base) { // scores of the slot's U keys, then the online-softmax update
            float sc[U];
            float bm = -INFINITY;
#pragma unroll
            for (int j = 0; j < U; j++) {
                const uint32_t t = base + j * keys_per_iter + w * KPW + slot;
                const Row kk = t == col_k ? k_col : kv[j];
                const float dot = group_sum<LPK>(row_dot(qv, kk));
                sc[j] = t < k_end ? score_of(dot, mk[j], p.scale) : -INFINITY;
                bm = fmaxf(bm, sc[j]);
            }
            const float nm = fmaxf(st.m, bm);
            if (nm > -INFINITY) { // per key slot; lanes of a slot agree
                const float alpha = st.m > -INFINITY ? expf(st.m - nm) : 0.0f;
                st.l *= alpha;
                st.acc = make_float4(st.acc.x * alpha, st.acc.y * alpha, st.acc.z * alpha, st.acc.w * alpha);
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const uint32_t t = base + j * keys_per_iter + w * KPW + slot;
                    const float wgt = sc[j] > -INFINITY ? expf(sc[j] - nm) : 0.0f;
                    const Row x = t == col_v ? v_col : vv[j];
                    st.l += wgt;
                    if (sc[j] > -INFINITY) row_axpy(st.acc, wgt, x); // rows of dead slots are never touched
                }
                st.m = nm;
            }
        };
        if (n_keys <= step_keys) { // the usual decode case: everything is already in registers
            if (n_keys) step(k_begin);
        } else {
            for (uint32_t base = k_begin; base < k_end; base += step_keys) {
                Row kn[U], vn[U]; // prefetch the next step (clamped: the last step re-reads live rows, L2 hits)
                float mn[U];
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const uint32_t s = min(base + step_keys + j * keys_per_iter + w * KPW + slot, last);
                    kn[j] = load_k(s);
                    vn[j] = load_v(s);
                    mn[j] = ldg1(p.mask + (uint64_t)s * p.mask_rs);
                }
                step(base);
#pragma unroll
                for (int j = 0; j < U; j++) kv[j] = kn[j], vv[j] = vn[j], mk[j] = mn[j];
            }
        }
        // ---- merge the key slots of the wave (two-pass: common max, one rescale per stream, plain sums)
        slots_merge<LPK>(st);
        if (lane < LPK) {
            part_acc[w * LPK + lane] = st.acc;
            if (lane == 0) part_ml[2 * w] = st.m, part_ml[2 * w + 1] = st.l;
        }
    }
    KS_STAMP(ts, 3); // scores, softmax, weighted V of this wave
    ks_barrier(); // (every wave of the workgroup: none has retired)
    if (w == 0) {
        if (ho && lane == 0) { // every wave of this workgroup has passed the hand-off (the barrier above)
            uint32_t* const seen = ho->seen + (hx * n_sp + sp) * 3;
            seen[0] = target[0], seen[1] = target[1], seen[2] = target[2];
        }
        SoftState r = st;
        if (NW > 1) { // slot g folds waves g, g + KPW, ...: every LDS read up front, independent exponentials (attention_decode.h)
            constexpr int MAXW = BLOCK / 64, NPS = MAXW / KPW > 0 ? MAXW / KPW : 1;
            static_assert(MAXW <= 16, "lanes 0..15 (one DPP row) hold the waves' maxima");
            float M = lane < NW ? part_ml[2 * lane] : -INFINITY;
            float2 ml[NPS];
            float4 pa[NPS];
#pragma unroll
            for (int q = 0; q < NPS; q++) {
                const uint32_t ww = slot + q * KPW, wc = ww < NW ? ww : 0;
                ml[q] = *(const float2*)&part_ml[2 * wc];
                pa[q] = part_acc[wc * LPK + li];
            }
            M = fmaxf(M, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(M), 0xB1, 0xF, 0xF, true)));
            M = fmaxf(M, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(M), 0x4E, 0xF, 0xF, true)));
            M = fmaxf(M, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(M), 0x141, 0xF, 0xF, true)));
            M = fmaxf(M, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(M), 0x140, 0xF, 0xF, true)));
            M = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(M)));
            r = SoftState{M, 0.f, make_float4(0.f, 0.f, 0.f, 0.f)};
#pragma unroll
            for (int q = 0; q < NPS; q++) {
                const bool live = slot + q * KPW < NW && ml[q].x > -INFINITY;
                const float fct = live ? expf(ml[q].x - M) : 0.0f;
                r.l += ml[q].y * fct;
                r.acc.x += pa[q].x * fct, r.acc.y += pa[q].y * fct, r.acc.z += pa[q].z * fct, r.acc.w += pa[q].w * fct;
            }
            slots_sum<LPK>(r);
        }
        bool is_last = true;
        if (n_active > 1) { // publish this chunk; the last arriver merges all of them in chunk order (attention_decode.h)
            constexpr uint32_t REC = DH + 4; // m, l, pad, pad, acc[DH]
            float* const head_buf = f.split_buf + (uint64_t)hx * n_sp * REC;
            float* const mine = head_buf + (uint64_t)sp * REC;
            if (lane < LPK) {
                if (lane == 0) split_put(mine, r.m, r.l);
                split_put(mine + 4 + 4 * lane, r.acc.x, r.acc.y);
                split_put(mine + 4 + 4 * lane + 2, r.acc.z, r.acc.w);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the write-through stores have left before the count
            uint32_t old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add((gu32*)(f.split_cnt + hx), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __shfl(old, 0, 64);
            is_last = old == n_active - 1;
            if (is_last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // compiler only: the loads below stay below the count
                r = SoftState{-INFINITY, 0.f, make_float4(0.f, 0.f, 0.f, 0.f)};
                for (uint32_t c = slot; c < n_active; c += KPW) {
                    const float* rec = head_buf + (uint64_t)c * REC;
                    float om, ol;
                    float4 oa;
                    split_get(rec, om, ol);
                    split_get(rec + 4 + 4 * li, oa.x, oa.y);
                    split_get(rec + 4 + 4 * li + 2, oa.z, oa.w);
                    soft_merge(r, om, ol, oa);
                }
                slots_merge<LPK>(r);
                if (lane == 0) __hip_atomic_store((gu32*)(f.split_cnt + hx), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-arm
            }
        }
        if (is_last && lane < LPK) {
            const float inv_l = r.l > 0.f ? 1.0f / r.l : 0.0f;
            const float4 o = make_float4(r.acc.x * inv_l, r.acc.y * inv_l, r.acc.z * inv_l, r.acc.w * inv_l);
            stg4(p.dst + 4 * lane, o); // dst_rs == 1, 16-byte aligned (planner)
            float* const d2 = p.dst2 + (uint64_t)d2_off + 4 * lane; // (d2_rs == 1, offset a multiple of d_head: planner)
            stg4(d2, o);
            *(float4*)(o_lds + 4 * lane) = o;
        }
        if (lane == 0) last_flag = is_last ? 1u : 0u;
    }
    KS_STAMP(ts, 4); // waves merged, head output stored
    ks_barrier();
    if (!last_flag) return;
    // ---- this head's partial of the O projection: out[n] += sum over the head's d_head rows (waves without a column group skip it)
    if (4 * w >= f.NB2_o) return;
#pragma unroll
    for (int j = 0; j < UO; j++) {
        const uint32_t gd = rho + j * (BLOCK / 16);
        const bool ok = gd < f.NB2_o;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; c++) ks_fma(uo[j][c], o_lds[32 * c + i16], o_lds[32 * c + 16 + i16], ok, a0, a1, a2, a3);
        if (ok) f.parts_out[(uint64_t)head * (16 * f.NB2_o) + gd * 16 + i16] = (a0 + a1) + (a2 + a3);
    }
    KS_STAMP(ts, 5);
    KS_FLUSH(ts, f.trace);
}
template <int LPK, bool KVQ, int UO>
__global__ void __launch_bounds__(kKsAttnBlock) ks_attn_o_kernel(KsAttnArgs f) {
    ks_attn_o_body<LPK, KVQ, UO>(f, blockIdx.x, blockIdx.y, nullptr);
}

// ── launch A: A1's workgroups and A2's in ONE launch (workgroups [0, n_proj) project, the rest attend: head-major, then split): the
// projections reach the attention through per-head-slice counters instead of a kernel boundary, and everything the attention can do
// without them (its record, the dynamic words, rope tables, K / V rows, its rows of W_o) overlaps the projection. All workgroups
// are 768 threads; the planner builds this launch only when the whole grid is resident at one workgroup per CU (the attention's
// workgroups spin on workgroups of the same grid), shrinking the split count to fit, and falls back to the two launches after a
// time-out (the context's hand-off word).
struct KsLayerAArgs {
    KsProjArgs pj;
    KsAttnArgs at;
    KsPublish pub;
    DecodeHandoff ho;
    uint32_t n_proj, n_heads;
    uint32_t dbg; // diagnostics build: 1 = the attention's workgroups exit at once, 2 = they skip the wait (timing only: wrong results)
};
template <int LPK, bool KVQ, int UO, int GP, int U>
__global__ void __launch_bounds__(kKsAttnBlock) ks_layer_a_kernel(KsLayerAArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[]; // (16-byte aligned: behind a kernel's static LDS the default is 4, and every b128 access of the carve-up would split)
    if (blockIdx.x < g.n_proj) {
        ks_proj_body<GP, U>(g.pj, smem, &g.pub);
    } else {
        const uint32_t b = blockIdx.x - g.n_proj; // head-major: the always-active split 0 of every head first
#ifdef ZGML_TRACE
        if (g.dbg == 1) return;
        if (g.dbg == 2) {
            ks_attn_o_body<LPK, KVQ, UO>(g.at, b % g.n_heads, b / g.n_heads, nullptr);
            return;
        }
#endif
        ks_attn_o_body<LPK, KVQ, UO>(g.at, b % g.n_heads, b / g.n_heads, &g.ho);
    }
}

// waves of a launch with a prologue: `rows_wanted` DPP rows (one unit per row where 16 waves allow it), never fewer threads than
// K / 4 (the prologue's quads)
uint32_t ks_pick_waves(uint32_t K, uint32_t rows_wanted) {
    uint32_t W = cdiv(rows_wanted, 4);
    const uint32_t w_min = cdiv(K / 4, 64);
    W = W < w_min ? w_min : W;
    W = W < 1 ? 1 : (W > 16 ? 16 : W);
    return W;
}

bool ks_vec_aligned(const KsNormIn& in) {
    auto a16 = [](const void* p) { return ((uintptr_t)p % 16) == 0; };
    return in.K && in.K % 4 == 0 && in.K / 4 <= 1024 && a16(in.x.base) && a16(in.x.parts) && a16(in.x.sum_dst) && a16(in.x.add_dst) && a16(in.gamma) &&
           a16(in.store_mid) && a16(in.store_x) && (in.x.n_parts == 0 || in.x.parts) && in.x.base;
}

} // namespace

bool ks_weight_ok(const QWeightDev& w) { return w.format == QW_Q4 && w.scale_f16 && !w.stream_nt && w.K <= 2048 && w.N % 32 == 0 && w.KC == (w.K + 31) / 32; }

bool ks_norm_ok(const KsNormIn& in) { return ks_vec_aligned(in); }

static uint32_t ks_proj_gp(const KsProjLaunch& L) { // column groups per workgroup: 2 unless a matrix has an odd number of them
    static const int env = getenv("ZGML_KS_PROJ_GP") ? atoi(getenv("ZGML_KS_PROJ_GP")) : 0;
    uint32_t gp = env == 1 || env == 2 || env == 4 ? (uint32_t)env : 2u;
    for (uint32_t t = 0; t < L.n_parts; t++)
        while (gp > 1 && (L.w[t].N / 16) % gp) gp >>= 1;
    return gp;
}

bool ks_proj_ok(const KsProjLaunch& L) {
    if (L.n_parts < 1 || L.n_parts > 3 || !ks_vec_aligned(L.in) || !L.in.gamma) return false;
    for (uint32_t t = 0; t < L.n_parts; t++)
        if (!ks_weight_ok(L.w[t]) || L.w[t].K != L.in.K || !L.dst[t]) return false;
    return true;
}

void launch_ks_proj(hipStream_t s, const KsProjLaunch& L) {
    const uint32_t gp = ks_proj_gp(L), KC = L.w[0].KC;
    KsProjArgs a{};
    a.in = L.in, a.n_parts = L.n_parts, a.KC = KC, a.trace = L.trace;
    uint32_t wgs = 0;
    for (uint32_t t = 0; t < 3; t++) {
        const uint32_t tt = t < L.n_parts ? t : 0;
        a.qs[t] = (const uint4*)L.w[tt].qs, a.sc[t] = (const uint32_t*)L.w[tt].sc, a.out[t] = L.dst[tt];
        a.wg_begin[t] = wgs;
        if (t < L.n_parts) wgs += (uint32_t)(L.w[t].N / 16) / gp;
    }
    static const int env_w = getenv("ZGML_KS_PROJ_WAVES") ? atoi(getenv("ZGML_KS_PROJ_WAVES")) : 0;
    uint32_t W = ks_pick_waves(L.in.K, KC * gp); // one unit per row when 4 W / gp >= KC
    if (env_w >= 1 && env_w <= 16 && (uint32_t)env_w * 64 >= L.in.K / 4) W = (uint32_t)env_w;
    const uint32_t n_slots = 4 * W / gp, U = cdiv(KC, n_slots);
    const size_t lds = ks_lds_bytes(KC, W * 64);
    using Fn = void (*)(KsProjArgs);
    Fn fn = nullptr;
#define KS_PROJ(G) (U <= 1 ? (Fn)ks_proj_kernel<G, 1> : U <= 2 ? (Fn)ks_proj_kernel<G, 2> : U <= 4 ? (Fn)ks_proj_kernel<G, 4> : (Fn)ks_proj_kernel<G, 8>)
    fn = gp == 4 ? KS_PROJ(4) : gp == 2 ? KS_PROJ(2) : KS_PROJ(1);
#undef KS_PROJ
    if (U > 8) { // (K <= 2048: KC <= 64, n_slots >= 8 — never reached)
        fprintf(stderr, "[zgml_hip] ERROR: ks_proj: %u units per row: not launched\n", U);
        return;
    }
    hipLaunchKernelGGL(fn, dim3(wgs), dim3(W * 64), lds, s, a);
}

uint32_t ks_mlp_parts(const KsMlpLaunch& L) { return (uint32_t)(L.gate.N / 32); }

static void ks_mlp_shape(const KsMlpLaunch& L, uint32_t& W, uint32_t& UG, uint32_t& UD) {
    const uint32_t KC = L.gate.KC, NB2_d = (uint32_t)(L.down.N / 16);
    static const int env_w = getenv("ZGML_KS_MLP_WAVES") ? atoi(getenv("ZGML_KS_MLP_WAVES")) : 0;
    // rows: the down projection wants one row per column group; gate / up want 4 W rows x UG chunks >= 4 KC: UG <= 2 is the aim
    W = ks_pick_waves(L.in.K, std::max(NB2_d, 2 * KC));
    if (env_w >= 1 && env_w <= 16 && (uint32_t)env_w * 64 >= L.in.K / 4) W = (uint32_t)env_w;
    UG = cdiv(KC, W), UD = cdiv(NB2_d, 4 * W);
}

bool ks_mlp_ok(const KsMlpLaunch& L) {
    if (!ks_vec_aligned(L.in) || !L.in.gamma || !ks_weight_ok(L.gate) || !ks_weight_ok(L.up) || !ks_weight_ok(L.down)) return false;
    if (L.gate.K != L.in.K || L.up.K != L.in.K || L.gate.N != L.up.N || L.down.K != L.gate.N || L.down.K % 32) return false;
    if (!L.gate_out || !L.up_out || !L.exp_out || !L.silu_out || !L.prod_out || !L.ones || !L.parts_out) return false;
    uint32_t W, UG, UD;
    ks_mlp_shape(L, W, UG, UD);
    return UG <= 4 && UD <= 2;
}

void launch_ks_mlp(hipStream_t s, const KsMlpLaunch& L) {
    uint32_t W, UG, UD;
    ks_mlp_shape(L, W, UG, UD);
    KsMlpArgs a{};
    a.in = L.in;
    a.qs_g = (const uint4*)L.gate.qs, a.qs_u = (const uint4*)L.up.qs, a.qs_d = (const uint4*)L.down.qs;
    a.sc_g = (const uint32_t*)L.gate.sc, a.sc_u = (const uint32_t*)L.up.sc, a.sc_d = (const uint32_t*)L.down.sc;
    a.KC = L.gate.KC, a.NB2_d = (uint32_t)(L.down.N / 16), a.KC_d = L.down.KC;
    a.gate_out = L.gate_out, a.up_out = L.up_out, a.exp_out = L.exp_out, a.silu_out = L.silu_out, a.prod_out = L.prod_out;
    a.ones = L.ones, a.parts_out = L.parts_out, a.trace = L.trace;
    using Fn = void (*)(KsMlpArgs);
    Fn fn = nullptr;
#define KS_MLP(D) (UG <= 1 ? (Fn)ks_mlp_kernel<1, D> : UG <= 2 ? (Fn)ks_mlp_kernel<2, D> : UG <= 3 ? (Fn)ks_mlp_kernel<3, D> : (Fn)ks_mlp_kernel<4, D>)
    fn = UD <= 1 ? KS_MLP(1) : KS_MLP(2);
#undef KS_MLP
    if (UG > 4 || UD > 2) { // (ks_mlp_ok refuses these: never reached)
        fprintf(stderr, "[zgml_hip] ERROR: ks_mlp: shape outside the kernel's unit counts: not launched\n");
        return;
    }
    hipLaunchKernelGGL(fn, dim3(ks_mlp_parts(L)), dim3(W * 64), ks_lds_bytes(a.KC, W * 64), s, a);
}

void launch_ks_norm(hipStream_t s, const KsNormIn& in) {
    const uint32_t KC = (in.K + 31) / 32, W = ks_pick_waves(in.K, std::min<uint32_t>(in.x.n_parts + 1, 16) * cdiv(in.K / 4, 64) * 4);
    hipLaunchKernelGGL(ks_norm_kernel, dim3(1), dim3(W * 64), ks_lds_bytes(KC, W * 64), s, in, KC);
}

// launch A (ks_layer_a_kernel): the two launches above as one. `counters` [32 * (n_heads + 2 n_kv)], `seen` [n_heads * splits * 3],
// zero-initialised by the caller; `idx` [n_heads][3] names each record's q / k / v counter. false: not a shape the fused kernel is
// built for (nothing launched: launch the two separately).
static bool ks_layer_a_shape(const KsProjLaunch& PL, const KsAttnOLaunch& AL, uint32_t& U, uint32_t& UO) {
    if (ks_proj_gp(PL) != 2 || PL.n_parts != 3) return false;
    const uint32_t W = kKsAttnBlock / 64;
    if ((uint32_t)W * 64 < PL.in.K / 4) return false;
    const uint32_t Wp = std::min(W, ks_pick_waves(PL.in.K, PL.w[0].KC * 2)); // the projecting waves of a workgroup (the rest retire)
    U = cdiv(PL.w[0].KC, 4 * Wp / 2), UO = cdiv((uint32_t)(AL.wo.N / 16), kKsAttnBlock / 16);
    return U <= 2 && UO <= 2 && (AL.d_head / 16) % 2 == 0;
}
bool ks_layer_a_ok(const KsProjLaunch& PL, const KsAttnOLaunch& AL) {
    uint32_t U, UO;
    return ks_proj_ok(PL) && ks_attn_o_ok(AL) && ks_layer_a_shape(PL, AL, U, UO);
}
uint32_t ks_layer_a_proj_wgs(const KsProjLaunch& PL) {
    uint32_t wgs = 0;
    for (uint32_t t = 0; t < PL.n_parts; t++) wgs += (uint32_t)(PL.w[t].N / 16) / 2;
    return wgs;
}
bool launch_ks_layer_a(hipStream_t s, const KsProjLaunch& PL, const KsAttnOLaunch& AL, uint32_t n_kv, uint32_t* counters, const uint32_t* idx, uint32_t* seen,
                       uint32_t* timeout) {
    uint32_t U, UO;
    if (!ks_layer_a_shape(PL, AL, U, UO)) return false;
    KsLayerAArgs g{};
    KsProjArgs& a = g.pj;
    a.in = PL.in, a.n_parts = PL.n_parts, a.KC = PL.w[0].KC, a.trace = PL.trace;
    a.n_threads = 64 * std::min<uint32_t>(kKsAttnBlock / 64, ks_pick_waves(PL.in.K, PL.w[0].KC * 2));
    uint32_t wgs = 0;
    for (uint32_t t = 0; t < 3; t++) {
        a.qs[t] = (const uint4*)PL.w[t].qs, a.sc[t] = (const uint32_t*)PL.w[t].sc, a.out[t] = PL.dst[t];
        a.wg_begin[t] = wgs;
        wgs += (uint32_t)(PL.w[t].N / 16) / 2;
    }
    KsAttnArgs& f = g.at;
    f.params = AL.dev, f.split_buf = AL.sp.buf, f.split_cnt = AL.sp.cnt, f.split_min_keys = AL.sp.min_keys;
    f.n_sp = AL.sp.splits ? AL.sp.splits : 1;
    f.qs_o = (const uint4*)AL.wo.qs, f.sc_o = (const uint32_t*)AL.wo.sc, f.KC_o = AL.wo.KC, f.NB2_o = (uint32_t)(AL.wo.N / 16);
    f.parts_out = AL.parts_out, f.trace = AL.trace;
    for (uint32_t h = 0; h < 64; h++) f.head_of[h] = AL.head_of[h];
    uint32_t shift = 0;
    while ((16u << shift) < AL.d_head) shift++;
    g.pub = KsPublish{counters, {0, AL.n_heads, AL.n_heads + n_kv}, shift};
    static const uint32_t poll_sleep = getenv("ZGML_HIP_HANDOFF_SLEEP") ? (uint32_t)atoi(getenv("ZGML_HIP_HANDOFF_SLEEP")) : 2u;
    g.ho = DecodeHandoff{counters, seen, idx, AL.n_heads, (AL.d_head / 16) / 2, timeout, poll_sleep, nullptr};
    g.n_proj = wgs, g.n_heads = AL.n_heads;
    static const uint32_t dbg = getenv("ZGML_KS_DEBUG_A") ? (uint32_t)atoi(getenv("ZGML_KS_DEBUG_A")) : 0u;
    g.dbg = dbg;
    using Fn = void (*)(KsLayerAArgs);
    Fn fn = nullptr;
#define KS_LA2(LPK_, KVQ_, UO_) (U <= 1 ? (Fn)ks_layer_a_kernel<LPK_, KVQ_, UO_, 2, 1> : (Fn)ks_layer_a_kernel<LPK_, KVQ_, UO_, 2, 2>)
#define KS_LA(LPK_, KVQ_) (UO <= 1 ? KS_LA2(LPK_, KVQ_, 1) : KS_LA2(LPK_, KVQ_, 2))
    if (AL.d_head == 64)
        fn = AL.kvq ? KS_LA(16, true) : KS_LA(16, false);
    else
        fn = AL.kvq ? KS_LA(32, true) : KS_LA(32, false);
#undef KS_LA
#undef KS_LA2
    hipLaunchKernelGGL(fn, dim3(wgs + AL.n_heads * f.n_sp), dim3(kKsAttnBlock), ks_lds_bytes(a.KC, kKsAttnBlock), s, g);
    return true;
}

bool ks_attn_o_ok(const KsAttnOLaunch& L) {
    if (!L.dev || !L.n_heads || L.n_heads > 64 || (L.d_head != 64 && L.d_head != 128) || !ks_weight_ok(L.wo) || L.wo.K != (uint64_t)L.n_heads * L.d_head || !L.parts_out) return false;
    return cdiv(L.wo.N / 16, kKsAttnBlock / 16) <= 4;
}

void launch_ks_attn_o(hipStream_t s, const KsAttnOLaunch& L) {
    KsAttnArgs f{};
    f.params = L.dev, f.split_buf = L.sp.buf, f.split_cnt = L.sp.cnt, f.split_min_keys = L.sp.min_keys;
    f.n_sp = L.sp.splits ? L.sp.splits : 1;
    f.qs_o = (const uint4*)L.wo.qs, f.sc_o = (const uint32_t*)L.wo.sc, f.KC_o = L.wo.KC, f.NB2_o = (uint32_t)(L.wo.N / 16);
    f.parts_out = L.parts_out, f.trace = L.trace;
    for (uint32_t h = 0; h < 64; h++) f.head_of[h] = L.head_of[h];
    const uint32_t UO = cdiv(f.NB2_o, kKsAttnBlock / 16);
    const dim3 grid(L.n_heads, f.n_sp);
    using Fn = void (*)(KsAttnArgs);
    Fn fn = nullptr;
#define KS_ATT(LPK_, KVQ_) (UO <= 1 ? (Fn)ks_attn_o_kernel<LPK_, KVQ_, 1> : UO <= 2 ? (Fn)ks_attn_o_kernel<LPK_, KVQ_, 2> : (Fn)ks_attn_o_kernel<LPK_, KVQ_, 4>)
    if (L.d_head == 64)
        fn = L.kvq ? KS_ATT(16, true) : KS_ATT(16, false);
    else
        fn = L.kvq ? KS_ATT(32, true) : KS_ATT(32, false);
#undef KS_ATT
    hipLaunchKernelGGL(fn, grid, dim3(kKsAttnBlock), 0, s, f);
}

} // namespace zgml
