// qmatvec.hip — block-quantized mat-vec / matmul for gfx950 (the hot kernel of the path).
//
// Semantics (src/quant.zig:475-578 == src/backend/reference.zig:530-565):
//     dst[m,n] = sum_k  x[m,k] * scales[(k*N+n)/bs] * f32(data[k*N+n])
// i.e. weights are [K,N] row-major int8 and ONE scale is shared by `bs` consecutive *flat*
// elements: for bs == 32 and N % 32 == 0 that is 32 consecutive n at a fixed k.
// No activation quantisation: x stays f32, products are f32, only the summation order differs from
// the reference's k-sequential loop.
//
// Device layout ("packed", chosen at compile_program time — the boundary hands int8 + f32 scales,
// src/backend.zig:259-266; re-packing is backend-internal):
//   NB = N/32 block-columns, KC = ceil(K/32) k-chunks (zero padded).
//   Q4 (all values in [-8,7]; GGUF Q4_0-sourced weights, 4.5 bit/weight like the file format):
//     qs: uint4[NB][KC][32]   item (j,c,nl) = the 32 nibbles of column n = 32j+nl for
//                             k = 32c..32c+31, two's-complement; dword d, byte b holds
//                             k_local = 8d+b in the low nibble and 8d+4+b in the high nibble.
//     sc: pair[NB][KC][16]    entry i = { scale(k=32c+i, j), scale(k=32c+16+i, j) }
//   Q8: qs: uint4[NB][2KC][32] item (j,h,nl) = 16 int8 of column n for k = 16h..16h+15
//       sc: scalar[NB][2KC][16] entry i = scale(k=16h+i, j)
//   Scales are stored as f16 when every scale is exactly representable (GGUF-sourced), else f32.
//
// Thread mapping: a wave64 is 4 DPP rows of 16 lanes. Row r of a wave owns 16 columns
// (n-half r&1 of block-column j) for one k-unit (a 32-k chunk for Q4, a 16-k half-chunk for Q8);
// rows 2,3 take the next unit. One wave-load is 1 KiB contiguous (16 B/lane). Lane i of a row
// computes t = scale(k_i) * x[k_i] for "its" k of the unit, and the 32 (16) products of a lane's
// column are   acc += f32(q[k,n]) * t[k]   with t[k] fetched from lane k of the same row by the
// DPP row_newbcast operand of v_fmac_f32 — no LDS round trip and no scalar broadcast for the
// per-k multiplier, although the scale grouping runs along n. Weights go HBM -> VGPR directly
// (GEMV: nothing to reuse, guide §5 "GEMV / M <= 16" row); only x is staged in LDS.
// A 256-thread workgroup (4 waves) covers one block-column and a K slice; K is split across
// workgroups (split-K) until the grid has >= ~2 workgroups per CU; partial slabs are combined in
// fixed split order by a second tiny kernel (deterministic, unlike float atomics).
#include "kernels.h"

#include <hip/hip_fp16.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace zgml {

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

// ── inline-asm helpers ──────────────────────────────────────────────────────────────────────
// The per-weight work is two VALU instructions: an SDWA convert (nibble or byte -> f32) and a
// v_fmac_f32 whose multiplier t[k] comes from lane k of the same 16-lane row through the DPP
// row_newbcast operand. Each 8-weight (Q4) / 4-weight (Q8) dword is ONE asm statement so hipcc
// neither pads the statement boundaries with s_nop nor serialises convert->fmac pairs through a
// single temporary; inside a statement producers and consumers are >= 4 instructions apart and
// two accumulators alternate, so nothing waits on the previous instruction.
// Hazard the assembler does not pad (guide §5.7): a VALU write of the DPP *source* VGPR (t) needs
// 2 wait states before the DPP read — dpp_fence() below provides them once per step.
#define ZGML_SDWA "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_"
#define ZGML_DPP(i) " row_newbcast:%" #i " row_mask:0xf bank_mask:0xf\n\t"

// one dword of a Q4 item = 8 two's-complement nibbles; byte b low nibble -> t[BASE+b], high
// nibble -> t[BASE+4+b]. v_cvt_off_f32_i4 yields q/16 (the caller folds the 16 into t).
template <int BASE>
__device__ __forceinline__ void q4_dword(float& accA, float& accB, uint32_t w, float t) {
    float c0, c1, c2, c3;
    uint32_t h;
    asm("v_cvt_off_f32_i4_sdwa %2, %7 " ZGML_SDWA "0\n\t"
        "v_cvt_off_f32_i4_sdwa %3, %7 " ZGML_SDWA "1\n\t"
        "v_cvt_off_f32_i4_sdwa %4, %7 " ZGML_SDWA "2\n\t"
        "v_cvt_off_f32_i4_sdwa %5, %7 " ZGML_SDWA "3\n\t"
        "v_lshrrev_b32 %6, 4, %7\n\t"
        "v_fmac_f32_dpp %0, %8, %2" ZGML_DPP(9)
        "v_cvt_off_f32_i4_sdwa %2, %6 " ZGML_SDWA "0\n\t"
        "v_fmac_f32_dpp %1, %8, %3" ZGML_DPP(10)
        "v_cvt_off_f32_i4_sdwa %3, %6 " ZGML_SDWA "1\n\t"
        "v_fmac_f32_dpp %0, %8, %4" ZGML_DPP(11)
        "v_cvt_off_f32_i4_sdwa %4, %6 " ZGML_SDWA "2\n\t"
        "v_fmac_f32_dpp %1, %8, %5" ZGML_DPP(12)
        "v_cvt_off_f32_i4_sdwa %5, %6 " ZGML_SDWA "3\n\t"
        "v_fmac_f32_dpp %0, %8, %2" ZGML_DPP(13)
        "v_fmac_f32_dpp %1, %8, %3" ZGML_DPP(14)
        "v_fmac_f32_dpp %0, %8, %4" ZGML_DPP(15)
        "v_fmac_f32_dpp %1, %8, %5" ZGML_DPP(16)
        : "+v"(accA), "+v"(accB), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(h)
        : "v"(w), "v"(t), "i"(BASE + 0), "i"(BASE + 1), "i"(BASE + 2), "i"(BASE + 3), "i"(BASE + 4),
          "i"(BASE + 5), "i"(BASE + 6), "i"(BASE + 7));
}

// one dword of a Q8 item = 4 signed bytes; byte b -> t[BASE+b]
template <int BASE>
__device__ __forceinline__ void q8_dword(float& accA, float& accB, uint32_t w, float t) {
    float c0, c1, c2, c3;
    asm("v_cvt_f32_i32_sdwa %2, sext(%6) " ZGML_SDWA "0\n\t"
        "v_cvt_f32_i32_sdwa %3, sext(%6) " ZGML_SDWA "1\n\t"
        "v_cvt_f32_i32_sdwa %4, sext(%6) " ZGML_SDWA "2\n\t"
        "v_cvt_f32_i32_sdwa %5, sext(%6) " ZGML_SDWA "3\n\t"
        "v_fmac_f32_dpp %0, %7, %2" ZGML_DPP(8)
        "v_fmac_f32_dpp %1, %7, %3" ZGML_DPP(9)
        "v_fmac_f32_dpp %0, %7, %4" ZGML_DPP(10)
        "v_fmac_f32_dpp %1, %7, %5" ZGML_DPP(11)
        : "+v"(accA), "+v"(accB), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3)
        : "v"(w), "v"(t), "i"(BASE + 0), "i"(BASE + 1), "i"(BASE + 2), "i"(BASE + 3));
}

// VALU write -> DPP read of the same VGPR needs 2 wait states; hipcc does not see inside asm.
__device__ __forceinline__ void dpp_fence(float& a, float& b) { asm volatile("s_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void dpp_fence(float& a) { asm volatile("s_nop 1" : "+v"(a)); }

__device__ __forceinline__ float to_f32(__half h) { return __half2float(h); }
__device__ __forceinline__ float to_f32(float f) { return f; }

template <typename ST>
struct Pair {
    ST a, b;
};

struct QMVArgs {
    const uint4* qs;
    const void* sc;
    const float* x; // input (offset applied); row m at x + m*in_rs
    float* out;     // dst (ksplit==1) or partial slabs
    uint32_t M, N, K, NB;
    uint32_t U;            // k-units per block column (Q4: KC, Q8: 2*KC)
    uint32_t units_per_wg; // multiple of 8
    uint32_t ksplit;
    uint32_t in_rs, out_rs; // out_rs: dst_rs when ksplit==1, else N
};

// Shared prologue: stage this workgroup's slice of x (zero beyond K) into LDS. 16-byte loads when
// the slice is aligned, all issued before the first LDS store (one latency, not one per pass).
__device__ __forceinline__ void stage_x(float* xs, const float* x, uint32_t k_begin, uint32_t k_count, uint32_t K) {
    const bool vec = ((uintptr_t)(x + k_begin) & 15) == 0;
    if (vec) {
        for (uint32_t i = threadIdx.x * 4; i < k_count; i += kBlock * 4) {
            const uint32_t k = k_begin + i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k + 3 < K) {
                v = *(const float4*)(x + k);
            } else {
                if (k < K) v.x = x[k];
                if (k + 1 < K) v.y = x[k + 1];
                if (k + 2 < K) v.z = x[k + 2];
            }
            *(float4*)(xs + i) = v;
        }
    } else {
        for (uint32_t i = threadIdx.x; i < k_count; i += kBlock) {
            const uint32_t k = k_begin + i;
            xs[i] = k < K ? x[k] : 0.0f;
        }
    }
}

// Shared epilogue: fold rows (r, r+2), then the 4 waves, in fixed order; 32 outputs per WG.
__device__ __forceinline__ void reduce_store(float acc, float* red, const QMVArgs& a, uint32_t j, uint32_t split,
                                             uint32_t m) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    acc += __shfl_xor(acc, 32, 64);
    if (lane < 32) red[w * 32 + lane] = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
        float v = red[threadIdx.x];
#pragma unroll
        for (int ww = 1; ww < kWaves; ww++) v += red[ww * 32 + threadIdx.x];
        const uint32_t n = j * 32 + threadIdx.x;
        if (a.ksplit == 1)
            a.out[(uint64_t)m * a.out_rs + n] = v;
        else
            a.out[((uint64_t)split * a.M + m) * a.N + n] = v;
    }
}

constexpr int kDepth = 4; // k-units a wave keeps in flight (4 x 1 KiB wave-loads)

template <typename ST>
__global__ void __launch_bounds__(kBlock) qmatvec_q4_kernel(QMVArgs a) {
    extern __shared__ float smem[];
    float* xs = smem;                        // units_per_wg * 32 floats
    float* red = smem + a.units_per_wg * 32; // kWaves * 32 floats
    const uint32_t j = blockIdx.x % a.NB, split = blockIdx.x / a.NB, m = blockIdx.y;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t row = lane >> 4, i = lane & 15, nl = lane & 31;
    const uint32_t u_begin = split * a.units_per_wg;
    const uint32_t u_end = min(a.U, u_begin + a.units_per_wg);
    const uint4* qs = a.qs + (uint64_t)j * a.U * 32 + nl;
    const Pair<ST>* sc = (const Pair<ST>*)a.sc + (uint64_t)j * a.U * 16 + i;
    const uint32_t n_steps = (u_end - u_begin + 2 * kWaves - 1) / (2 * kWaves);

    uint32_t u = u_begin + 2 * w + (row >> 1); // this lane's unit at step 0; += 2*kWaves per step
    uint4 wq[kDepth];
    Pair<ST> s2[kDepth];
    // first group of weight loads goes out before the x staging so HBM latency overlaps it
#pragma unroll
    for (int d = 0; d < kDepth; d++) {
        const uint32_t ud = u + d * 2 * kWaves;
        wq[d] = make_uint4(0, 0, 0, 0);
        s2[d] = Pair<ST>{ST(0), ST(0)};
        if (ud < u_end) {
            wq[d] = qs[(uint64_t)ud * 32];
            s2[d] = sc[(uint64_t)ud * 16];
        }
    }
    stage_x(xs, a.x + (uint64_t)m * a.in_rs, u_begin * 32, (u_end - u_begin) * 32, a.K);
    __syncthreads();

    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (uint32_t step = 0; step < n_steps; step += kDepth) {
        uint4 wn[kDepth];
        Pair<ST> sn[kDepth];
#pragma unroll
        for (int d = 0; d < kDepth; d++) { // next group (zero-filled past the slice end)
            const uint32_t ud = u + (kDepth + d) * 2 * kWaves;
            wn[d] = make_uint4(0, 0, 0, 0);
            sn[d] = Pair<ST>{ST(0), ST(0)};
            if (ud < u_end) {
                wn[d] = qs[(uint64_t)ud * 32];
                sn[d] = sc[(uint64_t)ud * 16];
            }
        }
#pragma unroll
        for (int d = 0; d < kDepth; d++) {
            const uint32_t ud = u + d * 2 * kWaves;
            float xa = 0.f, xb = 0.f;
            if (ud < u_end) {
                xa = xs[(ud - u_begin) * 32 + i];
                xb = xs[(ud - u_begin) * 32 + 16 + i];
            }
            // all 64 lanes run the DPP section (row_newbcast reads need their source lanes live);
            // out-of-range units contribute q = 0, t = 0
            float tA = (to_f32(s2[d].a) * 16.0f) * xa;
            float tB = (to_f32(s2[d].b) * 16.0f) * xb;
            dpp_fence(tA, tB);
            q4_dword<0>(acc0, acc1, wq[d].x, tA);
            q4_dword<8>(acc2, acc3, wq[d].y, tA);
            q4_dword<0>(acc0, acc1, wq[d].z, tB);
            q4_dword<8>(acc2, acc3, wq[d].w, tB);
        }
#pragma unroll
        for (int d = 0; d < kDepth; d++) {
            wq[d] = wn[d];
            s2[d] = sn[d];
        }
        u += kDepth * 2 * kWaves;
    }
    reduce_store((acc0 + acc1) + (acc2 + acc3), red, a, j, split, m);
}

template <typename ST>
__global__ void __launch_bounds__(kBlock) qmatvec_q8_kernel(QMVArgs a) {
    extern __shared__ float smem[];
    float* xs = smem;                        // units_per_wg * 16 floats
    float* red = smem + a.units_per_wg * 16; // kWaves * 32 floats
    const uint32_t j = blockIdx.x % a.NB, split = blockIdx.x / a.NB, m = blockIdx.y;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t row = lane >> 4, i = lane & 15, nl = lane & 31;
    const uint32_t u_begin = split * a.units_per_wg;
    const uint32_t u_end = min(a.U, u_begin + a.units_per_wg);
    const uint4* qs = a.qs + (uint64_t)j * a.U * 32 + nl;
    const ST* sc = (const ST*)a.sc + (uint64_t)j * a.U * 16 + i;
    const uint32_t n_steps = (u_end - u_begin + 2 * kWaves - 1) / (2 * kWaves);

    uint32_t u = u_begin + 2 * w + (row >> 1);
    uint4 wq[kDepth];
    ST s1[kDepth];
#pragma unroll
    for (int d = 0; d < kDepth; d++) {
        const uint32_t ud = u + d * 2 * kWaves;
        wq[d] = make_uint4(0, 0, 0, 0);
        s1[d] = ST(0);
        if (ud < u_end) {
            wq[d] = qs[(uint64_t)ud * 32];
            s1[d] = sc[(uint64_t)ud * 16];
        }
    }
    stage_x(xs, a.x + (uint64_t)m * a.in_rs, u_begin * 16, (u_end - u_begin) * 16, a.K);
    __syncthreads();

    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (uint32_t step = 0; step < n_steps; step += kDepth) {
        uint4 wn[kDepth];
        ST sn[kDepth];
#pragma unroll
        for (int d = 0; d < kDepth; d++) {
            const uint32_t ud = u + (kDepth + d) * 2 * kWaves;
            wn[d] = make_uint4(0, 0, 0, 0);
            sn[d] = ST(0);
            if (ud < u_end) {
                wn[d] = qs[(uint64_t)ud * 32];
                sn[d] = sc[(uint64_t)ud * 16];
            }
        }
#pragma unroll
        for (int d = 0; d < kDepth; d++) {
            const uint32_t ud = u + d * 2 * kWaves;
            const float xa = (ud < u_end) ? xs[(ud - u_begin) * 16 + i] : 0.f;
            float t = to_f32(s1[d]) * xa;
            dpp_fence(t);
            q8_dword<0>(acc0, acc1, wq[d].x, t);
            q8_dword<4>(acc2, acc3, wq[d].y, t);
            q8_dword<8>(acc0, acc1, wq[d].z, t);
            q8_dword<12>(acc2, acc3, wq[d].w, t);
        }
#pragma unroll
        for (int d = 0; d < kDepth; d++) {
            wq[d] = wn[d];
            s1[d] = sn[d];
        }
        u += kDepth * 2 * kWaves;
    }
    reduce_store((acc0 + acc1) + (acc2 + acc3), red, a, j, split, m);
}

// dst[m*dst_rs + n] = sum_s part[(s*M+m)*N + n], s ascending
__global__ void __launch_bounds__(kBlock) splitk_combine_kernel(float* __restrict__ dst, const float* __restrict__ part,
                                                                uint32_t M, uint32_t N, uint32_t ksplit,
                                                                uint32_t dst_rs) {
    uint32_t n = blockIdx.x * kBlock + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float v = part[(uint64_t)m * N + n];
    for (uint32_t s = 1; s < ksplit; s++) v += part[((uint64_t)s * M + m) * N + n];
    dst[(uint64_t)m * dst_rs + n] = v;
}

// Raw layout (any block size, any N): one thread per (m, n), k sequential — exactly the
// reference's loop order, coalesced along n. Used for odd shapes (e.g. the bs=4 conformance case).
__global__ void __launch_bounds__(kBlock) qmatmul_raw_kernel(const int8_t* __restrict__ data,
                                                             const float* __restrict__ scales, uint32_t bs,
                                                             QMatmulParams p) {
    uint32_t n = blockIdx.x * kBlock + threadIdx.x, m = blockIdx.y;
    if (n >= p.N) return;
    const float* x = p.input + (uint64_t)m * p.in_rs;
    float acc = 0.f;
    for (uint32_t k = 0; k < p.K; k++) {
        uint64_t flat = (uint64_t)k * p.N + n;
        float scale = scales[flat / bs] * x[k];
        acc = acc + (float)data[flat] * scale;
    }
    p.dst[(uint64_t)m * p.dst_rs + n] = acc;
}

// ── synthetic weights for the roofline ring (SURVEY §8d generator), written in packed form ──
__device__ __forceinline__ int synth_q4(uint64_t flat, uint32_t id) {
    return (int)((flat * 7 + (flat >> 5) * 3 + (uint64_t)id * 5) & 15) - 8; // stored nibble - 8
}
__device__ __forceinline__ int synth_q8(uint64_t flat, uint32_t id) {
    return (int)((flat * 13 + (uint64_t)id * 29) % 255) - 127;
}
__device__ __forceinline__ float synth_scale(uint64_t block, uint32_t id) {
    return 0.015625f * (1.0f + (float)((block + id) % 7) * 0.125f); // exact in f16
}

template <typename ST>
__global__ void __launch_bounds__(kBlock) synth_packed_kernel(QWeightDev w, uint32_t id) {
    const uint32_t NB = w.N / 32;
    const bool q4 = w.format == QW_Q4;
    const uint32_t U = q4 ? w.KC : 2 * w.KC;
    const uint64_t n_items = (uint64_t)NB * U * 32;
    for (uint64_t it = (uint64_t)blockIdx.x * kBlock + threadIdx.x; it < n_items; it += (uint64_t)gridDim.x * kBlock) {
        const uint32_t nl = it & 31;
        const uint64_t ju = it >> 5;
        const uint32_t u = ju % U, j = ju / U;
        const uint32_t n = j * 32 + nl;
        uint32_t d[4];
        for (int dd = 0; dd < 4; dd++) {
            uint32_t word = 0;
            for (int b = 0; b < 4; b++) {
                if (q4) {
                    uint32_t k_lo = u * 32 + 8 * dd + b, k_hi = k_lo + 4;
                    uint32_t lo = k_lo < w.K ? (uint32_t)synth_q4((uint64_t)k_lo * w.N + n, id) & 15u : 0u;
                    uint32_t hi = k_hi < w.K ? (uint32_t)synth_q4((uint64_t)k_hi * w.N + n, id) & 15u : 0u;
                    word |= (lo | (hi << 4)) << (8 * b);
                } else {
                    uint32_t k = u * 16 + 4 * dd + b;
                    uint32_t q = k < w.K ? (uint32_t)synth_q8((uint64_t)k * w.N + n, id) & 255u : 0u;
                    word |= q << (8 * b);
                }
            }
            d[dd] = word;
        }
        ((uint4*)w.qs)[it] = make_uint4(d[0], d[1], d[2], d[3]);
        if (nl < 16) {
            const uint32_t i = nl;
            if (q4) {
                uint32_t ka = u * 32 + i, kb = ka + 16;
                float sa = ka < w.K ? synth_scale((uint64_t)ka * NB + j, id) : 0.f;
                float sb = kb < w.K ? synth_scale((uint64_t)kb * NB + j, id) : 0.f;
                ((Pair<ST>*)w.sc)[ju * 16 + i] = Pair<ST>{ST(sa), ST(sb)};
            } else {
                uint32_t k = u * 16 + i;
                ((ST*)w.sc)[ju * 16 + i] = ST(k < w.K ? synth_scale((uint64_t)k * NB + j, id) : 0.f);
            }
        }
    }
}

// flags[0] != 0: some value outside [-8,7]; flags[1] != 0: some scale not exact in f16
__global__ void __launch_bounds__(kBlock) classify_kernel(const int8_t* __restrict__ data, uint64_t n_elems,
                                                          const float* __restrict__ scales, uint64_t n_blocks,
                                                          uint32_t* flags) {
    bool bad_q = false, bad_s = false;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock, t0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    for (uint64_t i = t0; i < n_elems; i += stride) bad_q |= data[i] < -8 || data[i] > 7;
    for (uint64_t b = t0; b < n_blocks; b += stride) bad_s |= __half2float(__float2half_rn(scales[b])) != scales[b];
    if (__any(bad_q) && (threadIdx.x & 63) == 0) atomicOr(&flags[0], 1u);
    if (__any(bad_s) && (threadIdx.x & 63) == 0) atomicOr(&flags[1], 1u);
}

// raw [K,N] int8 + f32 scale per 32 flat elements -> packed layout (see file header). One thread
// per 16-byte item; reads are 32-byte coalesced across the 32 columns of a block-column.
template <typename ST>
__global__ void __launch_bounds__(kBlock) pack_kernel(const int8_t* __restrict__ data,
                                                      const float* __restrict__ scales, QWeightDev w) {
    const uint32_t NB = w.N / 32;
    const bool q4 = w.format == QW_Q4;
    const uint32_t U = q4 ? w.KC : 2 * w.KC;
    const uint64_t n_items = (uint64_t)NB * U * 32;
    for (uint64_t it = (uint64_t)blockIdx.x * kBlock + threadIdx.x; it < n_items; it += (uint64_t)gridDim.x * kBlock) {
        const uint32_t nl = it & 31;
        const uint64_t ju = it >> 5;
        const uint32_t u = ju % U, j = ju / U;
        const uint32_t n = j * 32 + nl;
        uint32_t d[4];
        for (int dd = 0; dd < 4; dd++) {
            uint32_t word = 0;
            for (int b = 0; b < 4; b++) {
                if (q4) {
                    uint32_t k_lo = u * 32 + 8 * dd + b, k_hi = k_lo + 4;
                    uint32_t lo = k_lo < w.K ? (uint32_t)data[(uint64_t)k_lo * w.N + n] & 15u : 0u;
                    uint32_t hi = k_hi < w.K ? (uint32_t)data[(uint64_t)k_hi * w.N + n] & 15u : 0u;
                    word |= (lo | (hi << 4)) << (8 * b);
                } else {
                    uint32_t k = u * 16 + 4 * dd + b;
                    uint32_t q = k < w.K ? (uint32_t)data[(uint64_t)k * w.N + n] & 255u : 0u;
                    word |= q << (8 * b);
                }
            }
            d[dd] = word;
        }
        ((uint4*)w.qs)[it] = make_uint4(d[0], d[1], d[2], d[3]);
        if (nl < 16) {
            const uint32_t i = nl;
            if (q4) {
                uint32_t ka = u * 32 + i, kb = ka + 16;
                float sa = ka < w.K ? scales[(uint64_t)ka * NB + j] : 0.f;
                float sb = kb < w.K ? scales[(uint64_t)kb * NB + j] : 0.f;
                ((Pair<ST>*)w.sc)[ju * 16 + i] = Pair<ST>{ST(sa), ST(sb)};
            } else {
                uint32_t k = u * 16 + i;
                ((ST*)w.sc)[ju * 16 + i] = ST(k < w.K ? scales[(uint64_t)k * NB + j] : 0.f);
            }
        }
    }
}

inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

struct SplitPlan {
    uint32_t units_per_wg, ksplit;
};

// Choose the K split: enough workgroups to cover the chip ~2x, slices a multiple of 8 units
// (one unit pair per wave per step), LDS slice of x bounded to 32 KiB.
SplitPlan plan_split(const QWeightDev& w, uint32_t M) {
    const uint32_t NB = w.N / 32;
    const uint32_t U = w.format == QW_Q4 ? w.KC : 2 * w.KC;
    const uint32_t unit_k = w.format == QW_Q4 ? 32 : 16;
    const uint32_t max_units = (32 * 1024 / 4) / unit_k; // x slice <= 32 KiB of LDS
    const uint32_t target_wgs = 512;
    uint32_t ksplit = 1;
    uint64_t wgs = (uint64_t)NB * (M ? M : 1);
    if (wgs < target_wgs) ksplit = cdiv(target_wgs, wgs);
    uint32_t upw = cdiv(U, ksplit);
    upw = (upw + 7) / 8 * 8;
    if (upw < 8) upw = 8;
    if (upw > max_units) upw = max_units / 8 * 8;
    ksplit = cdiv(U, upw);
    return {upw, ksplit};
}

} // namespace

// ── device-side repack (compile_program time) ───────────────────────────────────────────────
void packed_bytes(QWFormat format, uint32_t scale_f16, uint64_t K, uint64_t N, uint64_t* qs_bytes,
                  uint64_t* sc_bytes) {
    const uint64_t NB = N / 32, KC = (K + 31) / 32;
    const uint64_t U = format == QW_Q4 ? KC : 2 * KC;
    *qs_bytes = NB * U * 32 * 16;
    const uint64_t n_sc = format == QW_Q4 ? NB * KC * 32 : NB * U * 16;
    *sc_bytes = n_sc * (scale_f16 ? 2 : 4);
}

uint32_t classify_qweight(hipStream_t s, const int8_t* raw_data, uint64_t n_elems, const float* raw_scales,
                          uint64_t n_blocks, uint32_t* flag_scratch) {
    hipMemsetAsync(flag_scratch, 0, 2 * sizeof(uint32_t), s);
    classify_kernel<<<1024, kBlock, 0, s>>>(raw_data, n_elems, raw_scales, n_blocks, flag_scratch);
    uint32_t flags[2] = {1, 1};
    hipMemcpyAsync(flags, flag_scratch, sizeof(flags), hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    return (flags[0] == 0 ? 1u : 0u) | (flags[1] == 0 ? 2u : 0u);
}

void launch_pack_qweight(hipStream_t s, const int8_t* raw_data, const float* raw_scales, const QWeightDev& out) {
    if (out.scale_f16)
        pack_kernel<__half><<<2048, kBlock, 0, s>>>(raw_data, raw_scales, out);
    else
        pack_kernel<float><<<2048, kBlock, 0, s>>>(raw_data, raw_scales, out);
}

uint64_t qmatmul_scratch_bytes(const QWeightDev& w, uint32_t M) {
    if (w.format == QW_RAW) return 0;
    SplitPlan sp = plan_split(w, M);
    return sp.ksplit > 1 ? (uint64_t)sp.ksplit * M * w.N * sizeof(float) : 0;
}

void launch_qmatmul(hipStream_t s, const QWeightDev& w, const QMatmulParams& p, float* scratch) {
    if (p.M == 0 || p.N == 0) return;
    if (w.format == QW_RAW) {
        dim3 grid(cdiv(p.N, kBlock), p.M);
        qmatmul_raw_kernel<<<grid, kBlock, 0, s>>>((const int8_t*)w.qs, (const float*)w.sc, w.bs, p);
        return;
    }
    const SplitPlan sp = plan_split(w, p.M);
    const bool q4 = w.format == QW_Q4;
    QMVArgs a{};
    a.qs = (const uint4*)w.qs;
    a.sc = w.sc;
    a.x = p.input;
    a.M = p.M, a.N = p.N, a.K = p.K, a.NB = p.N / 32;
    a.U = q4 ? w.KC : 2 * w.KC;
    a.units_per_wg = sp.units_per_wg;
    a.ksplit = sp.ksplit;
    a.in_rs = p.in_rs;
    if (sp.ksplit == 1) {
        a.out = p.dst;
        a.out_rs = p.dst_rs;
    } else {
        a.out = scratch;
        a.out_rs = p.N;
    }
    dim3 grid(a.NB * sp.ksplit, p.M);
    const size_t lds = ((size_t)sp.units_per_wg * (q4 ? 32 : 16) + kWaves * 32) * sizeof(float);
    if (q4) {
        if (w.scale_f16)
            qmatvec_q4_kernel<__half><<<grid, kBlock, lds, s>>>(a);
        else
            qmatvec_q4_kernel<float><<<grid, kBlock, lds, s>>>(a);
    } else {
        if (w.scale_f16)
            qmatvec_q8_kernel<__half><<<grid, kBlock, lds, s>>>(a);
        else
            qmatvec_q8_kernel<float><<<grid, kBlock, lds, s>>>(a);
    }
    if (sp.ksplit > 1) {
        dim3 g2(cdiv(p.N, kBlock), p.M);
        splitk_combine_kernel<<<g2, kBlock, 0, s>>>(p.dst, scratch, p.M, p.N, sp.ksplit, p.dst_rs);
    }
}

void launch_synth_packed(hipStream_t s, const QWeightDev& w, uint32_t matrix_id) {
    if (w.scale_f16)
        synth_packed_kernel<__half><<<2048, kBlock, 0, s>>>(w, matrix_id);
    else
        synth_packed_kernel<float><<<2048, kBlock, 0, s>>>(w, matrix_id);
}

} // namespace zgml
