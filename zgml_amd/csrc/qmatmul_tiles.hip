// qmatmul_tiles.hip — the M > 1 (prefill) forms of the block-quantized matmul for gfx950: tile kernels on the matrix cores, their
// launchers and the scratch they share. Semantics as qmatvec.hip's head says (src/quant.zig:475-578 == src/backend/reference.zig:530-565);
// the packed weight layouts are the ones qmatvec.hip documents and writes (QW_Q4 / QW_Q8; K-on-lanes weights never come here).
// Split out of qmatvec.hip in round 4 (one 3,700-line translation unit).
#include "qmv_common.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <utility>

namespace zgml {

namespace {

// ── M > 1 (prefill): tile kernel on the f32 matrix cores ─────────────────────────────────────
// With M rows of x the same weight stream feeds M dot products, so the op is a real contraction
// and goes to MFMA — in f32 (v_mfma_f32_16x16x4_f32), so products and sums stay f32 exactly as in
// the reference's loop (quant.zig:475-578); only the summation order differs. Same packed layout
// and the same thread mapping as the mat-vec: lane (i, r) of a wave holds column n = 16g+i of the
// row's k-unit u_r (16 B = 32 nibbles / 16 bytes). One MFMA step takes k_local = e of the FOUR
// units of the wave's rows as its 4 k-slots:  B[kslot=r][n=i] = f32(q[u_r, e][n]) and
// A[m=i][kslot=r] = x[m, k(u_r, e)] * scale(k(u_r, e), j)  — the per-(k, block-column) scale folds
// into the A operand exactly like the mat-vec's t[k]. R m-tiles of 16 rows share each converted B
// value. Waves of a workgroup split K; their partial tiles are summed through LDS in wave order.
// x: the A operand wants "lane = row m", i.e. 16 different rows per load instruction. Loading that
// straight from global memory costs 64 cache lines per instruction (measured: 5x slower than the
// MFMA bound), so each step's x chunk [16R rows][4*waves units] is staged through LDS with
// row-contiguous (coalesced) global loads (prefetched into registers one step ahead), rows padded
// by 4 floats so the per-lane ds_read_b128 of 16 different rows is bank-conflict free.
// A wave cannot overlap its own VALU work with its own MFMAs (measured, tools/exp/mfma.hip: every
// VALU instruction between two MFMAs adds its full issue time), so workgroups are 8 waves = 2 per
// SIMD, and the per-B-value VALU work (convert + scale) is shared by R m-tiles.
struct QMMArgs {
    const uint4* qs;
    const void* sc;
    const float* x;
    float* out;
    uint32_t M, K, U, in_rs, out_rs, NB2;
};

typedef float mfma_f4 __attribute__((ext_vector_type(4)));

// Single-instruction converts for the tile kernel (separate asm statements so hipcc can place
// them between the MFMAs): nibble -> q/16, byte -> q.
template <int BYTE>
__device__ __forceinline__ float cvt_nib(uint32_t w) {
    float c;
    if (BYTE == 0) asm("v_cvt_off_f32_i4_sdwa %0, %1 " ZGML_SDWA "0" : "=v"(c) : "v"(w));
    if (BYTE == 1) asm("v_cvt_off_f32_i4_sdwa %0, %1 " ZGML_SDWA "1" : "=v"(c) : "v"(w));
    if (BYTE == 2) asm("v_cvt_off_f32_i4_sdwa %0, %1 " ZGML_SDWA "2" : "=v"(c) : "v"(w));
    if (BYTE == 3) asm("v_cvt_off_f32_i4_sdwa %0, %1 " ZGML_SDWA "3" : "=v"(c) : "v"(w));
    return c;
}
template <int BYTE>
__device__ __forceinline__ float cvt_i8(uint32_t w) {
    float c;
    if (BYTE == 0) asm("v_cvt_f32_i32_sdwa %0, sext(%1) " ZGML_SDWA "0" : "=v"(c) : "v"(w));
    if (BYTE == 1) asm("v_cvt_f32_i32_sdwa %0, sext(%1) " ZGML_SDWA "1" : "=v"(c) : "v"(w));
    if (BYTE == 2) asm("v_cvt_f32_i32_sdwa %0, sext(%1) " ZGML_SDWA "2" : "=v"(c) : "v"(w));
    if (BYTE == 3) asm("v_cvt_f32_i32_sdwa %0, sext(%1) " ZGML_SDWA "3" : "=v"(c) : "v"(w));
    return c;
}
// One k-unit of one column held by a lane: 16 B of weights + the unit's scales. compute() turns
// it into 32 (16) B-operand values  b = f32(q) * scale  — for GGUF-sourced weights the product of
// a 4/8-bit integer and an f16 scale is exact in f32 — and feeds R MFMAs per value with the A
// operand x[m, k] read straight from the staged LDS chunk. (Q4 values come out of the convert as
// q/16; the kernel multiplies the finished tile by 16, exact.)
template <typename ST, bool Q4, bool NT>
struct TileUnit {
    static constexpr int KU = Q4 ? 32 : 16;             // k per unit
    static constexpr int NS = KU * (int)sizeof(ST) / 16; // uint4 of scales per unit
    static constexpr bool kHalf = sizeof(ST) == 2;
    uint4 wq;
    uint4 sraw[NS];
    __device__ __forceinline__ void load(const uint4* qs, const uint4* sc, uint32_t u, uint32_t U) {
        const uint32_t uc = min(u, U - 1); // clamped, unconditional; x of units >= U is staged as zero
        wq = wload<NT>(qs + (uint64_t)uc * 16);
#pragma unroll
        for (int q = 0; q < NS; q++) sraw[q] = sc[(uint64_t)uc * NS + q];
    }
    // scale of k_local applied to a converted value. The multiply is left to hipcc on purpose: the
    // value feeds an MFMA next, and gfx950 needs wait states between a VALU write and the MFMA read
    // that the compiler only inserts for producers it can see (an inline-asm v_fma_mix here read
    // stale registers).
    template <int K_LOCAL>
    __device__ __forceinline__ float scaled(float c) const {
        constexpr int E = Q4 ? 2 * (K_LOCAL % 16) + K_LOCAL / 16 : K_LOCAL; // element index in the unit's scale array
        return c * to_f32(((const ST*)sraw)[E]);
    }
    template <int R, int DD, int BY, int H>
    __device__ __forceinline__ void one(mfma_f4 (&acc)[R], const float4 (&xv)[R][2], uint32_t w) const {
        constexpr int K_LOCAL = Q4 ? 8 * DD + BY + 4 * H : 4 * DD + BY;
        const float b = scaled<K_LOCAL>(Q4 ? cvt_nib<BY>(w) : cvt_i8<BY>(w));
#pragma unroll
        for (int t = 0; t < R; t++) {
            constexpr int J = Q4 ? (BY + 4 * H) : BY; // position within the dword's 8 (4) k values
            const float4 v = xv[t][J / 4];
            const float xk = (J % 4) == 0 ? v.x : (J % 4) == 1 ? v.y : (J % 4) == 2 ? v.z : v.w;
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xk, b, acc[t], 0, 0, 0);
        }
    }
    template <int R, int DD>
    __device__ __forceinline__ void dword(mfma_f4 (&acc)[R], const float* xs, uint32_t tile_stride, uint32_t w) const {
        constexpr int KD = Q4 ? 8 : 4; // k values per weight dword
        float4 xv[R][2];
#pragma unroll
        for (int t = 0; t < R; t++)
#pragma unroll
            for (int c = 0; c < KD / 4; c++) xv[t][c] = *(const float4*)(xs + t * tile_stride + DD * KD + 4 * c);
        one<R, DD, 0, 0>(acc, xv, w);
        one<R, DD, 1, 0>(acc, xv, w);
        one<R, DD, 2, 0>(acc, xv, w);
        one<R, DD, 3, 0>(acc, xv, w);
        if (Q4) {
            const uint32_t h = w >> 4;
            one<R, DD, 0, 1>(acc, xv, h);
            one<R, DD, 1, 1>(acc, xv, h);
            one<R, DD, 2, 1>(acc, xv, h);
            one<R, DD, 3, 1>(acc, xv, h);
        }
    }
    // ── the same contraction on the XDL (bf16) matrix cores ──
    // The f32 MFMA shares the FP32 lanes with the VALU (tools/exp/mfma.hip), so the f32 form above is
    // bound by instruction issue. Here B = q itself (|q| <= 127: exact in bf16; Q4 as q/16) and the
    // per-(k, block-column) scale goes into A: t = x * scale in f32 exactly as the reference computes it,
    // then t is split into three bf16 pieces t = h1 + h2 + h3 (each residual is exact in f32, the third
    // piece leaves < 2^-26 |t|), and  sum_k t*q  is three v_mfma_f32_16x16x32_bf16 with exact products and
    // f32 accumulation. One MFMA takes k_local = 8g..8g+7 of the four units of the wave's rows as its
    // 32 k-values. All producers of MFMA operands are compiler-visible (cvt_pk), see scaled() above.
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    template <int K_LOCAL>
    __device__ __forceinline__ float scale_of() const {
        constexpr int E = Q4 ? 2 * (K_LOCAL % 16) + K_LOCAL / 16 : K_LOCAL;
        return to_f32(((const ST*)sraw)[E]);
    }
    template <int R, int G8>
    __device__ __forceinline__ void group8(mfma_f4 (&acc)[R], const float* xs, uint32_t tile_stride) const {
        const uint32_t w[4] = {wq.x, wq.y, wq.z, wq.w};
        float qf[8];
        if (Q4) { // dword G8: byte b low nibble = k_local 8*G8 + b, high nibble = 8*G8 + 4 + b
            const uint32_t lo = w[G8], hi = lo >> 4;
            qf[0] = cvt_nib<0>(lo), qf[1] = cvt_nib<1>(lo), qf[2] = cvt_nib<2>(lo), qf[3] = cvt_nib<3>(lo);
            qf[4] = cvt_nib<0>(hi), qf[5] = cvt_nib<1>(hi), qf[6] = cvt_nib<2>(hi), qf[7] = cvt_nib<3>(hi);
        } else { // dwords 2*G8, 2*G8+1: 8 consecutive int8
            const uint32_t w0 = w[(2 * G8) & 3], w1 = w[(2 * G8 + 1) & 3];
            qf[0] = cvt_i8<0>(w0), qf[1] = cvt_i8<1>(w0), qf[2] = cvt_i8<2>(w0), qf[3] = cvt_i8<3>(w0);
            qf[4] = cvt_i8<0>(w1), qf[5] = cvt_i8<1>(w1), qf[6] = cvt_i8<2>(w1), qf[7] = cvt_i8<3>(w1);
        }
        bf16x8 bq;
#pragma unroll
        for (int e = 0; e < 8; e++) bq[e] = (__bf16)qf[e];
        const float sk[8] = {scale_of<8 * G8 + 0>(), scale_of<8 * G8 + 1>(), scale_of<8 * G8 + 2>(), scale_of<8 * G8 + 3>(),
                             scale_of<8 * G8 + 4>(), scale_of<8 * G8 + 5>(), scale_of<8 * G8 + 6>(), scale_of<8 * G8 + 7>()};
#pragma unroll
        for (int t = 0; t < R; t++) {
            const float4 xa = *(const float4*)(xs + t * tile_stride + 8 * G8), xb = *(const float4*)(xs + t * tile_stride + 8 * G8 + 4);
            const float tv[8] = {xa.x * sk[0], xa.y * sk[1], xa.z * sk[2], xa.w * sk[3], xb.x * sk[4], xb.y * sk[5], xb.z * sk[6], xb.w * sk[7]};
            bf16x8 h1, h2, h3;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                h1[e] = (__bf16)tv[e];
                const float r1 = tv[e] - (float)h1[e];
                h2[e] = (__bf16)r1;
                const float r2 = r1 - (float)h2[e];
                h3[e] = (__bf16)r2;
            }
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, bq, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h2, bq, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h3, bq, acc[t], 0, 0, 0);
        }
    }
    template <int R>
    __device__ __forceinline__ void compute_xdl(mfma_f4 (&acc)[R], const float* xs, uint32_t tile_stride) const {
        group8<R, 0>(acc, xs, tile_stride);
        group8<R, 1>(acc, xs, tile_stride);
        if (Q4) {
            group8<R, 2>(acc, xs, tile_stride);
            group8<R, 3>(acc, xs, tile_stride);
        }
    }
    // xs: this lane's row of the staged chunk at its unit; m-tiles are `tile_stride` floats apart
    template <int R>
    __device__ __forceinline__ void compute(mfma_f4 (&acc)[R], const float* xs, uint32_t tile_stride) const {
        dword<R, 0>(acc, xs, tile_stride, wq.x);
        dword<R, 1>(acc, xs, tile_stride, wq.y);
        dword<R, 2>(acc, xs, tile_stride, wq.z);
        dword<R, 3>(acc, xs, tile_stride, wq.w);
    }
};

template <typename ST, bool Q4, int R, bool XVEC, bool NT, bool XDL>
__global__ void __launch_bounds__(512) qmatmul_tile_kernel(QMMArgs a) {
    using Unit = TileUnit<ST, Q4, NT>;
    constexpr int KU = Unit::KU;
    constexpr int NX = R * KU / 4; // float4 of x each thread stages per step
    extern __shared__ float smem[];
    const uint32_t g = column_group(blockIdx.x, a.NB2), m0 = blockIdx.y * 16 * R;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const uint32_t row = lane >> 4, i = lane & 15;
    const uint32_t stride = n_waves * 4;          // units per step
    const uint32_t chf = stride * KU;             // floats of k per step
    const uint32_t row_stride = chf + 4;          // padded LDS row
    const uint4* qs = a.qs + (uint64_t)g * a.U * 16 + i;
    const uint4* sc = (const uint4*)((const char*)a.sc + (uint64_t)(g >> 1) * a.U * Unit::NS * 16);
    const uint32_t n_steps = (a.U + stride - 1) / stride;

    // staging map: float4 slot f = tid + j*blockDim -> (row f / (chf/4), column 4*(f % (chf/4)))
    const uint32_t f4_per_row = chf / 4;
    uint32_t st_col[NX], st_lds[NX];
    const float* st_src[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) {
        const uint32_t f = threadIdx.x + j * blockDim.x, r = f / f4_per_row;
        st_col[j] = 4 * (f % f4_per_row);
        st_lds[j] = r * row_stride + st_col[j];
        st_src[j] = a.x + (uint64_t)min(m0 + r, a.M - 1) * a.in_rs;
    }
    float4 stage[NX];
    auto stage_load = [&](uint32_t step) {
#pragma unroll
        for (int j = 0; j < NX; j++) stage[j] = load_x4<XVEC>(st_src[j], step * chf + st_col[j], a.K);
    };
    auto stage_store = [&](uint32_t step) {
        if ((step + 1) * chf <= a.K) { // whole chunk inside K (uniform): no selects
#pragma unroll
            for (int j = 0; j < NX; j++) *(float4*)(smem + st_lds[j]) = stage[j];
        } else {
#pragma unroll
            for (int j = 0; j < NX; j++) *(float4*)(smem + st_lds[j]) = zero_tail(stage[j], step * chf + st_col[j], a.K);
        }
    };

    mfma_f4 acc[R];
#pragma unroll
    for (int t = 0; t < R; t++) acc[t] = mfma_f4{0.f, 0.f, 0.f, 0.f};
    const uint32_t ul = 4 * w + row; // this row's unit within a step
    const float* xs = smem + i * row_stride + ul * KU;
    stage_load(0);
    Unit cur;
    cur.load(qs, sc, ul, a.U);
    stage_store(0);
    __syncthreads();
    for (uint32_t st = 1; st < n_steps; st++) {
        stage_load(st); // next chunk -> registers while this one is consumed from LDS
        Unit nxt;
        nxt.load(qs, sc, ul + st * stride, a.U);
        __builtin_amdgcn_sched_barrier(0); // keep the prefetch ahead of the MFMA block (hipcc sinks it otherwise)
        if (XDL)
            cur.template compute_xdl<R>(acc, xs, 16 * row_stride);
        else
            cur.template compute<R>(acc, xs, 16 * row_stride);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        stage_store(st);
        __syncthreads();
        cur = nxt;
    }
    if (XDL)
        cur.template compute_xdl<R>(acc, xs, 16 * row_stride);
    else
        cur.template compute<R>(acc, xs, 16 * row_stride);
    __syncthreads();

    // D[m = 4*row + v][n = i] in acc[t][v]; fold the waves in fixed order (reuses the x buffer)
#pragma unroll
    for (int t = 0; t < R; t++)
#pragma unroll
        for (int v = 0; v < 4; v++) smem[((w * R + t) * 4 + v) * 64 + lane] = Q4 ? acc[t][v] * 16.0f : acc[t][v];
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < (uint32_t)R * 256; idx += blockDim.x) {
        const uint32_t t = idx >> 8, v = (idx >> 6) & 3, l = idx & 63;
        float sum = smem[(t * 4 + v) * 64 + l];
        for (uint32_t ww = 1; ww < n_waves; ww++) sum += smem[((ww * R + t) * 4 + v) * 64 + l];
        const uint32_t m = m0 + t * 16 + 4 * (l >> 4) + v, n = g * 16 + (l & 15);
        if (m < a.M) a.out[(uint64_t)m * a.out_rs + n] = sum;
    }
}

// ── XDL tile kernel, second form (Q4_0 with f16 block scales): the scale goes into B ──────────
// The form above pays ~6.5 VALU instructions per (row, k, 32-column block) to scale and split the A
// operand, in every workgroup, and stages x through LDS with two barriers per step: measured 21 us
// for 32 x 4096 x 4096 against an HBM time of 1.2 us. Here
//   * A = x itself, split ONCE per matmul (split_a_kernel) into three bf16 pieces by truncation
//     (x = h1 + h2 + h3 exactly: 8 + 8 + 8 significant bits) and laid out as the MFMA A operand, so a
//     lane's load is 16 contiguous bytes of a 1 KB wave-contiguous block (L2-resident, no LDS, no
//     barrier in the K loop);
//   * B = w = (q - 8)/16 * scale, exact in f32 (4-bit integer x 11-bit f16 scale = 15 significant
//     bits), split into two bf16 pieces w = b1 + b2 exactly (8 + 7 bits): 5 VALU instructions per
//     weight, independent of M;
//   * x*w = sum of the 3 x 2 piece products, each exact in the MFMA's f32 accumulator: six
//     v_mfma_f32_16x16x32_bf16 per 8-k group and m-tile, f32-level accuracy as before.
// Waves of a workgroup split the K steps (128 k each) and fold their tiles through LDS in wave order;
// a workgroup owns G column groups of one 32-column scale block and R m-tiles.
struct QMM2Part { // one weight of a grouped launch (q/k/v, gate/up: same rows, same K)
    const uint4* qs;
    const uint4* sc;
    float* out;
    uint32_t out_rs, NB2, block_begin; // first workgroup of the part
};
constexpr int kMaxQmmParts = 3;
constexpr int kMaxSpw = 12; // K steps per wave the weight preload is unrolled for (8 waves: K <= 12288)
struct QMM2Args {
    QMM2Part parts[kMaxQmmParts];
    const uint4* ap; // split_a_kernel output
    uint32_t n_parts, M, U, S; // S = K steps = ceil(U / 4)
    uint32_t spw_max;          // ceil(S / waves) <= kMaxSpw: sizes the weights' LDS region
};

// A pieces: ap[(((t * S + s) * 4 + j) * kAPieces + p) * 64 + lane] = the 8 bf16 of piece p that lane (i = lane % 16:
// row 16 t + i, r = lane / 16: unit 4 s + r) feeds to the MFMA of k_local 8 j .. 8 j + 7. Rows >= M and k >= K are zero.
__global__ void __launch_bounds__(256) split_a_kernel(const float* __restrict__ x, uint32_t M, uint32_t K, uint32_t in_rs, uint4* __restrict__ ap,
                                                      uint32_t S) {
    const uint32_t s = blockIdx.x, t = blockIdx.y, j = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, r = lane >> 4;
    const uint32_t m = 16 * t + i, k0 = (4 * s + r) * 32 + 8 * j;
    uint32_t h[kAPieces][8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float v = (m < M && k0 + e < K) ? x[(uint64_t)m * in_rs + k0 + e] : 0.0f;
        uint32_t hp[kAPieces];
        split_a_pieces(v, hp); // kernels.h: bf16 pieces by round-to-nearest-even, each in the high half of its dword
#pragma unroll
        for (int p = 0; p < kAPieces; p++) h[p][e] = hp[p];
    }
    uint4* dst = ap + ((((uint64_t)t * S + s) * 4 + j) * kAPieces) * 64 + lane;
#pragma unroll
    for (int p = 0; p < kAPieces; p++)
        dst[p * 64] = make_uint4((h[p][0] >> 16) | (h[p][1] & 0xFFFF0000u), (h[p][2] >> 16) | (h[p][3] & 0xFFFF0000u),
                                 (h[p][4] >> 16) | (h[p][5] & 0xFFFF0000u), (h[p][6] >> 16) | (h[p][7] & 0xFFFF0000u));
}

template <int CTRL>
__device__ __forceinline__ float row_bcast(int v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false)); }
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8_t as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8_t, v); }
// (hi16(a), hi16(b)) -> one dword of two bf16 (a in the low half)
__device__ __forceinline__ uint32_t pack_hi16(float a, float b) { return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u); }

// B pieces of one 8-k group: w = (q - 8)/16 * scale (exact in f32: 4 x 11 bits), split into bf16 hi + bf16 lo (exact).
// Written so that hipcc emits ~4.6 VALU instructions per weight instead of the ~8 of the xdl2 / xdl4 form:
//   * the scale broadcast rides in the multiply (v_mul_f32_dpp row_newbcast: full masks + bound_ctrl make the DPP move foldable);
//   * every convert takes its nibble through an SDWA byte select: the high nibbles come from ONE shifted copy of the dword
//     (kept opaque, or the shifts merge into byte-unaligned ones), and the `& 0xFF` is what lets the peephole pick BYTE_n;
//   * the residuals w - hi are formed two at a time (v_pk_add_f32 with a negated operand).
// x w = (a1 + a2 + a3)(b1 + b2) exactly (bf16 pieces, 8 bits each; b2 holds the <= 7 bits b1 leaves of the 15-bit weight). Five of the
// six piece products are computed: a3 b2 is <= 2^-16 x 2^-8 of |x w| — measured 2.9e-9 of sum |x w| at K = 4096 against 4.8e-8
// for a plain f32 matmul's own rounding and the tests' 2e-5 — and costs one MFMA in six. (3 = all six.)
constexpr int kXdlLoPieces = kAPieces < 2 ? kAPieces : 2; // A pieces that also meet the weight's LOW piece (round 4: both of the two)
template <int CTRL>
__device__ __forceinline__ float row_bcast_z(int v) { return __int_as_float(__builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true)); } // (no `old` to initialise)
typedef float xf32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t xu32x2 __attribute__((ext_vector_type(2)));

template <int J, int C>
__device__ __forceinline__ void xdl_prep_b(const uint4 (&wq)[C], int sb_lo, int sb_hi, uint4 (&b1)[C], uint4 (&b2)[C]) {
    const int sbits = J >> 1 ? sb_hi : sb_lo;
    constexpr int C0 = 0x150 + 8 * (J & 1); // the scale of k_local 8 J + e sits in lane 8 (J & 1) + e of this row
#pragma unroll
    for (int g = 0; g < C; g++) {
        const uint32_t lo = J == 0 ? wq[g].x : J == 1 ? wq[g].y : J == 2 ? wq[g].z : wq[g].w;
        uint32_t hi = lo >> 4;
        asm volatile("" : "+v"(hi));
#define XDL5_W(E, SRC, BYTE) (row_bcast_z<C0 + E>(sbits) * __builtin_amdgcn_cvt_off_f32_i4((int)(((SRC) >> (8 * BYTE)) & 0xFFu))) /* (q - 8)/16 * scale */
        const xf32x2 w01 = {XDL5_W(0, lo, 0), XDL5_W(1, lo, 1)}, w23 = {XDL5_W(2, lo, 2), XDL5_W(3, lo, 3)};
        const xf32x2 w45 = {XDL5_W(4, hi, 0), XDL5_W(5, hi, 1)}, w67 = {XDL5_W(6, hi, 2), XDL5_W(7, hi, 3)};
#undef XDL5_W
        auto top = [](xf32x2 v) { return __builtin_bit_cast(xf32x2, __builtin_bit_cast(xu32x2, v) & 0xFFFF0000u); };
        const xf32x2 l01 = w01 - top(w01), l23 = w23 - top(w23), l45 = w45 - top(w45), l67 = w67 - top(w67);
        b1[g] = make_uint4(pack_hi16(w01.x, w01.y), pack_hi16(w23.x, w23.y), pack_hi16(w45.x, w45.y), pack_hi16(w67.x, w67.y));
        b2[g] = make_uint4(pack_hi16(l01.x, l01.y), pack_hi16(l23.x, l23.y), pack_hi16(l45.x, l45.y), pack_hi16(l67.x, l67.y));
    }
}

// The K loop is software-pipelined by hand: while the MFMAs of 8-k group J run on the matrix pipe, the VALU
// prepares the B pieces of group J + 1 (a wave issues in order, so MFMAs and their own operand preparation
// only overlap if they are interleaved in program order; sched_group_barrier asks hipcc for 1 MFMA : 3 VALU).
// A is fetched one group ahead, the weights two steps ahead. Everything is compiler-visible (builtins, no
// inline asm) so the scheduler can move it and inserts the VALU -> MFMA / DPP wait states itself. The block
// scales: lane i of a row loads ONE dword (k_local i | 16 + i) and the multiply reads it from lane
// 8 (J & 1) + e of the row through DPP row_newbcast — 16 identical 64-byte loads and 32 converts per row before.
template <int R, int G, bool NT>
__global__ void __launch_bounds__(512) qmatmul_xdl2_kernel(QMM2Args a) {
    extern __shared__ float smem[];
    uint32_t pi = 0;
#pragma unroll
    for (uint32_t t = 1; t < (uint32_t)kMaxQmmParts; t++)
        if (t < a.n_parts && blockIdx.x >= a.parts[t].block_begin) pi = t;
    const QMM2Part& P = a.parts[pi];
    const uint32_t pb = blockIdx.x - P.block_begin;
    const uint32_t g0 = (G == 1 ? column_group(pb, P.NB2) : pb * G), t0 = blockIdx.y * R;
    // w in an SGPR: loop bounds that guard MFMAs must be scalar (a v_mfma ignores EXEC)
    const uint32_t lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = blockDim.x >> 6;
    const uint32_t row = lane >> 4, i = lane & 15;
    const uint4* qs = P.qs + (uint64_t)g0 * a.U * 16 + i;
    // 32 f16 scales per unit = 16 dwords; dword d = {k_local d, k_local 16 + d}
    const uint32_t* scd = (const uint32_t*)P.sc + (uint64_t)(g0 >> 1) * a.U * 16 + i;
    const uint4* ap = a.ap + (uint64_t)t0 * a.S * (4 * kAPieces) * 64 + lane;
    const uint64_t tile_stride = (uint64_t)a.S * (4 * kAPieces) * 64; // uint4 between m-tiles

    // one accumulator per (column group, m-tile, A piece): consecutive MFMAs never chain through the same registers
    mfma_f4 acc[G][R][kAPieces];
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int t = 0; t < R; t++)
#pragma unroll
            for (int p = 0; p < kAPieces; p++) acc[g][t][p] = mfma_f4{0.f, 0.f, 0.f, 0.f};

    struct BStep { // one K step of this lane: its unit's nibbles per column group, its dword of the unit's scales
        uint4 wq[G];
        uint32_t sd;
    };
    struct AGrp { // the A pieces of one 8-k group
        uint4 v[R][kAPieces];
    };
    struct BPieces { // w = (q - 8)/16 * scale of one 8-k group, as two bf16x8 operands per column group
        uint4 b1[G], b2[G];
    };
    auto load_b = [&](BStep& b, uint32_t s) {
        const uint32_t u = min(4 * s + row, a.U - 1); // units past the end: A is zero there
#pragma unroll
        for (int g = 0; g < G; g++) b.wq[g] = wload<NT>(qs + ((uint64_t)g * a.U + u) * 16);
        b.sd = scd[(uint64_t)u * 16];
    };
    auto load_a = [&](AGrp& x, uint32_t s, uint32_t j) {
#pragma unroll
        for (int t = 0; t < R; t++)
#pragma unroll
            for (int p = 0; p < kAPieces; p++) x.v[t][p] = ap[t * tile_stride + (((uint64_t)s * 4 + j) * kAPieces + p) * 64];
    };
    auto prep = [&](BPieces& o, const BStep& b, auto jc) {
        constexpr int J = decltype(jc)::value;
        const __half2 hh = __builtin_bit_cast(__half2, b.sd);
        const int sbits = __float_as_int(__half2float(J >> 1 ? hh.y : hh.x)); // this lane's scale of k_local (J >> 1) * 16 + i
        xdl_prep_b<J, G>(b.wq, sbits, sbits, o.b1, o.b2);
    };
    auto mfmas = [&](const BPieces& o, const AGrp& x) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            const bf16x8_t b1 = as_bf16x8(o.b1[g]), b2 = as_bf16x8(o.b2[g]);
#pragma unroll
            for (int p = 0; p < kAPieces; p++)
#pragma unroll
                for (int t = 0; t < R; t++) acc[g][t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(x.v[t][p]), b1, acc[g][t][p], 0, 0, 0);
#pragma unroll
            for (int p = 0; p < kXdlLoPieces; p++) // (the third A piece x the low B piece is not computed: kXdlLoPieces)
#pragma unroll
                for (int t = 0; t < R; t++) acc[g][t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(x.v[t][p]), b2, acc[g][t][p], 0, 0, 0);
        }
    };
    // one pipeline stage: fetch A of the group after next's predecessor, prepare B of the next group, multiply the current one
#define XDL2_STAGE(LOAD_A, PREP, MFMA)                                                  \
    do {                                                                               \
        LOAD_A;                                                                        \
        PREP;                                                                          \
        MFMA;                                                                          \
        __builtin_amdgcn_sched_group_barrier(0x020, kAPieces * R, 0); /* the A loads first */ \
        _Pragma("unroll") for (int k_ = 0; k_ < (kAPieces + kXdlLoPieces) * R * G; k_++) {    \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* 1 MFMA */            \
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); /* 4 VALU */            \
        }                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                             \
    } while (0)

    // The weights first, all of them: loads return in order, so a weight prefetch (HBM, ~2 us under load) issued
    // inside the K loop sits in front of every A load issued after it and each step then costs one HBM round trip
    // (measured: ~7000 cycles per wave-step against ~1900 of work). A wave's weights for ALL its steps are small
    // (<= kMaxSpw x (16 G + 4) B per lane): they are fetched up front in one straight-line block, parked in a
    // wave-private LDS region, and read back per step with ds_read (its own counter) — the vector-memory queue of
    // the loop then holds A loads only.
    const uint32_t spw = a.S > w ? (a.S - w + n_waves - 1) / n_waves : 0; // this wave's steps: w, w + n_waves, ...
    uint4* const lds_wq = (uint4*)smem + (size_t)w * a.spw_max * G * 64;  // [k][g][lane]
    uint32_t* const lds_sd = (uint32_t*)((uint4*)smem + (size_t)n_waves * a.spw_max * G * 64) + (size_t)w * a.spw_max * 64; // [k][lane]
    {
        BStep all[kMaxSpw];
#pragma unroll
        for (int k = 0; k < kMaxSpw; k++)
            if ((uint32_t)k < spw) load_b(all[k], w + k * n_waves); // (scalar guard) everything is waited for right below anyway
#pragma unroll
        for (int k = 0; k < kMaxSpw; k++)
            if ((uint32_t)k < spw) {
#pragma unroll
                for (int g = 0; g < G; g++) lds_wq[(k * G + g) * 64 + lane] = all[k].wq[g];
                lds_sd[k * 64 + lane] = all[k].sd;
            }
    }
    auto read_b = [&](BStep& b, uint32_t k) { // wave-private: no barrier, the wave's own ds ops are ordered
#pragma unroll
        for (int g = 0; g < G; g++) b.wq[g] = lds_wq[(k * G + g) * 64 + lane];
        b.sd = lds_sd[k * 64 + lane];
    };
    if (spw) {
        BStep cur, nx1;
        AGrp a0, a1;
        BPieces p0, p1;
        const uint32_t k_last = spw - 1;
        read_b(cur, 0);
        load_a(a0, w, 0);
        prep(p0, cur, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        for (uint32_t k = 0; k < spw; k++) {
            const uint32_t s = w + k * n_waves, kn = min(k + 1, k_last), sn = w + kn * n_waves; // clamped: the last prefetch re-reads live lines
            read_b(nx1, kn);
            __builtin_amdgcn_sched_barrier(0);
            XDL2_STAGE(load_a(a1, s, 1), prep(p1, cur, std::integral_constant<int, 1>{}), mfmas(p0, a0));
            XDL2_STAGE(load_a(a0, s, 2), prep(p0, cur, std::integral_constant<int, 2>{}), mfmas(p1, a1));
            XDL2_STAGE(load_a(a1, s, 3), prep(p1, cur, std::integral_constant<int, 3>{}), mfmas(p0, a0));
            XDL2_STAGE(load_a(a0, sn, 0), prep(p0, nx1, std::integral_constant<int, 0>{}), mfmas(p1, a1));
            cur = nx1;
        }
    }
#undef XDL2_STAGE
    __syncthreads(); // the reduction below reuses the weights' LDS region

    // D[m = 4 * row + v][n = i] in acc[g][t][v]; fold the waves in fixed order; the tile carries q/16: x 16 (exact)
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int t = 0; t < R; t++)
#pragma unroll
            for (int v = 0; v < 4; v++) smem[((w * (G * R) + g * R + t) * 4 + v) * 64 + lane] = kAPieces == 2 ? acc[g][t][0][v] + acc[g][t][kAPieces - 1][v] : (acc[g][t][0][v] + acc[g][t][1][v]) + acc[g][t][kAPieces - 1][v];
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < (uint32_t)(G * R) * 256; idx += blockDim.x) {
        const uint32_t gt = idx >> 8, v = (idx >> 6) & 3, l = idx & 63;
        float sum = smem[(gt * 4 + v) * 64 + l];
        for (uint32_t ww = 1; ww < n_waves; ww++) sum += smem[((ww * (G * R) + gt) * 4 + v) * 64 + l];
        const uint32_t g = gt / R, t = gt % R;
        const uint32_t m = (t0 + t) * 16 + 4 * (l >> 4) + v, n = (g0 + g) * 16 + (l & 15);
        if (m < a.M) P.out[(uint64_t)m * P.out_rs + n] = sum * 16.0f;
    }
}

// ── XDL tile kernel for M > 32 (prefill chunks of 64 .. 128+ tokens): RT m-tiles per workgroup ────────────────────
// At M = 32 the per-weight preparation of B (convert, scale, exact two-way bf16 split, packs: 5 VALU instructions per
// weight, independent of M) costs 3.3 VALU instructions per MFMA and does not hide under the matrix pipe (a 16x16x32
// MFMA holds the SIMD's vector issue for 8 of its 16 cycles): every M = 32 form measured lands at ~32 cycles per MFMA
// per SIMD (profiles/r02_exp_stream_kernel.txt). The preparation is per WEIGHT, so with RT = 8 m-tiles (128 rows) per
// workgroup one prepared B operand feeds 48 MFMAs (0.8 VALU per MFMA) and the kernel is paced by the matrix pipe.
// What grows with M is A: a workgroup that owns all of K pulls all of A (M K 6 B as bf16 pieces) through L2, so here
//   * a workgroup owns ONE scale block-column (C = 2 column groups, 32 columns), RT m-tiles and a SLICE of the K steps;
//     its 8 waves take whole steps (128 k);
//   * per 8-k group J the wave holds the A pieces of the RT tiles in registers (12 RT VGPRs, fetched half a group ahead
//     in two halves of RT / 2 tiles) and uses them for both column groups: 12 RT MFMAs per prepared B pair, chained on one
//     accumulator per (column group, m-tile) (a dependent 16x16x32 chain issues at the full rate);
//   * waves fold through LDS in wave order; a K split publishes partial tiles (write-through stores, drained, one
//     agent-scope add on the block's counter) and the LAST arriver sums the slices in slice order, scales, stores.
struct QMM4Args {
    QMM2Part parts[kMaxQmmParts]; // NB2 / block_begin in scale BLOCK-COLUMNS (pairs of groups) here
    const uint4* ap;              // split_a_kernel output (tiles padded to a multiple of RT)
    float* partial;               // [tile group][block-column][slice][2 RT 256] f32 (split-K only)
    uint32_t* counter;            // one word per (tile group, block-column), zero between launches
    uint32_t n_parts, M, U, S, SK, steps_per_slice;
};
constexpr int kX4Waves = 8;

template <int RT, int CB, bool NT>
__global__ void __launch_bounds__(512, 2) qmatmul_xdl4_kernel(QMM4Args a) {
    constexpr int C = 2, H = RT / 2; // H tiles per half
    constexpr uint32_t KW = kX4Waves / CB; // waves that split the K steps of one block-column
    extern __shared__ float smem[];
    const uint32_t cbx = blockIdx.x / a.SK, slice = blockIdx.x - cbx * a.SK; // launch-wide block-column, K slice
    uint32_t pi = 0;
#pragma unroll
    for (uint32_t t = 1; t < (uint32_t)kMaxQmmParts; t++)
        if (t < a.n_parts && cbx >= a.parts[t].block_begin) pi = t;
    const QMM2Part& P = a.parts[pi];
    const uint32_t lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // CB block-columns per workgroup: waves (kw, kw + KW, ...) take the SAME K steps on neighbouring block-columns, so
    // the second reader of a step's A pieces mostly hits this CU's L1 (A through L2 / CB)
    const uint32_t bcw = w / KW, kw = w - bcw * KW;
    const uint32_t cb = (cbx - P.block_begin) * CB + bcw, g0 = cb * C, t0 = blockIdx.y * RT;
    const uint32_t row = lane >> 4, i = lane & 15;
    const uint4* qs = P.qs + (uint64_t)g0 * a.U * 16 + i;
    const uint32_t* scd = (const uint32_t*)P.sc + (uint64_t)cb * a.U * 16 + i;
    const uint4* ap = a.ap + (uint64_t)t0 * a.S * (4 * kAPieces) * 64 + lane;
    const uint64_t tile_stride = (uint64_t)a.S * (4 * kAPieces) * 64;
    const uint32_t s_begin = slice * a.steps_per_slice, s_end = min(s_begin + a.steps_per_slice, a.S);

    mfma_f4 acc[C][RT];
#pragma unroll
    for (int g = 0; g < C; g++)
#pragma unroll
        for (int t = 0; t < RT; t++) acc[g][t] = mfma_f4{0.f, 0.f, 0.f, 0.f};

    struct AHalf {
        uint4 v[H][kAPieces];
    };
    auto load_a = [&](AHalf& x, uint32_t s, int J, int half) {
#pragma unroll
        for (int t = 0; t < H; t++)
#pragma unroll
            for (int p = 0; p < kAPieces; p++) x.v[t][p] = ap[(uint64_t)(half * H + t) * tile_stride + (((uint64_t)s * 4 + J) * kAPieces + p) * 64];
    };
    for (uint32_t s = s_begin + kw; s < s_end; s += KW) { // (scalar loop: kw, s are in SGPRs)
        const uint32_t u = min(4 * s + row, a.U - 1); // units past the end: A is zero there
        uint4 wq[C];
#pragma unroll
        for (int g = 0; g < C; g++) wq[g] = wload<NT>(qs + ((uint64_t)g * a.U + u) * 16);
        const uint32_t sd = scd[(uint64_t)u * 16];
        AHalf a0, a1;
        load_a(a0, s, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const __half2 hh = __builtin_bit_cast(__half2, sd);
        const int sb_lo = __float_as_int(__half2float(hh.x)), sb_hi = __float_as_int(__half2float(hh.y)); // scales of k_local i, 16 + i
        uint4 b1[C], b2[C];
        auto prep = [&](auto jc) { xdl_prep_b<decltype(jc)::value, C>(wq, sb_lo, sb_hi, b1, b2); };
        auto mfmas = [&](const AHalf& x, int half) {
#pragma unroll
            for (int g = 0; g < C; g++)
#pragma unroll
                for (int t = 0; t < H; t++)
#pragma unroll
                    for (int p = 0; p < kAPieces; p++) {
                        acc[g][half * H + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(x.v[t][p]), as_bf16x8(b1[g]), acc[g][half * H + t], 0, 0, 0);
                        if (p < kXdlLoPieces) acc[g][half * H + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(x.v[t][p]), as_bf16x8(b2[g]), acc[g][half * H + t], 0, 0, 0);
                    }
        };
        // eight half-groups: the A pieces of the next half are requested before the MFMAs of the current one
#define XDL4_HALF(CUR, NXT, J, HALF, NJ, NHALF, PREP)          \
    do {                                                       \
        load_a(NXT, s, NJ, NHALF);                             \
        __builtin_amdgcn_sched_barrier(0);                     \
        PREP;                                                  \
        mfmas(CUR, HALF);                                      \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)
        XDL4_HALF(a0, a1, 0, 0, 0, 1, prep(std::integral_constant<int, 0>{}));
        XDL4_HALF(a1, a0, 0, 1, 1, 0, (void)0);
        XDL4_HALF(a0, a1, 1, 0, 1, 1, prep(std::integral_constant<int, 1>{}));
        XDL4_HALF(a1, a0, 1, 1, 2, 0, (void)0);
        XDL4_HALF(a0, a1, 2, 0, 2, 1, prep(std::integral_constant<int, 2>{}));
        XDL4_HALF(a1, a0, 2, 1, 3, 0, (void)0);
        XDL4_HALF(a0, a1, 3, 0, 3, 1, prep(std::integral_constant<int, 3>{}));
        mfmas(a1, 1);
#undef XDL4_HALF
    }

    // D[m = 4 * row + v][n = i] in acc[g][t][v]; fold the KW waves of each block-column in wave order
    constexpr uint32_t TILE = C * RT * 256; // floats of one block-column's tile
#pragma unroll
    for (int g = 0; g < C; g++)
#pragma unroll
        for (int t = 0; t < RT; t++)
#pragma unroll
            for (int v = 0; v < 4; v++) smem[((bcw * KW + kw) * (C * RT) + g * RT + t) * 256 + v * 64 + lane] = acc[g][t][v];
    __syncthreads();
    uint32_t* const flag = (uint32_t*)(smem + (size_t)kX4Waves * TILE);
    constexpr uint32_t NQ = CB * TILE / 512; // values per thread: the workgroup's CB tiles over 512 threads
    float sum[NQ];
#pragma unroll
    for (uint32_t q = 0; q < NQ; q++) {
        const uint32_t idx = threadIdx.x + q * 512, bc = idx / TILE, e = idx - bc * TILE;
        float v = smem[(bc * KW) * TILE + e];
        for (uint32_t ww = 1; ww < KW; ww++) v += smem[(bc * KW + ww) * TILE + e];
        sum[q] = v;
    }
    if (a.SK > 1) {
        using gf32 = __attribute__((address_space(1))) float;
        using gu32 = __attribute__((address_space(1))) unsigned int;
        float* const mine = a.partial + ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * (CB * TILE); // [tile group][column block][slice]
#pragma unroll
        for (uint32_t q = 0; q < NQ; q++) __hip_atomic_store((gf32*)(mine + threadIdx.x + q * 512), sum[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains before the count
        __syncthreads();
        uint32_t* const cnt = a.counter + blockIdx.y * (gridDim.x / a.SK) + cbx;
        if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != a.SK - 1) return; // not the last slice of this block-column
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // compiler only: the loads below stay below the count
        const float* const base = a.partial + ((uint64_t)blockIdx.y * gridDim.x + (uint64_t)cbx * a.SK) * (CB * TILE);
#pragma unroll
        for (uint32_t q = 0; q < NQ; q++) {
            float v = 0.f;
            for (uint32_t sl = 0; sl < a.SK; sl++) // slice order, not arrival order
                v += __hip_atomic_load((gf32*)(base + (uint64_t)sl * (CB * TILE) + threadIdx.x + q * 512), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sum[q] = v;
        }
        if (threadIdx.x == 0) __hip_atomic_store((gu32*)cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-arm
    }
#pragma unroll
    for (uint32_t q = 0; q < NQ; q++) {
        const uint32_t idx = threadIdx.x + q * 512, bc = idx / TILE, e = idx - bc * TILE, gt = e >> 8, v = (e >> 6) & 3, l = e & 63;
        const uint32_t g = gt / RT, t = gt % RT;
        const uint32_t m = (t0 + t) * 16 + 4 * (l >> 4) + v, n = (((cbx - P.block_begin) * CB + bc) * C + g) * 16 + (l & 15);
        if (m < a.M) P.out[(uint64_t)m * P.out_rs + n] = sum[q] * 16.0f; // the tile carries q/16
    }
}

// ── XDL tile kernel for M <= 32 with the A pieces SHARED through LDS ──────────────────────────────────────────────────
// What bounds qmatmul_xdl2_kernel at M = 32 is not the matrix pipe: a workgroup there owns 16 or 32 columns and all of K, so
// every workgroup pulls the whole A operand (M K 6 B as bf16 pieces, 786 KB at 32 x 4096) out of L2 — 200-540 MB per launch,
// and the launches run at exactly that traffic over ~16 TB/s (o 12.5 us, q/k/v 32 us, gate/up 52 us, down 30 us). Here
//   * a workgroup-column is WAVES scale block-columns (32 columns each, one per wave: 256 columns at 8 waves) x RT = 1 or 2
//     m-tiles (all of M); all waves of a workgroup walk the SAME K steps;
//   * a step's A pieces (RT x 4 kAPieces KB) are fetched from L2 ONCE per workgroup into a double-buffered LDS region (one barrier
//     per step) and every wave reads its MFMA operands from there: A through L2 / (8 x the old form);
//   * each wave prepares B for its own block-column only (every weight still prepared exactly once per launch) and keeps the
//     weights of the next two steps in registers; no cross-wave fold — a wave's accumulators ARE its output tile;
//   * the (workgroup-column, K step) pairs of the launch are ONE list cut into equal runs, one per workgroup (as many
//     workgroups as the chip holds, whatever N is): a run covers the tail of one column and the head of the next at most.
//     A run that is not a whole column publishes its partial tile (write-through stores, drain, one agent-scope add on the
//     column's counter); the LAST arriver sums the column's runs in run order — deterministic — scales and stores.
struct QMM5Args {
    QMM2Part parts[kMaxQmmParts]; // NB2 = block-columns (N / 32) of the part, block_begin = its first workgroup-column
    const uint4* ap;              // split_a_kernel output
    float* partial;               // [workgroup][segment 0 / 1][wave][2 RT 256] f32
    uint32_t* counter;            // one word per workgroup-column, zero between launches
    uint32_t n_parts, M, U, S, run, total; // total = workgroup-columns x S (column, step) pairs, run = pairs per workgroup (<= S)
    uint64_t* trace;                       // trace builds (ZGML_QMM_XDL5_TRACE=1): 8 wall-clock stamps per workgroup and segment
};
#ifdef ZGML_TRACE // (never in product builds: the stamps serialise)
#define X5_STAMP(K) \
    if (a.trace && threadIdx.x == 0) a.trace[(blockIdx.x * 2 + seg) * 8 + (K)] = wall_clock64()
#else
#define X5_STAMP(K) (void)0
#endif

template <int RT, int WAVES, bool NT>
__global__ void __launch_bounds__(WAVES * 64) qmatmul_xdl5_kernel(QMM5Args a) {
    constexpr int C = 2;
    constexpr uint32_t APS = 4 * kAPieces * 64; // uint4 of one m-tile's A pieces per 128-k step: [J][piece][lane]
    constexpr uint32_t T = WAVES * 64, STEP = RT * APS; // uint4 of one step's A pieces: [tile][J][piece][lane]
    constexpr uint32_t NA = (STEP + T - 1) / T;
    constexpr uint32_t WT = C * RT * 256; // floats of one wave's tile
    extern __shared__ uint4 lds_a[];      // [2][STEP], then one flag word
    uint32_t* const flag = (uint32_t*)(lds_a + 2 * STEP);
    const uint32_t lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t row = lane >> 4, i = lane & 15;
    const uint64_t tile_stride = (uint64_t)a.S * APS;
    const uint32_t u_begin = blockIdx.x * a.run, u_end = min(u_begin + a.run, a.total);

    for (uint32_t u0 = u_begin, seg = 0; u0 < u_end; seg++) { // at most two segments (run <= S)
        const uint32_t wgc = u0 / a.S, s_begin = u0 - wgc * a.S, s_end = min(a.S, s_begin + (u_end - u0)), s_last = s_end - 1;
        u0 += s_end - s_begin;
        uint32_t pi = 0;
#pragma unroll
        for (uint32_t t = 1; t < (uint32_t)kMaxQmmParts; t++)
            if (t < a.n_parts && wgc >= a.parts[t].block_begin) pi = t;
        const QMM2Part& P = a.parts[pi];
        const uint32_t cb_raw = (wgc - P.block_begin) * WAVES + w;
        const bool live = cb_raw < P.NB2;               // (a part whose block-columns do not fill its last workgroup-column)
        const uint32_t cb = live ? cb_raw : P.NB2 - 1;  // idle waves shadow the last column: loads stay in bounds
        const uint4* qs = P.qs + (uint64_t)(cb * C) * a.U * 16 + i;
        const uint32_t* scd = (const uint32_t*)P.sc + (uint64_t)cb * a.U * 16 + i;

        // two accumulators per (column group, m-tile), one per B piece: with the tiles taken one after the other (their A
        // registers are refilled in between) that makes FOUR independent MFMA chains, each touched every fourth MFMA — a wave
        // issues in order, and a dependent 16-cycle MFMA right behind its producer stalls the VALU work queued behind it too
        // (measured with two chains: MFMA pipe 45 % busy, 41 % of the wave cycles waiting to issue)
        mfma_f4 acc[C][RT], acc2[C][RT];
#pragma unroll
        for (int g = 0; g < C; g++)
#pragma unroll
            for (int t = 0; t < RT; t++) acc[g][t] = acc2[g][t] = mfma_f4{0.f, 0.f, 0.f, 0.f};

        struct WStep { // one K step of this lane: its unit's nibbles for the two column groups, its dword of the unit's scales
            uint4 wq[C];
            uint32_t sd;
        };
        struct BP {
            uint4 b1[C], b2[C];
        };
        auto load_w = [&](WStep& b, uint32_t s) {
            const uint32_t u = min(4 * s + row, a.U - 1); // units past the end: A is zero there
#pragma unroll
            for (int g = 0; g < C; g++) b.wq[g] = wload<NT>(qs + ((uint64_t)g * a.U + u) * 16);
            b.sd = scd[(uint64_t)u * 16];
        };
        // (named registers, not an array: hipcc left a uint4[NA] in scratch memory, whose reloads wait for every older load)
        uint4 ar0 = {}, ar1 = {}, ar2 = {};
        static_assert(NA <= 3, "A pieces of a step per thread");
        auto a_src = [&](uint32_t s, uint32_t q) -> const uint4* {
            const uint32_t idx = threadIdx.x + q * T;
            return a.ap + (uint64_t)(idx / APS) * tile_stride + (uint64_t)s * APS + idx % APS;
        };
        auto fetch_a = [&](uint32_t s) {
            ar0 = *a_src(s, 0);
            if (NA > 1 && (STEP >= 2 * T || threadIdx.x + T < STEP)) ar1 = *a_src(s, 1);
            if (NA > 2 && (STEP >= 3 * T || threadIdx.x + 2 * T < STEP)) ar2 = *a_src(s, 2);
        };
        auto park_a = [&](uint32_t buf) {
            lds_a[buf * STEP + threadIdx.x] = ar0;
            if (NA > 1 && (STEP >= 2 * T || threadIdx.x + T < STEP)) lds_a[buf * STEP + threadIdx.x + T] = ar1;
            if (NA > 2 && (STEP >= 3 * T || threadIdx.x + 2 * T < STEP)) lds_a[buf * STEP + threadIdx.x + 2 * T] = ar2;
        };
        uint4 xa[RT][kAPieces]; // the A pieces of the current 8-k group, refilled tile by tile after their last use
        auto read_a = [&](uint32_t buf, int J, int t) {
#pragma unroll
            for (int p = 0; p < kAPieces; p++) xa[t][p] = lds_a[buf * STEP + ((t * 4 + J) * kAPieces + p) * 64 + lane];
        };
        auto mfmas = [&](const BP& o, int t) {
#pragma unroll
            for (int p = 0; p < kAPieces; p++) {
#pragma unroll
                for (int g = 0; g < C; g++) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(xa[t][p]), as_bf16x8(o.b1[g]), acc[g][t], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < C; g++)
                    if (p < kXdlLoPieces) acc2[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(xa[t][p]), as_bf16x8(o.b2[g]), acc2[g][t], 0, 0, 0);
            }
        };
        auto scale_bits = [](const WStep& b, int& lo, int& hi) {
            const __half2 hh = __builtin_bit_cast(__half2, b.sd);
            lo = __float_as_int(__half2float(hh.x)), hi = __float_as_int(__half2float(hh.y)); // scales of k_local i, 16 + i
        };

        WStep wa, wb, wc; // the weights of steps s, s + 1, s + 2: three names rotating (a register copy would wait for the load)
        BP p0, p1;
        int sl, sh;
        X5_STAMP(0);
        load_w(wa, s_begin);
        fetch_a(s_begin);
        load_w(wb, min(s_begin + 1, s_last));
        if (seg) __syncthreads(); // the previous segment's last reads of the A buffers
        park_a(0);
        __syncthreads();
        scale_bits(wa, sl, sh);
        read_a(0, 0, 0);
        if (RT > 1) read_a(0, 0, RT - 1);
        xdl_prep_b<0, C>(wa.wq, sl, sh, p0.b1, p0.b2);
        X5_STAMP(1);
        __builtin_amdgcn_sched_barrier(0);
        // one stage: the MFMAs of group J, tile by tile, each tile's A registers refilled for group JN = J + 1 once its MFMAs
        // are issued, and the B preparation of group JN. The order inside a stage is the compiler's: matrix and vector
        // instructions of a SIMD do not overlap (tools/exp/stage_parts.hip: 24 MFMAs + their B preparation cost the SUM, 32.0
        // cycles per MFMA on a busy chip; forcing 1 MFMA : 3 VALU with sched_group_barrier: 34.9)
        constexpr bool kX5Interleave = false;
#define XDL5_STAGE(CUR, NXT, BUF_N, JN, WSRC)                                          \
    do {                                                                               \
        xdl_prep_b<JN, C>(WSRC.wq, sl, sh, NXT.b1, NXT.b2);                            \
        mfmas(CUR, 0);                                                                 \
        read_a(BUF_N, JN, 0);                                                          \
        if (RT > 1) {                                                                  \
            mfmas(CUR, RT - 1);                                                        \
            read_a(BUF_N, JN, RT - 1);                                                 \
        }                                                                              \
        if (kX5Interleave) {                                                           \
            _Pragma("unroll") for (int t_ = 0; t_ < RT; t_++) {                        \
                _Pragma("unroll") for (int k_ = 0; k_ < (kAPieces + kXdlLoPieces) * C; k_++) { \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* 1 MFMA */    \
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); /* 3 VALU */    \
                }                                                                      \
                __builtin_amdgcn_sched_group_barrier(0x100, kAPieces, 0); /* tile refill */   \
            }                                                                          \
        }                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                             \
    } while (0)
        // one K step: A of step s + 1 and the weights of step s + 2 requested first (in that order: the A pieces are waited
        // for below with the weight loads still in flight), three stages, the hand-over of the A buffers, the fourth stage
#define XDL5_STEP(S_, CUR, NXT, LD)                                                                   \
    do {                                                                                              \
        fetch_a(min((S_) + 1, s_last)); /* clamped: the last prefetches re-read live lines */          \
        load_w(LD, min((S_) + 2, s_last));                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        XDL5_STAGE(p0, p1, buf, 1, CUR);                                                              \
        XDL5_STAGE(p1, p0, buf, 2, CUR);                                                              \
        XDL5_STAGE(p0, p1, buf, 3, CUR);                                                              \
        park_a(buf ^ 1); /* every wave is past its last read of that buffer since the previous barrier */ \
        __syncthreads();                                                                              \
        scale_bits(NXT, sl, sh);                                                                      \
        XDL5_STAGE(p1, p0, buf ^ 1, 0, NXT); /* (after the last step: a wasted preparation) */        \
        buf ^= 1;                                                                                     \
    } while (0)
        uint32_t buf = 0;
        for (uint32_t s = s_begin;; s += 3) {
            XDL5_STEP(s, wa, wb, wc);
            if (s + 1 >= s_end) break;
            XDL5_STEP(s + 1, wb, wc, wa);
            if (s + 2 >= s_end) break;
            XDL5_STEP(s + 2, wc, wa, wb);
            if (s + 3 >= s_end) break;
        }
#undef XDL5_STEP
#undef XDL5_STAGE
        X5_STAMP(2);
#pragma unroll
        for (int g = 0; g < C; g++)
#pragma unroll
            for (int t = 0; t < RT; t++)
#pragma unroll
                for (int v = 0; v < 4; v++) acc[g][t][v] += acc2[g][t][v];
        // D[m = 4 * row + v][n = i] in acc[g][t][v]
        if (s_end - s_begin != a.S) { // part of a column: publish, count, the last arriver sums the column's runs in run order
            using gf32 = __attribute__((address_space(1))) float;
            using gu32 = __attribute__((address_space(1))) unsigned int;
            float* const mine = a.partial + (((uint64_t)blockIdx.x * 2 + seg) * WAVES + w) * WT + lane;
#pragma unroll
            for (int g = 0; g < C; g++)
#pragma unroll
                for (int t = 0; t < RT; t++)
#pragma unroll
                    for (int v = 0; v < 4; v++) __hip_atomic_store((gf32*)(mine + ((g * RT + t) * 4 + v) * 64), acc[g][t][v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains before the count
            X5_STAMP(3);
            __syncthreads();
            const uint32_t b_lo = (wgc * a.S) / a.run, b_hi = (wgc * a.S + a.S - 1) / a.run; // the workgroups whose runs touch this column
            uint32_t* const cnt = a.counter + wgc;
            if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const bool last = *flag == b_hi - b_lo;
            __syncthreads(); // (the flag word is reused by the next segment)
            X5_STAMP(4);
            if (!last) continue;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // compiler only: the loads below stay below the count
#pragma unroll
            for (int g = 0; g < C; g++)
#pragma unroll
                for (int t = 0; t < RT; t++)
#pragma unroll
                    for (int v = 0; v < 4; v++) acc[g][t][v] = 0.f;
            // run order, not arrival order; the loads of kBatch runs are in flight together (one run at a time cost a memory
            // round trip per run: 12 us of a 26 us launch at 16 runs per column)
            constexpr uint32_t kBatch = 8;
            for (uint32_t b0 = b_lo; b0 <= b_hi; b0 += kBatch) {
                float pv[kBatch][C * RT * 4];
#pragma unroll
                for (uint32_t q = 0; q < kBatch; q++) {
                    const uint32_t b = min(b0 + q, b_hi);              // (past the end: a live line again, dropped below)
                    const uint32_t bseg = (b * a.run) / a.S != wgc;    // the column is that workgroup's second one
                    const float* const src = a.partial + (((uint64_t)b * 2 + bseg) * WAVES + w) * WT + lane;
#pragma unroll
                    for (int e = 0; e < C * RT * 4; e++) pv[q][e] = __hip_atomic_load((gf32*)(src + e * 64), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (uint32_t q = 0; q < kBatch; q++) {
                    const bool in = b0 + q <= b_hi;
#pragma unroll
                    for (int g = 0; g < C; g++)
#pragma unroll
                        for (int t = 0; t < RT; t++)
#pragma unroll
                            for (int v = 0; v < 4; v++) acc[g][t][v] += in ? pv[q][(g * RT + t) * 4 + v] : 0.f;
                }
            }
            if (threadIdx.x == 0) __hip_atomic_store((gu32*)cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-arm
        }
        if (live) {
#pragma unroll
            for (int g = 0; g < C; g++)
#pragma unroll
                for (int t = 0; t < RT; t++)
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const uint32_t m = t * 16 + 4 * row + v, n = (cb * C + g) * 16 + i;
                        if (m < a.M) P.out[(uint64_t)m * P.out_rs + n] = acc[g][t][v] * 16.0f; // the tile carries q/16
                    }
        }
#ifdef ZGML_TRACE
        if (a.trace) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        X5_STAMP(5);
#endif
    }
}

#ifdef ZGML_TRACE
// ── XDL tile kernel for M > 32 with the A pieces SHARED through LDS (round 4) ─────────────────────────────────────────────────────────
// qmatmul_xdl4_kernel gives every 32-column workgroup its own pass over A (128 x 4096 rows: 2 MB of pieces per workgroup, 1.4 GB
// per gate / up launch through L2): q/k/v and gate/up at M = 128 run at 2.4-2.8 x their MFMA time (57.6 / 87.6 us against 20.5 /
// 36.7). This is the M <= 32 form above (qmatmul_xdl5_kernel: 8 waves = 8 block-columns = 256 columns per workgroup-column, one
// list of (workgroup-column, K step) pairs cut into equal runs, partial tiles published and summed in run order by the last
// arriver) at RT = 4 or 8 m-tiles per workgroup, with what the larger tile needs:
//   * a step's A pieces (RT x 8 KB, double-buffered: 128 KB at RT = 8) come through registers in TWO halves, each fetched one
//     stage before it is parked (a whole step's 8 uint4 per thread would not fit beside the accumulators);
//   * the m-tiles are taken in PAIRS: the pair's A registers (2 tiles x 2 pieces) are a ring of two slots, the next pair is read
//     from LDS while the current one multiplies (16 MFMAs over four accumulators: every accumulator is touched each fourth MFMA);
//   * one accumulator per (column group, m-tile): 16 independent chains, no second set for the low B piece;
//   * tile groups of RT m-tiles (M > 16 RT) are blockIdx.y: every group has its own partial tiles and counters.
// MEASURED SLOWER than the K-split kernel it was meant to replace (profiles/r04_exp_xdl7.txt: 4096 x 22016 at M = 128 114-116 us
// against 84-87, 4096 x 12288 89-93 against 58; M = 64: 58.6 / 43.8 against 49.5 / 34.0), parity green. The counters say why the
// A traffic was the wrong target: 5.64 M MFMAs (90 M pipe cycles) + 18.3 M vector instructions on 1024 SIMDs are 159 k cycles per
// SIMD when matrix and vector work serialise — which they do (tools/exp/stage_parts.hip) — i.e. 64 us at best for this
// instruction mix, 48 us for the K-split kernel's (1.4 vector instructions per MFMA there, 3.25 here: ring copies, the staging
// stores, address arithmetic); both kernels sit at ~57 % of that bound. Diagnostics build only (ZGML_QMM_XDL7=1).
struct QMM7Args {
    QMM2Part parts[kMaxQmmParts]; // NB2 = block-columns (N / 32) of the part, block_begin = its first workgroup-column
    const uint4* ap;              // split_a_kernel output (tiles padded to a multiple of RT)
    float* partial;               // [tile group][workgroup][segment 0 / 1][wave][2 RT 256] f32
    uint32_t* counter;            // [tile group][workgroup-column], zero between launches
    uint32_t n_parts, M, U, S, run, total, wg_cols;
};
constexpr uint32_t kX7Waves = 8;

template <int RT, bool NT>
__global__ void __launch_bounds__(kX7Waves * 64) qmatmul_xdl7_kernel(QMM7Args a) {
    constexpr int C = 2, WAVES = kX7Waves, NP = RT / 2; // NP tile pairs
    static_assert(RT == 4 || RT == 8, "m-tiles per workgroup");
    constexpr uint32_t APS = 4 * kAPieces * 64;          // uint4 of one m-tile's A pieces per 128-k step: [J][piece][lane]
    constexpr uint32_t T = WAVES * 64, STEP = RT * APS;  // uint4 of one step's A pieces: [tile][J][piece][lane]
    constexpr uint32_t NA = STEP / T, NH = NA / 2;       // uint4 per thread and step, per half
    static_assert(STEP % (2 * T) == 0, "two halves of whole rounds");
    constexpr uint32_t WT = C * RT * 256;                // floats of one wave's tile
    extern __shared__ uint4 lds_a[];                     // [2][STEP], then one flag word
    uint32_t* const flag = (uint32_t*)(lds_a + 2 * STEP);
    const uint32_t lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t row = lane >> 4, i = lane & 15;
    const uint64_t tile_stride = (uint64_t)a.S * APS;
    const uint32_t tg = blockIdx.y;
    const uint4* const ap_g = a.ap + (uint64_t)tg * RT * tile_stride;
    const uint32_t u_begin = blockIdx.x * a.run, u_end = min(u_begin + a.run, a.total);

    for (uint32_t u0 = u_begin, seg = 0; u0 < u_end; seg++) { // at most two segments (run <= S)
        const uint32_t wgc = u0 / a.S, s_begin = u0 - wgc * a.S, s_end = min(a.S, s_begin + (u_end - u0)), s_last = s_end - 1;
        u0 += s_end - s_begin;
        uint32_t pi = 0;
#pragma unroll
        for (uint32_t t = 1; t < (uint32_t)kMaxQmmParts; t++)
            if (t < a.n_parts && wgc >= a.parts[t].block_begin) pi = t;
        const QMM2Part& P = a.parts[pi];
        const uint32_t cb_raw = (wgc - P.block_begin) * WAVES + w;
        const bool live = cb_raw < P.NB2;
        const uint32_t cb = live ? cb_raw : P.NB2 - 1; // idle waves shadow the last column: loads stay in bounds
        const uint4* qs = P.qs + (uint64_t)(cb * C) * a.U * 16 + i;
        const uint32_t* scd = (const uint32_t*)P.sc + (uint64_t)cb * a.U * 16 + i;

        mfma_f4 acc[C][RT];
#pragma unroll
        for (int g = 0; g < C; g++)
#pragma unroll
            for (int t = 0; t < RT; t++) acc[g][t] = mfma_f4{0.f, 0.f, 0.f, 0.f};

        struct WStep {
            uint4 wq[C];
            uint32_t sd;
        };
        struct BP {
            uint4 b1[C], b2[C];
        };
        auto load_w = [&](WStep& b, uint32_t s) {
            const uint32_t u = min(4 * s + row, a.U - 1); // units past the end: A is zero there
#pragma unroll
            for (int g = 0; g < C; g++) b.wq[g] = wload<NT>(qs + ((uint64_t)g * a.U + u) * 16);
            b.sd = scd[(uint64_t)u * 16];
        };
        uint4 ah0 = {}, ah1 = {}, ah2 = {}, ah3 = {}; // one half of a step's A pieces (named registers: see the M <= 32 form)
        static_assert(NH <= 4, "A pieces of half a step per thread");
        auto a_src = [&](uint32_t s, uint32_t q) -> const uint4* {
            const uint32_t idx = threadIdx.x + q * T;
            return ap_g + (uint64_t)(idx / APS) * tile_stride + (uint64_t)s * APS + idx % APS;
        };
        auto fetch_half = [&](uint32_t s, uint32_t h) {
            ah0 = *a_src(s, h * NH);
            if (NH > 1) ah1 = *a_src(s, h * NH + 1);
            if (NH > 2) ah2 = *a_src(s, h * NH + 2);
            if (NH > 3) ah3 = *a_src(s, h * NH + 3);
        };
        auto park_half = [&](uint32_t buf, uint32_t h) {
            uint4* const d = lds_a + buf * STEP + h * NH * T + threadIdx.x;
            d[0] = ah0;
            if (NH > 1) d[T] = ah1;
            if (NH > 2) d[2 * T] = ah2;
            if (NH > 3) d[3 * T] = ah3;
        };
        uint4 xa[2][2][kAPieces]; // ring of two tile PAIRS: [slot][tile of the pair][piece]
        auto read_pair = [&](uint32_t buf, int J, int pr, int slot) {
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int p = 0; p < kAPieces; p++) xa[slot][tt][p] = lds_a[buf * STEP + (((2 * pr + tt) * 4 + J) * kAPieces + p) * 64 + lane];
        };
        auto mfma_pair = [&](const BP& o, int pr, int slot) { // 16 MFMAs over four accumulators
#pragma unroll
            for (int p = 0; p < kAPieces; p++) {
#pragma unroll
                for (int tt = 0; tt < 2; tt++)
#pragma unroll
                    for (int g = 0; g < C; g++)
                        acc[g][2 * pr + tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(xa[slot][tt][p]), as_bf16x8(o.b1[g]), acc[g][2 * pr + tt], 0, 0, 0);
#pragma unroll
                for (int tt = 0; tt < 2; tt++)
#pragma unroll
                    for (int g = 0; g < C; g++)
                        if (p < kXdlLoPieces)
                            acc[g][2 * pr + tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(xa[slot][tt][p]), as_bf16x8(o.b2[g]), acc[g][2 * pr + tt], 0, 0, 0);
            }
        };
        auto scale_bits = [](const WStep& b, int& lo, int& hi) {
            const __half2 hh = __builtin_bit_cast(__half2, b.sd);
            lo = __float_as_int(__half2float(hh.x)), hi = __float_as_int(__half2float(hh.y)); // scales of k_local i, 16 + i
        };

        WStep wa, wb, wc; // the weights of steps s, s + 1, s + 2: three names rotating
        BP p0, p1;
        int sl, sh;
        // prologue: the first step's A pieces (both halves), the weights of the first two steps
        load_w(wa, s_begin);
        fetch_half(s_begin, 0);
        if (seg) __syncthreads(); // the previous segment's last reads of the A buffers
        park_half(0, 0);
        fetch_half(s_begin, 1);
        load_w(wb, min(s_begin + 1, s_last));
        park_half(0, 1);
        __syncthreads();
        scale_bits(wa, sl, sh);
        xdl_prep_b<0, C>(wa.wq, sl, sh, p0.b1, p0.b2);
        read_pair(0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // one stage = the 8-k group J of the step: the B pieces of group J + 1 are prepared (vector work beside the MFMAs), the tile
        // pairs multiply one after the other, each one's successor (the next pair of this group, or the first pair of the next
        // group / step) read from LDS meanwhile
#define XDL7_STAGE(CUR, NXT, BUF, J, JN, WSRC, CHAIN)                                  \
    do {                                                                               \
        xdl_prep_b<JN, C>(WSRC.wq, sl, sh, NXT.b1, NXT.b2);                            \
        __builtin_amdgcn_sched_barrier(0);                                             \
        _Pragma("unroll") for (int pr_ = 0; pr_ < NP; pr_++) {                         \
            if (pr_ + 1 < NP)                                                          \
                read_pair(BUF, J, pr_ + 1, (pr_ + 1) & 1);                             \
            else if (CHAIN)                                                            \
                read_pair(BUF, JN, 0, NP & 1);                                         \
            __builtin_amdgcn_sched_barrier(0); /* (hipcc otherwise sinks the reads to their first use: one LDS round trip per pair exposed) */ \
            mfma_pair(CUR, pr_, pr_ & 1);                                              \
            __builtin_amdgcn_sched_barrier(0);                                         \
        }                                                                              \
    } while (0)
        static_assert((NP & 1) == 0, "the ring's slot of a group's first pair is 0 in every stage");
        // one K step: the first half of A(s + 1) and the weights of step s + 2 requested first; half 0 parked and half 1 requested
        // behind the first stage, half 1 parked behind the third. EVERY LDS read of the step (the fourth stage's pairs included) lies
        // in front of the barrier: the next step parks into this buffer right after its first stage. The next step's first pair is
        // read behind the barrier (a bubble of one LDS round trip per step and wave; the SIMD's other wave fills it)
#define XDL7_STEP(S_, CUR, NXT, LD)                                                                   \
    do {                                                                                              \
        const uint32_t sn_ = min((S_) + 1, s_last); /* clamped: the last prefetches re-read live lines */ \
        fetch_half(sn_, 0);                                                                           \
        load_w(LD, min((S_) + 2, s_last));                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        XDL7_STAGE(p0, p1, buf, 0, 1, CUR, true);                                                     \
        park_half(buf ^ 1, 0); /* every wave is past its last read of that buffer since the previous barrier */ \
        fetch_half(sn_, 1);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        XDL7_STAGE(p1, p0, buf, 1, 2, CUR, true);                                                     \
        XDL7_STAGE(p0, p1, buf, 2, 3, CUR, true);                                                     \
        park_half(buf ^ 1, 1);                                                                        \
        scale_bits(NXT, sl, sh);                                                                      \
        XDL7_STAGE(p1, p0, buf, 3, 0, NXT, false); /* (after the last step: a wasted preparation) */   \
        __syncthreads();                                                                              \
        buf ^= 1;                                                                                     \
        read_pair(buf, 0, 0, 0);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
        uint32_t buf = 0;
        for (uint32_t s = s_begin;; s += 3) {
            XDL7_STEP(s, wa, wb, wc);
            if (s + 1 >= s_end) break;
            XDL7_STEP(s + 1, wb, wc, wa);
            if (s + 2 >= s_end) break;
            XDL7_STEP(s + 2, wc, wa, wb);
            if (s + 3 >= s_end) break;
        }
#undef XDL7_STEP
#undef XDL7_STAGE
        // D[m = 4 * row + v][n = i] in acc[g][t][v]
        if (s_end - s_begin != a.S) { // part of a column: publish, count, the last arriver sums the column's runs in run order
            using gf32 = __attribute__((address_space(1))) float;
            using gu32 = __attribute__((address_space(1))) unsigned int;
            float* const part_g = a.partial + (uint64_t)tg * gridDim.x * (2 * WAVES * WT);
            float* const mine = part_g + (((uint64_t)blockIdx.x * 2 + seg) * WAVES + w) * WT + lane;
#pragma unroll
            for (int g = 0; g < C; g++)
#pragma unroll
                for (int t = 0; t < RT; t++)
#pragma unroll
                    for (int v = 0; v < 4; v++) __hip_atomic_store((gf32*)(mine + ((g * RT + t) * 4 + v) * 64), acc[g][t][v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains before the count
            __syncthreads();
            const uint32_t b_lo = (wgc * a.S) / a.run, b_hi = (wgc * a.S + a.S - 1) / a.run; // the workgroups whose runs touch this column
            uint32_t* const cnt = a.counter + tg * a.wg_cols + wgc;
            if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const bool last = *flag == b_hi - b_lo;
            __syncthreads(); // (the flag word is reused by the next segment)
            if (!last) continue;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // compiler only: the loads below stay below the count
#pragma unroll
            for (int g = 0; g < C; g++)
#pragma unroll
                for (int t = 0; t < RT; t++)
#pragma unroll
                    for (int v = 0; v < 4; v++) acc[g][t][v] = 0.f;
            // run order, not arrival order; the loads of kBatch runs are in flight together
            constexpr uint32_t kBatch = RT == 8 ? 2 : 4;
            for (uint32_t b0 = b_lo; b0 <= b_hi; b0 += kBatch) {
                float pv[kBatch][C * RT * 4];
#pragma unroll
                for (uint32_t q = 0; q < kBatch; q++) {
                    const uint32_t b = min(b0 + q, b_hi);           // (past the end: a live line again, dropped below)
                    const uint32_t bseg = (b * a.run) / a.S != wgc; // the column is that workgroup's second one
                    const float* const src = part_g + (((uint64_t)b * 2 + bseg) * WAVES + w) * WT + lane;
#pragma unroll
                    for (int e = 0; e < C * RT * 4; e++) pv[q][e] = __hip_atomic_load((gf32*)(src + e * 64), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (uint32_t q = 0; q < kBatch; q++) {
                    const bool in = b0 + q <= b_hi;
#pragma unroll
                    for (int g = 0; g < C; g++)
#pragma unroll
                        for (int t = 0; t < RT; t++)
#pragma unroll
                            for (int v = 0; v < 4; v++) acc[g][t][v] += in ? pv[q][(g * RT + t) * 4 + v] : 0.f;
                }
            }
            if (threadIdx.x == 0) __hip_atomic_store((gu32*)cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-arm
        }
        if (live) {
#pragma unroll
            for (int g = 0; g < C; g++)
#pragma unroll
                for (int t = 0; t < RT; t++)
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const uint32_t m = (tg * RT + t) * 16 + 4 * row + v, n = (cb * C + g) * 16 + i;
                        if (m < a.M) P.out[(uint64_t)m * P.out_rs + n] = acc[g][t][v] * 16.0f; // the tile carries q/16
                    }
        }
    }
}

#endif // ZGML_TRACE (qmatmul_xdl7_kernel)

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

} // namespace

// Scratch of one M > 1 launch: the pre-split A operand of the second XDL form (split_a_kernel): per
// 16-row tile and 128-k step 4 * kAPieces KB (4 groups x kAPieces pieces x 64 lanes x 16 B). Launches are serialised on one
// stream, so one block serves all of a program's quantized matmuls.
static bool xdl2_enabled() {
    static const bool on = !(getenv("ZGML_QMM_XDL2") && atoi(getenv("ZGML_QMM_XDL2")) == 0);
    return on;
}
static bool xdl2_applies(const QWeightDev& w, uint32_t M) { // K <= 12288: a wave's weights (<= 12 steps) are preloaded into LDS
    return M > 1 && w.format == QW_Q4 && w.scale_f16 && xdl2_enabled() && (w.KC + 3) / 4 <= 8 * 12;
}
// scratch of an M > 1 launch: [A pieces][split-K partial tiles of the M > 32 form]; the partial region is sized for a
// grouped launch of kMaxQmmParts weights of this shape (the runtime allocates the maximum over a program's weights, so
// the widest weight bounds every group it is part of). The arrival counters of the K split live in the kQmmScratchHead
// bytes in FRONT of the scratch pointer (kernels.h): a fixed place, so launches of different shapes that share the block
// can never scribble partial tiles or A pieces over each other's counters. A launch only splits K when it has fewer
// (tile group, block-column) pairs than counter words.
static uint32_t xdl_tile_pad(uint32_t M) { return M > 64 ? 8 : (M > 32 ? 4 : (M > 16 ? 2 : 1)); } // m-tiles per workgroup
static uint64_t xdl_a_bytes(const QWeightDev& w, uint32_t M) {
    const uint64_t S = (w.KC + 3) / 4, tiles = (M + 15) / 16, R = xdl_tile_pad(M);
    return (tiles + R - 1) / R * R * S * (4 * kAPieces) * 1024;
}
static uint64_t xdl4_partial_bytes(const QWeightDev& w, uint32_t M) { // one part, at most 4 K slices
    const uint64_t RT = xdl_tile_pad(M), groups = ((M + 15) / 16 + RT - 1) / RT;
    return groups * (w.N / 32) * 4 * (2 * RT * 256 * 4);
}
// M <= 32 (qmatmul_xdl5_kernel): workgroups of the launch = CUs x this; two partial tiles of 8 waves each per workgroup
constexpr uint32_t kX5Waves = 8;
static uint32_t xdl5_wgs() {
    static const int per_cu = getenv("ZGML_QMM_XDL5_WGS_PER_CU") ? std::max(1, atoi(getenv("ZGML_QMM_XDL5_WGS_PER_CU"))) : 1;
    return device_cus() * (uint32_t)per_cu;
}
static bool xdl5_enabled() {
    static const bool on = !(getenv("ZGML_QMM_XDL5") && atoi(getenv("ZGML_QMM_XDL5")) == 0);
    return on;
}
static uint64_t xdl5_partial_bytes(uint32_t M) {
    if (M > 32 || !xdl5_enabled()) return 0;
    return (uint64_t)xdl5_wgs() * 2 * kX5Waves * (2 * xdl_tile_pad(M) * 256) * sizeof(float);
}
// M > 32 (qmatmul_xdl7_kernel, diagnostics build only, opt-in): tile groups x workgroups per group <= max(workgroups of the launch, groups)
static bool xdl7_enabled() {
#ifdef ZGML_TRACE
    static const bool on = getenv("ZGML_QMM_XDL7") && atoi(getenv("ZGML_QMM_XDL7")) != 0;
    return on;
#else
    return false;
#endif
}
static uint64_t xdl7_partial_bytes(uint32_t M) {
    if (M <= 32 || !xdl7_enabled()) return 0;
    const uint64_t RT = xdl_tile_pad(M), groups = ((M + 15) / 16 + RT - 1) / RT;
    return std::max<uint64_t>(xdl5_wgs(), groups) * 2 * 8 * (2 * RT * 256) * sizeof(float);
}
uint64_t qmatmul_scratch_bytes(const QWeightDev& w, uint32_t M) {
    if (!xdl2_applies(w, M)) return 0;
    return xdl_a_bytes(w, M) + std::max({(uint64_t)kMaxQmmParts * xdl4_partial_bytes(w, M), xdl5_partial_bytes(M), xdl7_partial_bytes(M)});
}

namespace {

using TileFn = void (*)(QMMArgs);
template <typename ST, bool Q4, bool NT, bool XDL>
TileFn pick_tile_nt(bool two, bool xvec) {
    if (two) return xvec ? qmatmul_tile_kernel<ST, Q4, 2, true, NT, XDL> : qmatmul_tile_kernel<ST, Q4, 2, false, NT, XDL>;
    return xvec ? qmatmul_tile_kernel<ST, Q4, 1, true, NT, XDL> : qmatmul_tile_kernel<ST, Q4, 1, false, NT, XDL>;
}
template <typename ST, bool Q4>
TileFn pick_tile(bool two, bool xvec, bool nt) {
#ifdef ZGML_TRACE // diagnostics build only: ZGML_QMM_XDL=0 keeps the contraction on the f32 MFMA (same results up to rounding order; slower)
    static const bool xdl = !(getenv("ZGML_QMM_XDL") && atoi(getenv("ZGML_QMM_XDL")) == 0);
    if (!xdl) return nt ? pick_tile_nt<ST, Q4, true, false>(two, xvec) : pick_tile_nt<ST, Q4, false, false>(two, xvec);
#endif
    return nt ? pick_tile_nt<ST, Q4, true, true>(two, xvec) : pick_tile_nt<ST, Q4, false, true>(two, xvec);
}

template <int R, bool NT>
void launch_xdl2_rg(hipStream_t s, const QMM2Args& a, uint32_t G, dim3 grid, uint32_t waves, size_t lds) {
    using Fn2 = void (*)(QMM2Args);
    const Fn2 fn = G == 2 ? (Fn2)qmatmul_xdl2_kernel<R, 2, NT> : (Fn2)qmatmul_xdl2_kernel<R, 1, NT>;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(fn, grid, dim3(waves * 64), lds, s, a);
}

// second XDL form: split x once, then the B-scaled tile kernel (Q4_0, f16 scales); n >= 1 weights that
// read the same rows share one launch (blockIdx.x ranges)
// M > 32: RT = 4 or 8 m-tiles per workgroup, one scale block-column wide, K split to fill the chip
void launch_xdl4(hipStream_t s, const QWeightDev* w, const QMatmulParams* p, uint32_t n, float* scratch, uint32_t S, uint32_t RT, uint32_t tiles) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    }
    QMM4Args a{};
    uint32_t blocks = 0;
    // block-columns per workgroup (kernel template CB): pairing waves on neighbouring block-columns so that the second
    // reader of a step's A pieces hits L1 was measured SLOWER (128 x 4096 x 4096: 47.4 vs 32.5 us) — one per workgroup
    constexpr uint32_t CB = 1;
    for (uint32_t t = 0; t < n; t++) {
        a.parts[t] = {(const uint4*)w[t].qs, (const uint4*)w[t].sc, p[t].dst, p[t].dst_rs, p[t].N / 32 / CB, blocks};
        blocks += p[t].N / 32 / CB;
    }
    const uint32_t groups = tiles / RT, KW = kX4Waves / CB;
    // K split: at least one workgroup per CU, at most 4 slices (the scratch regions are sized for 4), whole steps per wave
    static const int env_sk = getenv("ZGML_QMM_XDL4_SK") ? atoi(getenv("ZGML_QMM_XDL4_SK")) : 0;
    uint32_t SK = env_sk > 0 ? (uint32_t)env_sk : cdiv((uint32_t)n_cu, blocks * groups);
    SK = std::max(1u, std::min({SK, 4u, cdiv(S, KW)}));
    if ((uint64_t)blocks * groups * sizeof(uint32_t) > kQmmCounterBytes) SK = 1; // one counter word per (tile group, block-column)
    const uint32_t steps_per_slice = cdiv(S, SK);
    SK = cdiv(S, steps_per_slice);
    const uint64_t a_bytes = xdl_a_bytes(w[0], p[0].M);
    a.ap = (const uint4*)scratch;
    a.partial = (float*)((char*)scratch + a_bytes);
    a.counter = (uint32_t*)((char*)scratch - kQmmScratchHead);
    a.n_parts = n, a.M = p[0].M, a.U = w[0].KC, a.S = S, a.SK = SK, a.steps_per_slice = steps_per_slice;
    const dim3 grid(blocks * SK, groups);
    const size_t lds = (size_t)kX4Waves * 2 * RT * 256 * sizeof(float) + 64;
    const bool nt = w[0].stream_nt != 0 && groups == 1;
    using Fn4 = void (*)(QMM4Args);
    const Fn4 fn = RT == 8   ? (nt ? (Fn4)qmatmul_xdl4_kernel<8, 1, true> : (Fn4)qmatmul_xdl4_kernel<8, 1, false>)
                   : RT == 4 ? (nt ? (Fn4)qmatmul_xdl4_kernel<4, 1, true> : (Fn4)qmatmul_xdl4_kernel<4, 1, false>)
                             : (nt ? (Fn4)qmatmul_xdl4_kernel<2, 1, true> : (Fn4)qmatmul_xdl4_kernel<2, 1, false>);
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(fn, grid, dim3(kX4Waves * 64), lds, s, a);
}

// M <= 32: the shared-A form. false = not applicable (more workgroup-columns than counter words)
bool launch_xdl5(hipStream_t s, const QWeightDev* w, const QMatmulParams* p, uint32_t n, float* scratch, uint32_t S, uint32_t RT) {
    QMM5Args a{};
    uint32_t wg_cols = 0;
    for (uint32_t t = 0; t < n; t++) {
        a.parts[t] = {(const uint4*)w[t].qs, (const uint4*)w[t].sc, p[t].dst, p[t].dst_rs, p[t].N / 32, wg_cols};
        wg_cols += cdiv(p[t].N / 32, kX5Waves);
    }
    if ((uint64_t)wg_cols * sizeof(uint32_t) > kQmmCounterBytes) return false;
    // narrow outputs stay with qmatmul_xdl2_kernel: every run of a column costs the column's fan-in one more device-scope
    // atomic on one word (measured 2.5 + 0.6 us x runs per column: 12.5 us at 4096 x 4096, 16 runs), and below ~40
    // workgroup-columns (N < 10240) that outweighs the A traffic saved (4096 x 8192: 27.8 against 20.9 us; x 12288: 32.1 against 33.9)
    static const int min_cols = getenv("ZGML_QMM_XDL5_MIN_COLS") ? atoi(getenv("ZGML_QMM_XDL5_MIN_COLS")) : 40;
    if ((int)wg_cols < min_cols) return false;
    static const int min_run = getenv("ZGML_QMM_XDL5_MIN_RUN") ? std::max(1, atoi(getenv("ZGML_QMM_XDL5_MIN_RUN"))) : 2;
    const uint32_t total = wg_cols * S;
    const uint32_t run = std::min(S, std::max(cdiv(total, xdl5_wgs()), std::min((uint32_t)min_run, S)));
    a.ap = (const uint4*)scratch;
    a.partial = (float*)((char*)scratch + xdl_a_bytes(w[0], p[0].M));
    a.counter = (uint32_t*)((char*)scratch - kQmmScratchHead);
    a.n_parts = n, a.M = p[0].M, a.U = w[0].KC, a.S = S, a.run = run, a.total = total;
    const uint32_t grid = cdiv(total, run); // <= xdl5_wgs() unless every run is a whole column (no partial tiles then)
    const size_t lds = 2 * (size_t)RT * (4 * kAPieces * 64) * 16 + 64;
    const bool nt = w[0].stream_nt != 0;
    using Fn5 = void (*)(QMM5Args);
    const Fn5 fn = RT == 2 ? (nt ? (Fn5)qmatmul_xdl5_kernel<2, kX5Waves, true> : (Fn5)qmatmul_xdl5_kernel<2, kX5Waves, false>)
                           : (nt ? (Fn5)qmatmul_xdl5_kernel<1, kX5Waves, true> : (Fn5)qmatmul_xdl5_kernel<1, kX5Waves, false>);
#ifdef ZGML_TRACE
    static const bool tr = getenv("ZGML_QMM_XDL5_TRACE") && atoi(getenv("ZGML_QMM_XDL5_TRACE")) != 0;
    static uint64_t* tbuf = nullptr;
    static uint32_t t_grid, t_cols, t_run, t_units;
    if (tr) { // every launch stamps into one buffer (graph replays included); the LAST launch's stamps are printed at exit
        if (!tbuf) {
            hipHostMalloc((void**)&tbuf, 4096 * 16 * sizeof(uint64_t));
            memset(tbuf, 0, 4096 * 16 * sizeof(uint64_t));
            atexit([] {
                hipDeviceSynchronize();
                uint64_t t0 = UINT64_MAX, t_end = 0;
                for (uint32_t b = 0; b < t_grid; b++) {
                    t0 = std::min(t0, tbuf[b * 16]);
                    for (int k = 0; k < 16; k++) t_end = std::max(t_end, tbuf[b * 16 + k]);
                }
                fprintf(stderr, "[x5 trace] K units %u, %u workgroup-columns, grid %u, run %u: span %.2f us (100 MHz clock; thread 0 of each workgroup)\n", t_units, t_cols,
                        t_grid, t_run, (t_end - t0) * 0.01);
                const char* names[6] = {"start", "prologue done", "K loop done", "partial drained", "counted", "output stored"};
                for (int seg = 0; seg < 2; seg++)
                    for (int k = 0; k < 6; k++) {
                        double mn = 1e30, mx = 0, sum = 0;
                        uint32_t n = 0;
                        for (uint32_t b = 0; b < t_grid; b++) {
                            const uint64_t v = tbuf[(b * 2 + seg) * 8 + k];
                            if (v >= t0 && v) mn = std::min(mn, (v - t0) * 0.01), mx = std::max(mx, (v - t0) * 0.01), sum += (v - t0) * 0.01, n++;
                        }
                        if (n) fprintf(stderr, "[x5 trace]   segment %d %-16s n %4u  min %7.2f  avg %7.2f  max %7.2f us\n", seg, names[k], n, mn, sum / n, mx);
                    }
            });
        }
        a.trace = tbuf, t_grid = grid, t_cols = wg_cols, t_run = run, t_units = a.U;
    }
#endif
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kX5Waves * 64), lds, s, a);
    return true;
}

#ifdef ZGML_TRACE
// M > 32: the shared-A form at RT = 4 / 8 m-tiles per workgroup. false = not applicable (narrow outputs: the K-split kernel above keeps them)
bool launch_xdl7(hipStream_t s, const QWeightDev* w, const QMatmulParams* p, uint32_t n, float* scratch, uint32_t S, uint32_t RT, uint32_t tiles) {
    QMM7Args a{};
    uint32_t wg_cols = 0;
    for (uint32_t t = 0; t < n; t++) {
        a.parts[t] = {(const uint4*)w[t].qs, (const uint4*)w[t].sc, p[t].dst, p[t].dst_rs, p[t].N / 32, wg_cols};
        wg_cols += cdiv(p[t].N / 32, kX7Waves);
    }
    const uint32_t groups = tiles / RT;
    if ((uint64_t)groups * wg_cols * sizeof(uint32_t) > kQmmCounterBytes) return false;
    // narrow outputs: every run of a column costs the fan-in a partial tile of 16 RT KB per wave to write and to read back; below ~40
    // workgroup-columns (N < 10240) there are many runs per column and the K-split kernel wins (as for the M <= 32 form)
    static const int min_cols = getenv("ZGML_QMM_XDL7_MIN_COLS") ? atoi(getenv("ZGML_QMM_XDL7_MIN_COLS")) : 40;
    if ((int)wg_cols < min_cols) return false;
    static const int min_run = getenv("ZGML_QMM_XDL7_MIN_RUN") ? std::max(1, atoi(getenv("ZGML_QMM_XDL7_MIN_RUN"))) : 2;
    const uint32_t total = wg_cols * S, wgs_g = std::max(1u, xdl5_wgs() / groups);
    const uint32_t run = std::min(S, std::max(cdiv(total, wgs_g), std::min((uint32_t)min_run, S)));
    a.ap = (const uint4*)scratch;
    a.partial = (float*)((char*)scratch + xdl_a_bytes(w[0], p[0].M));
    a.counter = (uint32_t*)((char*)scratch - kQmmScratchHead);
    a.n_parts = n, a.M = p[0].M, a.U = w[0].KC, a.S = S, a.run = run, a.total = total, a.wg_cols = wg_cols;
    const dim3 grid(cdiv(total, run), groups); // grid.x <= wgs_g unless every run is a whole column (no partial tiles then)
    const size_t lds = 2 * (size_t)RT * (4 * kAPieces * 64) * 16 + 64;
    const bool nt = w[0].stream_nt != 0 && groups == 1;
    using Fn7 = void (*)(QMM7Args);
    const Fn7 fn = RT == 8 ? (nt ? (Fn7)qmatmul_xdl7_kernel<8, true> : (Fn7)qmatmul_xdl7_kernel<8, false>)
                           : (nt ? (Fn7)qmatmul_xdl7_kernel<4, true> : (Fn7)qmatmul_xdl7_kernel<4, false>);
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(fn, grid, dim3(kX7Waves * 64), lds, s, a);
    return true;
}
#endif

void launch_xdl2(hipStream_t s, const QWeightDev* w, const QMatmulParams* p, uint32_t n, float* scratch) {
    const uint32_t U = w[0].KC, S = cdiv(U, 4), R = xdl_tile_pad(p[0].M), tiles = cdiv(cdiv(p[0].M, 16), R) * R;
    if (!p[0].reuse_split) hipLaunchKernelGGL(split_a_kernel, dim3(S, tiles), dim3(256), 0, s, p[0].input, p[0].M, p[0].K, p[0].in_rs, (uint4*)scratch, S);
    static const bool xdl4_on = !(getenv("ZGML_QMM_XDL4") && atoi(getenv("ZGML_QMM_XDL4")) == 0);
#ifdef ZGML_TRACE
    if (R >= 4 && xdl7_enabled() && launch_xdl7(s, w, p, n, scratch, S, R, tiles)) return;
#endif
    if (R >= 4 && xdl4_on) {
        launch_xdl4(s, w, p, n, scratch, S, R, tiles);
        return;
    }
    if (R <= 2 && tiles == R && xdl5_enabled() && launch_xdl5(s, w, p, n, scratch, S, R)) return;
    // narrow outputs at 17-32 rows: the K-split kernel of the M > 32 form at two m-tiles when it splits K at least two ways with
    // >= 16 steps per slice — a workgroup then pulls half (a quarter) of A instead of all of it and the fan-in has 2-4 arrivals
    // (4096 x 4096: 15.6 -> 14.5 us, 11008 x 4096: 31.8 -> 26.5; 2048 x 2048, one step per wave: 9.9 -> 10.3, hence the floor)
    static const bool x4_m32 = !(getenv("ZGML_QMM_XDL4_M32") && atoi(getenv("ZGML_QMM_XDL4_M32")) == 0);
    if (R == 2 && tiles == R && x4_m32 && xdl4_on) {
        uint32_t bcs = 0;
        for (uint32_t t = 0; t < n; t++) bcs += p[t].N / 32;
        const uint32_t sk = std::min({cdiv(device_cus(), bcs), 4u, cdiv(S, (uint32_t)kX4Waves)});
        if (sk >= 2 && S / sk >= 16 && (uint64_t)bcs * sizeof(uint32_t) <= kQmmCounterBytes) {
            launch_xdl4(s, w, p, n, scratch, S, R, tiles);
            return;
        }
    }
    const uint32_t R2 = std::min(R, 2u); // (switch off: the M = 32 form over pairs of tiles; the A pieces are laid out per tile)
    static const int env_g = getenv("ZGML_QMM_XDL2_G") ? atoi(getenv("ZGML_QMM_XDL2_G")) : 0;
    uint32_t total_nb2 = 0;
    for (uint32_t t = 0; t < n; t++) total_nb2 += p[t].N / 16;
    const uint32_t G = env_g ? (uint32_t)env_g : (total_nb2 >= 512 ? 2 : 1);
    static const int env_w = getenv("ZGML_QMM_WAVES") ? atoi(getenv("ZGML_QMM_WAVES")) : 8;
    const uint32_t waves = std::max(std::min<uint32_t>(S, (uint32_t)env_w), cdiv(S, kMaxSpw)), spw_max = cdiv(S, waves); // <= kMaxSpw (xdl2_applies: S <= 96)
    // LDS: the waves' weights (spw_max steps x (G KB of nibbles + 256 B of scales) each), reused by the final reduction
    auto b_bytes = [&](uint32_t g) { return (size_t)waves * spw_max * (g * 1024 + 256); };
    uint32_t Gs = G;
    if (b_bytes(Gs) > 150 * 1024) Gs = 1; // long K: one column group per workgroup keeps the weights in LDS
    QMM2Args a{};
    uint32_t blocks = 0;
    for (uint32_t t = 0; t < n; t++) {
        a.parts[t] = {(const uint4*)w[t].qs, (const uint4*)w[t].sc, p[t].dst, p[t].dst_rs, p[t].N / 16, blocks};
        blocks += p[t].N / 16 / Gs;
    }
    a.ap = (const uint4*)scratch, a.n_parts = n, a.M = p[0].M, a.U = U, a.S = S, a.spw_max = spw_max;
    const dim3 grid(blocks, tiles / R2);
    const size_t lds = std::max((size_t)waves * Gs * R2 * 256 * sizeof(float), b_bytes(Gs));
    const bool nt = w[0].stream_nt != 0 && tiles / R2 == 1;
    if (R2 == 2)
        nt ? launch_xdl2_rg<2, true>(s, a, Gs, grid, waves, lds) : launch_xdl2_rg<2, false>(s, a, Gs, grid, waves, lds);
    else
        nt ? launch_xdl2_rg<1, true>(s, a, Gs, grid, waves, lds) : launch_xdl2_rg<1, false>(s, a, Gs, grid, waves, lds);
}

void launch_tile(hipStream_t s, const QWeightDev& w, const QMatmulParams& p, bool xvec, float* scratch) {
    if (scratch && xdl2_applies(w, p.M)) {
        launch_xdl2(s, &w, &p, 1, scratch);
        return;
    }
    const bool q4 = w.format == QW_Q4, two = p.M > 16;
    const uint32_t R = two ? 2 : 1, KU = q4 ? 32 : 16;
    QMMArgs a{(const uint4*)w.qs, w.sc, p.input, p.dst, p.M, p.K, q4 ? w.KC : 2 * w.KC, p.in_rs, p.dst_rs, p.N / 16};
    uint32_t waves = cdiv(a.U, 4);
    static const int env_w = getenv("ZGML_QMM_WAVES") ? atoi(getenv("ZGML_QMM_WAVES")) : 8;
    if (waves > (uint32_t)env_w) waves = env_w; // x chunk per step = 16R rows x 4*waves units in LDS
    const bool nt = w.stream_nt != 0 && cdiv(p.M, 16 * R) == 1; // several m-tiles re-read the weights: keep them cached
    const TileFn fn = w.scale_f16 ? (q4 ? pick_tile<__half, true>(two, xvec, nt) : pick_tile<__half, false>(two, xvec, nt))
                                  : (q4 ? pick_tile<float, true>(two, xvec, nt) : pick_tile<float, false>(two, xvec, nt));
    dim3 grid(a.NB2, cdiv(p.M, 16 * R));
    const size_t x_bytes = 16ull * R * (waves * 4 * KU + 4) * sizeof(float);
    const size_t red_bytes = (size_t)waves * R * 256 * sizeof(float);
    const size_t lds = x_bytes > red_bytes ? x_bytes : red_bytes;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    hipLaunchKernelGGL(fn, grid, dim3(waves * 64), lds, s, a);
}

} // namespace

bool qmatmul_can_group(const QWeightDev& a, const QMatmulParams& pa, const QWeightDev& b, const QMatmulParams& pb) {
    // one launch computes both: neither output may overlap the other output or the shared input rows
    auto span = [](const float* base, uint32_t M, uint32_t rs, uint32_t n) { return std::make_pair(base, base + (uint64_t)(M - 1) * rs + n); };
    auto apart = [](std::pair<const float*, const float*> x, std::pair<const float*, const float*> y) { return x.second <= y.first || y.second <= x.first; };
    const auto da = span(pa.dst, pa.M, pa.dst_rs, pa.N), db = span(pb.dst, pb.M, pb.dst_rs, pb.N), in = span(pa.input, pa.M, pa.in_rs, pa.K);
    return xdl2_applies(a, pa.M) && xdl2_applies(b, pb.M) && a.KC == b.KC && a.K == b.K && pa.input == pb.input && pa.M == pb.M &&
           pa.K == pb.K && pa.in_rs == pb.in_rs && a.stream_nt == b.stream_nt && (pa.N / 16) % 2 == 0 && (pb.N / 16) % 2 == 0 &&
           apart(da, db) && apart(da, in) && apart(db, in);
}
uint32_t qmatmul_max_group() { return (uint32_t)kMaxQmmParts; }

void launch_qmatmul_group(hipStream_t s, const QWeightDev* w, const QMatmulParams* p, uint32_t n, float* scratch) {
    if (n == 1 || !scratch) {
        for (uint32_t t = 0; t < n; t++) launch_qmatmul(s, w[t], p[t], scratch);
        return;
    }
    launch_xdl2(s, w, p, n, scratch);
}

void launch_qmatmul(hipStream_t s, const QWeightDev& w, const QMatmulParams& p, float* scratch) {
    if (p.M == 0 || p.N == 0) return;
    if (w.format == QW_RAW) {
        dim3 grid(cdiv(p.N, kBlock), p.M);
        qmatmul_raw_kernel<<<grid, kBlock, 0, s>>>((const int8_t*)w.qs, (const float*)w.sc, w.bs, p);
        return;
    }
    if (w.format == QW_W8A8) { // (opt-in) the reference's W8A8 arm: compile_program only gives it to weights every use of which is a dense M = 1 row
        if (p.M != 1) {
            fprintf(stderr, "[zgml_hip] ERROR: an M = %u matmul over a W8A8 (mat-vec only) weight: not launched\n", p.M);
            return;
        }
        launch_w8a8_matvec(s, w, p.input, p.dst);
        return;
    }
    if (w.format == QW_Q4K && p.M != 1) { // compile_program only gives this layout to weights every use of which has M = 1
        fprintf(stderr, "[zgml_hip] ERROR: an M = %u matmul over a K-on-lanes (mat-vec only) weight: not launched\n", p.M);
        return;
    }
    const bool xvec = ((uintptr_t)p.input % 16 == 0) && (p.K % 4 == 0) && (p.M == 1 || p.in_rs % 4 == 0);
    static const int tile_min_m = getenv("ZGML_QMM_TILE_MIN_M") ? atoi(getenv("ZGML_QMM_TILE_MIN_M")) : 2;
    if (p.M >= (uint32_t)tile_min_m && p.M > 1) {
        launch_tile(s, w, p, xvec, scratch);
        return;
    }
    launch_qmatvec_rows(s, w, p, xvec);
}

} // namespace zgml
