// dense_f16.hip — dense matmul with f16-promoted weights (the one place the path is a real
// contraction on 16-bit data, so the one place the XDL matrix cores are used).
//
// Semantics (the reference's f16 weight promotion: src/backend/wgpu.zig:1071-1104 packs the B
// operand of a `matmul` op that has an initial upload to f16 [K,N]; src/backend/metal.zig:680-760
// `matvec_f16` / `matmul_f16` then compute
//     M == 1:  C[n]   = sum_k A[k]            * f32(B16[k,n])      (A stays f32)
//     M  > 1:  C[m,n] = sum_k f16(A[m,k])     * B16[k,n]           (A rounded to half, f32 accumulate)
// Same here; only the summation order differs.
//
// Device layout of a promoted weight ("MFMA-packed", chosen at compile_program time):
//   Bp: half[N/16][KC][64][8],  KC = ceil(K/32), item (g, c, lane = 16r+i) = the 8 values
//       B[k = 32c + 8r + e][n = 16g + i], e = 0..7 (zero for k >= K)
// which is exactly the B operand of v_mfma_f32_16x16x32_f16 for lane (i, r): one 16-byte load per
// lane, 1 KiB contiguous per wave, and no LDS or shuffles on the weight stream.
//
// Thread mapping (same shape as the quantized mat-vec): a workgroup owns one 16-column group and all
// of K; its waves split K. Per step a wave takes DEPTH consecutive 32-k chunks (4 KiB of weights in
// flight per wave, next step prefetched; DEPTH = 8: with 4 the step count doubled and every step exposed
// the x-load latency, 4096^2 at M = 32 15.3 -> 12.2 us), x for the step is staged through LDS by row-contiguous
// loads — as f16 [16R rows][chunk] for the MFMA form (ds_read_b128 = one A operand), as f32 for the
// M == 1 form (v_fma_mix_f32: f32 x times the f16 weight, no rounding of x).
#include "kernels.h"

#include <hip/hip_fp16.h>
#include <stdlib.h>

#include <algorithm>
#include <utility>

namespace zgml {

namespace {

constexpr int kBlock = 256;

typedef float mfma_f4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct F16Args {
    const uint4* bp;
    const float* a;
    float* out;
    uint32_t M, K, KC, a_rs, out_rs, NB2;
};

inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

__device__ __forceinline__ uint32_t column_group(uint32_t b, uint32_t NB2) {
    // groups b and b+8 land on the same XCD (round-robin dispatch); nothing to share here, keep it linear
    (void)NB2;
    return b;
}

template <bool XVEC>
__device__ __forceinline__ float4 load_a4(const float* x, uint32_t i, uint32_t K) {
    if (XVEC) return *(const float4*)(x + (i < K ? i : 0));
    const uint32_t last = K - 1;
    return make_float4(x[min(i, last)], x[min(i + 1, last)], x[min(i + 2, last)], x[min(i + 3, last)]);
}
__device__ __forceinline__ float4 zero_tail(float4 v, uint32_t i, uint32_t K) {
    return make_float4(i < K ? v.x : 0.f, i + 1 < K ? v.y : 0.f, i + 2 < K ? v.z : 0.f, i + 3 < K ? v.w : 0.f);
}

__device__ __forceinline__ uint2 pack_half4(float4 v) {
    const __half2 lo = __floats2half2_rn(v.x, v.y), hi = __floats2half2_rn(v.z, v.w);
    uint2 r;
    r.x = *(const uint32_t*)&lo, r.y = *(const uint32_t*)&hi;
    return r;
}

// R m-tiles of 16 rows (MFMA form), or R == 0: the M == 1 mat-vec form. G column groups per workgroup:
// wave w = (column group w / KW, K slice w % KW); the G wave sets share every staged x chunk, which is
// what the M > 1 form is bound by (x re-staged per 16 columns costs ~1/3 of the kernel at M = 32).
template <int R, bool XVEC, int G, bool NT>
__global__ void __launch_bounds__(512) dense_f16_kernel(F16Args a) {
    // 32-k chunks per wave per step. 8 halves the step count (every step exposes one x-load latency:
    // 4096^2 at M = 1 9.7 -> 8.5 us); the R = 2 form cannot afford the staging registers for that and
    // runs 8 K-slice waves of depth 4 instead (same 1024-k step).
    constexpr int kDepth = R == 2 ? 4 : 8;
    constexpr bool MV = R == 0;
    constexpr int ROWS = MV ? 1 : 16 * R;
    constexpr int NX = MV ? 1 : 2 * kDepth * R / G; // float4 of x each thread stages per step (16R rows x KW*kDepth*32 k)
    extern __shared__ float smem[];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n_kw = (blockDim.x >> 6) / G;
    const uint32_t w = wv % n_kw, cgi = wv / n_kw; // K slice, column group within the workgroup
    const uint32_t g_raw = blockIdx.x * G + cgi, g = min(g_raw, a.NB2 - 1), m0 = blockIdx.y * ROWS;
    const uint32_t n_waves = n_kw;
    const uint32_t row = lane >> 4, i = lane & 15;
    const uint32_t step_chunks = n_waves * kDepth, chf = step_chunks * 32;
    // LDS row: f16 form chf halves + 8 pad (= (chf + 8) / 2 floats); f32 form chf floats
    const uint32_t row_floats = MV ? chf : (chf + 8) / 2;
    const uint32_t buf_floats = ROWS * row_floats;
    const uint4* bp = a.bp + ((uint64_t)g * a.KC) * 64 + lane;
    const uint32_t n_steps = (a.KC + step_chunks - 1) / step_chunks;

    const uint32_t f4_per_row = chf / 4;
    uint32_t st_col[NX], st_lds[NX];
    const float* st_src[NX];
    bool st_on[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) {
        const uint32_t f = threadIdx.x + j * blockDim.x, r = f / f4_per_row;
        st_on[j] = r < (uint32_t)ROWS; // mat-vec form: only chf/4 threads stage
        st_col[j] = 4 * (f % f4_per_row);
        st_lds[j] = MV ? st_col[j] : r * row_floats + st_col[j] / 2;
        st_src[j] = a.a + (uint64_t)min(m0 + min(r, (uint32_t)ROWS - 1), a.M - 1) * a.a_rs;
    }
    float4 stage[NX];
    auto stage_load = [&](uint32_t step) {
#pragma unroll
        for (int j = 0; j < NX; j++) stage[j] = load_a4<XVEC>(st_src[j], step * chf + st_col[j], a.K);
    };
    auto stage_store = [&](uint32_t step) {
        float* dst = smem + (step & 1) * buf_floats;
        const bool full = (step + 1) * chf <= a.K; // whole chunk inside K (uniform): no selects
#pragma unroll
        for (int j = 0; j < NX; j++) {
            const float4 v = full ? stage[j] : zero_tail(stage[j], step * chf + st_col[j], a.K);
            if (MV) {
                if (st_on[j]) *(float4*)(dst + st_lds[j]) = v;
            } else {
                *(uint2*)(dst + st_lds[j]) = pack_half4(v); // f16(A), round to nearest even
            }
        }
    };
    auto load_b = [&](uint4 (&b)[kDepth], uint32_t step) {
#pragma unroll
        for (int d = 0; d < kDepth; d++) { // clamped, unconditional; x of chunks >= KC is staged as zero
            const uint32_t c = min(step * step_chunks + w * kDepth + d, a.KC - 1);
            if (NT) { // weight set beyond the Infinity Cache: read-once stream, do not allocate (see qmatvec.hip wload)
                typedef unsigned int u4v __attribute__((ext_vector_type(4)));
                const u4v v = __builtin_nontemporal_load((const u4v*)(bp + (uint64_t)c * 64));
                b[d] = make_uint4(v.x, v.y, v.z, v.w);
            } else {
                b[d] = bp[(uint64_t)c * 64];
            }
        }
    };

    mfma_f4 acc[MV ? 1 : R];
#pragma unroll
    for (int t = 0; t < (MV ? 1 : R); t++) acc[t] = mfma_f4{0.f, 0.f, 0.f, 0.f};
    float acc_mv = 0.f;

    auto compute = [&](const uint4 (&b)[kDepth], uint32_t step) {
        const float* xs = smem + (step & 1) * buf_floats;
#pragma unroll
        for (int d = 0; d < kDepth; d++) {
            const uint32_t k_local = (w * kDepth + d) * 32 + 8 * row; // this lane's 8 k within the step
            if (MV) {
                const float4 x0 = *(const float4*)(xs + k_local), x1 = *(const float4*)(xs + k_local + 4);
                const __half2* h = (const __half2*)&b[d];
                // fmaf(f32, f32(f16), f32) -> v_fma_mix_f32
                acc_mv = fmaf(x0.x, __low2float(h[0]), acc_mv);
                acc_mv = fmaf(x0.y, __high2float(h[0]), acc_mv);
                acc_mv = fmaf(x0.z, __low2float(h[1]), acc_mv);
                acc_mv = fmaf(x0.w, __high2float(h[1]), acc_mv);
                acc_mv = fmaf(x1.x, __low2float(h[2]), acc_mv);
                acc_mv = fmaf(x1.y, __high2float(h[2]), acc_mv);
                acc_mv = fmaf(x1.z, __low2float(h[3]), acc_mv);
                acc_mv = fmaf(x1.w, __high2float(h[3]), acc_mv);
            } else {
                const half8 bv = *(const half8*)&b[d];
#pragma unroll
                for (int t = 0; t < (MV ? 1 : R); t++) {
                    const half8 av = *(const half8*)(xs + (t * 16 + i) * row_floats + k_local / 2);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
    };

    uint4 cur[kDepth];
    stage_load(0);
    load_b(cur, 0);
    stage_store(0);
    __syncthreads();
    for (uint32_t st = 1; st < n_steps; st++) {
        uint4 nxt[kDepth];
        stage_load(st);
        load_b(nxt, st);
        __builtin_amdgcn_sched_barrier(0); // keep the prefetch ahead of the compute block
        compute(cur, st - 1);
        __builtin_amdgcn_sched_barrier(0);
        stage_store(st);
        __syncthreads();
#pragma unroll
        for (int d = 0; d < kDepth; d++) cur[d] = nxt[d];
    }
    compute(cur, n_steps - 1);
    __syncthreads();

    if (MV) {
        // fold the 4 k-rows of the wave, then the waves, in fixed order
        acc_mv += __shfl_xor(acc_mv, 16, 64);
        acc_mv += __shfl_xor(acc_mv, 32, 64);
        if (lane < 16) smem[w * 16 + lane] = acc_mv;
        __syncthreads();
        if (threadIdx.x < 16) {
            float v = smem[threadIdx.x];
            for (uint32_t ww = 1; ww < n_waves; ww++) v += smem[ww * 16 + threadIdx.x];
            a.out[(uint64_t)m0 * a.out_rs + g * 16 + threadIdx.x] = v;
        }
    } else {
        constexpr int RT = MV ? 1 : R;
#pragma unroll
        for (int t = 0; t < RT; t++)
#pragma unroll
            for (int v = 0; v < 4; v++) smem[((wv * RT + t) * 4 + v) * 64 + lane] = acc[t][v];
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < (uint32_t)G * RT * 256; idx += blockDim.x) {
            const uint32_t cg = idx / (RT * 256), rem = idx % (RT * 256);
            const uint32_t t = rem >> 8, v = (rem >> 6) & 3, l = rem & 63;
            const uint32_t w0 = cg * n_kw; // this column group's K-slice waves, summed in wave order
            float sum = smem[((w0 * RT + t) * 4 + v) * 64 + l];
            for (uint32_t ww = 1; ww < n_kw; ww++) sum += smem[(((w0 + ww) * RT + t) * 4 + v) * 64 + l];
            const uint32_t gg = blockIdx.x * G + cg;
            const uint32_t m = m0 + t * 16 + 4 * (l >> 4) + v, n = gg * 16 + (l & 15);
            if (m < a.M && gg < a.NB2) a.out[(uint64_t)m * a.out_rs + n] = sum;
        }
    }
}

// ── M > 1, second form: A pre-rounded and pre-laid-out, no LDS in the K loop ─────────────────
// The staged form above re-reads f32 x through LDS in every workgroup (rounding it to f16 each time) with a
// barrier per step: 13.3 us for 32 x 4096 x 4096 against 4.2 us of weight streaming. Here pack_a_f16_kernel
// rounds x to f16 ONCE per matmul (same round-to-nearest-even, so the products are the same numbers) into
// the MFMA A-operand layout  Ap: half[tiles][KC][64][8], item (t, c, lane = 16 r + i) = A[16 t + i][32 c + 8 r + e],
// and a lane's A operand is one 16-byte load of a wave-contiguous 1 KB block (L2-resident), exactly like
// its B operand. Waves split the 32-k chunks round-robin and keep DEPTH chunks (B and A) in flight.
struct F16Part2 { // one weight of a grouped launch (q/k/v, gate/up: same rows, same K)
    const uint4* bp;
    float* out;
    uint32_t out_rs, block_begin; // first workgroup (column group) of the part
};
constexpr int kMaxF16Parts = 3;
struct F16Args2 {
    F16Part2 parts[kMaxF16Parts];
    const uint4* ap;
    uint32_t n_parts, M, KC;
};

__global__ void __launch_bounds__(kBlock) pack_a_f16_kernel(const float* __restrict__ x, uint32_t M, uint32_t K, uint32_t a_rs, uint32_t KC,
                                                            uint32_t tiles, uint4* __restrict__ ap) {
    const uint64_t n_items = (uint64_t)tiles * KC * 64;
    for (uint64_t it = (uint64_t)blockIdx.x * kBlock + threadIdx.x; it < n_items; it += (uint64_t)gridDim.x * kBlock) {
        const uint32_t lane = it & 63, i = lane & 15, r = lane >> 4;
        const uint64_t tc = it >> 6;
        const uint32_t c = tc % KC, t = tc / KC, m = 16 * t + i, k0 = 32 * c + 8 * r;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (m < M && k0 + e < K) ? x[(uint64_t)m * a_rs + k0 + e] : 0.f;
        const uint2 lo = pack_half4(make_float4(v[0], v[1], v[2], v[3])), hi = pack_half4(make_float4(v[4], v[5], v[6], v[7]));
        ap[it] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}

// SINK (diagnostics build, ZGML_F16_SINK=1 / 2): the same kernel with its MFMAs replaced by a vector-ALU sink of the operands
// (1: xor of every loaded dword into the accumulators) or with the A loads gone as well (2) — WRONG results by construction;
// separates what the matrix instructions cost the stream from what the memory path costs.
template <int R, bool NT, int CG = 1, int SINK = 0>
__global__ void __launch_bounds__(512) dense_f16_tile2_kernel(F16Args2 a) {
    // chunks in flight per wave, (CG + R) x 16 B per lane each (8 measured no faster at R = 2). R = 4 / 8 (prefill chunks of
    // 64 / 128 tokens): all m-tiles in one workgroup, so the weights are read ONCE per matmul instead of once per tile pair
    // (Llama-2-7B at 128 tokens: 13.5 GB of f16 weights per chunk instead of 54 GB); 3 / 2 chunks in flight keep the ring
    // within the register budget of two waves per SIMD.
    // CG (round 4): column groups per workgroup. Every workgroup re-reads the whole pre-laid-out A operand from L2 (R x 1 KiB per
    // 32-k chunk) next to ONE KiB of weights per column group: at CG = 1 and R = 2 two of every three bytes a CU's vector-memory
    // path returns are A, and that path (64 B / clk per CU whatever the hit level), not HBM, paces the launch (3.6 TB/s at
    // 32 x 4096 x 22016). With CG column groups per workgroup one A chunk feeds CG x R MFMAs: A : B = R : CG.
    constexpr int DEPTH = R >= 8 ? 2 : (R >= 4 ? 3 : (CG >= 4 ? 3 : 4));
    extern __shared__ float smem[];
    // w in an SGPR: the chunk guard below must be a scalar branch — a v_mfma ignores EXEC, so a predicated
    // (if-converted) guard would still accumulate the clamped duplicate chunks
    const uint32_t lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = blockDim.x >> 6;
    uint32_t pi = 0;
#pragma unroll
    for (uint32_t t = 1; t < (uint32_t)kMaxF16Parts; t++)
        if (t < a.n_parts && blockIdx.x >= a.parts[t].block_begin) pi = t;
    const F16Part2& P = a.parts[pi];
    const uint32_t g = (blockIdx.x - P.block_begin) * CG, t0 = blockIdx.y * R; // first column group of this workgroup
    const uint4* bp = P.bp + (uint64_t)g * a.KC * 64 + lane;
    const uint4* ap = a.ap + (uint64_t)t0 * a.KC * 64 + lane;
    const uint64_t tile_stride = (uint64_t)a.KC * 64;
    struct Chunk {
        uint4 b[CG];
        uint4 av[R];
    };
    auto load = [&](Chunk& x, uint32_t c) { // clamped, unconditional: a repeated chunk is never used twice (the loop bounds decide)
        const uint32_t cc = min(c, a.KC - 1);
#pragma unroll
        for (int j = 0; j < CG; j++) {
            if (NT) {
                typedef unsigned int u4v __attribute__((ext_vector_type(4)));
                const u4v v = __builtin_nontemporal_load((const u4v*)(bp + j * tile_stride + (uint64_t)cc * 64));
                x.b[j] = make_uint4(v.x, v.y, v.z, v.w);
            } else {
                x.b[j] = bp[j * tile_stride + (uint64_t)cc * 64];
            }
        }
#pragma unroll
        for (int t = 0; t < R; t++) x.av[t] = SINK == 2 ? make_uint4(cc, 1u, 2u, 3u) : ap[t * tile_stride + (uint64_t)cc * 64];
    };
    mfma_f4 acc[R][CG];
#pragma unroll
    for (int t = 0; t < R; t++)
#pragma unroll
        for (int j = 0; j < CG; j++) acc[t][j] = mfma_f4{0.f, 0.f, 0.f, 0.f};
    // wave w takes chunks w, w + n_waves, ...; ring of DEPTH chunks
    Chunk ring[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) load(ring[d], w + d * n_waves);
    for (uint32_t c = w; c < a.KC; c += DEPTH * n_waves) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const Chunk cur = ring[d];
            load(ring[d], c + (d + DEPTH) * n_waves);
            __builtin_amdgcn_sched_barrier(0); // the refill is issued before the MFMAs of this chunk
            if (c + d * n_waves < a.KC) { // (wave-uniform)
#pragma unroll
                for (int j = 0; j < CG; j++) {
                    const half8 bv = __builtin_bit_cast(half8, cur.b[j]);
#pragma unroll
                    for (int t = 0; t < R; t++) {
                        if (SINK) { // four vector-ALU instructions per (tile, column group) instead of one MFMA
                            const uint4 bq = cur.b[j], aq = cur.av[t];
                            acc[t][j][0] = __uint_as_float(__float_as_uint(acc[t][j][0]) ^ bq.x ^ aq.x), acc[t][j][1] = __uint_as_float(__float_as_uint(acc[t][j][1]) ^ bq.y ^ aq.y);
                            acc[t][j][2] = __uint_as_float(__float_as_uint(acc[t][j][2]) ^ bq.z ^ aq.z), acc[t][j][3] = __uint_as_float(__float_as_uint(acc[t][j][3]) ^ bq.w ^ aq.w);
                        } else {
                            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, cur.av[t]), bv, acc[t][j], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int t = 0; t < R; t++)
#pragma unroll
        for (int j = 0; j < CG; j++)
#pragma unroll
            for (int v = 0; v < 4; v++) smem[(((w * R + t) * CG + j) * 4 + v) * 64 + lane] = acc[t][j][v];
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < (uint32_t)R * CG * 256; idx += blockDim.x) {
        const uint32_t t = idx / (CG * 256), j = (idx >> 8) % CG, v = (idx >> 6) & 3, l = idx & 63;
        float sum = smem[((t * CG + j) * 4 + v) * 64 + l];
        for (uint32_t ww = 1; ww < n_waves; ww++) sum += smem[(((ww * R + t) * CG + j) * 4 + v) * 64 + l]; // waves in order: deterministic
        const uint32_t m = (t0 + t) * 16 + 4 * (l >> 4) + v, n = (g + j) * 16 + (l & 15);
        if (m < a.M) P.out[(uint64_t)m * P.out_rs + n] = sum;
    }
}

constexpr int kT3Chunks = 4; // (the experiment below) 32-k chunks per stage: 512 threads x 16 B = 4 chunks x R = 2 tiles x 1 KiB
constexpr int kT3Waves = 8;
#ifdef ZGML_TRACE // diagnostics build only: MEASURED NO FASTER than the form above (tools/f16_m32_sweep.sh)
// ── M <= 32, third form (round 4 experiment): A SHARED through LDS, K split over workgroups ─────────────────────────────
// Result: 32 x 4096 x 22016 46.2 us against 42.6 (two column groups per workgroup above), 4096^2 15.5 against 13.2, K = 11008
// 25.3 against 27.4 at 8 slices: taking seven eighths of the A traffic out of the vector-memory path did NOT raise the stream
// (4.0-4.5 TB/s either way, a read-only stream of the same access shape does 6.5) — the A re-read is not what paces the launch.
// The hypothesis it was built on: what paces the form above (3.6 TB/s at 32 x 4096 x 22016) is the A operand:
// every workgroup (16 columns) re-reads the whole pre-laid-out A from L2 — R KiB per 1 KiB of weights — through the CU's one
// vector-memory return path. Here a workgroup owns 8 column groups (128 columns: one per wave) and a SLICE of K:
//   * all eight waves work on the SAME 32-k chunks, so a chunk's A operand (R KiB) is fetched from L2 ONCE per workgroup, by
//     one 16-byte load per thread for a stage of kT3Chunks chunks, parked in LDS (two stages: one barrier per stage) and read
//     back by every wave with ds_read_b128: A : B through the vector-memory path drops from R : 1 to R : 8;
//   * a wave streams only its own column group's weights (1 KiB per chunk, two stages in flight) and owns its 16 x 16R output
//     tile: no cross-wave fold;
//   * the grid is (column tiles, K slices): enough workgroups for every CU although a tile is 128 columns wide. A K split
//     publishes partial tiles (write-through stores, drained, one agent-scope add on the tile's counter); the LAST arriver
//     sums the slices IN SLICE ORDER (deterministic: two executions agree bit for bit) and stores.
struct F16Args3 {
    F16Part2 parts[kMaxF16Parts]; // block_begin in COLUMN TILES (8 column groups)
    const uint4* ap;              // pack_a_f16_kernel output: [tile][KC][64] uint4
    float* partial;               // [column tile][slice][wave][R][4][64] floats
    uint32_t* counter;            // one word per column tile, zero between launches
    uint32_t n_parts, M, KC, SK, chunks_per_slice;
};
template <int R, bool NT>
__global__ void __launch_bounds__(kT3Waves * 64) dense_f16_tile3_kernel(F16Args3 a) {
    static_assert(R * kT3Chunks * 64 == kT3Waves * 64, "one 16-byte A load per thread and stage");
    __shared__ uint4 lds_a[3][kT3Chunks * R * 64]; // three stages: the one being multiplied, the next (already parked), the one after (in flight)
    __shared__ uint32_t flag;
    const uint32_t lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t pi = 0;
#pragma unroll
    for (uint32_t t = 1; t < (uint32_t)kMaxF16Parts; t++)
        if (t < a.n_parts && blockIdx.x >= a.parts[t].block_begin) pi = t;
    const F16Part2& P = a.parts[pi];
    const uint32_t tile = blockIdx.x - P.block_begin, g = tile * kT3Waves + w, slice = blockIdx.y;
    const uint32_t c_begin = slice * a.chunks_per_slice, c_end = min(c_begin + a.chunks_per_slice, a.KC), c_last = a.KC - 1;
    const uint4* const bp = P.bp + (uint64_t)g * a.KC * 64 + lane;
    // this thread's share of a stage's A: chunk (tid / (R * 64)), tile ((tid / 64) % R), lane
    const uint32_t a_cc = threadIdx.x / (R * 64), a_t = (threadIdx.x >> 6) % R;
    const uint4* const ap = a.ap + (uint64_t)a_t * a.KC * 64 + lane;
    auto load_a = [&](uint32_t c0) { return ap[(uint64_t)min(c0 + a_cc, c_last) * 64]; };
    struct BStage {
        uint4 b[kT3Chunks];
    };
    auto load_b = [&](BStage& x, uint32_t c0) { // clamped, unconditional; chunks past the slice's end are never multiplied
#pragma unroll
        for (int j = 0; j < kT3Chunks; j++) {
            const uint4* const src = bp + (uint64_t)min(c0 + j, c_last) * 64;
            if (NT) {
                typedef unsigned int u4v __attribute__((ext_vector_type(4)));
                const u4v v = __builtin_nontemporal_load((const u4v*)src);
                x.b[j] = make_uint4(v.x, v.y, v.z, v.w);
            } else {
                x.b[j] = *src;
            }
        }
    };
    mfma_f4 acc[R];
#pragma unroll
    for (int t = 0; t < R; t++) acc[t] = mfma_f4{0.f, 0.f, 0.f, 0.f};
    // two stages of weights in flight per wave beside the one being multiplied (with one, a workgroup keeps 4 KiB per wave in
    // flight and the stream starves: Little's law)
    BStage cur, n1, n2;
    uint4 a0 = load_a(c_begin);
    load_b(cur, c_begin);
    uint4 a1 = load_a(c_begin + kT3Chunks);
    load_b(n1, c_begin + kT3Chunks);
    lds_a[0][threadIdx.x] = a0;
    __syncthreads();
    uint32_t buf = 0;
    for (uint32_t c0 = c_begin; c0 < c_end; c0 += kT3Chunks) {
        const uint4 a2 = load_a(c0 + 2 * kT3Chunks);
        load_b(n2, c0 + 2 * kT3Chunks);
        __builtin_amdgcn_sched_barrier(0); // the stage after next is requested before this one is multiplied
#pragma unroll
        for (int j = 0; j < kT3Chunks; j++) {
            if (c0 + j < c_end) { // (scalar: c0, c_end are wave-uniform)
                const half8 bv = __builtin_bit_cast(half8, cur.b[j]);
#pragma unroll
                for (int t = 0; t < R; t++)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, lds_a[buf][(j * R + t) * 64 + lane]), bv, acc[t], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t nb = buf == 2 ? 0 : buf + 1;
        lds_a[nb][threadIdx.x] = a1; // (that buffer was last read two stages ago: every wave has passed a barrier since)
        __syncthreads();
        cur = n1, n1 = n2, a1 = a2;
        buf = nb;
    }
    // D[m = 16 t + 4 (lane / 16) + v][n = 16 g + lane % 16] in acc[t][v]
    if (a.SK > 1) {
        using gf32 = __attribute__((address_space(1))) float;
        using gu32 = __attribute__((address_space(1))) unsigned int;
        constexpr uint32_t WT = R * 256; // floats of one wave's tile
        float* const mine = a.partial + (((uint64_t)blockIdx.x * a.SK + slice) * kT3Waves + w) * WT + lane;
#pragma unroll
        for (int t = 0; t < R; t++)
#pragma unroll
            for (int v = 0; v < 4; v++) __hip_atomic_store((gf32*)(mine + (t * 4 + v) * 64), acc[t][v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains before the count
        __syncthreads();
        if (threadIdx.x == 0) flag = __hip_atomic_fetch_add((gu32*)(a.counter + blockIdx.x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (flag != a.SK - 1) return; // not the last slice of this column tile
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // compiler only: the loads below stay below the count
#pragma unroll
        for (int t = 0; t < R; t++) acc[t] = mfma_f4{0.f, 0.f, 0.f, 0.f};
        const float* const base = a.partial + (((uint64_t)blockIdx.x * a.SK) * kT3Waves + w) * WT + lane;
        for (uint32_t s0 = 0; s0 < a.SK; s0 += 4) { // slice order, four slices' loads in flight together
            float pv[4][R * 4];
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
                const uint32_t sl = min(s0 + q, a.SK - 1);
#pragma unroll
                for (int e = 0; e < R * 4; e++)
                    pv[q][e] = __hip_atomic_load((gf32*)(base + (uint64_t)sl * kT3Waves * WT + e * 64), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (uint32_t q = 0; q < 4; q++)
#pragma unroll
                for (int t = 0; t < R; t++)
#pragma unroll
                    for (int v = 0; v < 4; v++) acc[t][v] += s0 + q < a.SK ? pv[q][t * 4 + v] : 0.f;
        }
        if (threadIdx.x == 0) __hip_atomic_store((gu32*)(a.counter + blockIdx.x), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-arm
    }
#pragma unroll
    for (int t = 0; t < R; t++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const uint32_t m = t * 16 + 4 * (lane >> 4) + v, n = g * 16 + (lane & 15);
            if (m < a.M) P.out[(uint64_t)m * P.out_rs + n] = acc[t][v];
        }
}

#endif // ZGML_TRACE

#ifdef ZGML_TRACE // diagnostics build only (ZGML_F16_TILE4=1): parity green, MEASURED within +-5 % of the shipped form on wide outputs and
                  // 28 % slower at 4096 x 4096 (profiles/r04_f16_m32_tile4.txt) — the A loads it removes are not all that paces the stream
// ── M <= 32, fourth form (round 4 experiment): A STATIONARY in LDS ──────────────────────────────────────────────────────
// The sink experiment (tools/f16_sink.sh) says what paces the forms above: not the MFMAs (replaced by a VALU sink: unchanged),
// the A operand's GLOBAL loads (gone as well: 5.6 TB/s instead of 3.6-4.4) — R KiB of L2 hits per KiB of weights through the CU's
// one vector-memory path. Here that path carries weights only:
//   * one workgroup per CU (16 waves; its dynamic LDS is sized to keep a second one out) owns a K SLICE of <= 64 chunks and
//     loads that slice of the pre-laid-out A (<= 128 KiB) into LDS ONCE; one barrier, then no A load and no barrier again;
//   * a wave's work is a list of TASKS = (column group, this slice): the column groups cb, cb + n_cwg, ... of the workgroup's
//     column class are dealt to its waves round-robin; per 32-k chunk a wave issues one 16-byte weight load per lane (1 KiB per
//     wave-instruction, DEPTH in flight, the ring runs on across task boundaries), reads its two A operands from LDS
//     (ds_read_b128, lane-contiguous: conflict-free) and issues two MFMAs; it owns its 16 x 32 output tile: no cross-wave fold;
//   * K slices: as many as give every CU ~10 tasks (the tasks, not the workgroups, carry the parallelism). A split K publishes
//     each task's partial tile (write-through stores, drained by the storing wave, one agent-scope add on the column group's
//     counter); the LAST arriver of a column group sums ALL slices in slice order — its own included — so two executions agree
//     bit for bit, stores, and re-arms the counter.
#endif
constexpr int kT4Waves = 16, kT4MaxSliceChunks = 64, kT4Depth = 4;
#ifdef ZGML_TRACE
struct F16Args4 {
    F16Part2 parts[kMaxF16Parts]; // block_begin in COLUMN GROUPS
    const uint4* ap;              // pack_a_f16_kernel output: [tile][KC][64] uint4
    float* partial;               // [column group][slice][2][4][64] floats
    uint32_t* counter;            // one word per column group, zero between launches
    uint32_t n_parts, M, KC, SK, chunks_per_slice, n_cwg, total_groups;
};
template <bool NT, int kT4Depth = 4>
__global__ void __launch_bounds__(kT4Waves * 64) dense_f16_tile4_kernel(F16Args4 a) {
    constexpr int R = 2;
    extern __shared__ uint4 lds_a4[]; // [chunk of the slice][R][64]
    const uint32_t lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t slice = blockIdx.x % a.SK, cb = blockIdx.x / a.SK;
    const uint32_t c_begin = slice * a.chunks_per_slice, c_end = min(c_begin + a.chunks_per_slice, a.KC), S = c_end - c_begin;
    // this wave's tasks: column groups cb + (w + kT4Waves * r) * n_cwg, r = 0, 1, ... (< total_groups)
    const uint32_t g_first = cb + w * a.n_cwg, g_step = kT4Waves * a.n_cwg;
    const uint32_t n_tasks = g_first < a.total_groups ? (a.total_groups - 1 - g_first) / g_step + 1 : 0;
    auto part_of = [&](uint32_t g, uint32_t& gl) -> const F16Part2& { // (<= 3 parts: selects, no loop-carried lookups)
        uint32_t pi = 0;
#pragma unroll
        for (uint32_t t = 1; t < (uint32_t)kMaxF16Parts; t++)
            if (t < a.n_parts && g >= a.parts[t].block_begin) pi = t;
        gl = g - a.parts[pi].block_begin;
        return a.parts[pi];
    };
    auto b_src = [&](uint32_t task, uint32_t c) -> const uint4* { // weights of chunk c_begin + c of task `task` (clamped: in bounds)
        const uint32_t g = min(g_first + task * g_step, a.total_groups - 1);
        uint32_t gl;
        const F16Part2& P = part_of(g, gl);
        return P.bp + ((uint64_t)gl * a.KC + min(c_begin + c, a.KC - 1)) * 64 + lane;
    };
    auto wl = [&](const uint4* src) {
        if (NT) {
            typedef unsigned int u4v __attribute__((ext_vector_type(4)));
            const u4v v = __builtin_nontemporal_load((const u4v*)src);
            return make_uint4(v.x, v.y, v.z, v.w);
        }
        return *src;
    };
    // the weight ring starts before the A slice is in LDS (its first loads fly while A is fetched)
    uint4 ring[kT4Depth];
    uint32_t lt = 0, lc = 0; // (task, chunk) of the next load
    auto advance = [&](uint32_t& t, uint32_t& c) {
        if (++c == S) c = 0, t++;
    };
    if (n_tasks) {
#pragma unroll
        for (int d = 0; d < kT4Depth; d++) {
            ring[d] = wl(b_src(min(lt, n_tasks - 1), lc));
            advance(lt, lc);
        }
    }
    // A slice -> LDS: item (c, t, lane) of the slice from ap[t][c_begin + c][lane]
    for (uint32_t i = threadIdx.x; i < S * R * 64; i += kT4Waves * 64) {
        const uint32_t l = i & 63, t = (i >> 6) % R, c = i / (R * 64);
        lds_a4[i] = a.ap[((uint64_t)t * a.KC + c_begin + c) * 64 + l];
    }
    __syncthreads();
    if (!n_tasks) return;
    using gf32 = __attribute__((address_space(1))) float;
    using gu32 = __attribute__((address_space(1))) unsigned int;
    for (uint32_t task = 0; task < n_tasks; task++) {
        mfma_f4 acc[R] = {mfma_f4{0.f, 0.f, 0.f, 0.f}, mfma_f4{0.f, 0.f, 0.f, 0.f}};
        for (uint32_t c0 = 0; c0 < S; c0 += kT4Depth) {
#pragma unroll
            for (int d = 0; d < kT4Depth; d++) {
                const uint4 bq = ring[d];
                ring[d] = wl(b_src(min(lt, n_tasks - 1), lc)); // refill the slot just read (runs on into the next task; clamped past the end)
                advance(lt, lc);
                __builtin_amdgcn_sched_barrier(0);
                if (c0 + d < S) { // (scalar)
                    const half8 bv = __builtin_bit_cast(half8, bq);
#pragma unroll
                    for (int t = 0; t < R; t++)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, lds_a4[((c0 + d) * R + t) * 64 + lane]), bv, acc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // (S need not be a multiple of the depth: the ring slots consumed past S belong to the next task's first chunks — realign)
        if (S % kT4Depth) { // rare (K / 32 not a multiple of 4 per slice): restart the ring at the next task's first chunk
            lt = task + 1, lc = 0;
#pragma unroll
            for (int d = 0; d < kT4Depth; d++) {
                ring[d] = wl(b_src(min(lt, n_tasks - 1), lc));
                advance(lt, lc);
            }
        }
        const uint32_t g = g_first + task * g_step;
        uint32_t gl;
        const F16Part2& P = part_of(g, gl);
        if (a.SK > 1) {
            constexpr uint32_t WT = R * 256;
            float* const mine = a.partial + ((uint64_t)g * a.SK + slice) * WT + lane;
#pragma unroll
            for (int t = 0; t < R; t++)
#pragma unroll
                for (int v = 0; v < 4; v++) __hip_atomic_store((gf32*)(mine + (t * 4 + v) * 64), acc[t][v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's stores have left before it counts itself in
            uint32_t old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add((gu32*)(a.counter + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old != a.SK - 1) continue; // not the last slice of this column group
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // compiler only: the loads below stay below the count
            acc[0] = acc[1] = mfma_f4{0.f, 0.f, 0.f, 0.f};
            const float* const base = a.partial + (uint64_t)g * a.SK * WT + lane;
            for (uint32_t s0 = 0; s0 < a.SK; s0 += 4) { // slice order; four slices' loads in flight together
                float pv[4][R * 4];
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    const uint32_t sl = min(s0 + q, a.SK - 1);
#pragma unroll
                    for (int e = 0; e < R * 4; e++) pv[q][e] = __hip_atomic_load((gf32*)(base + (uint64_t)sl * WT + e * 64), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (uint32_t q = 0; q < 4; q++)
#pragma unroll
                    for (int t = 0; t < R; t++)
#pragma unroll
                        for (int v = 0; v < 4; v++) acc[t][v] += s0 + q < a.SK ? pv[q][t * 4 + v] : 0.f;
            }
            if (lane == 0) __hip_atomic_store((gu32*)(a.counter + g), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-arm
        }
#pragma unroll
        for (int t = 0; t < R; t++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const uint32_t m = t * 16 + 4 * (lane >> 4) + v, n = gl * 16 + (lane & 15);
                if (m < a.M) P.out[(uint64_t)m * P.out_rs + n] = acc[t][v];
            }
    }
}

#endif // ZGML_TRACE (tile4)

// f32 B (any strides, device memory) -> MFMA-packed f16. One thread per 16-byte item.
__global__ void __launch_bounds__(kBlock) pack_f16_kernel(const float* __restrict__ b, uint32_t b_rs, uint32_t b_cs,
                                                          uint32_t K, uint32_t N, uint32_t KC, uint4* __restrict__ out) {
    const uint64_t n_items = (uint64_t)(N / 16) * KC * 64;
    for (uint64_t it = (uint64_t)blockIdx.x * kBlock + threadIdx.x; it < n_items; it += (uint64_t)gridDim.x * kBlock) {
        const uint32_t lane = it & 63, i = lane & 15, r = lane >> 4;
        const uint64_t gc = it >> 6;
        const uint32_t c = gc % KC, g = gc / KC;
        const uint32_t n = g * 16 + i;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t k = c * 32 + 8 * r + e;
            v[e] = k < K ? b[(uint64_t)k * b_rs + (uint64_t)n * b_cs] : 0.f;
        }
        const uint2 lo = pack_half4(make_float4(v[0], v[1], v[2], v[3])), hi = pack_half4(make_float4(v[4], v[5], v[6], v[7]));
        out[it] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}

} // namespace

uint64_t f16_packed_bytes(uint64_t K, uint64_t N) { return (N / 16) * ((K + 31) / 32) * 64 * 16; }

bool f16_packable(uint64_t K, uint64_t N) { return K > 0 && N > 0 && N % 16 == 0; }

void launch_pack_f16(hipStream_t s, const float* b, uint32_t b_rs, uint32_t b_cs, uint32_t K, uint32_t N, void* out) {
    pack_f16_kernel<<<2048, kBlock, 0, s>>>(b, b_rs, b_cs, K, N, (K + 31) / 32, (uint4*)out);
}

static uint32_t f16_tiles_per_wg(uint32_t M) { // m-tiles a workgroup of the A-pre-laid-out kernel carries
    static const bool wide = !(getenv("ZGML_F16_TILE2_WIDE") && atoi(getenv("ZGML_F16_TILE2_WIDE")) == 0);
    return M > 64 && wide ? 8 : (M > 32 && wide ? 4 : (M > 16 ? 2 : 1));
}
// the pre-laid-out A operand of an M x K matmul has no padding rows / columns (a producer may then write it in place of
// pack_a_f16_kernel: kernels.h a_f16_index)
bool dense_f16_a_unpadded(uint32_t M, uint32_t K) {
    return M > 1 && M % 16 == 0 && K % 32 == 0 && (M / 16) % f16_tiles_per_wg(M) == 0 && dense_f16_scratch_bytes(M, K) != 0;
}
static uint64_t f16_a_bytes(uint32_t M, uint32_t K) { // the pre-laid-out A operand (rounded up to 256 B)
    const uint64_t tiles = (M + 15) / 16, R = f16_tiles_per_wg(M);
    return ((tiles + R - 1) / R * R * ((K + 31) / 32) * 1024 + 255) / 256 * 256;
}
// the shared-A form's K split: at most kT3MaxWgs workgroups per launch, one partial tile (8 waves x R x 1 KiB) each
constexpr uint64_t kT3MaxWgs = 1024;
static bool f16_tile3_on() { // the shared-A K-split experiment: diagnostics build only, and only on request
#ifdef ZGML_TRACE
    static const bool on = getenv("ZGML_F16_TILE3") && atoi(getenv("ZGML_F16_TILE3")) != 0;
    return on;
#else
    return false;
#endif
}
static bool f16_tile4_on();
uint64_t dense_f16_scratch_bytes(uint32_t M, uint32_t K) {
    static const bool on = !(getenv("ZGML_F16_TILE2") && atoi(getenv("ZGML_F16_TILE2")) == 0);
    if (!on || M <= 1) return 0;
    return f16_a_bytes(M, K) + (M <= 32 && M > 16 && (f16_tile3_on() || f16_tile4_on()) ? kT3MaxWgs * kT3Waves * 2 * 1024 : 0);
}

// M in (16, 32], every part a whole number of 128-column tiles: the shared-A form with a K split. false: not applicable.
static bool launch_dense_f16_tile3(hipStream_t s, const DenseF16Params* p, uint32_t n, uint32_t KC) {
#ifndef ZGML_TRACE
    (void)s, (void)p, (void)n, (void)KC;
    return false;
#else
    if (!f16_tile3_on() || p[0].M <= 16 || p[0].M > 32) return false;
    F16Args3 a{};
    uint32_t tiles = 0;
    for (uint32_t t = 0; t < n; t++) {
        if ((p[t].N / 16) % kT3Waves != 0) return false;
        a.parts[t] = {(const uint4*)p[t].bp, p[t].dst, p[t].dst_rs, tiles};
        tiles += p[t].N / 16 / kT3Waves;
    }
    if ((uint64_t)tiles * sizeof(uint32_t) > kQmmScratchHead) return false; // one counter word per column tile
    // K slices: ~2.5 workgroups per CU, whole stages per slice, at least 4 stages each, no more partial tiles than the scratch holds
    static const int env_sk = getenv("ZGML_F16_TILE3_SK") ? atoi(getenv("ZGML_F16_TILE3_SK")) : 0;
    const uint32_t stages = cdiv(KC, kT3Chunks);
    uint32_t SK = env_sk > 0 ? (uint32_t)env_sk : std::max(1u, std::min(cdiv(640u, tiles), stages / 4));
    SK = std::max(1u, std::min<uint32_t>(SK, (uint32_t)(kT3MaxWgs / tiles)));
    uint32_t cps = cdiv(stages, SK) * kT3Chunks; // chunks per slice
    SK = cdiv(KC, cps);
    a.ap = (const uint4*)p[0].scratch, a.n_parts = n, a.M = p[0].M, a.KC = KC, a.SK = SK, a.chunks_per_slice = cps;
    a.partial = (float*)((char*)p[0].scratch + f16_a_bytes(p[0].M, p[0].K));
    a.counter = (uint32_t*)((char*)p[0].scratch - kQmmScratchHead);
    const bool nt = p[0].stream_nt != 0;
    hipLaunchKernelGGL((nt ? dense_f16_tile3_kernel<2, true> : dense_f16_tile3_kernel<2, false>), dim3(tiles, SK), dim3(kT3Waves * 64), 0, s, a);
    return true;
#endif
}

static bool f16_tile4_on() { // the A-stationary experiment: diagnostics build only, and only on request
#ifdef ZGML_TRACE
    static const bool on = getenv("ZGML_F16_TILE4") && atoi(getenv("ZGML_F16_TILE4")) != 0;
    return on;
#else
    return false;
#endif
}
static uint32_t f16_device_cus() {
    static const uint32_t n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return (uint32_t)cus;
    }();
    return n;
}
// M in (16, 32]: the A-stationary form. false: not applicable (launch nothing).
static bool launch_dense_f16_tile4(hipStream_t s, const DenseF16Params* p, uint32_t n, uint32_t KC) {
#ifndef ZGML_TRACE
    (void)s, (void)p, (void)n, (void)KC;
    return false;
#else
    if (!f16_tile4_on() || p[0].M <= 16 || p[0].M > 32) return false;
    F16Args4 a{};
    uint32_t groups = 0;
    for (uint32_t t = 0; t < n; t++) {
        a.parts[t] = {(const uint4*)p[t].bp, p[t].dst, p[t].dst_rs, groups};
        groups += p[t].N / 16;
    }
    if ((uint64_t)groups * sizeof(uint32_t) > kQmmScratchHead || groups == 0) return false; // one counter word per column group
    const uint32_t cus = f16_device_cus();
    // K slices: every CU ~10 tasks (column group x slice), a slice of at most 64 chunks (its A: 128 KiB of LDS) and at least 8
    static const int env_sk = getenv("ZGML_F16_TILE4_SK") ? atoi(getenv("ZGML_F16_TILE4_SK")) : 0;
    const uint32_t sk_min = cdiv(KC, (uint32_t)kT4MaxSliceChunks), sk_max = std::max(sk_min, KC / 8);
    uint32_t SK = env_sk > 0 ? (uint32_t)env_sk : (10u * cus + groups / 2) / groups;
    SK = std::max(sk_min, std::min(SK, sk_max));
    SK = std::min(SK, cus);
    uint32_t cps = cdiv(KC, SK);
    cps = cdiv(cps, (uint32_t)kT4Depth) * kT4Depth; // whole ring rounds per slice (the last slice may be shorter)
    if (cps > (uint32_t)kT4MaxSliceChunks) cps = kT4MaxSliceChunks;
    SK = cdiv(KC, cps);
    if ((uint64_t)groups * SK * 2 * 1024 > kT3MaxWgs * kT3Waves * 2 * 1024) return false; // partial tiles must fit the scratch's partial region
    a.ap = (const uint4*)p[0].scratch, a.n_parts = n, a.M = p[0].M, a.KC = KC, a.SK = SK, a.chunks_per_slice = cps;
    a.n_cwg = std::max(1u, cus / SK), a.total_groups = groups;
    a.partial = (float*)((char*)p[0].scratch + f16_a_bytes(p[0].M, p[0].K));
    a.counter = (uint32_t*)((char*)p[0].scratch - kQmmScratchHead);
    const bool nt = p[0].stream_nt != 0;
    const size_t lds = std::max<size_t>((size_t)cps * 2 * 1024, 96 * 1024); // >= 96 KiB: never two workgroups on one CU
    using Fn4 = void (*)(F16Args4);
    // (round 5) weight loads in flight per wave: a wave walks its task's 64 chunks one after the other, so with 4 in flight and
    // ~1.6 us of loaded latency a chunk costs >= 0.4 us whatever else happens: ZGML_F16_TILE4_DEPTH=8 doubles the ring
    static const int env_depth = getenv("ZGML_F16_TILE4_DEPTH") ? atoi(getenv("ZGML_F16_TILE4_DEPTH")) : 4;
    const bool deep = env_depth >= 8 && cps % 8 == 0;
    const Fn4 fn = deep ? (nt ? (Fn4)dense_f16_tile4_kernel<true, 8> : (Fn4)dense_f16_tile4_kernel<false, 8>)
                        : (nt ? (Fn4)dense_f16_tile4_kernel<true, 4> : (Fn4)dense_f16_tile4_kernel<false, 4>);
    static bool attr_set[4] = {false, false, false, false};
    const int ai = (deep ? 2 : 0) + (nt ? 1 : 0);
    if (!attr_set[ai]) attr_set[ai] = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    hipLaunchKernelGGL(fn, dim3(a.n_cwg * SK), dim3(kT4Waves * 64), lds, s, a);
    return true;
#endif
}

static void launch_dense_f16_tile2(hipStream_t s, const DenseF16Params* p, uint32_t n) {
    const uint32_t KC = (p[0].K + 31) / 32, R = f16_tiles_per_wg(p[0].M), tiles = cdiv(cdiv(p[0].M, 16), R) * R;
    if (!p[0].reuse_a) {
        const uint64_t items = (uint64_t)tiles * KC * 64;
        pack_a_f16_kernel<<<(uint32_t)std::min<uint64_t>(2048, (items + kBlock - 1) / kBlock), kBlock, 0, s>>>(p[0].a, p[0].M, p[0].K, p[0].a_rs, KC,
                                                                                                             tiles, (uint4*)p[0].scratch);
    }
    if (R == 2 && tiles == 2 && launch_dense_f16_tile3(s, p, n, KC)) return;
    if (R == 2 && tiles == 2 && launch_dense_f16_tile4(s, p, n, KC)) return;
    static const int env_w = getenv("ZGML_F16_TILE2_WAVES") ? atoi(getenv("ZGML_F16_TILE2_WAVES")) : 8;
    const uint32_t waves = std::max(1u, std::min<uint32_t>(KC, (uint32_t)env_w));
    // column groups per workgroup (R <= 2 only: the wide-M forms already amortise A over 4 / 8 m-tiles): as many as keep the grid
    // at >= ~1.3 workgroups per CU, every part a whole number of workgroups
    uint32_t total_groups = 0;
    for (uint32_t t = 0; t < n; t++) total_groups += p[t].N / 16;
    static const int env_cg = getenv("ZGML_F16_TILE2_CG") ? atoi(getenv("ZGML_F16_TILE2_CG")) : 0;
    uint32_t CG = 1;
    if (R <= 2) {
        // measured at M = 32, K = 4096 (tools/f16_m32_sweep.sh, us per launch incl. the A pack; CG = 1 / 2 / 4): N = 4096 13.6 / 16.8 / 22.9,
        // N = 12288 29.9 / 30.0 / 28.5, N = 22016 51.3 / 44.0 / 48.4, N = 32000 68.5 / 55.4 / 56.1; K = 11008, N = 4096 27.8 / 35.3 / 47.2:
        // two groups per workgroup once that still leaves >= 2.5 workgroups per CU (CG = 4 runs one workgroup per CU: 142 registers)
        {
            bool fits = total_groups / 2 >= 640;
            for (uint32_t t = 0; t < n; t++) fits = fits && (p[t].N / 16) % 2 == 0;
            if (fits) CG = 2;
        }
        if (env_cg == 1 || env_cg == 2 || env_cg == 4) {
            bool fits = true;
            for (uint32_t t = 0; t < n; t++) fits = fits && (p[t].N / 16) % (uint32_t)env_cg == 0;
            if (fits) CG = (uint32_t)env_cg;
        }
    }
    F16Args2 a{};
    uint32_t blocks = 0;
    for (uint32_t t = 0; t < n; t++) {
        a.parts[t] = {(const uint4*)p[t].bp, p[t].dst, p[t].dst_rs, blocks};
        blocks += p[t].N / 16 / CG;
    }
    a.ap = (const uint4*)p[0].scratch, a.n_parts = n, a.M = p[0].M, a.KC = KC;
    const dim3 grid(blocks, tiles / R);
    const size_t lds = (size_t)waves * R * CG * 256 * sizeof(float);
    const bool nt = p[0].stream_nt != 0 && tiles / R == 1;
    using Fn2 = void (*)(F16Args2);
#define ZGML_T2(RV) (CG == 4 ? (nt ? (Fn2)dense_f16_tile2_kernel<RV, true, 4> : (Fn2)dense_f16_tile2_kernel<RV, false, 4>)                    \
                     : CG == 2 ? (nt ? (Fn2)dense_f16_tile2_kernel<RV, true, 2> : (Fn2)dense_f16_tile2_kernel<RV, false, 2>)                 \
                               : (nt ? (Fn2)dense_f16_tile2_kernel<RV, true, 1> : (Fn2)dense_f16_tile2_kernel<RV, false, 1>))
    Fn2 fn = R == 8   ? (nt ? (Fn2)dense_f16_tile2_kernel<8, true> : (Fn2)dense_f16_tile2_kernel<8, false>)
             : R == 4 ? (nt ? (Fn2)dense_f16_tile2_kernel<4, true> : (Fn2)dense_f16_tile2_kernel<4, false>)
             : R == 2 ? ZGML_T2(2)
                      : ZGML_T2(1);
#undef ZGML_T2
#ifdef ZGML_TRACE // the MFMA-sink experiment (wrong results): diagnostics build only
    static const int sink = getenv("ZGML_F16_SINK") ? atoi(getenv("ZGML_F16_SINK")) : 0;
    if (sink && R == 2 && nt) {
        if (CG == 2)
            fn = sink == 2 ? (Fn2)dense_f16_tile2_kernel<2, true, 2, 2> : (Fn2)dense_f16_tile2_kernel<2, true, 2, 1>;
        else if (CG == 1)
            fn = sink == 2 ? (Fn2)dense_f16_tile2_kernel<2, true, 1, 2> : (Fn2)dense_f16_tile2_kernel<2, true, 1, 1>;
    }
#endif
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(fn, grid, dim3(waves * 64), lds, s, a);
}

bool dense_f16_can_group(const DenseF16Params& a, const DenseF16Params& b) {
    auto span = [](const float* base, uint32_t M, uint32_t rs, uint32_t n) { return std::make_pair(base, base + (uint64_t)(M - 1) * rs + n); };
    auto apart = [](std::pair<const float*, const float*> x, std::pair<const float*, const float*> y) { return x.second <= y.first || y.second <= x.first; };
    const auto da = span(a.dst, a.M, a.dst_rs, a.N), db = span(b.dst, b.M, b.dst_rs, b.N), in = span(a.a, a.M, a.a_rs, a.K);
    return a.M > 1 && a.scratch && a.scratch == b.scratch && dense_f16_scratch_bytes(a.M, a.K) && a.a == b.a && a.M == b.M && a.K == b.K &&
           a.a_rs == b.a_rs && a.stream_nt == b.stream_nt && apart(da, db) && apart(da, in) && apart(db, in);
}
uint32_t dense_f16_max_group() { return (uint32_t)kMaxF16Parts; }

void launch_dense_f16_group(hipStream_t s, const DenseF16Params* p, uint32_t n) {
    if (n == 1) {
        launch_dense_f16(s, p[0]);
        return;
    }
    launch_dense_f16_tile2(s, p, n);
}

void launch_dense_f16(hipStream_t s, const DenseF16Params& p) {
    if (p.M == 0 || p.N == 0) return;
    if (p.M > 1 && p.scratch && dense_f16_scratch_bytes(p.M, p.K)) {
        launch_dense_f16_tile2(s, &p, 1);
        return;
    }
    F16Args a{(const uint4*)p.bp, p.a, p.dst, p.M, p.K, (p.K + 31) / 32, p.a_rs, p.dst_rs, p.N / 16};
    const uint32_t R = p.M == 1 ? 0 : (p.M > 16 ? 2 : 1);
    // see kDepth in the kernel; the 8-wave R = 2 form only pays while the grid is about one round (N <= 6144)
    const uint32_t depth = R == 2 ? 4 : 8, max_kw = (R == 2 && a.NB2 <= 384) ? 8 : 4;
    uint32_t waves = cdiv(a.KC, depth);
    if (waves > max_kw) waves = max_kw;
    const bool xvec = ((uintptr_t)p.a % 16 == 0) && (p.K % 4 == 0) && (p.M == 1 || p.a_rs % 4 == 0);
    const uint32_t chf = waves * depth * 32;
    using Fn = void (*)(F16Args);
    const bool nt = p.stream_nt != 0;
    if (R == 0) {
        const size_t lds = 2ull * chf * sizeof(float);
        const Fn fn = nt ? (xvec ? (Fn)dense_f16_kernel<0, true, 1, true> : (Fn)dense_f16_kernel<0, false, 1, true>)
                         : (xvec ? (Fn)dense_f16_kernel<0, true, 1, false> : (Fn)dense_f16_kernel<0, false, 1, false>);
        if (lds > 64 * 1024) hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(fn, dim3(a.NB2, 1), dim3(waves * 64), lds, s, a);
        return;
    }
    // two column groups per workgroup share the staged x chunk — only worth it when the grid stays several
    // rounds deep (measured at M = 32: N = 32000 75 -> 67 us; N = 11008 neutral, N = 4096 slower); the
    // 8-wave R = 2 form has no room for a second wave set
    static const int env_g = getenv("ZGML_F16_GROUPS") ? atoi(getenv("ZGML_F16_GROUPS")) : 0;
    uint32_t G = env_g ? (uint32_t)env_g : (a.NB2 >= 1536 ? 2u : 1u);
    if (waves * G > 8) G = 1;
    size_t lds = 2ull * 16 * R * ((chf + 8) / 2) * sizeof(float);
    const size_t red = (size_t)waves * G * R * 256 * sizeof(float);
    if (red > lds) lds = red;
    const dim3 grid(cdiv(a.NB2, G), cdiv(p.M, 16 * R));
    Fn fn;
#define ZGML_F16_PICK(NTV)                                                                                                   \
    (G == 2 ? (R == 2 ? (xvec ? (Fn)dense_f16_kernel<2, true, 2, NTV> : (Fn)dense_f16_kernel<2, false, 2, NTV>)               \
                      : (xvec ? (Fn)dense_f16_kernel<1, true, 2, NTV> : (Fn)dense_f16_kernel<1, false, 2, NTV>))              \
            : (R == 2 ? (xvec ? (Fn)dense_f16_kernel<2, true, 1, NTV> : (Fn)dense_f16_kernel<2, false, 1, NTV>)               \
                      : (xvec ? (Fn)dense_f16_kernel<1, true, 1, NTV> : (Fn)dense_f16_kernel<1, false, 1, NTV>)))
    fn = nt ? ZGML_F16_PICK(true) : ZGML_F16_PICK(false);
#undef ZGML_F16_PICK
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(fn, grid, dim3(waves * G * 64), lds, s, a);
}

} // namespace zgml
