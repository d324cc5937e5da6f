// kernels_generic.hip — gfx950 kernels for the DeviceOp kinds that are not the quantized
// mat-vec: elementwise / fused chains / row-wise norms / reduce / repeat / slice_assign / rope /
// attention / dense matmul / argmax / copy. One launch per op; fused variants live in fused.hip.
//
// Semantics follow the reference executor src/backend/reference.zig (cited per kernel). These ops
// are tiny at decode (d_model..d_ff elements): they are latency-bound, so the kernels favour
// simple full-wave shapes (64-lane shuffles, one workgroup per row) over tiling.
#include "kernels.h"
#include <stdlib.h>

#include <hip/hip_fp16.h>
#include <math.h>

#include "attention_decode.h"

namespace zgml {

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// all-threads block reductions over 256 threads (4 waves); `red` has >= 4 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) t = fmaxf(t, red[i]);
    return t;
}

__device__ __forceinline__ float gelu_f(float a) {
    float kk = 0.7978845608f * (a + 0.044715f * a * a * a);
    return 0.5f * a * (1.0f + tanhf(kk));
}

// unary/binary op table: reference.zig:201-273 (+ true sgn/step, forward.zig:959-1007)
__device__ __forceinline__ float apply_unary(uint32_t op, float a) {
    switch (op) {
        case ZGML_OP_NEG: return -a;
        case ZGML_OP_ABS: return fabsf(a);
        case ZGML_OP_SGN: return a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f);
        case ZGML_OP_STEP: return a > 0.f ? 1.f : 0.f;
        case ZGML_OP_RELU: return fmaxf(a, 0.f);
        case ZGML_OP_SQRT: return sqrtf(a);
        case ZGML_OP_RECIP: return 1.0f / a;
        case ZGML_OP_EXP: return expf(a);
        case ZGML_OP_LOG: return logf(a);
        case ZGML_OP_GELU: return gelu_f(a);
        default: return a; // reference: unknown tag copies src0 (:271)
    }
}

__global__ void __launch_bounds__(kBlock) elementwise_kernel(uint32_t op, float* __restrict__ dst,
                                                             const float* s0, const float* s1, uint32_t n) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float a = s0[i];
    float r;
    if (op == ZGML_OP_ADD)
        r = a + s1[i];
    else if (op == ZGML_OP_MUL)
        r = a * s1[i];
    else
        r = apply_unary(op, a);
    dst[i] = r;
}

// reference.zig:275-307
__global__ void __launch_bounds__(kBlock) fused_elementwise_kernel(FusedParams p) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= p.n) return;
    float v = p.src[i];
#pragma unroll 1
    for (uint32_t s = 0; s < p.n_steps; s++) {
        uint32_t op = p.steps[s].op;
        if (op == ZGML_OP_ADD) {
            float o = p.steps[s].secondary[i];
            v = p.steps[s].swapped ? o + v : v + o;
        } else if (op == ZGML_OP_MUL) {
            float o = p.steps[s].secondary[i];
            v = p.steps[s].swapped ? o * v : v * o;
        } else {
            v = apply_unary(op, v);
        }
    }
    p.dst[i] = v;
}

// several elementwise / fused_elementwise ops in a row (same arithmetic, same order, per element)
// the bf16 pieces of x in the A-operand layout of the XDL tile kernels (kernels.h: split_a_pieces — split_a_kernel's arithmetic)
__device__ __forceinline__ void store_a_pieces(uint16_t* ap, uint32_t S, uint32_t m, uint32_t j, float v) {
    if (S & kApF16) { // f16-promoted matmul: one half, round to nearest even (pack_a_f16_kernel's rounding)
        ap[a_f16_index(S & ~kApF16, m, j)] = __half_as_ushort(__float2half_rn(v));
        return;
    }
    uint32_t h[kAPieces];
    split_a_pieces(v, h);
    uint16_t* const d = ap + a_piece_index(S, m, j);
#pragma unroll
    for (int p = 0; p < kAPieces; p++) d[512 * p] = (uint16_t)(h[p] >> 16);
}

__global__ void __launch_bounds__(kBlock) eltwise_chain_kernel(EltChainParams p) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= p.n) return;
    float v = p.src[i];
#pragma unroll 1
    for (uint32_t s = 0; s < p.n_steps; s++) {
        const uint32_t op = p.steps[s].op;
        if (op == ZGML_OP_ADD) {
            float o = p.steps[s].secondary[i];
            v = p.steps[s].swapped ? o + v : v + o;
        } else if (op == ZGML_OP_MUL) {
            float o = p.steps[s].secondary[i];
            v = p.steps[s].swapped ? o * v : v * o;
        } else {
            v = apply_unary(op, v);
        }
        if (p.steps[s].store) p.steps[s].store[i] = v;
    }
    if (p.ap) { // 64 consecutive lanes = 64 consecutive k: 128 contiguous bytes per piece and wave
        const uint32_t m = i / p.ap_cols;
        store_a_pieces(p.ap, p.ap_S, m, i - m * p.ap_cols, v);
    }
}

// reference.zig:309-327 with the finite-shift guard of forward.zig:1306-1322 (all -inf -> zeros)
__global__ void __launch_bounds__(kBlock) softmax_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                         uint32_t cols) {
    __shared__ float red[8];
    const float* s = src + (uint64_t)blockIdx.x * cols;
    float* d = dst + (uint64_t)blockIdx.x * cols;
    float m = -INFINITY;
    for (uint32_t j = threadIdx.x; j < cols; j += kBlock) m = fmaxf(m, s[j]);
    m = block_max(m, red);
    float sum = 0;
    for (uint32_t j = threadIdx.x; j < cols; j += kBlock) {
        float shifted = s[j] - m;
        float e = isfinite(shifted) ? expf(shifted) : 0.0f;
        d[j] = e;
        sum += e;
    }
    sum = block_sum(sum, red);
    float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
    for (uint32_t j = threadIdx.x; j < cols; j += kBlock) d[j] *= inv;
}

// The same chain with every step's second operand requested UP FRONT: the form above reads a step's operand only when the
// chain reaches it, one memory round trip per binary step (silu(gate) * up over 32 x 11008: 2 round trips for 1.4 MB, 5.9 us).
// V = 4: four consecutive elements per thread (16-byte loads and stores, 8-byte piece stores) — only for long chains, a short
// one is latency bound and wants the threads (32 x 11008: 4.68 ms per prefill chunk with V = 4 against 4.62 with V = 1).
// Per element the operations and their order are the scalar kernel's: bit-identical outputs. The launcher takes this form
// only when no operand is an EARLIER step's output (the chain's own stores would have to land first); V = 4 also needs
// n % 4 == 0 and 16-byte aligned pointers.
template <int V>
__global__ void __launch_bounds__(kBlock) eltwise_chain_pre_kernel(EltChainParams p) {
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    if (q * V >= p.n) return;
    float v[V], sec[kMaxChainSteps][V];
    auto load = [&](float (&d)[V], const float* src) {
        if (V == 4) {
            const float4 t = ((const float4*)src)[q];
            d[0] = t.x, d[V > 1 ? 1 : 0] = t.y, d[V > 2 ? 2 : 0] = t.z, d[V > 3 ? 3 : 0] = t.w;
        } else {
            d[0] = src[q];
        }
    };
    load(v, p.src);
#pragma unroll
    for (int s = 0; s < kMaxChainSteps; s++)
        if ((uint32_t)s < p.n_steps && p.steps[s].secondary) load(sec[s], p.steps[s].secondary);
#pragma unroll
    for (int s = 0; s < kMaxChainSteps; s++) {
        if ((uint32_t)s >= p.n_steps) break;
        const uint32_t op = p.steps[s].op;
#pragma unroll
        for (int e = 0; e < V; e++) {
            if (op == ZGML_OP_ADD)
                v[e] = p.steps[s].swapped ? sec[s][e] + v[e] : v[e] + sec[s][e];
            else if (op == ZGML_OP_MUL)
                v[e] = p.steps[s].swapped ? sec[s][e] * v[e] : v[e] * sec[s][e];
            else
                v[e] = apply_unary(op, v[e]);
        }
        if (p.steps[s].store) {
            if (V == 4)
                ((float4*)p.steps[s].store)[q] = make_float4(v[0], v[V > 1 ? 1 : 0], v[V > 2 ? 2 : 0], v[V > 3 ? 3 : 0]);
            else
                p.steps[s].store[q] = v[0];
        }
    }
    if (!p.ap) return;
    const uint32_t i = q * V, m = i / p.ap_cols, j = i - m * p.ap_cols;
    if (V == 1) {
        store_a_pieces(p.ap, p.ap_S, m, j, v[0]);
    } else if (p.ap_S & kApF16) { // ap_cols % 4 == 0: four consecutive k of one row, inside one 8-k group
        uint16_t h[4];
#pragma unroll
        for (int e = 0; e < 4; e++) h[e] = __half_as_ushort(__float2half_rn(v[e % V]));
        *(uint2*)(p.ap + a_f16_index(p.ap_S & ~kApF16, m, j)) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
    } else {
        uint32_t hp[4][kAPieces];
#pragma unroll
        for (int e = 0; e < 4; e++) split_a_pieces(v[e % V], hp[e]); // store_a_pieces' arithmetic
        uint16_t* const d = p.ap + a_piece_index(p.ap_S, m, j);
#pragma unroll
        for (int pc = 0; pc < kAPieces; pc++)
            *(uint2*)(d + 512 * pc) = make_uint2((hp[0][pc] >> 16) | (hp[1][pc] & 0xFFFF0000u), (hp[2][pc] >> 16) | (hp[3][pc] & 0xFFFF0000u));
    }
}

// reference.zig:329-347
__global__ void __launch_bounds__(kBlock) layernorm_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                           uint32_t cols, float eps) {
    __shared__ float red[8];
    const float* s = src + (uint64_t)blockIdx.x * cols;
    float* d = dst + (uint64_t)blockIdx.x * cols;
    float mu = 0;
    for (uint32_t j = threadIdx.x; j < cols; j += kBlock) mu += s[j];
    mu = block_sum(mu, red) / (float)cols;
    float v = 0;
    for (uint32_t j = threadIdx.x; j < cols; j += kBlock) {
        float diff = s[j] - mu;
        v += diff * diff;
    }
    v = block_sum(v, red);
    float inv_std = 1.0f / sqrtf(v / (float)cols + eps);
    for (uint32_t j = threadIdx.x; j < cols; j += kBlock) d[j] = (s[j] - mu) * inv_std;
}

// reference.zig:349-374: inv = 1/sqrt(ss/cols + eps)
__global__ void __launch_bounds__(kBlock) rmsnorm_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                         uint32_t cols, float eps) {
    __shared__ float red[8];
    const float* s = src + (uint64_t)blockIdx.x * cols;
    float* d = dst + (uint64_t)blockIdx.x * cols;
    float ss = 0;
    for (uint32_t j = threadIdx.x; j < cols; j += kBlock) ss += s[j] * s[j];
    ss = block_sum(ss, red);
    float inv = 1.0f / sqrtf(ss / (float)cols + eps);
    for (uint32_t j = threadIdx.x; j < cols; j += kBlock) d[j] = s[j] * inv;
}

// [add ->] rmsnorm [-> mul] of one row per workgroup: the same element-to-thread mapping and the same block
// reduction as elementwise_kernel / rmsnorm_kernel, so every stored value is bit-identical to the three-launch
// form. NPT > 0: the row fits NPT elements per thread — everything is loaded up front with independent loads
// (a row is one dependent chain otherwise: 32 rows x 4096 columns took ~25 us) and kept in registers between
// the two passes. NPT == 0: any width, looping.
// gridDim.y > 1 (NPT > 0, few rows: a 32-row prefill chunk is 32 workgroups otherwise): every workgroup of a row loads the whole
// row and forms the same sum in the same order, but stores — the add's result, the normalised row, the product, the A pieces —
// only for its share of the thread-strided column chunks (chunk k = columns 256 k .. 256 k + 255): the stores and the
// piece transpose are what the launch spends its time on behind the first round trip.
template <int NPT>
__global__ void __launch_bounds__(kBlock) row_chain_kernel(RowChainParams p) {
    __shared__ float red[8];
    const uint64_t base = (uint64_t)blockIdx.x * p.cols;
    const float* s = p.src + base;
    float* nd = p.norm_dst + base;
    float ss = 0;
    if (NPT > 0) {
        constexpr int N = NPT > 0 ? NPT : 1;
        __shared__ __attribute__((aligned(16))) uint16_t pl[kAPieces * N * kBlock]; // A pieces of the row (RowChainParams::ap)
        float v[N], mo[N];
        const uint32_t last = p.cols - 1;
        const uint32_t chunks = (p.cols + kBlock - 1) / kBlock; // (<= N; the launcher's split divides it)
        const uint32_t k_lo = blockIdx.y * chunks / gridDim.y, k_hi = (blockIdx.y + 1) * chunks / gridDim.y; // this workgroup's column chunks
        if (p.add_dst) {
            const float *a0 = p.a0 + base, *a1 = p.a1 + base;
            float w[N];
#pragma unroll
            for (int k = 0; k < N; k++) v[k] = a0[min(threadIdx.x + k * kBlock, last)];
#pragma unroll
            for (int k = 0; k < N; k++) w[k] = a1[min(threadIdx.x + k * kBlock, last)];
#pragma unroll
            for (int k = 0; k < N; k++) v[k] = v[k] + w[k];
        } else {
#pragma unroll
            for (int k = 0; k < N; k++) v[k] = s[min(threadIdx.x + k * kBlock, last)];
        }
        if (p.mul_dst) {
#pragma unroll
            for (int k = 0; k < N; k++) mo[k] = p.mul_other[base + min(threadIdx.x + k * kBlock, last)];
        }
#pragma unroll
        for (int k = 0; k < N; k++) {
            const uint32_t j = threadIdx.x + k * kBlock;
            if (j < p.cols) {
                if (p.add_dst && (uint32_t)k >= k_lo && (uint32_t)k < k_hi) p.add_dst[base + j] = v[k];
                ss += v[k] * v[k];
            }
        }
        ss = block_sum(ss, red);
        const float inv = 1.0f / sqrtf(ss / (float)p.cols + p.eps);
#pragma unroll
        for (int k = 0; k < N; k++) {
            const uint32_t j = threadIdx.x + k * kBlock;
            if (j < p.cols && (uint32_t)k >= k_lo && (uint32_t)k < k_hi) {
                const float nv = v[k] * inv;
                nd[j] = nv;
                const float xv = p.mul_dst ? nv * mo[k] : nv;
                if (p.mul_dst) p.mul_dst[base + j] = xv;
                if (p.ap && (p.ap_S & kApF16)) { // f16 A operand: one half per element
                    pl[j] = __half_as_ushort(__float2half_rn(xv));
                } else if (p.ap) { // the bf16 pieces (kernels.h: split_a_pieces), transposed through LDS so that they leave as 16-byte stores
                    uint32_t hp[kAPieces];
                    split_a_pieces(xv, hp);
#pragma unroll
                    for (int pc = 0; pc < kAPieces; pc++) pl[pc * N * kBlock + j] = (uint16_t)(hp[pc] >> 16);
                }
            }
        }
        if (p.ap) { // (uniform) cols % 128 == 0 here (planner)
            __syncthreads();
            const uint32_t c_lo = min(k_lo * kBlock, p.cols), c_hi = min(k_hi * kBlock, p.cols), groups = (c_hi - c_lo) / 8; // (cols % 128 == 0)
            const bool f16 = (p.ap_S & kApF16) != 0;
            for (uint32_t u = threadIdx.x; u < (f16 ? 1u : (uint32_t)kAPieces) * groups; u += kBlock) {
                const uint32_t piece = u / groups, j0 = c_lo + (u - piece * groups) * 8;
                const uint4 val = *(const uint4*)(pl + piece * (N * kBlock) + j0);
                uint16_t* const d = f16 ? p.ap + a_f16_index(p.ap_S & ~kApF16, blockIdx.x, j0) : p.ap + a_piece_index(p.ap_S, blockIdx.x, j0) + piece * 512;
                *(uint4*)d = val;
            }
        }
        return;
    }
    if (p.add_dst) {
        const float *a0 = p.a0 + base, *a1 = p.a1 + base;
        float* ad = p.add_dst + base;
        for (uint32_t j = threadIdx.x; j < p.cols; j += kBlock) {
            const float v = a0[j] + a1[j];
            ad[j] = v;
            ss += v * v;
        }
    } else {
        for (uint32_t j = threadIdx.x; j < p.cols; j += kBlock) ss += s[j] * s[j];
    }
    ss = block_sum(ss, red);
    const float inv = 1.0f / sqrtf(ss / (float)p.cols + p.eps);
    // a thread re-reads only the elements it wrote itself above (same j): no barrier needed
    if (p.mul_dst) {
        const float* mo = p.mul_other + base;
        float* md = p.mul_dst + base;
        for (uint32_t j = threadIdx.x; j < p.cols; j += kBlock) {
            const float nv = s[j] * inv;
            nd[j] = nv;
            md[j] = nv * mo[j];
            if (p.ap) store_a_pieces(p.ap, p.ap_S, blockIdx.x, j, nv * mo[j]);
        }
    } else {
        for (uint32_t j = threadIdx.x; j < p.cols; j += kBlock) {
            nd[j] = s[j] * inv;
            if (p.ap) store_a_pieces(p.ap, p.ap_S, blockIdx.x, j, s[j] * inv);
        }
    }
}

// reference.zig:376-389: one wave per output
__global__ void __launch_bounds__(kBlock) reduce_kernel(uint32_t op, float* __restrict__ dst,
                                                        const float* __restrict__ src, uint32_t n_out,
                                                        uint32_t reduce_size) {
    uint32_t out = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (out >= n_out) return;
    const float* s = src + (uint64_t)out * reduce_size;
    uint32_t lane = threadIdx.x & 63;
    if (op == ZGML_OP_MAX) {
        float v = -INFINITY;
        for (uint32_t k = lane; k < reduce_size; k += 64) v = fmaxf(v, s[k]);
        v = wave_max(v);
        if (lane == 0) dst[out] = v;
    } else {
        float v = 0;
        for (uint32_t k = lane; k < reduce_size; k += 64) v += s[k];
        v = wave_sum(v);
        if (lane == 0) dst[out] = v;
    }
}

// reference.zig:391-433 (bit-exact copy semantics; the three fast paths are special cases of the
// generic index math and are kept only because they avoid the div/mod chain)
__global__ void __launch_bounds__(kBlock) repeat_kernel(const RepeatParams* __restrict__ params) {
    const RepeatParams& p = params[blockIdx.y];
    uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    if (gid >= p.n) return;
    float* d = p.dst + p.dst_offset;
    const float* s = p.src + p.src_offset;
    if (p.mode == 1) {
        d[gid] = s[0];
    } else if (p.mode == 2) {
        d[gid] = s[gid];
    } else if (p.mode == 3) {
        d[gid] = s[gid % p.src_n];
    } else {
        uint32_t idx = gid, src_idx = p.src_offset;
#pragma unroll
        for (int dim = 3; dim >= 0; dim--) {
            uint32_t coord = idx / p.dst_strides[dim];
            idx = idx % p.dst_strides[dim];
            src_idx += (coord % p.src_ne[dim]) * p.src_strides[dim];
        }
        d[gid] = p.src[src_idx];
    }
}

// reference.zig:435-455; dst_offset is read from the program's dynamic-parameter block so a
// captured graph stays valid across KV positions.
__device__ __forceinline__ void slice_assign_body(const SliceAssignParams& p, uint32_t gid) {
    if (gid >= p.rows * p.cols) return;
    uint32_t row = gid % p.rows, col = gid / p.rows;
    uint32_t doff = *p.dyn_dst_offset;
    p.dst[(uint64_t)doff + (uint64_t)row * p.dst_row_stride + (uint64_t)col * p.dst_col_stride] =
        p.src[(uint64_t)row * p.src_row_stride + (uint64_t)col * p.src_col_stride];
}

// reference.zig:457-478 (DeviceOp convention: sin at cs + pair + half_d). With dst2 set the rotated
// values are also stored through the fused slice_assign (src = this rope's dense [2*half_d, seq]
// output: row = pair or pair + half_d, col = col).
__device__ __forceinline__ void rope_body(const RopeParams& p, uint32_t gid) {
    if (gid >= p.half_d * p.seq_len) return;
    uint32_t pair = gid % p.half_d, col = gid / p.half_d;
    float x_lo = p.src[(uint64_t)pair * p.src_rs + (uint64_t)col * p.src_cs];
    float x_hi = p.src[(uint64_t)(pair + p.half_d) * p.src_rs + (uint64_t)col * p.src_cs];
    float c = p.cs[pair + (uint64_t)col * p.cs_cs];
    float s = p.cs[pair + p.half_d + (uint64_t)col * p.cs_cs];
    const float lo = x_lo * c - x_hi * s, hi = x_hi * c + x_lo * s;
    p.dst[pair + (uint64_t)col * 2 * p.half_d] = lo;
    p.dst[pair + p.half_d + (uint64_t)col * 2 * p.half_d] = hi;
    if (p.dst2) {
        const uint64_t doff = *p.dyn_dst2_off;
        p.dst2[doff + (uint64_t)pair * p.d2_rs + (uint64_t)col * p.d2_cs] = lo;
        p.dst2[doff + (uint64_t)(pair + p.half_d) * p.d2_rs + (uint64_t)col * p.d2_cs] = hi;
    }
}

__global__ void __launch_bounds__(kBlock) move_kernel(const MoveParams* __restrict__ params) {
    const MoveParams& p = params[blockIdx.y];
    const uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    if (gid >= p.n_elems) return;
    if (p.kind == 0)
        rope_body(p.rope, gid);
    else
        slice_assign_body(p.sa, gid);
}

// reference.zig:568-672. One workgroup per query column; keys are processed in tiles of 256 with
// an online softmax (so there is no seq_kv cap, unlike the 4096-entry score buffers of the Metal
// and WGSL kernels). Masked (non-finite mask) and non-finite-score keys are skipped exactly as the
// reference does; a query with no valid key yields zeros.
// head output element (row r of query column qi), plus the fused row store when present
__device__ __forceinline__ void store_out(const AttentionParams& p, uint32_t qi, uint32_t r, float v) {
    p.dst[(uint64_t)qi * p.dst_cs + (uint64_t)r * p.dst_rs] = v;
    if (p.dst2) p.dst2[(uint64_t)*p.dyn_dst2_off + (uint64_t)r * p.d2_rs + (uint64_t)qi * p.d2_cs] = v;
}

__global__ void __launch_bounds__(kBlock) attention_kernel(const AttentionParams* __restrict__ params) {
    const AttentionParams p = params[blockIdx.y];
    if (blockIdx.x >= p.seq_q) return; // batched ops may have fewer queries than the grid
    __shared__ float q_s[512];
    __shared__ float w_s[kBlock];
    __shared__ float red[8];
    __shared__ float acc_s[kBlock]; // cross-group reduction of the V accumulation
    const uint32_t qi = blockIdx.x, tid = threadIdx.x, dh = p.d_head;
    const uint32_t seq_kv = *p.dyn_seq_kv;
    for (uint32_t r = tid; r < dh; r += kBlock) q_s[r] = p.q[(uint64_t)qi * p.q_cs + (uint64_t)r * p.q_rs];
    __syncthreads();

    // V accumulation layout: dh <= 256 -> G = 256/dh key groups, thread (g, r); else 2 rows/thread.
    const bool grouped = dh <= kBlock;
    const uint32_t G = grouped ? kBlock / dh : 1;
    const uint32_t g = grouped ? tid / dh : 0;
    const uint32_t r0 = grouped ? tid % dh : tid;
    const bool v_active = grouped ? (g < G) : true;
    float acc0 = 0.f, acc1 = 0.f;
    float m = -INFINITY, l = 0.f;

    for (uint32_t s0 = 0; s0 < seq_kv; s0 += kBlock) {
        uint32_t s = s0 + tid;
        bool valid = s < seq_kv;
        float score = -INFINITY;
        if (valid) {
            float mask_add = p.mask ? p.mask[(uint64_t)qi * p.mask_cs + (uint64_t)s * p.mask_rs] : 0.0f;
            if (isfinite(mask_add)) {
                const float* kp = p.k + (uint64_t)s * p.k_cs;
                float dot = 0.f;
                for (uint32_t r = 0; r < dh; r++) dot += q_s[r] * kp[(uint64_t)r * p.k_rs];
                score = dot * p.scale + mask_add;
                if (!isfinite(score)) score = -INFINITY;
            }
        }
        valid = score > -INFINITY;
        float tile_max = block_max(score, red);
        float new_m = fmaxf(m, tile_max);
        if (new_m == -INFINITY) continue; // uniform: nothing valid so far
        float alpha = (m == -INFINITY) ? 0.0f : expf(m - new_m);
        float w = valid ? expf(score - new_m) : 0.0f;
        w_s[tid] = w;
        float tile_sum = block_sum(w, red); // contains the barrier that publishes w_s
        l = l * alpha + tile_sum;
        m = new_m;
        uint32_t tile_n = min((uint32_t)kBlock, seq_kv - s0);
        if (v_active) {
            if (grouped) {
                float a = acc0 * alpha;
                for (uint32_t t = g; t < tile_n; t += G)
                    a += w_s[t] * p.v[(uint64_t)r0 * p.v_rs + (uint64_t)(s0 + t) * p.v_cs];
                acc0 = a;
            } else {
                float a0 = acc0 * alpha, a1 = acc1 * alpha;
                for (uint32_t t = 0; t < tile_n; t++) {
                    float wt = w_s[t];
                    if (r0 < dh) a0 += wt * p.v[(uint64_t)r0 * p.v_rs + (uint64_t)(s0 + t) * p.v_cs];
                    if (r0 + kBlock < dh)
                        a1 += wt * p.v[(uint64_t)(r0 + kBlock) * p.v_rs + (uint64_t)(s0 + t) * p.v_cs];
                }
                acc0 = a0;
                acc1 = a1;
            }
        }
        __syncthreads(); // w_s reused next tile
    }

    float inv_l = l > 0.f ? 1.0f / l : 0.0f;
    if (grouped) {
        acc_s[tid] = v_active ? acc0 : 0.f;
        __syncthreads();
        if (tid < dh) {
            float a = 0.f;
            for (uint32_t gg = 0; gg < G; gg++) a += acc_s[gg * dh + tid];
            store_out(p, qi, tid, a * inv_l);
        }
    } else {
        if (r0 < dh) store_out(p, qi, r0, acc0 * inv_l);
        if (r0 + kBlock < dh) store_out(p, qi, r0 + kBlock, acc1 * inv_l);
    }
}

// Decode-shaped fast path: unit row strides on q/k/v/dst and d_head a power of two in [4, 256]
// (checked on the host). A key is handled by LPK = d_head/4 adjacent lanes holding one float4 each
// (one coalesced row read per key), so a wave scores 64/LPK keys per iteration and the 16 waves of
// the workgroup stride over the keys. The loop is unrolled by kAttnUnroll with all K (and later all V)
// row loads issued before the first use: at decode lengths the kernel is a chain of dependent
// memory round trips, not bandwidth. Scores of a tile (<= kTile keys) live in LDS; tiles chain with
// the online-softmax rescale, so seq_kv is unbounded. P.V uses the same lane layout (float4 of the
// head dimension per lane), folded across lane groups by shuffles and across waves through LDS.
constexpr uint32_t kTile = 2048;

__device__ __forceinline__ float block_max_n(float v, float* red, int n_waves) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int i = 0; i < n_waves; i++) t = fmaxf(t, red[i]);
    return t;
}
__device__ __forceinline__ float block_sum_n(float v, float* red, int n_waves) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < n_waves; i++) t += red[i];
    return t;
}

__global__ void __launch_bounds__(kAttnBlock) attention_dense_kernel(const AttentionParams* __restrict__ params) {
    const AttentionParams p = params[blockIdx.y];
    if (blockIdx.x >= p.seq_q) return;
    __shared__ float sc_s[kTile];
    __shared__ float red[32];
    __shared__ float4 acc_s[kAttnBlock / 64 * 64];
    constexpr int NW = kAttnBlock / 64;
    const uint32_t qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, dh = p.d_head;
    const uint32_t LPK = dh >> 2, KPW = 64 / LPK, sub = lane / LPK, li = lane % LPK;
    const uint32_t seq_kv = *p.dyn_seq_kv;
    const float4 qv = *(const float4*)(p.q + (uint64_t)qi * p.q_cs + 4 * li);
    const uint32_t keys_per_iter = KPW * NW;
    float m = -INFINITY, l = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

    for (uint32_t t0 = 0; t0 < seq_kv; t0 += kTile) {
        const uint32_t tn = min(kTile, seq_kv - t0);
        // scores
        float tmax = -INFINITY;
        for (uint32_t base = 0; base < tn; base += keys_per_iter * kAttnUnroll) {
            float4 kv[kAttnUnroll];
            float mk[kAttnUnroll];
#pragma unroll
            for (int j = 0; j < kAttnUnroll; j++) { // all loads first (clamped, unconditional)
                const uint32_t t = base + j * keys_per_iter + w * KPW + sub;
                const uint32_t s = t0 + min(t, tn - 1);
                kv[j] = *(const float4*)(p.k + (uint64_t)s * p.k_cs + 4 * li);
                mk[j] = p.mask ? p.mask[(uint64_t)qi * p.mask_cs + (uint64_t)s * p.mask_rs] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < kAttnUnroll; j++) {
                const uint32_t t = base + j * keys_per_iter + w * KPW + sub;
                float dot = qv.x * kv[j].x + qv.y * kv[j].y + qv.z * kv[j].z + qv.w * kv[j].w;
                for (uint32_t off = LPK >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
                float score = -INFINITY;
                if (t < tn) {
                    if (isfinite(mk[j])) {
                        score = dot * p.scale + mk[j];
                        if (!isfinite(score)) score = -INFINITY;
                    }
                    if (li == 0) sc_s[t] = score;
                }
                tmax = fmaxf(tmax, score);
            }
        }
        tmax = block_max_n(tmax, red, NW); // barriers also publish sc_s
        const float new_m = fmaxf(m, tmax);
        if (new_m == -INFINITY) continue; // uniform: no valid key yet
        const float alpha = (m == -INFINITY) ? 0.0f : expf(m - new_m);
        // weights (in place) and their sum
        float tsum = 0.f;
        for (uint32_t t = tid; t < tn; t += kAttnBlock) {
            const float sc = sc_s[t];
            const float wgt = sc > -INFINITY ? expf(sc - new_m) : 0.0f;
            sc_s[t] = wgt;
            tsum += wgt;
        }
        tsum = block_sum_n(tsum, red, NW); // barriers publish the weights
        l = l * alpha + tsum;
        m = new_m;
        // P.V
        acc = make_float4(acc.x * alpha, acc.y * alpha, acc.z * alpha, acc.w * alpha);
        for (uint32_t base = 0; base < tn; base += keys_per_iter * kAttnUnroll) {
            float4 vv[kAttnUnroll];
#pragma unroll
            for (int j = 0; j < kAttnUnroll; j++) {
                const uint32_t t = base + j * keys_per_iter + w * KPW + sub;
                vv[j] = *(const float4*)(p.v + (uint64_t)(t0 + min(t, tn - 1)) * p.v_cs + 4 * li);
            }
#pragma unroll
            for (int j = 0; j < kAttnUnroll; j++) {
                const uint32_t t = base + j * keys_per_iter + w * KPW + sub;
                const float wgt = t < tn ? sc_s[t] : 0.0f;
                acc.x += wgt * vv[j].x;
                acc.y += wgt * vv[j].y;
                acc.z += wgt * vv[j].z;
                acc.w += wgt * vv[j].w;
            }
        }
        __syncthreads(); // sc_s reused by the next tile
    }
    // fold the KPW lane groups of each wave, then the waves
    for (uint32_t off = LPK; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off, 64);
        acc.y += __shfl_xor(acc.y, off, 64);
        acc.z += __shfl_xor(acc.z, off, 64);
        acc.w += __shfl_xor(acc.w, off, 64);
    }
    acc_s[tid] = acc;
    __syncthreads();
    if (tid < LPK) {
        float4 a = acc_s[tid];
        for (int ww = 1; ww < NW; ww++) {
            const float4 b = acc_s[ww * 64 + tid];
            a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
        }
        const float inv_l = l > 0.f ? 1.0f / l : 0.0f;
        store_out(p, qi, 4 * tid + 0, a.x * inv_l);
        store_out(p, qi, 4 * tid + 1, a.y * inv_l);
        store_out(p, qi, 4 * tid + 2, a.z * inv_l);
        store_out(p, qi, 4 * tid + 3, a.w * inv_l);
    }
}

// ── attention of seq_q > 1 query columns (prefill), dense layouts ─────────────────────────────
// attention_dense_kernel spends a 1024-thread workgroup with a score buffer and four block reductions per key
// tile on every (query, head): 23 us per layer for a 32-token chunk whose queries see 32 keys. This is the
// decode kernel's streaming scheme without its fused ropes / cache stores: a key is LPK lanes holding one
// float4 of the head dimension each, every key slot runs its own online softmax, waves the context does not
// need retire at the top, one barrier, two-pass merges. One workgroup per (query column, op).
template <int LPK>
__global__ void __launch_bounds__(kAttnBlock) attention_rows_kernel(const AttentionParams* __restrict__ params, const float* __restrict__ zero_word) {
    constexpr int KPW = 64 / LPK, U = kAttnUnroll;
    const AttentionParams& p = params[blockIdx.y];
    const uint32_t qi = blockIdx.x;
    if (qi >= p.seq_q) return; // batched ops may have fewer queries than the grid
    __shared__ __attribute__((aligned(8))) float part_ml[2 * (kAttnBlock / 64)];
    __shared__ float4 part_acc[(kAttnBlock / 64) * LPK];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t slot = lane / LPK, li = lane % LPK, d0 = 4 * li;
    const uint32_t seq_kv = *p.dyn_seq_kv;
    const uint32_t d2_off = p.dst2 ? *p.dyn_dst2_off : 0;
    const float4 qv = *(const float4*)(p.q + (uint64_t)qi * p.q_cs + d0);
    const float* mask = p.mask ? p.mask + (uint64_t)qi * p.mask_cs : zero_word; // no mask: a zero word with stride 0
    const uint32_t mask_rs = p.mask ? p.mask_rs : 0;
    uint32_t NW = (seq_kv + KPW * U - 1) / (KPW * U);
    const uint32_t max_w = blockDim.x >> 6; // 4 when the grid alone fills the chip (many queries), else 16
    NW = NW < 1 ? 1 : (NW > max_w ? max_w : NW);
    if (w >= NW) return;
    const uint32_t keys_per_iter = KPW * NW, step_keys = keys_per_iter * U, last = seq_kv ? seq_kv - 1 : 0;
    float4 kv[U], vv[U];
    float mk[U];
#pragma unroll
    for (int j = 0; j < U; j++) { // clamped to live rows, unconditional
        const uint32_t s = min(j * keys_per_iter + w * KPW + slot, last);
        kv[j] = *(const float4*)(p.k + (uint64_t)s * p.k_cs + d0);
        vv[j] = *(const float4*)(p.v + (uint64_t)s * p.v_cs + d0);
        mk[j] = mask[(uint64_t)s * mask_rs];
    }
    SoftState st{-INFINITY, 0.f, make_float4(0.f, 0.f, 0.f, 0.f)};
    auto step = [&](uint32_t base) {
        float sc[U];
        float bm = -INFINITY;
#pragma unroll
        for (int j = 0; j < U; j++) {
            const uint32_t t = base + j * keys_per_iter + w * KPW + slot;
            const float dot = group_sum<LPK>(qv.x * kv[j].x + qv.y * kv[j].y + qv.z * kv[j].z + qv.w * kv[j].w);
            sc[j] = t < seq_kv ? score_of(dot, mk[j], p.scale) : -INFINITY;
            bm = fmaxf(bm, sc[j]);
        }
        const float nm = fmaxf(st.m, bm);
        if (nm > -INFINITY) {
            const float alpha = st.m > -INFINITY ? expf(st.m - nm) : 0.0f;
            st.l *= alpha;
            st.acc = make_float4(st.acc.x * alpha, st.acc.y * alpha, st.acc.z * alpha, st.acc.w * alpha);
#pragma unroll
            for (int j = 0; j < U; j++) {
                const float wgt = sc[j] > -INFINITY ? expf(sc[j] - nm) : 0.0f;
                st.l += wgt;
                if (sc[j] > -INFINITY) { // rows of dead slots are never touched
                    st.acc.x += wgt * vv[j].x;
                    st.acc.y += wgt * vv[j].y;
                    st.acc.z += wgt * vv[j].z;
                    st.acc.w += wgt * vv[j].w;
                }
            }
            st.m = nm;
        }
    };
    if (seq_kv <= step_keys) {
        if (seq_kv) step(0);
    } else {
        for (uint32_t base = 0; base < seq_kv; base += step_keys) {
            float4 kn[U], vn[U];
            float mn[U];
#pragma unroll
            for (int j = 0; j < U; j++) {
                const uint32_t s = min(base + step_keys + j * keys_per_iter + w * KPW + slot, last);
                kn[j] = *(const float4*)(p.k + (uint64_t)s * p.k_cs + d0);
                vn[j] = *(const float4*)(p.v + (uint64_t)s * p.v_cs + d0);
                mn[j] = mask[(uint64_t)s * mask_rs];
            }
            step(base);
#pragma unroll
            for (int j = 0; j < U; j++) kv[j] = kn[j], vv[j] = vn[j], mk[j] = mn[j];
        }
    }
    slots_merge<LPK>(st);
    SoftState r = st;
    if (NW > 1) {
        if (lane < LPK) {
            part_acc[w * LPK + lane] = st.acc;
            if (lane == 0) part_ml[2 * w] = st.m, part_ml[2 * w + 1] = st.l;
        }
        __syncthreads();
        if (w != 0) return;
        constexpr int MAXW = kAttnBlock / 64, NPS = MAXW / KPW > 0 ? MAXW / KPW : 1;
        float M = lane < NW ? part_ml[2 * lane] : -INFINITY;
        float2 ml[NPS];
        float4 pa[NPS];
#pragma unroll
        for (int i = 0; i < NPS; i++) {
            const uint32_t ww = slot + i * KPW, wc = ww < NW ? ww : 0;
            ml[i] = *(const float2*)&part_ml[2 * wc];
            pa[i] = part_acc[wc * LPK + li];
        }
        M = fmaxf(M, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(M), 0xB1, 0xF, 0xF, true)));
        M = fmaxf(M, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(M), 0x4E, 0xF, 0xF, true)));
        M = fmaxf(M, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(M), 0x141, 0xF, 0xF, true)));
        M = fmaxf(M, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(M), 0x140, 0xF, 0xF, true)));
        M = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(M)));
        r = SoftState{M, 0.f, make_float4(0.f, 0.f, 0.f, 0.f)};
#pragma unroll
        for (int i = 0; i < NPS; i++) {
            const bool live = slot + i * KPW < NW && ml[i].x > -INFINITY;
            const float f = live ? expf(ml[i].x - M) : 0.0f;
            r.l += ml[i].y * f;
            r.acc.x += pa[i].x * f, r.acc.y += pa[i].y * f, r.acc.z += pa[i].z * f, r.acc.w += pa[i].w * f;
        }
        slots_sum<LPK>(r);
    }
    if (lane < LPK) {
        const float inv_l = r.l > 0.f ? 1.0f / r.l : 0.0f;
        const float o[4] = {r.acc.x * inv_l, r.acc.y * inv_l, r.acc.z * inv_l, r.acc.w * inv_l};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            p.dst[(uint64_t)qi * p.dst_cs + (uint64_t)(4 * lane + e) * p.dst_rs] = o[e];
            if (p.dst2) p.dst2[(uint64_t)d2_off + (uint64_t)(4 * lane + e) * p.d2_rs + (uint64_t)qi * p.d2_cs] = o[e];
        }
    }
}

// ── quantised KV cache (extension ops; src/quant.zig:645-1091) ───────────────────────────────
// storeColumn: one wave per op; lane r handles elements r, r+64, ... of each block. Same arithmetic
// as quantizeInput (:320-341): scale = absmax/127 (1 if the block is all zero), q = trunc(clamp(v * (127/absmax))).
__global__ void __launch_bounds__(64) kvq_store_kernel(const KvqStoreParams* __restrict__ params) {
    const KvqStoreParams& p = params[blockIdx.x];
    const uint32_t col = *p.dyn_col;
    if (col >= p.n_cols) return; // never write outside the cache
    const uint32_t dh = p.d_head, bs = p.block_size, bpc = dh / bs, lane = threadIdx.x;
    int8_t* qd = (int8_t*)p.cache + (uint64_t)col * dh;
    float* sc = p.cache + (uint64_t)p.n_cols * dh / 4 + (uint64_t)col * bpc;
    for (uint32_t b = 0; b < bpc; b++) {
        float mx = 0.f;
        for (uint32_t i = lane; i < bs; i += 64) mx = fmaxf(mx, fabsf(p.src[b * bs + i]));
        mx = wave_max(mx);
        const float scale = mx > 0.f ? mx / 127.0f : 1.0f, inv = mx > 0.f ? 127.0f / mx : 0.0f;
        if (lane == 0) sc[b] = scale;
        for (uint32_t i = lane; i < bs; i += 64) {
            const float v = fminf(fmaxf(p.src[b * bs + i] * inv, -127.0f), 127.0f);
            qd[b * bs + i] = (int8_t)(int)v; // @intFromFloat truncates toward zero
        }
    }
}

// attentionQuantized: one workgroup per (query, op), shaped like attention_decode_kernel: a key is
// handled by LPK = d_head/4 lanes, each dequantising 4 consecutive dims of the int8 row (one 4-byte
// load); a wave streams KPW keys x U per step with the next step's rows prefetched; waves the context
// does not need retire at the top; per-slot online softmax merged at the end (one barrier). Only the
// summation / rescale order differs from the reference's Bs = 8 flash tiles.
template <int LPK>
__global__ void __launch_bounds__(kAttnBlock) kvq_attention_kernel(const KvqAttentionParams* __restrict__ params, float* split_buf, uint32_t* split_cnt,
                                                                   uint32_t split_min_keys) {
    constexpr int DH = 4 * LPK, KPW = 64 / LPK, U = kAttnUnroll, MAXW = kAttnBlock / 64;
    const KvqAttentionParams& p = params[blockIdx.y];
    if (blockIdx.x >= p.seq_q) return;
    __shared__ __attribute__((aligned(8))) float part_ml[2 * MAXW];
    __shared__ float4 part_acc[MAXW * LPK];
    const uint32_t qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t slot = lane / LPK, li = lane % LPK, d0 = 4 * li;
    uint32_t seq_kv = *p.dyn_seq_kv;
    const uint32_t room_k = p.n_cols > p.k_col_start ? p.n_cols - p.k_col_start : 0, room_v = p.n_cols > p.v_col_start ? p.n_cols - p.v_col_start : 0;
    seq_kv = min(seq_kv, min(room_k, room_v)); // never read outside the caches
    // long contexts: gridDim.z workgroups per (query, op) split the keys exactly as attention_decode_kernel does
    const uint32_t sp = blockIdx.z;
    uint32_t n_active = 1, k_begin = 0, k_end = seq_kv;
    if (gridDim.z > 1) {
        n_active = seq_kv / split_min_keys;
        n_active = n_active < 1 ? 1 : (n_active > gridDim.z ? gridDim.z : n_active);
        if (sp >= n_active) return;
        const uint32_t chunk = (seq_kv + n_active - 1) / n_active;
        k_begin = min(sp * chunk, seq_kv), k_end = min(k_begin + chunk, seq_kv);
    }
    const uint32_t n_keys = k_end - k_begin;
    uint32_t NW = (n_keys + KPW * U - 1) / (KPW * U);
    NW = NW < 1 ? 1 : (NW > (uint32_t)MAXW ? (uint32_t)MAXW : NW);
    if (w >= NW) return;
    const uint32_t keys_per_iter = KPW * NW, step_keys = keys_per_iter * U, last = k_end ? k_end - 1 : 0;
    const uint32_t bpc = DH / p.block_size, blk = d0 / p.block_size;
    const int8_t* kq = (const int8_t*)p.k_cache + (uint64_t)p.k_col_start * DH + d0;
    const float* ks = p.k_cache + (uint64_t)p.n_cols * DH / 4 + (uint64_t)p.k_col_start * bpc + blk;
    const int8_t* vq = (const int8_t*)p.v_cache + (uint64_t)p.v_col_start * DH + d0;
    const float* vs = p.v_cache + (uint64_t)p.n_cols * DH / 4 + (uint64_t)p.v_col_start * bpc + blk;
    const float* mask = p.mask ? p.mask + (uint64_t)qi * p.mask_cs : nullptr;
    const float4 qv = *(const float4*)(p.q + (uint64_t)qi * p.q_cs + d0);
    struct Rows {
        uint32_t kw[U], vw[U];
        float ksc[U], vsc[U], mk[U];
    };
    auto load = [&](Rows& r, uint32_t base) { // clamped to live rows, unconditional
#pragma unroll
        for (int j = 0; j < U; j++) {
            const uint64_t t = min(base + j * keys_per_iter + w * KPW + slot, last);
            r.kw[j] = *(const uint32_t*)(kq + t * DH), r.vw[j] = *(const uint32_t*)(vq + t * DH);
            r.ksc[j] = ks[t * bpc], r.vsc[j] = vs[t * bpc];
            r.mk[j] = mask ? mask[t * p.mask_rs] : 0.0f;
        }
    };
    SoftState st{-INFINITY, 0.f, make_float4(0.f, 0.f, 0.f, 0.f)};
    auto step = [&](const Rows& r, uint32_t base) {
        float sc[U];
        float bm = -INFINITY;
#pragma unroll
        for (int j = 0; j < U; j++) {
            const uint32_t t = base + j * keys_per_iter + w * KPW + slot, kw = r.kw[j];
            // dequantised dot: the block scale applies to this lane's 4 dims (block_size % 4 == 0)
            float dot = (qv.x * (float)(int8_t)(kw & 255) + qv.y * (float)(int8_t)((kw >> 8) & 255) +
                         qv.z * (float)(int8_t)((kw >> 16) & 255) + qv.w * (float)(int8_t)(kw >> 24)) * r.ksc[j];
            dot = group_sum<LPK>(dot);
            sc[j] = t < k_end ? score_of(dot, r.mk[j], p.scale) : -INFINITY;
            bm = fmaxf(bm, sc[j]);
        }
        const float nm = fmaxf(st.m, bm);
        if (nm > -INFINITY) {
            const float alpha = st.m > -INFINITY ? expf(st.m - nm) : 0.0f;
            st.l *= alpha;
            st.acc = make_float4(st.acc.x * alpha, st.acc.y * alpha, st.acc.z * alpha, st.acc.w * alpha);
#pragma unroll
            for (int j = 0; j < U; j++) {
                if (sc[j] > -INFINITY) {
                    const float wgt = expf(sc[j] - nm), ws = wgt * r.vsc[j];
                    const uint32_t vw = r.vw[j];
                    st.l += wgt;
                    st.acc.x += ws * (float)(int8_t)(vw & 255);
                    st.acc.y += ws * (float)(int8_t)((vw >> 8) & 255);
                    st.acc.z += ws * (float)(int8_t)((vw >> 16) & 255);
                    st.acc.w += ws * (float)(int8_t)(vw >> 24);
                }
            }
            st.m = nm;
        }
    };
    Rows cur;
    load(cur, k_begin);
    if (n_keys <= step_keys) {
        if (n_keys) step(cur, k_begin);
    } else {
        for (uint32_t base = k_begin; base < k_end; base += step_keys) {
            Rows nxt;
            load(nxt, base + step_keys);
            step(cur, base);
            cur = nxt;
        }
    }
    slots_merge<LPK>(st);
    SoftState r = st;
    if (NW > 1) {
        if (lane < LPK) {
            part_acc[w * LPK + lane] = st.acc;
            if (lane == 0) part_ml[2 * w] = st.m, part_ml[2 * w + 1] = st.l;
        }
        __syncthreads();
        if (w != 0) return;
        constexpr int NPS = MAXW / KPW > 0 ? MAXW / KPW : 1;
        static_assert(MAXW == 16, "lanes 0..15 (one DPP row) hold the waves' maxima");
        float M = lane < NW ? part_ml[2 * lane] : -INFINITY;
        float2 ml[NPS];
        float4 pa[NPS];
#pragma unroll
        for (int i = 0; i < NPS; i++) {
            const uint32_t ww = slot + i * KPW, wc = ww < NW ? ww : 0;
            ml[i] = *(const float2*)&part_ml[2 * wc];
            pa[i] = part_acc[wc * LPK + li];
        }
        M = fmaxf(M, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(M), 0xB1, 0xF, 0xF, true)));
        M = fmaxf(M, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(M), 0x4E, 0xF, 0xF, true)));
        M = fmaxf(M, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(M), 0x141, 0xF, 0xF, true)));
        M = fmaxf(M, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(M), 0x140, 0xF, 0xF, true)));
        M = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(M)));
        r = SoftState{M, 0.f, make_float4(0.f, 0.f, 0.f, 0.f)};
#pragma unroll
        for (int i = 0; i < NPS; i++) {
            const bool live = slot + i * KPW < NW && ml[i].x > -INFINITY;
            const float f = live ? expf(ml[i].x - M) : 0.0f;
            r.l += ml[i].y * f;
            r.acc.x += pa[i].x * f, r.acc.y += pa[i].y * f, r.acc.z += pa[i].z * f, r.acc.w += pa[i].w * f;
        }
        slots_sum<LPK>(r);
    }
    if (n_active > 1) { // (wave 0) publish this chunk; the last arriver merges all of them in chunk order
        constexpr uint32_t REC = DH + 4;
        float* const head_buf = split_buf + ((uint64_t)blockIdx.y * gridDim.x + qi) * gridDim.z * REC;
        float* const mine = head_buf + (uint64_t)sp * REC;
        uint32_t* const cnt = split_cnt + (uint64_t)blockIdx.y * gridDim.x + qi;
        if (lane < LPK) {
            if (lane == 0) split_put(mine, r.m, r.l);
            split_put(mine + 4 + 4 * lane, r.acc.x, r.acc.y);
            split_put(mine + 4 + 4 * lane + 2, r.acc.z, r.acc.w);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __shfl(old, 0, 64);
        if (old != n_active - 1) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
        if (lane == 0) __hip_atomic_store((gu32*)cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane < LPK) {
        const float inv_l = r.l > 0.f ? 1.0f / r.l : 0.0f;
        *(float4*)(p.dst + (uint64_t)qi * p.dst_cs + 4 * lane) = make_float4(r.acc.x * inv_l, r.acc.y * inv_l, r.acc.z * inv_l, r.acc.w * inv_l);
    }
}

// ── dense f32 matmul (reference.zig:480-497 -> forward.blasSgemm index contract) ───────────
// C[m*dst_rs + n] = sum_k A[m*a_rs + k*a_cs] * B[k*b_rs + n*b_cs]

// B is N-contiguous (b_cs == 1): one thread per n, coalesced across lanes, k sequential.
template <typename BT>
__global__ void __launch_bounds__(kBlock) dense_ncontig_kernel(DenseMatmulParams p) {
    uint32_t n = blockIdx.x * kBlock + threadIdx.x, m = blockIdx.y;
    if (n >= p.N) return;
    const float* a = p.a + (uint64_t)m * p.a_rs;
    const BT* b = (const BT*)p.b + n;
    float acc = 0.f;
    for (uint32_t k = 0; k < p.K; k++) acc = fmaf(a[(uint64_t)k * p.a_cs], (float)b[(uint64_t)k * p.b_rs], acc);
    p.dst[(uint64_t)m * p.dst_rs + n] = acc;
}

// B is K-contiguous (b_rs == 1, the tied-LM-head layout `trans1`): one wave per output n, lanes
// stride k with 16-byte loads, 64-lane shuffle reduction. This is the largest single read of a
// SmolLM decode step (576 x 49152 f32 = 113 MB), so loads are vectorised and a workgroup takes 4
// consecutive n (4 contiguous rows of B).
template <typename BT, int VEC>
__global__ void __launch_bounds__(kBlock) dense_kcontig_kernel(DenseMatmulParams p) {
    uint32_t n = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), m = blockIdx.y;
    if (n >= p.N) return;
    uint32_t lane = threadIdx.x & 63;
    const float* a = p.a + (uint64_t)m * p.a_rs;
    const BT* b = (const BT*)p.b + (uint64_t)n * p.b_cs;
    float acc = 0.f;
    if (VEC == 4) {
        // caller guarantees a_cs == 1, K % 4 == 0 and 16-byte (f32) / 8-byte (f16) alignment
        for (uint32_t k = lane * 4; k < p.K; k += 256) {
            float4 av = *(const float4*)(a + k);
            float b0, b1, b2, b3;
            if (sizeof(BT) == 4) {
                float4 bv = *(const float4*)((const float*)b + k);
                b0 = bv.x, b1 = bv.y, b2 = bv.z, b3 = bv.w;
            } else {
                const __half2* hp = (const __half2*)((const __half*)b + k);
                float2 lo = __half22float2(hp[0]), hi = __half22float2(hp[1]);
                b0 = lo.x, b1 = lo.y, b2 = hi.x, b3 = hi.y;
            }
            acc = fmaf(av.x, b0, acc);
            acc = fmaf(av.y, b1, acc);
            acc = fmaf(av.z, b2, acc);
            acc = fmaf(av.w, b3, acc);
        }
    } else {
        for (uint32_t k = lane; k < p.K; k += 64) acc = fmaf(a[(uint64_t)k * p.a_cs], (float)b[k], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) p.dst[(uint64_t)m * p.dst_rs + n] = acc;
}

template <typename BT>
__global__ void __launch_bounds__(kBlock) dense_strided_kernel(DenseMatmulParams p) {
    uint32_t n = blockIdx.x * kBlock + threadIdx.x, m = blockIdx.y;
    if (n >= p.N) return;
    const float* a = p.a + (uint64_t)m * p.a_rs;
    const BT* b = (const BT*)p.b + (uint64_t)n * p.b_cs;
    float acc = 0.f;
    for (uint32_t k = 0; k < p.K; k++) acc = fmaf(a[(uint64_t)k * p.a_cs], (float)b[(uint64_t)k * p.b_rs], acc);
    p.dst[(uint64_t)m * p.dst_rs + n] = acc;
}

// ── argmax: first index of the maximum (strict >), nn.zig:122-138 ──────────────────────────
constexpr int kArgBlocks = 256;

__device__ __forceinline__ void arg_combine(float& bv, int64_t& bi, float v, int64_t i) {
    if (v > bv || (v == bv && i < bi)) {
        bv = v;
        bi = i;
    }
}

__global__ void __launch_bounds__(kBlock) argmax_stage1(const float* __restrict__ v, uint64_t n, float* out_val,
                                                        int64_t* out_idx) {
    __shared__ float sv[kBlock];
    __shared__ int64_t si[kBlock];
    float bv = -INFINITY;
    int64_t bi = INT64_MAX;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        float x = v[i];
        if (x > bv || bi == INT64_MAX) { // strict >: earlier index wins within a thread's ascending walk
            bv = x;
            bi = (int64_t)i;
        }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) arg_combine(sv[threadIdx.x], si[threadIdx.x], sv[threadIdx.x + off], si[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out_val[blockIdx.x] = sv[0];
        out_idx[blockIdx.x] = si[0];
    }
}

__global__ void __launch_bounds__(kBlock) argmax_stage2(const float* vals, const int64_t* idxs, int nblk,
                                                        int64_t* out, ArgmaxAdvance adv) {
    __shared__ float sv[kBlock];
    __shared__ int64_t si[kBlock];
    float bv = -INFINITY;
    int64_t bi = INT64_MAX;
    for (int i = threadIdx.x; i < nblk; i += kBlock)
        if (idxs[i] != INT64_MAX) {
            if (bi == INT64_MAX) {
                bv = vals[i];
                bi = idxs[i];
            } else {
                arg_combine(bv, bi, vals[i], idxs[i]);
            }
        }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            if (si[threadIdx.x + off] != INT64_MAX) {
                if (si[threadIdx.x] == INT64_MAX) {
                    sv[threadIdx.x] = sv[threadIdx.x + off];
                    si[threadIdx.x] = si[threadIdx.x + off];
                } else {
                    arg_combine(sv[threadIdx.x], si[threadIdx.x], sv[threadIdx.x + off], si[threadIdx.x + off]);
                }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int64_t next = si[0] == INT64_MAX ? -1 : si[0];
        *out = next;
        if (adv.state) { // the resident loop's advance step
            const uint32_t n = adv.state[2];
            if (n < adv.cap) adv.tokens[n] = next;
            adv.state[0] = (uint32_t)next;
            adv.state[1] += 1;
            adv.state[2] = n + 1;
        }
    }
}

__global__ void __launch_bounds__(kBlock) copy_f4_kernel(float4* __restrict__ dst, const float4* __restrict__ src,
                                                         uint64_t n4) {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * kBlock)
        dst[i] = src[i];
}
// The HBM calibration copy (bench.py: extra.copy_kernel_*): U 16-byte loads in flight per lane before the first store, each
// wave-instruction a contiguous 1 KiB, non-temporal on both sides (every byte is touched once). The plain loop above keeps one
// load in flight per lane: 4.8 TB/s (read + write); this form 5.0-5.1 — a copy tops out near 5 TB/s on the boxes of this pool
// (the guide quotes 6.3 for its float4 copy), while READ-only streams go higher: the Q4_0 mat-vec at 4096 x 32000 on two
// streams reads 6.07 TB/s (tools/overlap_bench.py).
template <int U, bool NT>
__global__ void __launch_bounds__(kBlock) copy_f4_unrolled_kernel(float4* __restrict__ dst, const float4* __restrict__ src, uint64_t n4) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4v v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load((const f4v*)src + i + u * stride) : *((const f4v*)src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NT)
                __builtin_nontemporal_store(v[u], (f4v*)dst + i + u * stride);
            else
                *((f4v*)dst + i + u * stride) = v[u];
        }
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

// Two more calibration forms (ZGML_COPY_VARIANT=1 / 2): ONE float4 per thread on a grid of n4 / 256 workgroups (the textbook copy:
// the dispatcher, not a loop, walks the buffer), and a READ-ONLY stream (U loads in flight per lane, a sum per lane kept alive by
// one conditional store) — the mat-vec is a read stream, so this is the ceiling it is to be held against.
__global__ void __launch_bounds__(kBlock) copy_f4_one_kernel(float4* __restrict__ dst, const float4* __restrict__ src, uint64_t n4) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}
template <int U>
__global__ void __launch_bounds__(kBlock) read_f4_kernel(float* __restrict__ sink, const float4* __restrict__ src, uint64_t n4) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4v v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load((const f4v*)src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    const float t = acc.x + acc.y + acc.z + acc.w;
    if (t == 123456.789f) sink[blockIdx.x] = t; // (never true for the benchmark's data: keeps the loads alive)
}

// Access-pattern calibration for the mat-vec (ZGML_COPY_VARIANT=3 / 4, read-only, 4 x 16 B in flight per lane, 256-thread workgroups,
// `grid` workgroups of one wave-set each): 3 = every workgroup streams its OWN contiguous region of n4 / grid items (the packed
// weight layout: column-group-major), 4 = the workgroups' 4 KiB chunks interleaved (chunk c of workgroup b at (c * grid + b)): the
// resident workgroups together sweep ONE window of the buffer, as the hardware-dispatched one-float4-per-thread copy does.
template <bool INTERLEAVED>
__global__ void __launch_bounds__(kBlock) read_regions_kernel(float* __restrict__ sink, const float4* __restrict__ src, uint64_t n4) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const uint64_t per = n4 / gridDim.x / kBlock * kBlock; // items of one workgroup (a multiple of 256)
    const uint64_t chunks = per / kBlock;                  // 4 KiB chunks (one item per lane)
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    for (uint64_t c = 0; c + 3 < chunks; c += 4) {
        f4v v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t idx = INTERLEAVED ? ((c + u) * gridDim.x + blockIdx.x) * kBlock + threadIdx.x : (uint64_t)blockIdx.x * per + (c + u) * kBlock + threadIdx.x;
            v[u] = __builtin_nontemporal_load((const f4v*)src + idx);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc += v[u];
    }
    const float t = acc.x + acc.y + acc.z + acc.w;
    if (t == 123456.789f) sink[blockIdx.x] = t;
}

__global__ void __launch_bounds__(kBlock) f32_to_f16_kernel(__half* __restrict__ dst, const float* __restrict__ src,
                                                            uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
        dst[i] = __float2half_rn(src[i]);
}

inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

} // namespace

void launch_elementwise(hipStream_t s, uint32_t op, float* dst, const float* s0, const float* s1, uint32_t n) {
    if (n == 0) return;
    elementwise_kernel<<<cdiv(n, kBlock), kBlock, 0, s>>>(op, dst, s0, s1, n);
}

void launch_fused_elementwise(hipStream_t s, const FusedParams& p) {
    if (p.n == 0) return;
    fused_elementwise_kernel<<<cdiv(p.n, kBlock), kBlock, 0, s>>>(p);
}

void launch_eltwise_chain(hipStream_t s, const EltChainParams& p) {
    if (p.n == 0) return;
    // ZGML_HIP_ELT_PRELOAD=0: the step-by-step form always
    static const bool pre_on = !(getenv("ZGML_HIP_ELT_PRELOAD") && atoi(getenv("ZGML_HIP_ELT_PRELOAD")) == 0);
    static const uint32_t vec_min = getenv("ZGML_HIP_ELT_VEC4_MIN") ? (uint32_t)atol(getenv("ZGML_HIP_ELT_VEC4_MIN")) : (1u << 20);
    auto a16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    bool pre = pre_on;
    bool vec = p.n >= vec_min && p.n % 4 == 0 && a16(p.src) && (!p.ap || (p.ap_cols % 4 == 0 && a16(p.ap)));
    for (uint32_t t = 0; pre && t < p.n_steps; t++) {
        vec = vec && a16(p.steps[t].secondary) && a16(p.steps[t].store);
        for (uint32_t e = 0; pre && p.steps[t].secondary && e < t; e++) // an operand written by an earlier step of this chain: keep the step-by-step reads
            if (p.steps[e].store && p.steps[t].secondary < p.steps[e].store + p.n && p.steps[e].store < p.steps[t].secondary + p.n) pre = false;
    }
    if (pre && vec)
        eltwise_chain_pre_kernel<4><<<cdiv(p.n / 4, kBlock), kBlock, 0, s>>>(p);
    else if (pre)
        eltwise_chain_pre_kernel<1><<<cdiv(p.n, kBlock), kBlock, 0, s>>>(p);
    else
        eltwise_chain_kernel<<<cdiv(p.n, kBlock), kBlock, 0, s>>>(p);
}

void launch_softmax(hipStream_t s, float* dst, const float* src, uint32_t rows, uint32_t cols) {
    if (rows == 0) return;
    softmax_kernel<<<rows, kBlock, 0, s>>>(dst, src, cols);
}

void launch_layernorm(hipStream_t s, float* dst, const float* src, uint32_t rows, uint32_t cols, float eps) {
    if (rows == 0) return;
    layernorm_kernel<<<rows, kBlock, 0, s>>>(dst, src, cols, eps);
}

void launch_rmsnorm(hipStream_t s, float* dst, const float* src, uint32_t rows, uint32_t cols, float eps) {
    if (rows == 0) return;
    if (cols <= 32 * kBlock) { // the register-resident row kernel (same mapping and reduction: bit-identical), all loads up front
        RowChainParams rc;
        rc.src = src, rc.norm_dst = dst, rc.cols = cols, rc.eps = eps;
        launch_row_chain(s, rc, rows);
        return;
    }
    rmsnorm_kernel<<<rows, kBlock, 0, s>>>(dst, src, cols, eps);
}

void launch_row_chain(hipStream_t s, const RowChainParams& p, uint32_t rows) {
    if (rows == 0) return;
    // few rows: several workgroups per row share its stores (row_chain_kernel): as many as keep the grid near one workgroup per CU,
    // at most one per 256-column chunk the row really has (ZGML_HIP_ROW_SPLIT=1 switches it off)
    static const int split_env = getenv("ZGML_HIP_ROW_SPLIT") ? atoi(getenv("ZGML_HIP_ROW_SPLIT")) : 0;
    const uint32_t chunks = cdiv(p.cols, kBlock);
    uint32_t split = 1;
    while (split * 2 <= chunks && rows * split * 2 <= 256 && chunks % (split * 2) == 0) split *= 2;
    if (split_env > 0) split = std::min<uint32_t>((uint32_t)split_env, chunks);
    { // The workgroups of a row each re-read the WHOLE row (a0, a1, src, mul_other) and store only their own column chunks (add_dst,
      // norm_dst, mul_dst). One workgroup per row is safe in place (a thread reads its elements before it writes them); several
      // are not: a sibling may already have overwritten what this one still has to read — x.addInplace(y) followed by rmsNorm is
      // a legal op stream (src/tensor/api.zig:132) and would get a double add and a wrong sum of squares. So: any
      // written span overlapping any span the siblings read -> one workgroup per row.
        const size_t n = (size_t)rows * p.cols;
        auto overlap = [n](const float* a, const float* b) { return a && b && a < b + n && b < a + n; };
        const float* const reads[4] = {p.a0, p.a1, p.src, p.mul_other};
        const float* const writes[3] = {p.add_dst, p.norm_dst, p.mul_dst};
        for (const float* w : writes)
            for (const float* r : reads)
                if (overlap(w, r) && !(w == p.add_dst && r == p.src && p.a0)) // (src == add_dst behind a pre-add is the chain itself: every workgroup recomputes the add from a0 / a1 and never reads add_dst back)
                    split = 1;
    }
    const dim3 grid(rows, p.cols <= 32 * kBlock ? split : 1);
    if (p.cols <= 4 * kBlock)
        row_chain_kernel<4><<<grid, kBlock, 0, s>>>(p);
    else if (p.cols <= 16 * kBlock)
        row_chain_kernel<16><<<grid, kBlock, 0, s>>>(p);
    else if (p.cols <= 32 * kBlock)
        row_chain_kernel<32><<<grid, kBlock, 0, s>>>(p);
    else
        row_chain_kernel<0><<<grid, kBlock, 0, s>>>(p);
}

void launch_reduce(hipStream_t s, uint32_t op, float* dst, const float* src, uint32_t n_out, uint32_t reduce_size) {
    if (n_out == 0) return;
    reduce_kernel<<<cdiv(n_out, kBlock / 64), kBlock, 0, s>>>(op, dst, src, n_out, reduce_size);
}

// Batched forms: `dev_params` is a device array of n_ops parameter records (blockIdx.y = op);
// max_elems / max_seq_q size grid.x for the largest op, smaller ops early-exit.
void launch_repeat_batch(hipStream_t s, const RepeatParams* dev_params, uint32_t n_ops, uint32_t max_elems) {
    if (!n_ops || !max_elems) return;
    repeat_kernel<<<dim3(cdiv(max_elems, kBlock), n_ops), kBlock, 0, s>>>(dev_params);
}

void launch_move_batch(hipStream_t s, const MoveParams* dev_params, uint32_t n_ops, uint32_t max_elems) {
    if (!n_ops || !max_elems) return;
    move_kernel<<<dim3(cdiv(max_elems, kBlock), n_ops), kBlock, 0, s>>>(dev_params);
}

void launch_kvq_store_batch(hipStream_t s, const KvqStoreParams* dev_params, uint32_t n_ops) {
    if (n_ops) kvq_store_kernel<<<n_ops, 64, 0, s>>>(dev_params);
}

// all ops of a launch share d_head (the planner groups them); d_head in {16, 32, 64, 128, 256}
void launch_kvq_attention_batch(hipStream_t s, const KvqAttentionParams* dev_params, uint32_t n_ops, uint32_t max_seq_q, uint32_t d_head,
                                const AttnSplit& sp) {
    if (!n_ops || !max_seq_q) return;
    const dim3 grid(max_seq_q, n_ops, max_seq_q == 1 && sp.splits > 1 ? sp.splits : 1);
#define AKVQ(L) kvq_attention_kernel<L><<<grid, kAttnBlock, 0, s>>>(dev_params, sp.buf, sp.cnt, sp.min_keys)
    switch (d_head) {
        case 16: AKVQ(4); break;
        case 32: AKVQ(8); break;
        case 64: AKVQ(16); break;
        case 128: AKVQ(32); break;
        case 256: AKVQ(64); break;
        default: break;
    }
#undef AKVQ
}

void launch_attention_decode_batch(hipStream_t s, const AttnDecodeParams* dev_params, uint32_t n_heads, uint32_t d_head,
                                   const AttnSplit& sp, bool kvq) {
    if (!n_heads) return;
    const dim3 grid(n_heads, sp.splits ? sp.splits : 1); // x = head: the always-active split 0 of every head is dispatched first
#define ADEC(L, Q) attention_decode_kernel<L, Q><<<grid, kAttnBlock, 0, s>>>(dev_params, sp.buf, sp.cnt, sp.min_keys)
    if (kvq) { // quantised KV caches (block 32): d_head a multiple of 32
        switch (d_head) {
            case 32: ADEC(8, true); break;
            case 64: ADEC(16, true); break;
            case 128: ADEC(32, true); break;
            case 256: ADEC(64, true); break;
            default: break;
        }
        return;
    }
    // d_head 128 (f32 KV): 4-wave workgroups — at short context the chunk needs few waves anyway and a 256-thread workgroup
    // dispatches and merges faster, at long context the splits carry the parallelism: Llama-2-7B 802 -> 812 tok/s, position 1900
    // 605 -> 622 (ZGML_HIP_ATTN_DECODE_BLOCK=1024 / 256 forces either for d_head 64 and 128)
    static const int block_env = getenv("ZGML_HIP_ATTN_DECODE_BLOCK") ? atoi(getenv("ZGML_HIP_ATTN_DECODE_BLOCK")) : 0;
    const bool small_block = block_env == 256 || (block_env == 0 && d_head == 128);
    if (small_block && (d_head == 64 || d_head == 128)) {
        if (d_head == 64)
            attention_decode_kernel<16, false, 256><<<grid, 256, 0, s>>>(dev_params, sp.buf, sp.cnt, sp.min_keys);
        else
            attention_decode_kernel<32, false, 256><<<grid, 256, 0, s>>>(dev_params, sp.buf, sp.cnt, sp.min_keys);
        return;
    }
    switch (d_head) { // all heads of a launch share d_head (checked by the planner)
        case 8: ADEC(2, false); break;
        case 16: ADEC(4, false); break;
        case 32: ADEC(8, false); break;
        case 64: ADEC(16, false); break;
        case 128: ADEC(32, false); break;
        case 256: ADEC(64, false); break;
        default: break;
    }
#undef ADEC
}

void launch_attention_batch(hipStream_t s, const AttentionParams* dev_params, uint32_t n_ops, uint32_t max_seq_q,
                            bool all_dense, uint32_t rows_d_head, const float* zero_word, const AttnPieceSink& sink) {
    if (!n_ops || !max_seq_q) return;
    static const bool rows_on = !(getenv("ZGML_HIP_ATTN_ROWS") && atoi(getenv("ZGML_HIP_ATTN_ROWS")) == 0);
    if (all_dense && rows_d_head && zero_word && rows_on && attention_tiles_applies(max_seq_q, rows_d_head)) {
        launch_attention_tiles(s, dev_params, n_ops, max_seq_q, rows_d_head, zero_word, sink);
        return;
    }
    if (all_dense && rows_d_head && zero_word && rows_on) { // every op dense with this d_head: the streaming kernel
        const dim3 grid(max_seq_q, n_ops);
        // enough (query, head) workgroups to fill the chip: 4 waves each (a 16-wave workgroup whose context needs 2
        // still pays for launching 16); few workgroups: all 16 so a long context is spread over more waves
        const uint32_t block = (uint64_t)max_seq_q * n_ops >= 256 ? 256 : kAttnBlock;
#define AROWS(L) attention_rows_kernel<L><<<grid, block, 0, s>>>(dev_params, zero_word)
        switch (rows_d_head) {
            case 8: AROWS(2); return;
            case 16: AROWS(4); return;
            case 32: AROWS(8); return;
            case 64: AROWS(16); return;
            case 128: AROWS(32); return;
            case 256: AROWS(64); return;
            default: break;
        }
#undef AROWS
    }
    if (all_dense)
        attention_dense_kernel<<<dim3(max_seq_q, n_ops), kAttnBlock, 0, s>>>(dev_params);
    else
        attention_kernel<<<dim3(max_seq_q, n_ops), kBlock, 0, s>>>(dev_params);
}

void launch_dense_matmul(hipStream_t s, const DenseMatmulParams& p) {
    if (p.M == 0 || p.N == 0) return;
    const bool f16 = p.b_f16 != 0;
    if (p.b_cs == 1 && p.b_rs != 1) {
        dim3 grid(cdiv(p.N, kBlock), p.M);
        if (f16)
            dense_ncontig_kernel<__half><<<grid, kBlock, 0, s>>>(p);
        else
            dense_ncontig_kernel<float><<<grid, kBlock, 0, s>>>(p);
    } else if (p.b_rs == 1) {
        dim3 grid(cdiv(p.N, kBlock / 64), p.M);
        const uint32_t esz = f16 ? 2 : 4;
        bool vec = p.a_cs == 1 && (p.K % 4) == 0 && (p.a_rs % 4) == 0 && (p.b_cs % 4) == 0 &&
                   ((uintptr_t)p.a % 16) == 0 && ((uintptr_t)p.b % (4 * esz)) == 0;
        if (f16) {
            if (vec)
                dense_kcontig_kernel<__half, 4><<<grid, kBlock, 0, s>>>(p);
            else
                dense_kcontig_kernel<__half, 1><<<grid, kBlock, 0, s>>>(p);
        } else {
            if (vec)
                dense_kcontig_kernel<float, 4><<<grid, kBlock, 0, s>>>(p);
            else
                dense_kcontig_kernel<float, 1><<<grid, kBlock, 0, s>>>(p);
        }
    } else {
        dim3 grid(cdiv(p.N, kBlock), p.M);
        if (f16)
            dense_strided_kernel<__half><<<grid, kBlock, 0, s>>>(p);
        else
            dense_strided_kernel<float><<<grid, kBlock, 0, s>>>(p);
    }
}

namespace {
// one element of the flat prep index space (kernels.h: ResidentPrepArgs) at position `pos` with the chunk's tokens `tokens`
__device__ __forceinline__ void resident_prep_element(const ResidentPrepArgs& a, uint32_t i, uint32_t pos, const uint32_t* tokens, uint32_t token0, bool use_token0) {
    if (i < a.T * a.d) {
        const uint32_t j = i / a.d, e = i - j * a.d;
        a.tok_in[i] = a.embed[(uint64_t)(use_token0 ? token0 : tokens[j]) * a.d + e];
        return;
    }
    i -= a.T * a.d;
    if (i < a.T * a.max_seq) {
        const uint32_t j = i / a.max_seq, sidx = i - j * a.max_seq;
        a.mask[i] = sidx <= pos + j ? 0.0f : -INFINITY;
        return;
    }
    i -= a.T * a.max_seq;
    if (i < a.n_rope * a.T * 2 * a.dh) {
        const uint32_t l = i / (a.T * 2 * a.dh), rem = i - l * (a.T * 2 * a.dh), j = rem / (2 * a.dh), e = rem - j * 2 * a.dh;
        a.rope_bufs[l][rem] = e < a.dh ? a.cos[(uint64_t)(pos + j) * a.dh + e] : a.sin[(uint64_t)(pos + j) * a.dh + e - a.dh];
        return;
    }
    i -= a.n_rope * a.T * 2 * a.dh;
    if (i < a.n_ops) {
        if (a.dyn_kind[i] == 1) a.dyn[i] = a.dyn_base[i] + pos * a.dyn_stride[i];
        if (a.dyn_kind[i] == 2) a.dyn[i] = pos + a.T;
    }
}
__global__ void __launch_bounds__(256) resident_prep_kernel(ResidentPrepArgs a) {
    resident_prep_element(a, blockIdx.x * 256 + threadIdx.x, a.state[1], a.tokens, 0, false);
}

// argmax_stage1 + (last arriver) argmax_stage2 + advance + the next token's prep (kernels.h: launch_argmax_tail)
__global__ void __launch_bounds__(kBlock) argmax_tail_kernel(const float* __restrict__ v, uint64_t n, float* vals, int64_t* idxs, uint32_t* cnt, int64_t* out,
                                                             ArgmaxAdvance adv, ResidentPrepArgs prep, uint32_t prep_total, uint32_t has_prep) {
    __shared__ float sv[kBlock];
    __shared__ int64_t si[kBlock];
    __shared__ uint32_t bc[2];
    using gu32 = __attribute__((address_space(1))) uint32_t;
    using gu64 = __attribute__((address_space(1))) unsigned long long;
    // The next token's patches that depend on its POSITION only — mask column, RoPE rows, dynamic words: everything but the
    // embedding row — are written here by ALL workgroups (the position is known: state[1] + 1, read before this workgroup's
    // arrival is counted, i.e. before the last arriver advances it; the plan that read the current token's patches has finished).
    // Round 5's first form left the whole prep to the last arriver: one workgroup walking 8-16 K elements alone was a longer chain
    // than the two launches it replaced. What is left for the last arriver is the embedding row of the token it has just found.
    const uint32_t prep_tok = has_prep ? prep.T * prep.d : 0;
    if (has_prep && adv.state) {
        const uint32_t pos_next = adv.state[1] + 1;
        if (pos_next < prep.max_seq)
            for (uint32_t i = prep_tok + blockIdx.x * kBlock + threadIdx.x; i < prep_total; i += gridDim.x * kBlock) resident_prep_element(prep, i, pos_next, nullptr, 0, true);
    }
    float bv = -INFINITY;
    int64_t bi = INT64_MAX;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) { // (as argmax_stage1)
        const float x = v[i];
        if (x > bv || bi == INT64_MAX) {
            bv = x;
            bi = (int64_t)i;
        }
    }
    sv[threadIdx.x] = bv, si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) arg_combine(sv[threadIdx.x], si[threadIdx.x], sv[threadIdx.x + off], si[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) { // publish this workgroup's pair write-through, drain, count (the guide's fan-in form: sc1 payload + vmcnt(0) + agent atomic)
        __hip_atomic_store((gu32*)vals + blockIdx.x, __float_as_uint(sv[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((gu64*)idxs + blockIdx.x, (unsigned long long)si[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bc[0] = __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!bc[0]) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // compiler only: the agent-scope loads below stay below the count
    bv = -INFINITY, bi = INT64_MAX; // ---- stage 2 (as argmax_stage2), every pair read with agent-scope loads
    for (uint32_t i = threadIdx.x; i < gridDim.x; i += kBlock) {
        const int64_t ix = (int64_t)__hip_atomic_load((gu64*)idxs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float vx = __uint_as_float(__hip_atomic_load((gu32*)vals + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (ix != INT64_MAX) {
            if (bi == INT64_MAX)
                bv = vx, bi = ix;
            else
                arg_combine(bv, bi, vx, ix);
        }
    }
    sv[threadIdx.x] = bv, si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off && si[threadIdx.x + off] != INT64_MAX) {
            if (si[threadIdx.x] == INT64_MAX)
                sv[threadIdx.x] = sv[threadIdx.x + off], si[threadIdx.x] = si[threadIdx.x + off];
            else
                arg_combine(sv[threadIdx.x], si[threadIdx.x], sv[threadIdx.x + off], si[threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int64_t next = si[0] == INT64_MAX ? -1 : si[0];
        *out = next;
        uint32_t pos_next = 0;
        if (adv.state) { // the resident loop's advance step
            const uint32_t k = adv.state[2];
            if (k < adv.cap) adv.tokens[k] = next;
            adv.state[0] = (uint32_t)next;
            pos_next = adv.state[1] + 1;
            adv.state[1] = pos_next;
            adv.state[2] = k + 1;
        }
        __hip_atomic_store((gu32*)cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-arm
        bc[0] = (uint32_t)next, bc[1] = pos_next;
    }
    __syncthreads();
    if (!has_prep) return;
    const uint32_t token = bc[0], pos = bc[1];
    if (pos >= prep.max_seq || token >= 0x7FFFFFFFu) return; // (the context is full, or no finite logit: nothing to prepare)
    for (uint32_t i = threadIdx.x; i < prep_tok; i += kBlock) resident_prep_element(prep, i, pos, nullptr, token, true); // the embedding row(s)
}
} // namespace

void launch_resident_prep(hipStream_t s, const ResidentPrepArgs& a, uint32_t total) {
    if (total) resident_prep_kernel<<<(total + 255) / 256, 256, 0, s>>>(a);
}

void launch_argmax_tail(hipStream_t s, const float* v, uint64_t n, float* scratch_val, int64_t* scratch_idx, uint32_t* cnt, int64_t* out,
                        const ArgmaxAdvance& adv, const ResidentPrepArgs* prep, uint32_t prep_total) {
    int nblk = (int)(n / (kBlock * 4) + 1);
    if (nblk > kArgBlocks) nblk = kArgBlocks;
    argmax_tail_kernel<<<nblk, kBlock, 0, s>>>(v, n, scratch_val, scratch_idx, cnt, out, adv, prep ? *prep : ResidentPrepArgs{}, prep_total, prep ? 1u : 0u);
}

void launch_argmax(hipStream_t s, const float* v, uint64_t n, float* scratch_val, int64_t* scratch_idx, int64_t* out, const ArgmaxAdvance& adv) {
    int nblk = (int)(n / (kBlock * 4) + 1);
    if (nblk > kArgBlocks) nblk = kArgBlocks;
    argmax_stage1<<<nblk, kBlock, 0, s>>>(v, n, scratch_val, scratch_idx);
    argmax_stage2<<<1, kBlock, 0, s>>>(scratch_val, scratch_idx, nblk, out, adv);
}

void launch_copy_f4(hipStream_t s, void* dst, const void* src, uint64_t bytes) {
    uint64_t n4 = bytes / 16;
    if (n4 == 0) return;
    // ZGML_COPY_VARIANT: 0 = the plain loop, else U | NT << 8 | blocks-per-CU << 16 (experiments); default: 8 in flight, nt, 32 blocks per CU (sweep: 4.5-5.1 TB/s over all variants on this part)
    static const int variant = getenv("ZGML_COPY_VARIANT") ? (int)strtol(getenv("ZGML_COPY_VARIANT"), nullptr, 0) : (8 | 1 << 8 | 32 << 16);
    if (variant == 0) {
        copy_f4_kernel<<<2048, kBlock, 0, s>>>((float4*)dst, (const float4*)src, n4);
        return;
    }
    if (variant == 1) { // one float4 per thread
        copy_f4_one_kernel<<<(unsigned)((n4 + kBlock - 1) / kBlock), kBlock, 0, s>>>((float4*)dst, (const float4*)src, n4);
        return;
    }
    if ((variant & 0xFF) == 3 || (variant & 0xFF) == 4) { // regions / interleaved regions; workgroups = variant >> 8 (default 688)
        const unsigned grid = (unsigned)(variant >> 8) ? (unsigned)(variant >> 8) : 688u;
        if ((variant & 0xFF) == 3)
            read_regions_kernel<false><<<grid, kBlock, 0, s>>>((float*)dst, (const float4*)src, n4);
        else
            read_regions_kernel<true><<<grid, kBlock, 0, s>>>((float*)dst, (const float4*)src, n4);
        return;
    }
    if (variant == 2) { // read-only stream (dst is only the never-written sink)
        read_f4_kernel<8><<<256 * 16, kBlock, 0, s>>>((float*)dst, (const float4*)src, n4);
        return;
    }
    const int U = variant & 0xFF, nt = (variant >> 8) & 1, bpc = (variant >> 16) & 0xFF ? (variant >> 16) & 0xFF : 8;
    const dim3 grid(256 * bpc);
    if (U >= 8)
        nt ? copy_f4_unrolled_kernel<8, true><<<grid, kBlock, 0, s>>>((float4*)dst, (const float4*)src, n4) : copy_f4_unrolled_kernel<8, false><<<grid, kBlock, 0, s>>>((float4*)dst, (const float4*)src, n4);
    else if (U >= 4)
        nt ? copy_f4_unrolled_kernel<4, true><<<grid, kBlock, 0, s>>>((float4*)dst, (const float4*)src, n4) : copy_f4_unrolled_kernel<4, false><<<grid, kBlock, 0, s>>>((float4*)dst, (const float4*)src, n4);
    else
        nt ? copy_f4_unrolled_kernel<2, true><<<grid, kBlock, 0, s>>>((float4*)dst, (const float4*)src, n4) : copy_f4_unrolled_kernel<2, false><<<grid, kBlock, 0, s>>>((float4*)dst, (const float4*)src, n4);
}

void launch_f32_to_f16(hipStream_t s, void* dst, const float* src, uint64_t n) {
    if (n == 0) return;
    f32_to_f16_kernel<<<1024, kBlock, 0, s>>>((__half*)dst, src, n);
}

} // namespace zgml
