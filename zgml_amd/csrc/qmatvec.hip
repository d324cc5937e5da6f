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
//   NB = N/32 scale block-columns, NB2 = N/16 column groups, KC = ceil(K/32) k-chunks (zero padded).
//   Q4 (all values in [-8,7]; GGUF Q4_0-sourced weights, 4.5 bit/weight like the file format):
//     qs: uint4[NB2][KC][16]  item (g,c,i) = the 32 nibbles of column n = 16g+i for
//                             k = 32c..32c+31, two's-complement; dword d, byte b holds
//                             k_local = 8d+b in the low nibble and 8d+4+b in the high nibble.
//     sc: pair[NB][KC][16]    entry i = { scale(k=32c+i, j), scale(k=32c+16+i, j) }, j = g/2
//   Q8: qs: uint4[NB2][2KC][16] item (g,h,i) = 16 int8 of column n for k = 16h..16h+15
//       sc: scalar[NB][2KC][16] entry i = scale(k=16h+i, j)
//   Scales are stored as f16 when every scale is exactly representable (GGUF-sourced), else f32.
//
// Thread mapping: a wave64 is 4 DPP rows of 16 lanes. A workgroup owns ONE 16-column group g and
// ALL of K (so there is no cross-workgroup reduction and the mat-vec is a single launch: on this
// chip a dependent launch boundary costs ~1.7 us, as much as streaming 9.4 MB).
// Row r of wave w takes k-unit u = 4w + r (+ 4*waves per step); a unit is a 32-k chunk for Q4 and
// a 16-k half-chunk for Q8. The four rows of a wave read four consecutive units = 1 KiB contiguous
// (16 B/lane). Lane i of a row computes t = scale(k_i) * x[k_i] for "its" k of the row's unit,
// and the 32 (16) products of a lane's column are   acc += f32(q[k,n]) * t[k]   with t[k] fetched
// from lane k of the same row by the DPP row_newbcast operand of v_fmac_f32 — no LDS round trip
// and no scalar broadcast for the per-k multiplier, although the scale grouping runs along n.
// Weights go HBM -> VGPR directly (GEMV: nothing to reuse, guide §5 "GEMV / M <= 16" row); only x
// is staged in LDS. The two column groups of a scale block-column read the same scales; their
// workgroups are placed 8 blocks apart so they share an XCD L2 (speed only, never correctness).
#include "qmv_common.h"
#include "attention_decode.h"

#include <hip/hip_fp16.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>

namespace zgml {

namespace {



struct QMVPartDev {
    const uint4* qs;
    const void* sc;
    float* out;           // row m at out + m*out_rs
    uint32_t NB2;         // 16-column groups of this matrix
    uint32_t block_begin; // first blockIdx.x of this part
    uint32_t out_rs;
    uint32_t n_epi;
    QmvEpiStep epi[kMaxEpiSteps];
    uint32_t epi_kind; // kEpiGeneric, or a recognised chain that runs as straight-line code (same operations, same order)
};
// NEG, EXP [store], ADD operand, RECIP, MUL by the part's own output [store] — the gate projection's SiLU chain
// (silu = nn.zig:38-44 as zgml_amd/host/llama_decode.cpp emits it). The generic interpreter spends ~0.2 us per step on
// scalar loads, compares and branches at the very end of the kernel (in-kernel stamps: 1.5 us of tail against 0.5).
constexpr uint32_t kEpiGeneric = 0, kEpiSilu = 1;
// ADD vector [store]: the residual add behind the O / down projections (single-matrix launches; the vector is the operand
// the kernel requests at its start, `pre0`)
constexpr uint32_t kEpiResidual = 2;
// MUL vector [store]: one multiply by a vector (zgml_hip_qmatvec_chain_bench's bounded-magnitude epilogue), same treatment
constexpr uint32_t kEpiMulVec = 3;

struct QMVArgs {
    QMVPartDev parts[kMaxQmvParts];
    uint32_t n_parts;
    QmvPrologue pro; // pro.a is the input vector for kind NONE
    QmvNextNorm next; // residual-epilogue launches: side outputs for the next launch's rmsnorm prologue
    float* pair_out;  // PAIR launches: where silu(part 0) * part 1 goes (nullptr: not a pair launch)
    uint32_t M, K;
    uint32_t U;      // k-units per column group (Q4: KC, Q8: 2*KC)
    uint32_t in_rs;
    uint32_t x_vec; // x rows are 16-byte aligned and K % 4 == 0
    unsigned long long* trace; // diagnostics: stamps of workgroup (0,0), else nullptr
};

// x staging in two halves so the weight loads can sit between them: x_fetch() issues this thread's
// global loads of x FIRST (vmcnt is in-order: if x were loaded after the weights, the wait before
// the LDS store would also wait for every weight load and serialise load and compute), x_commit()
// stores them to LDS after the weight loads have been issued. Every load here and in the kernels
// is UNCONDITIONAL (addresses are clamped, values are zeroed by selects): hipcc counts vmcnt
// exactly only across straight-line code — one load under a divergent branch and it falls back to
// vmcnt(0) everywhere, which serialises the weight stream behind the LDS store (measured:
// load and compute times simply added up).
constexpr int kXRegs = 4; // float4 per thread held in registers (covers K <= 16 * blockDim)

struct XRegs {
    float4 v[kXRegs];
};


template <bool XVEC>
__device__ __forceinline__ XRegs x_fetch(const float* x, uint32_t K, uint32_t bdim) {
    XRegs r;
#pragma unroll
    for (int j = 0; j < kXRegs; j++) r.v[j] = load_x4<XVEC>(x, (threadIdx.x + j * bdim) * 4, K);
    return r;
}

__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 scale4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }

// k_count is a multiple of 16; xs has 4 spare floats at xs[k_count..] for the out-of-range lanes.
// KIND selects the prologue (kernels.h QmvPrologueKind); `inv` is the rmsnorm factor for kind 2.
// Workgroup (0,0) also stores the prologue's intermediates to global memory so the buffers of
// the absorbed ops hold what the unfused plan would have written.
template <bool XVEC, bool PRO>
__device__ __forceinline__ void x_commit(float* xs, const XRegs& ra, const XRegs& rb, const QmvPrologue& pro, float inv,
                                         uint32_t k_count, uint32_t K, const float* xa_row, uint32_t bdim) {
    const bool owner = PRO && blockIdx.x == 0 && blockIdx.y == 0;
#pragma unroll
    for (int j = 0; j < kXRegs; j++) {
        const uint32_t i = (threadIdx.x + j * bdim) * 4;
        float4 v = zero_tail(ra.v[j], i, K);
        if (PRO && pro.kind == QMV_PRO_RMSNORM_MUL) {
            v = scale4(v, inv);
            if (owner && pro.store_mid && i < K) {
                if (i + 3 < K) *(float4*)(pro.store_mid + i) = v;
                else { pro.store_mid[i] = v.x; if (i + 1 < K) pro.store_mid[i + 1] = v.y; if (i + 2 < K) pro.store_mid[i + 2] = v.z; }
            }
        }
        if (PRO) {
            v = mul4(v, zero_tail(rb.v[j], i, K));
            if (owner && pro.store_x && i < K) {
                if (i + 3 < K) *(float4*)(pro.store_x + i) = v;
                else { pro.store_x[i] = v.x; if (i + 1 < K) pro.store_x[i + 1] = v.y; if (i + 2 < K) pro.store_x[i + 2] = v.z; }
            }
        }
        *(float4*)(xs + (i < k_count ? i : k_count)) = v;
    }
    if (!PRO) // only for K > 16 * blockDim (prologues are limited to the register window)
        for (uint32_t i = (threadIdx.x + kXRegs * bdim) * 4; i < k_count; i += bdim * 4)
            *(float4*)(xs + i) = zero_tail(load_x4<XVEC>(xa_row, i, K), i, K);
}

// kEpiSilu: the five steps of the interpreter above for exactly that chain (the host checked the pattern)
__device__ __forceinline__ void run_epilogue_silu(const QMVPartDev& part, uint32_t n, float raw, float ones) {
    const float e = expf(-raw);  // NEG, EXP
    part.epi[1].store[n] = e;
    const float s = 1.0f / (e + ones); // ADD (commutative), RECIP
    part.epi[4].store[n] = s * raw;    // MUL by the part's own output (commutative)
}


// sum of squares of the (zero-tailed) register window, fixed reduction order: lanes, then waves
__device__ __forceinline__ float block_sumsq(const XRegs& r, uint32_t K, float* red, uint32_t bdim) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < kXRegs; j++) {
        const uint32_t i = (threadIdx.x + j * bdim) * 4;
        const float4 v = zero_tail(r.v[j], i, K);
        ss += v.x * v.x;
        ss += v.y * v.y;
        ss += v.z * v.z;
        ss += v.w * v.w;
    }
    return block_total(ss, red, bdim);
}

__device__ __forceinline__ float epi_unary(uint32_t op, float a) {
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
        case ZGML_OP_GELU: {
            const float kk = 0.7978845608f * (a + 0.044715f * a * a * a);
            return 0.5f * a * (1.0f + tanhf(kk));
        }
        default: return a;
    }
}


// the part's elementwise epilogue on one output value (residual add, SiLU chain, ...)
// `pre0` / `have_pre0`: the operand of step 0, fetched at kernel start under the weight stream (non-grouped launches:
// the residual add of the O / down projections) instead of by a dependent load behind the reduction.
__device__ __forceinline__ void run_epilogue(const QMVPartDev& part, uint32_t n, float v, const float* out_row, float pre0 = 0.f,
                                             bool have_pre0 = false) {
    const float raw = v; // an operand that is this part's own output (SiLU's final gate * sigmoid) stays in the register
    // fully unrolled: `part.epi[e]` with a run-time e is a scalar load from the argument block inside
    // the loop, i.e. one dependent ~0.3 us round trip per step at the very end of the kernel
#pragma unroll
    for (uint32_t e = 0; e < (uint32_t)kMaxEpiSteps; e++) { // M == 1 whenever n_epi != 0
        if (e >= part.n_epi) break;
        const QmvEpiStep st = part.epi[e];
        if (st.op == ZGML_OP_ADD) {
            const float o = st.operand == out_row ? raw : (e == 0 && have_pre0 ? pre0 : st.operand[n]);
            v = st.swapped ? o + v : v + o;
        } else if (st.op == ZGML_OP_MUL) {
            const float o = st.operand == out_row ? raw : (e == 0 && have_pre0 ? pre0 : st.operand[n]);
            v = st.swapped ? o * v : v * o;
        } else {
            v = epi_unary(st.op, v);
        }
        if (st.store) st.store[n] = v;
    }
}

// A launch that also holds the decode attention's workgroups (qkv_attn_kernel below) hands the projections over without a
// kernel boundary: the 16 owning lanes store their outputs write-through (agent scope), the wave drains, and lane 0 bumps
// the counter of the head slice this column group belongs to (attention_decode.h: DecodeHandoff).
struct QmvPublish {
    uint32_t* cnt;         // [n_heads | n_kv | n_kv]
    uint32_t base[3];      // first counter of part 0 / 1 / 2 (q, k, v)
    uint32_t groups_shift; // log2(d_head / 16): column group -> head slice
    uint32_t drop;         // diagnostics (ZGML_HIP_DEBUG_DROP_PUBLISH=1): column group 0 of part 0 never signals — its consumers time out
};

// Fold the 4 rows of each wave, then the waves, in fixed order; 16 outputs per workgroup. The 16
// owning lanes then run the part's epilogue. `pi` is the (wave-uniform) part index: every use of
// `a.parts[...]` below is a COMPILE-TIME index so the kernel-argument loads are issued up front —
// a run-time index into the argument block is a dependent scalar load that costs ~1 us at the
// start and again in the tail of a 5 us kernel.
// The waves' rows are folded by wave 0 with 4 independent LDS reads per lane (row r of the wave takes waves r, r + 4,
// r + 8, r + 12) and one more row fold: the serial loop over the waves it replaces was one dependent LDS round trip per
// wave (in-kernel stamps: 0.64 us of tail with 9-16 waves; SmolLM-135M +3.0 %, Llama-2-7B +1.1 % on the same box).
// Measured with it and NOT kept: requesting the operands of all epilogue steps at kernel start (the SiLU chain's constant
// vector; -1 %: two more loads per lane and the part selects cost more than the one dependent load they save), and the
// prologue kind / x alignment from the preloaded head instead of the argument block (+0.7 % / -0.9 %).
// KON (the K-on-lanes body below): the caller has already folded its wave's lanes into red[w * 16 + column] (and the wave's sum of
// squares into red[kMaxWaves * 16 + w]); the finished sums are scaled by `kon->post` (the fp8 converts' 2^9) and, with a
// deferred rmsnorm prologue, by 1 / sqrt(mean(x^2) + eps); lanes 0-15 also store this workgroup's slice of the prologue's side
// outputs (KonTail).
struct KonTail {
    float post;        // 512: the weights were accumulated as q * 2^-9
    bool norm;         // rmsnorm prologue: scale by the factor computed from the folded sum of squares
    float eps;
    uint32_t K;
    // the prologue's side outputs (the buffers of the absorbed ops): every workgroup stores `slice` elements from `k0`;
    // lanes 0-15 hold a[k0 + lane] / b[k0 + lane] (requested at kernel start) when slice <= 16
    float a_s, b_s;
    uint32_t k0, slice;
    bool has_pro;
    // PRENORM: the factor comes from the producing launch's partial sums of squares; lane l of wave 0 holds partials
    // l, l + 64, l + 128, l + 192 (requested at kernel start, zero beyond n_ssq)
    bool prenorm;
    float pp[4];
    // ... or the finished factor, when the caller folded the partials under the weight stream (n_ssq <= 256: prenorm_factor)
    bool inv_known = false;
    float inv_pre = 1.0f;
};
// PRENORM: 1 / sqrt(mean(a^2) + eps) from the producing launch's <= 256 partial sums of squares, four per lane (pp: partials
// l, l + 64, l + 128, l + 192, zero beyond n_ssq), folded in a fixed order — the same value in every wave of every workgroup.
// The partials are requested BEFORE the weights (loads return in order) and folded while the weights fly: in the tail, behind
// the cross-wave barrier, the same arithmetic cost 0.45 us per launch (stamps, SmolLM-135M).
__device__ __forceinline__ float prenorm_factor(const float (&pp)[4], uint32_t K, float eps) {
    const float ss = rows_sum4(row16_sum((pp[0] + pp[1]) + (pp[2] + pp[3])));
    return 1.0f / sqrtf(ss / (float)K + eps); // reference.zig:365
}
// wave 0 of a K-on-lanes workgroup, behind the barrier: the deferred rmsnorm factor (1 without a norm) and this workgroup's slice
// of the prologue's side outputs
__device__ __forceinline__ float kon_factor_and_slices(const KonTail& kt, const QMVArgs& a, const float* red, uint32_t n_waves, uint32_t lane) {
    const uint32_t col = lane & 15;
    float inv = 1.0f;
    if (kt.inv_known) {
        inv = kt.inv_pre;
    } else if (kt.prenorm) { // the producer's partials, folded in a fixed order (the same value in every workgroup of the launch)
        float s4 = (kt.pp[0] + kt.pp[1]) + (kt.pp[2] + kt.pp[3]);
        for (uint32_t j = lane + 256; j < a.pro.n_ssq; j += 64) s4 += a.pro.ssq[j];
        const float ss = rows_sum4(row16_sum(s4));
        inv = 1.0f / sqrtf(ss / (float)kt.K + kt.eps); // reference.zig:365
    } else if (kt.norm) { // the waves' sums of squares, in wave order (the same value in every workgroup of the launch)
        const float sv = red[kMaxWaves * 16 + min(col, n_waves - 1)];
        const float ss = row16_sum(col < n_waves ? sv : 0.f);
        inv = 1.0f / sqrtf(ss / (float)kt.K + kt.eps); // reference.zig:365
    }
    if (kt.has_pro && lane < 16) { // this workgroup's slice of the absorbed ops' outputs (mid = a * inv, x = mid * b: reference order)
        const QmvPrologue& pr = a.pro;
        if (kt.slice <= 16) {
            const uint32_t k = kt.k0 + lane;
            if (lane < kt.slice && k < kt.K) {
                const float mid = kt.norm ? kt.a_s * inv : kt.a_s;
                if (kt.norm && pr.store_mid) pr.store_mid[k] = mid;
                if (pr.store_x) pr.store_x[k] = mid * kt.b_s;
            }
        } else { // few workgroups, long vector: a loop over the slice (not a shape the LLaMA plans produce)
            for (uint32_t k = kt.k0 + lane; k < min(kt.k0 + kt.slice, kt.K); k += 16) {
                const float mid = kt.norm ? pr.a[k] * inv : pr.a[k];
                if (kt.norm && pr.store_mid) pr.store_mid[k] = mid;
                if (pr.store_x) pr.store_x[k] = mid * pr.b[k];
            }
        }
    }
    return inv;
}
template <bool GROUPED, bool KON = false>
__device__ __forceinline__ void reduce_store(float acc, float* red, const QMVArgs& a, uint32_t pi, float* out0, uint32_t g, uint32_t m,
                                             float pre0, bool have_pre0, uint32_t n_waves, const QmvPublish* pub = nullptr, const KonTail* kon = nullptr,
                                             float pre_g = 0.f) {
    const uint32_t lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (!KON) {
        acc = rows_sum4(acc);
        if (lane < 16) red[w * 16 + lane] = acc;
    }
    // a recognised chain's vector operand is requested before the barrier (the stream is over: a load under a branch
    // costs nothing here) so that it lands under the cross-wave fold
    // (grouped launches only — gate / up: single-matrix launches carry residual adds, and every scalar load in this tail counts)
    uint32_t kind = kEpiGeneric;
    float ones = 0.f;
    if (GROUPED) {
        kind = pi == 1 ? a.parts[1].epi_kind : pi == 2 ? a.parts[2].epi_kind : pi == 3 ? a.parts[3].epi_kind : a.parts[0].epi_kind;
        if (kind == kEpiSilu && w == 0) {
            const float* const ones_p = pi == 1 ? a.parts[1].epi[2].operand : pi == 2 ? a.parts[2].epi[2].operand : pi == 3 ? a.parts[3].epi[2].operand : a.parts[0].epi[2].operand;
            ones = ones_p[g * 16 + (lane & 15)];
        }
    }
    __syncthreads();
    if (w != 0) return;
    const uint32_t r = lane >> 4, col = lane & 15;
    float part[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) part[j] = red[min(r + 4 * j, n_waves - 1) * 16 + col]; // clamped, unconditional
    float v = 0.f;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) v += r + 4 * j < n_waves ? part[j] : 0.f;
    v = rows_sum4(v);
    if (KON) v *= kon->post * kon_factor_and_slices(*kon, a, red, n_waves, lane);
    if (!KON && kon) v *= kon_factor_and_slices(*kon, a, red, n_waves, lane); // (the n-on-lanes body under PRENORM: the factor and the slices only)
    if (lane < 16) {
        const uint32_t n = g * 16 + threadIdx.x;
        // part 0's output pointer is a preloaded argument; further parts' come from the argument block
        // (static indices), long arrived by now; the row stride only matters for M > 1
        float* out = out0;
        if (GROUPED) out = pi == 1 ? a.parts[1].out : pi == 2 ? a.parts[2].out : pi == 3 ? a.parts[3].out : out;
        float* out_row = out;
        if (m) {
            uint32_t out_rs = a.parts[0].out_rs;
            if (GROUPED) out_rs = pi == 1 ? a.parts[1].out_rs : pi == 2 ? a.parts[2].out_rs : pi == 3 ? a.parts[3].out_rs : out_rs;
            out_row = out + (uint64_t)m * out_rs;
        }
        if (pub) // (fused launches carry no epilogues on these parts: planner)
            __hip_atomic_store((__attribute__((address_space(1))) float*)out_row + n, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            out_row[n] = v;
        if (GROUPED && kind == kEpiSilu) {
            if (pi == 0)
                run_epilogue_silu(a.parts[0], n, v, ones);
            else if (pi == 1)
                run_epilogue_silu(a.parts[1], n, v, ones);
            else if (pi == 2)
                run_epilogue_silu(a.parts[2], n, v, ones);
            else
                run_epilogue_silu(a.parts[3], n, v, ones);
        } else if (!GROUPED && have_pre0 && a.parts[0].epi_kind == kEpiResidual) {
            const float h = v + pre0; // (commutative: the interpreter's swapped flag does not matter)
            a.parts[0].epi[0].store[n] = h;
            if (a.next.xg_out) { // the next launch's rmsnorm -> mul(gamma) prologue, prepared here (QmvNextNorm)
                a.next.xg_out[n] = h * pre_g;
                const float sq = row16_sum(h * h);
                if (lane == 0) a.next.ssq_out[g] = sq;
            }
        } else if (!GROUPED && have_pre0 && a.parts[0].epi_kind == kEpiMulVec) {
            a.parts[0].epi[0].store[n] = v * pre0;
        } else if (!GROUPED || pi == 0)
            run_epilogue(a.parts[0], n, v, out_row, pre0, !GROUPED && have_pre0);
        else if (pi == 1)
            run_epilogue(a.parts[1], n, v, out_row);
        else if (pi == 2)
            run_epilogue(a.parts[2], n, v, out_row);
        else
            run_epilogue(a.parts[3], n, v, out_row);
    }
    if (pub) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the write-through stores of this wave's 16 lanes have left
#endif
        if (lane == 0 && !(pub->drop && pi == 0 && g == 0))
            __hip_atomic_fetch_add((__attribute__((address_space(1))) uint32_t*)pub->cnt + 32 * (pub->base[pi < 3 ? pi : 2] + (g >> pub->groups_shift)), 1u,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// DEPTH = k-units a lane keeps in flight per group (host picks 1/2/4 from the step count so short
// K does not issue clamped duplicate loads). Loop shape: load group 0; for every further group
// {load next; compute current}; compute last — the prefetch is unconditional inside the loop so
// hipcc can keep counted vmcnt waits (a prefetch under a branch would degrade them to vmcnt(0)).


// XD ("x direct"): x is not staged through LDS; each lane loads the x value(s) of "its" k of the
// unit together with the unit's weights (dword loads, L2-resident vector), so the kernel has no
// x round trip + LDS store + barrier before its first FMA. PROMUL: x = a * b (the absorbed
// elementwise mul), computed per lane; workgroup (0,0) stores it.
struct XDirect {
    const float* a;
    const float* b;
    float* store_x;   // nullptr unless this workgroup owns the prologue's side outputs
    float* store_mid; // rmsnorm prologue: the normalised vector before the gain
    uint32_t K;
    float inv;        // rmsnorm factor (norm == true)
    bool norm;
    // x of one k under the prologue: a * b, or (a * inv) * b for rmsnorm -> repeat(gamma) -> mul
    __device__ __forceinline__ float pro_value(float av, float bv, uint32_t k, bool live) const {
        float v = av;
        if (norm) {
            v = av * inv;
            if (store_mid && live) store_mid[k] = v;
        }
        v = v * bv;
        if (store_x && live) store_x[k] = v;
        return v;
    }
};

// sum of squares of x[0, K) over the whole workgroup, fixed order, in two halves so the kernel can
// issue these loads BEFORE its weight loads (vmcnt is in order: loads issued after the weights would
// make the reduction wait for the HBM stream). `vec`: x is 16-byte aligned and K % 4 == 0
// (host-checked): 4 float4 per thread cover 16*T elements; else 4 dwords per thread cover 4*T.
struct SumsqRegs {
    float4 v[4];
};
__device__ __forceinline__ SumsqRegs sumsq_fetch(const float* x, uint32_t K, bool vec, uint32_t T) {
    SumsqRegs r;
    if (vec) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t k = 4 * (threadIdx.x + j * T);
            r.v[j] = *(const float4*)(x + (k < K ? k : 0));
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) r.v[j] = make_float4(x[min(threadIdx.x + j * T, K - 1)], 0.f, 0.f, 0.f);
    }
    return r;
}
__device__ __forceinline__ float block_sumsq_direct(const SumsqRegs& r, const float* x, uint32_t K, float* red, bool vec, uint32_t T) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float4 v = r.v[j];
            if (4 * (threadIdx.x + j * T) < K) s[j] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        for (uint32_t k = 4 * threadIdx.x + 16 * T; k < K; k += 4 * T) { // K > 16 * T: the rest, one load at a time
            const float4 v = *(const float4*)(x + k);
            s[0] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (threadIdx.x + j * T < K) s[j] += r.v[j].x * r.v[j].x;
        for (uint32_t k = threadIdx.x + 4 * T; k < K; k += T) s[0] += x[k] * x[k];
    }
    return block_total((s[0] + s[1]) + (s[2] + s[3]), red, T);
}

template <typename ST, int DEPTH, bool XD, bool PROMUL, bool NT>
struct Q4Group {
    uint4 wq[DEPTH];
    Pair<ST> s2[DEPTH];
    float xa[XD ? DEPTH : 1], xb[XD ? DEPTH : 1], ya[XD && PROMUL ? DEPTH : 1], yb[XD && PROMUL ? DEPTH : 1];
    // WHAT: 0 = weights and x, 1 = weights only, 2 = x only; AGENT: x through agent-scope loads (a consumer of values
    // stored earlier in the SAME launch: never a stale line of this XCD's L2)
    template <int WHAT = 0, bool AGENT = false>
    __device__ __forceinline__ void load(const uint4* qs, const Pair<ST>* sc, uint32_t u, uint32_t stride, uint32_t u_last,
                                         const XDirect& xd, uint32_t i) {
        using gf32 = const __attribute__((address_space(1))) float;
#pragma unroll
        for (int d = 0; d < DEPTH; d++) { // clamped, unconditional
            const uint32_t ud = min(u + d * stride, u_last);
            if (XD && WHAT != 1) {
                const uint32_t ka = min(ud * 32 + i, xd.K - 1), kb = min(ud * 32 + 16 + i, xd.K - 1);
                if (AGENT) {
                    xa[d] = __hip_atomic_load((gf32*)xd.a + ka, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    xb[d] = __hip_atomic_load((gf32*)xd.a + kb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    xa[d] = xd.a[ka], xb[d] = xd.a[kb];
                }
                if (PROMUL) ya[d] = xd.b[ka], yb[d] = xd.b[kb];
            }
            if (WHAT != 2) {
                wq[d] = wload<NT>(qs + (uint64_t)ud * 16);
                s2[d] = sc[(uint64_t)ud * 16];
            }
        }
    }
    __device__ __forceinline__ void compute(const float* xs, uint32_t u, uint32_t stride, uint32_t U, uint32_t i,
                                            float& acc0, float& acc1, float& acc2, float& acc3, const XDirect& xd) const {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const uint32_t ud = u + d * stride;
            const bool ok = ud < U;
            const uint32_t uc = min(ud, U - 1);
            float va, vb;
            if (XD) {
                const uint32_t ka = ud * 32 + i, kb = ka + 16;
                va = PROMUL ? xd.pro_value(xa[d], ya[d], ka, ok && ka < xd.K) : xa[d];
                vb = PROMUL ? xd.pro_value(xb[d], yb[d], kb, ok && kb < xd.K) : xb[d];
                va = ka < xd.K ? va : 0.f, vb = kb < xd.K ? vb : 0.f;
            } else {
                va = xs[uc * 32 + i], vb = xs[uc * 32 + 16 + i];
            }
            // all 64 lanes run the DPP section (row_newbcast reads need their source lanes live);
            // out-of-range units get t = 0 (their clamped weights are finite, so 0 * q = 0).
            // cvt_nib yields q/16: the 16 is folded into t (exact power of two)
            float tA = ok ? (to_f32(s2[d].a) * 16.0f) * va : 0.f;
            float tB = ok ? (to_f32(s2[d].b) * 16.0f) * vb : 0.f;
            dpp_fence(tA, tB);
            q4_dword<0>(acc0, acc1, wq[d].x, tA);
            q4_dword<8>(acc2, acc3, wq[d].y, tA);
            q4_dword<0>(acc0, acc1, wq[d].z, tB);
            q4_dword<8>(acc2, acc3, wq[d].w, tB);
        }
    }
};

template <typename ST, int DEPTH, bool XD, bool PROMUL, bool NT>
struct Q8Group {
    uint4 wq[DEPTH];
    ST s1[DEPTH];
    float xa[XD ? DEPTH : 1], ya[XD && PROMUL ? DEPTH : 1];
    template <int WHAT = 0, bool AGENT = false>
    __device__ __forceinline__ void load(const uint4* qs, const ST* sc, uint32_t u, uint32_t stride, uint32_t u_last,
                                         const XDirect& xd, uint32_t i) {
        using gf32 = const __attribute__((address_space(1))) float;
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const uint32_t ud = min(u + d * stride, u_last);
            if (XD && WHAT != 1) {
                const uint32_t ka = min(ud * 16 + i, xd.K - 1);
                xa[d] = AGENT ? __hip_atomic_load((gf32*)xd.a + ka, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : xd.a[ka];
                if (PROMUL) ya[d] = xd.b[ka];
            }
            if (WHAT != 2) {
                wq[d] = wload<NT>(qs + (uint64_t)ud * 16);
                s1[d] = sc[(uint64_t)ud * 16];
            }
        }
    }
    __device__ __forceinline__ void compute(const float* xs, uint32_t u, uint32_t stride, uint32_t U, uint32_t i,
                                            float& acc0, float& acc1, float& acc2, float& acc3, const XDirect& xd) const {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const uint32_t ud = u + d * stride;
            const bool ok = ud < U;
            float va;
            if (XD) {
                const uint32_t ka = ud * 16 + i;
                va = PROMUL ? xd.pro_value(xa[d], ya[d], ka, ok && ka < xd.K) : xa[d];
                va = ka < xd.K ? va : 0.f;
            } else {
                va = xs[min(ud, U - 1) * 16 + i];
            }
            float t = ok ? to_f32(s1[d]) * va : 0.f;
            dpp_fence(t);
            q8_dword<0>(acc0, acc1, wq[d].x, t);
            q8_dword<4>(acc2, acc3, wq[d].y, t);
            q8_dword<8>(acc0, acc1, wq[d].z, t);
            q8_dword<12>(acc2, acc3, wq[d].w, t);
        }
    }
};

// Q4: unit = 32 k, UNIT_X = 32 floats of x per unit; Q8: unit = 16 k.
// PRO: the launch has a prologue (second vector b, optional rmsnorm); GROUPED: more than one matrix.
// The leading 14 argument dwords are what the kernel needs before it can issue its first loads; the
// build preloads them into SGPRs (-amdgpu-kernarg-preload-count=16), so the x and weight streams
// start without waiting for an s_load round trip to the argument block. A grouped launch (q/k/v,
// gate/up) whose weights sit back to back in the weight arenas (compile_program packs them in index
// order) finds its part from NB2_0..2 alone: part t starts at block sum(NB2_u, u < t) and at the
// same offset, in column groups, inside the arenas. `a` carries the rest (outputs, prologue,
// epilogues, non-contiguous parts), needed late or rarely.
#define QMV_HEAD_PARAMS                                                                                                 \
    const uint4 *__restrict__ qs0, const void *__restrict__ sc0, float *__restrict__ out0, const float *__restrict__ xa_base, \
        const float *__restrict__ xb_base, uint32_t in_rs, uint32_t K, uint32_t nb2_0_flags /* NB2_0 (20 bits) | (waves - 1) << 20 | n_parts << 24 | contiguous << 28 | rmsnorm prologue << 29 | x_vec << 30 */, \
        uint32_t nb2_12 /* NB2_1 | NB2_2 << 16 */
// The consumer side of an in-launch hand-off (the O projection behind the decode attention, qkv_attn_o_kernel): the
// weights do not depend on the producers and are requested first; then wave 0 waits (bounded) for the producers' counter
// to pass this workgroup's private `seen` word + `need`, and x is read with agent-scope loads.
struct QmvWait {
    const uint32_t* cnt; // one monotonic counter, bumped once by every producer
    uint32_t* seen;      // [workgroups of this part]
    uint32_t need;
    uint32_t* timeout;
    uint32_t poll_sleep = 10; // s_sleep units (64 clocks) between two polls
    uint32_t pre_sleep = 0;   // experiments: s_sleep units before the weights are requested
};

// PROM: 0 = no prologue, 1 = in-kernel prologue (second vector b, optional rmsnorm), 2 = PRENORM (x-direct launches only): xa_base
// is the PRODUCER's a * gamma with its partial sums of a^2 right behind it (runtime.hip: arm_prenorm), xb_base the original a;
// the finished sums are scaled by 1 / sqrt(mean(a^2) + eps) (the mat-vec is linear in x) and every workgroup stores a 16-element
// slice of the absorbed ops' outputs — no second vector, no reduction over x and no barrier in front of the first FMA
__device__ __forceinline__ void kon_pair_finish(const float* red, const QMVArgs& a, float* out0, uint32_t g, uint32_t n_waves, const KonTail& kt, float ones); // (below)
// PAIR (x-direct launches only; the gate / up launch of a SwiGLU block whose next launch would multiply silu(gate) and up as its
// prologue — runtime.hip: arm_pair): the workgroup computes the same 16 columns of BOTH matrices (two contiguous parts of equal
// shape; x is loaded once) and wave 0 stores the plan's buffers and the product itself (kon_pair_finish), so the down
// projection streams one vector with no prologue. One workgroup per column group of part 0.
template <typename ST, bool XVEC, int DEPTH, bool Q4, int PROM, bool GROUPED, bool XD, bool NT, bool CONSUME = false, bool PAIR = false>
__device__ __forceinline__ void qmatvec_body(QMV_HEAD_PARAMS, const QMVArgs& a, const uint32_t bx, const QmvPublish* pub, const QmvWait* wt = nullptr) {
    constexpr bool PRO = PROM == 1;
    static_assert(PROM != 2 || (XD && !CONSUME), "PRENORM: x-direct launches only");
    static_assert(!PAIR || (XD && Q4 && GROUPED && !CONSUME && PROM != 1), "PAIR: x-direct Q4_0 gate / up launches without an in-kernel prologue");
    constexpr uint32_t UNIT_X = Q4 ? 32 : 16;
    using ScaleT = typename std::conditional<Q4, Pair<ST>, ST>::type;
    using Group = typename std::conditional<Q4, Q4Group<ST, DEPTH, XD, PRO, NT>, Q8Group<ST, DEPTH, XD, PRO, NT>>::type;
    extern __shared__ float smem[];
#ifdef ZGML_TRACE // build with -DZGML_TRACE: the stamps serialise the kernel-argument loads, so never in product builds
#define QMV_STAMP(i) do { if (a.trace && bx == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.trace[i] = wall_clock64(); } while (0)
#else
#define QMV_STAMP(i) do { } while (0)
#endif
    QMV_STAMP(0);
    const uint32_t U = Q4 ? (K + 31) >> 5 : ((K + 31) >> 5) * 2; // units per column group (KC or 2 KC)
    float* xs = smem;                     // U * UNIT_X floats (+4 spare)
    float* red = XD ? smem : smem + U * UNIT_X + 4; // waves * 16 floats
    // which matrix of the group this workgroup belongs to (wave-uniform); fields are selected from
    // statically indexed argument loads
    uint32_t pi = 0;
    const uint4* qs_base = qs0;
    const void* sc_base = sc0;
    // the workgroup size comes with the preloaded head too: blockDim is a hidden kernel argument, i.e. one more scalar-load
    // round trip in front of the first load
    const uint32_t n_waves = ((nb2_0_flags >> 20) & 0xFu) + 1, bdim = n_waves * 64;
    const uint32_t NB2_0 = nb2_0_flags & 0xFFFFFu, n_parts = (nb2_0_flags >> 24) & 0xF, NB2_1 = nb2_12 & 0xFFFFu, NB2_2 = nb2_12 >> 16;
    uint32_t NB2 = NB2_0, block_begin = 0;
    if (GROUPED && !PAIR) {
        if ((nb2_0_flags >> 28) & 1) { // contiguous parts: everything from preloaded scalars
            const uint32_t b1 = NB2_0, b2 = b1 + NB2_1; // (<= 3 parts: a fourth part's size would be an argument-block load in front of the first load)
            if (n_parts > 1 && bx >= b1) pi = 1, block_begin = b1, NB2 = NB2_1;
            if (n_parts > 2 && bx >= b2) pi = 2, block_begin = b2, NB2 = NB2_2;
            qs_base = qs0 + (uint64_t)block_begin * U * 16;
            sc_base = (const ScaleT*)sc0 + (uint64_t)(block_begin >> 1) * U * 16;
        } else {
#pragma unroll
            for (uint32_t t = 1; t < (uint32_t)kMaxQmvParts; t++) {
                const bool take = t < n_parts && bx >= a.parts[t].block_begin;
                pi = take ? t : pi;
                qs_base = take ? a.parts[t].qs : qs_base;
                sc_base = take ? a.parts[t].sc : sc_base;
                NB2 = take ? a.parts[t].NB2 : NB2;
                block_begin = take ? a.parts[t].block_begin : block_begin;
            }
        }
    }
    const uint32_t g = column_group(bx - block_begin, NB2), m = blockIdx.y;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t row = lane >> 4, i = lane & 15;
    const uint32_t stride = n_waves * 4; // units per step (4 rows per wave)
    const uint4* qs = qs_base + (uint64_t)g * U * 16 + i;
    const ScaleT* sc = (const ScaleT*)sc_base + (uint64_t)(g >> 1) * U * 16 + i;
    const uint32_t n_groups = (U + stride * DEPTH - 1) / (stride * DEPTH);
    const uint32_t u_last = U - 1;

    const float* xa_row = xa_base + (uint64_t)m * in_rs;
    uint32_t u = 4 * w + row; // this row's unit in step 0
    const bool pro_owner = XD && PRO && bx == 0 && blockIdx.y == 0;
    XDirect xd{xa_row, xb_base, nullptr, nullptr, K, 1.0f, false};
    Group cur;
    // PAIR: part 1 follows part 0 in the weight arenas (same shape): its column group g sits NB2_0 groups further on
    const uint4* const qs_b = qs + (uint64_t)NB2_0 * U * 16;
    const ScaleT* const sc_b = sc + (uint64_t)(NB2_0 >> 1) * U * 16;
    Group cur_b;
    // prologue kind and x alignment come with the preloaded head (bits 29 / 30), and the owner's side-output pointers are
    // read after the loads are issued: anything taken from the argument block before that is an s_load round trip
    // (~0.7 us on its first line) in front of the kernel's first load
    const bool norm = PRO && ((nb2_0_flags >> 29) & 1) != 0;
    const bool x_vec = ((nb2_0_flags >> 30) & 1) != 0;
    float pn_pp[4] = {0.f, 0.f, 0.f, 0.f}, pn_inv = 1.0f; // PRENORM
    if (XD) {
        SumsqRegs sq;
        if (norm) sq = sumsq_fetch(xa_row, K, x_vec, bdim); // before the weights (in-order vmcnt)
        if (CONSUME) { // weights first, then the hand-off, then x (QmvWait); no prologue in this mode (planner)
            cur.template load<1>(qs, sc, u, stride, u_last, xd, i);
            const uint32_t target = wt->seen[bx] + wt->need;
            if (w == 0) {
                uint32_t spins = 0;
                while ((int32_t)(__hip_atomic_load((const __attribute__((address_space(1))) uint32_t*)wt->cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                    if (++spins > 400000u) { // bounded: never hang the device; the caller sees the flag
                        if (threadIdx.x == 0) __hip_atomic_store(wt->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(10); // ~0.3 us between polls (2: 1724, 10: 1741, 40: 1728 tok/s): 36 workgroups polling ONE line back to back starve the line
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // the agent-scope x loads below stay behind the poll
            if (n_waves > 1) __syncthreads();
            if (threadIdx.x == 0) wt->seen[bx] = target; // (every wave read it before the barrier)
            cur.template load<2, true>(qs, sc, u, stride, u_last, xd, i);
        } else {
            if (PROM == 2) { // the producer's partial sums of squares (K / 16 <= 256 of them, right behind its vector): BEFORE the weights
                const uint32_t n_ssq = K >> 4;
                const float* const ssq = xa_base + K;
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t q = lane + 64 * j;
                    const float v = ssq[min(q, n_ssq - 1)];
                    pn_pp[j] = q < n_ssq ? v : 0.f;
                }
            }
            cur.load(qs, sc, u, stride, u_last, xd, i);
            if (PAIR) cur_b.template load<1>(qs_b, sc_b, u, stride, u_last, xd, i); // (weights only: x is the other group's)
        }
        QMV_STAMP(1); // loads issued
        __builtin_amdgcn_sched_barrier(0); // argument-block reads below wait while the loads above fly
        if (PROM == 2) pn_inv = prenorm_factor(pn_pp, K, a.pro.eps); // (every wave: uniform code, folded while the weights fly)
        if (pro_owner) xd.store_x = a.pro.store_x, xd.store_mid = a.pro.store_mid;
        if (norm) { // the weights are in flight while the vector is reduced
            const float ss = block_sumsq_direct(sq, xa_row, K, red, x_vec, bdim);
            xd.inv = 1.0f / sqrtf(ss / (float)K + a.pro.eps); // reference.zig:365
            xd.norm = true;
        }
        QMV_STAMP(2); // (+ sum of squares)
        QMV_STAMP(3);
    } else {
        const XRegs xa = x_fetch<XVEC>(xa_row, K, bdim);
        const XRegs xb = PRO ? x_fetch<XVEC>(xb_base, K, bdim) : xa;
        cur.load(qs, sc, u, stride, u_last, xd, i);
        QMV_STAMP(1); // loads issued
        float inv = 1.0f;
        if (norm) {
            const float ss = block_sumsq(xa, K, red, bdim);
            inv = 1.0f / sqrtf(ss / (float)K + a.pro.eps); // reference.zig:365
        }
        QMV_STAMP(2); // x arrived (+ sum of squares)
        x_commit<XVEC, PRO>(xs, xa, xb, a.pro, inv, U * UNIT_X, K, xa_row, bdim);
        __syncthreads();
        QMV_STAMP(3); // x staged
    }

    // Touch every 64-byte line of the argument block now (one scalar dword each, consumed after the stream). The epilogue's
    // step descriptors are read from it behind the reduction, under the part selects, where hipcc issues their s_loads
    // late — each a miss in the kernel's tail; with the lines already in the scalar cache they hit (SmolLM-135M +1 %,
    // Llama-2-7B unchanged). Unconditional: under `if (K <= 2048)` the scalar loads sit behind a branch and the counted
    // waits of the whole kernel degrade (-5 %). Fetching the whole descriptor block into SGPRs with three
    // s_load_dwordx16 under the cross-wave fold gave nothing on top of this.
    uint32_t arg_touch = 0;
#if defined(__HIP_DEVICE_COMPILE__) // (the builtin's pointer type differs in the host pass)
    {
        const auto kargs = (const __attribute__((address_space(4))) uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();
        constexpr uint32_t kArgDwords = (56 + sizeof(QMVArgs)) / 4; // QMV_HEAD_PARAMS (5 pointers + 4 dwords), then QMVArgs
#pragma unroll
        for (uint32_t o = 16; o < kArgDwords; o += 16) arg_touch |= kargs[o];
    }
#endif
    // PRENORM: this workgroup's 16-element slice of the absorbed ops' inputs (workgroup bx: k = 16 bx ..; the planner checked
    // that the launch has >= K / 16 workgroups), requested under the weight stream; stored by wave 0 behind the fold (reduce_store)
    KonTail ptail{1.0f, true, 0.f, K, 0.f, 0.f, 16u * bx, 16u, true, true, {0.f, 0.f, 0.f, 0.f}};
    if (PROM == 2) {
        const uint32_t ks = min(ptail.k0 + i, K - 1);
        ptail.a_s = xb_base[ks];
        ptail.b_s = a.pro.b[ks];
        ptail.has_pro = ptail.k0 < K && blockIdx.y == 0;
        ptail.inv_known = !(nb2_0_flags >> 31), ptail.inv_pre = pn_inv; // (bit 31, ZGML_HIP_PRENORM_EARLY=0: the fold in the tail, for A/B runs)
        ptail.eps = a.pro.eps;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) ptail.pp[j] = pn_pp[j];
    }
    // single-matrix launches (O / down projection): the operand of the first epilogue step (the residual) is requested
    // now, under the weight stream, by every lane for its column (unconditional: a load under a branch would degrade
    // the counted waits); the 16 owning lanes use it behind the reduction instead of a dependent ~0.5 us load
    float pre0 = 0.f, pre_g = 0.f;
    bool have_pre0 = false;
    if (!GROUPED) {
        const float* const op0 = a.parts[0].epi[0].operand;
        have_pre0 = m == 0 && a.parts[0].n_epi != 0 && op0 != nullptr && op0 != out0;
        pre0 = (have_pre0 ? op0 : xa_row)[have_pre0 ? g * 16 + i : 0];
        const float* const gp = a.next.xg_out ? a.next.gamma : nullptr; // (the next launch's gain, for the side outputs of reduce_store)
        pre_g = (gp ? gp : xa_row)[gp ? g * 16 + i : 0];
    }
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    float bcc0 = 0.f, bcc1 = 0.f, bcc2 = 0.f, bcc3 = 0.f; // PAIR: part 1's sums
    float ones = 0.f;
    if (PAIR) ones = a.parts[0].epi[2].operand[g * 16 + i]; // the SiLU chain's constant vector, requested under the stream
    auto pair_x = [&](Group& b, const Group& src) { // part 1 multiplies the x values part 0's group loaded
        if constexpr (PAIR) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) b.xa[d] = src.xa[d], b.xb[d] = src.xb[d];
        }
    };
    for (uint32_t gi = 1; gi < n_groups; gi++) {
        Group nxt, nxt_b;
        nxt.template load<0, CONSUME>(qs, sc, u + DEPTH * stride, stride, u_last, xd, i);
        if (PAIR) nxt_b.template load<1>(qs_b, sc_b, u + DEPTH * stride, stride, u_last, xd, i);
        cur.compute(xs, u, stride, U, i, acc0, acc1, acc2, acc3, xd);
        if (PAIR) {
            pair_x(cur_b, cur);
            cur_b.compute(xs, u, stride, U, i, bcc0, bcc1, bcc2, bcc3, xd);
            cur_b = nxt_b;
        }
        cur = nxt;
        u += DEPTH * stride;
    }
    cur.compute(xs, u, stride, U, i, acc0, acc1, acc2, acc3, xd);
    if (PAIR) {
        pair_x(cur_b, cur);
        cur_b.compute(xs, u, stride, U, i, bcc0, bcc1, bcc2, bcc3, xd);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"s"(arg_touch)); // (keeps the touches alive; long arrived)
#endif
    QMV_STAMP(4); // weights streamed
    if constexpr (PAIR) { // both matrices' sums folded by wave 0, which stores the plan's buffers and the product (kon_pair_finish)
        const float va = rows_sum4((acc0 + acc1) + (acc2 + acc3)), vb = rows_sum4((bcc0 + bcc1) + (bcc2 + bcc3));
        if (lane < 16) red[w * 16 + lane] = va, red[(kMaxWaves + 1 + w) * 16 + lane] = vb;
        if (PROM != 2) ptail.has_pro = false, ptail.norm = false, ptail.prenorm = false;
        __syncthreads();
        if (w == 0) kon_pair_finish(red, a, out0, g, n_waves, ptail, ones);
    } else {
        reduce_store<GROUPED>((acc0 + acc1) + (acc2 + acc3), red, a, pi, out0, g, m, pre0, have_pre0, n_waves, pub, PROM == 2 ? &ptail : nullptr, pre_g);
    }
    QMV_STAMP(5);
#undef QMV_STAMP
}

template <typename ST, bool XVEC, int DEPTH, bool Q4, bool PRO, bool GROUPED, bool XD, bool NT>
__global__ void __launch_bounds__(1024) qmatvec_kernel(QMV_HEAD_PARAMS, QMVArgs a) {
    qmatvec_body<ST, XVEC, DEPTH, Q4, PRO, GROUPED, XD, NT>(qs0, sc0, out0, xa_base, xb_base, in_rs, K, nb2_0_flags, nb2_12, a, blockIdx.x, nullptr);
}

// PRENORM launches of this form: Q4_0 with f16 scales, x direct, default cache policy (the short-K models whose weights stay
// in the Infinity Cache: SmolLM-135M's 576 / 1536; the K-on-lanes kernels below carry their own PRENORM form for K > 2048)
template <int DEPTH, bool GROUPED>
__global__ void __launch_bounds__(1024) qmatvec_prenorm_kernel(QMV_HEAD_PARAMS, QMVArgs a) {
    qmatvec_body<__half, false, DEPTH, true, 2, GROUPED, true, false>(qs0, sc0, out0, xa_base, xb_base, in_rs, K, nb2_0_flags, nb2_12, a, blockIdx.x, nullptr);
}

// gate / up in pairs, n-on-lanes form (SmolLM-class models; LDS: two column sets of wave sums)
template <int DEPTH, int PROM>
__global__ void __launch_bounds__(1024) qmatvec_pair_kernel(QMV_HEAD_PARAMS, QMVArgs a) {
    qmatvec_body<__half, false, DEPTH, true, PROM, true, true, false, false, true>(qs0, sc0, out0, xa_base, xb_base, in_rs, K, nb2_0_flags, nb2_12, a, blockIdx.x, nullptr);
}

// ── K ON LANES (QW_Q4K, round 3): the M = 1 mat-vec of GGUF-Q4_0-sourced weights ───────────────────────────────────────
// The form above pays two VALU instructions per weight (SDWA convert + DPP fmac) plus addressing: 3.26 per weight by the SQ
// counters, and at one to three waves per SIMD that issue time shows (profiles/r02_qmatvec_pmc_series.csv: 45 % of the wave
// cycles at 4096 x 11008 are issue stalls). Here a lane owns k instead of n:
//   qs: uint4[NB2][P]  P = ceil(K / 2); item (g, p) = the 32 nibbles of k = 2p, 2p + 1 x the 16 columns of group g, stored
//                      OFFSET-BINARY (q = w + 8 in 0..15: the GGUF nibble itself). dword dd: k = 2p + (dd >> 1), byte b holds
//                      column 16g + 8 (dd & 1) + b in the low nibble and + 4 + b in the high nibble.
//   sc: half2[NB][P]   { scale(2p, j), scale(2p + 1, j) }
// so the multiplier t = scale * x is a per-lane scalar (no DPP broadcast), a byte 0x0N is the fp8 (e4m3) value N * 2^-9
// (denormals and the first binade are linear), v_cvt_pk_f32_fp8 turns two masked bytes into two f32 and one v_pk_fma_f32
// (t broadcast by op_sel) adds both products: 1 + 3/8 VALU instructions per weight instead of 2 + 1/8, ~1.8 with everything.
// The -8 of the offset-binary form is taken out per lane, acc -= 8 * sum(t), mirroring the accumulation chain term by term
// (an all-zero weight block gives exactly 0), before anything is folded across lanes. 64 consecutive lanes read 1 KiB of one
// column group; a workgroup still owns 16 columns and all of K, so grid, epilogues, grouping and the in-launch publish are
// those of the form above. The 16 per-lane column sums are folded across the wave by the gfx950 row swaps (16 values -> 8 ->
// 4 registers), one row fold each, and land in LDS as red[wave][column].
// The rmsnorm -> mul(gamma) prologue is DEFERRED: the mat-vec is linear in x, every k of the vector belongs to exactly one lane
// of the workgroup, so the lanes sum x^2 beside the products and 1 / sqrt(mean + eps) scales the 16 finished sums — no second
// pass over x, no barrier in front of the first FMA. The absorbed ops' own outputs (x * inv, x * inv * gamma) are stored in
// slices, 1 / workgroups of the vector each, by the lanes that store the results.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// one dword of an item: 8 nibbles of one k -> four column pairs of the accumulators. ONE asm statement: left to hipcc the
// converts of all dwords (and items) are hoisted in front of the FMAs, which costs hundreds of live registers (the pair
// kernel spilled 200 of them); here six temporaries are live and converts and FMAs alternate. `tt.x` = t, broadcast to both
// halves of the packed FMA by op_sel_hi (tt.y is never read).
__device__ __forceinline__ void kon_dword(f32x2& a0, f32x2& a1, f32x2& a2, f32x2& a3, uint32_t dw, f32x2 tt) {
    uint32_t lo, hi;
    f32x2 c0, c1, c2, c3;
    asm("v_and_b32_e32 %4, 0xf0f0f0f, %10\n\t"
        "v_lshrrev_b32_e32 %5, 4, %10\n\t"
        "v_cvt_pk_f32_fp8_e32 %6, %4\n\t"
        "v_and_b32_e32 %5, 0xf0f0f0f, %5\n\t"
        "v_cvt_pk_f32_fp8_sdwa %7, %4 src0_sel:WORD_1\n\t"
        "v_cvt_pk_f32_fp8_e32 %8, %5\n\t"
        "v_pk_fma_f32 %0, %6, %11, %0 op_sel_hi:[1,0,1]\n\t"
        "v_cvt_pk_f32_fp8_sdwa %9, %5 src0_sel:WORD_1\n\t"
        "v_pk_fma_f32 %1, %7, %11, %1 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %2, %8, %11, %2 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %3, %9, %11, %3 op_sel_hi:[1,0,1]"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(lo), "=&v"(hi), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3)
        : "v"(dw), "v"(tt));
}
__device__ __forceinline__ void kon_dword(f32x2 (&acc)[8], int c, uint32_t dw, float t) {
    f32x2 tt;
    tt.x = t, tt.y = t;
    kon_dword(acc[c], acc[c + 1], acc[c + 2], acc[c + 3], dw, tt);
}
// PAIR: the same k-pair and columns of a SECOND matrix (same shape, `pair_off` items / scale pairs further on) ride along: x is
// loaded once and both column sets are accumulated (gate / up: kon_pair_finish)
template <bool NT, bool PRO, bool XV, bool PAIR>
struct KonItem { // one lane's share of a step: k = 2p, 2p + 1 x 16 columns
    uint4 wq, wq_b;
    uint32_t s2, s2_b;
    f32x2 xa, xb;
    // unconditional, clamped (counted vmcnt waits need straight-line loads); XV: the vectors are 8-byte aligned and K is even
    __device__ __forceinline__ void load(const uint4* qs, const uint32_t* sc, const float* a, const float* b, uint32_t p, uint32_t p_last, uint32_t K,
                                         uint64_t pair_q, uint64_t pair_s) {
        const uint32_t pd = min(p, p_last);
        if (XV) {
            xa = *(const f32x2*)(a + 2 * pd);
            if (PRO) xb = *(const f32x2*)(b + 2 * pd);
        } else {
            const uint32_t k1 = min(2 * pd + 1, K - 1);
            xa = f32x2{a[2 * pd], a[k1]};
            if (PRO) xb = f32x2{b[2 * pd], b[k1]};
        }
        s2 = sc[pd];
        if (PAIR) s2_b = sc[pair_s + pd];
        wq = wload<NT>(qs + pd);
        if (PAIR) wq_b = wload<NT>(qs + pair_q + pd);
    }
    // the consumer side of an in-launch hand-off (the O projection behind the decode attention): weights and scales do not depend on
    // the producers and are requested first; x is read afterwards with agent-scope loads (never a stale L1 / L2 line). XV only.
    __device__ __forceinline__ void load_w(const uint4* qs, const uint32_t* sc, uint32_t p, uint32_t p_last) {
        const uint32_t pd = min(p, p_last);
        s2 = sc[pd];
        wq = wload<NT>(qs + pd);
    }
    __device__ __forceinline__ void load_x_agent(const float* a, uint32_t p, uint32_t p_last) {
        const uint32_t pd = min(p, p_last);
        const unsigned long long v = __hip_atomic_load((const __attribute__((address_space(1))) unsigned long long*)(a + 2 * pd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xa = f32x2{__uint_as_float((uint32_t)v), __uint_as_float((uint32_t)(v >> 32))};
    }
    __device__ __forceinline__ void compute(uint32_t pd, uint32_t P, uint32_t K, f32x2 (&acc)[8], f32x2 (&acc_b)[8], float& T, float& T_b, float& SS) const {
        const bool ok0 = pd < P, ok1 = ok0 && (XV || 2 * pd + 1 < K);
        float x0 = ok0 ? xa.x : 0.f, x1 = ok1 ? xa.y : 0.f;
        if (PRO) {
            SS += x0 * x0;
            SS += x1 * x1;
            x0 *= xb.x, x1 *= xb.y;
        }
        const __half2 h = *(const __half2*)&s2;
        const float t0 = __low2float(h) * x0, t1 = __high2float(h) * x1; // (scale * x as the reference rounds it, reference.zig:552)
        T += t0;
        T += t1;
        kon_dword(acc, 0, wq.x, t0);
        kon_dword(acc, 4, wq.y, t0);
        kon_dword(acc, 0, wq.z, t1);
        kon_dword(acc, 4, wq.w, t1);
        if (PAIR) {
            const __half2 hb = *(const __half2*)&s2_b;
            const float u0 = __low2float(hb) * x0, u1 = __high2float(hb) * x1;
            T_b += u0;
            T_b += u1;
            kon_dword(acc_b, 0, wq_b.x, u0);
            kon_dword(acc_b, 4, wq_b.y, u0);
            kon_dword(acc_b, 0, wq_b.z, u1);
            kon_dword(acc_b, 4, wq_b.w, u1);
        }
    }
};

// (a, b) -> lanes 0-31: a summed over lane ^ 32, lanes 32-63: b summed over lane ^ 32
__device__ __forceinline__ float kon_fold32(float a, float b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// rows 0 / 2: a summed over lane ^ 16, rows 1 / 3: b summed over lane ^ 16
__device__ __forceinline__ float kon_fold16(float a, float b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// PROM: 0 = no prologue, 1 = in-kernel (x = a * b, optionally the deferred rmsnorm), 2 = PRENORM (xa_base is the producer's a * b,
// the factor comes from its partial sums of squares)
// the 16 per-lane column sums of a wave -> red[column] (lanes i < 4 of each row write; see the fold comment in the body)
__device__ __forceinline__ void kon_fold_wave(f32x2 (&acc)[8], float T, float* red_w, uint32_t r, uint32_t i) {
    // w = q - 8, per lane, before any cross-lane sum: the sums hold t * q / 512, so acc -= T * 8 / 512. With every q = 8 the
    // chain acc <- fl(acc + t / 64) equals T / 64 term by term (a power of two commutes with rounding): exactly 0 comes out
    const f32x2 TT = f32x2{T, T}, m64 = f32x2{-0.015625f, -0.015625f};
#pragma unroll
    for (int c = 0; c < 8; c++) acc[c] = __builtin_elementwise_fma(TT, m64, acc[c]);
    // acc[c] = columns (2c, 2c + 1): after kon_fold32 over column pairs and kon_fold16 over those, register m holds, in row r,
    // the partial of column 4m + {0, 2, 1, 3}[r]; one row fold finishes it
    float a8[8];
#pragma unroll
    for (int m2 = 0; m2 < 8; m2++) a8[m2] = kon_fold32(acc[m2].x, acc[m2].y);
    float a4[4];
#pragma unroll
    for (int m4 = 0; m4 < 4; m4++) a4[m4] = row16_sum(kon_fold16(a8[2 * m4], a8[2 * m4 + 1]));
    const float v = (i & 2) ? ((i & 1) ? a4[3] : a4[2]) : ((i & 1) ? a4[1] : a4[0]);
    if (i < 4) red_w[4 * i + ((r & 1) * 2 + (r >> 1))] = v;
}

// PAIR launches (gate / up): part 0 carries the SiLU chain, part 1 is plain, and the elementwise product silu(gate) * up that
// the plan's NEXT mat-vec (the down projection) would recompute as its prologue is stored here, `pair_out` (runtime.hip:
// arm_pair) — both operands of that product are in this workgroup's registers. Wave 0, behind the barrier.
__device__ __forceinline__ void kon_pair_finish(const float* red, const QMVArgs& a, float* out0, uint32_t g, uint32_t n_waves, const KonTail& kt, float ones) {
    const uint32_t lane = threadIdx.x & 63, r = lane >> 4, col = lane & 15;
    float pa[4], pb[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) { // clamped, unconditional
        const uint32_t ww = min(r + 4 * j, n_waves - 1);
        pa[j] = red[ww * 16 + col], pb[j] = red[(kMaxWaves + 1 + ww) * 16 + col];
    }
    float va = 0.f, vb = 0.f;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) va += r + 4 * j < n_waves ? pa[j] : 0.f, vb += r + 4 * j < n_waves ? pb[j] : 0.f;
    va = rows_sum4(va), vb = rows_sum4(vb);
    const float f = kt.post * kon_factor_and_slices(kt, a, red, n_waves, lane);
    va *= f, vb *= f;
    if (lane < 16) {
        const uint32_t n = g * 16 + lane;
        out0[n] = va;
        a.parts[1].out[n] = vb;
        const float e = expf(-va); // NEG, EXP [store], ADD ones, RECIP, MUL by the gate [store]: run_epilogue_silu
        a.parts[0].epi[1].store[n] = e;
        const float sl = (1.0f / (e + ones)) * va;
        a.parts[0].epi[4].store[n] = sl;
        a.pair_out[n] = sl * vb;
    }
}

template <int DEPTH, int PROM, bool GROUPED, bool XV, bool NT, bool PAIR = false, bool CONSUME = false>
__device__ __forceinline__ void qmatvec_kon_body(QMV_HEAD_PARAMS, const QMVArgs& a, const uint32_t bx, const uint32_t n_blocks, const QmvPublish* pub,
                                                 const QmvWait* wt = nullptr) {
    constexpr bool PRO = PROM == 1;
    static_assert(!CONSUME || (PROM == 0 && !GROUPED && !PAIR && XV), "the consumer form: one matrix, no prologue, aligned x");
    extern __shared__ float smem[];
    float* red = smem; // kMaxWaves * 16 column sums + kMaxWaves sums of squares (PAIR: the second matrix's sums from row kMaxWaves + 1)
#ifdef ZGML_TRACE
#define QMV_STAMP(i) do { if (a.trace && (bx == 0 || bx == n_blocks - 1) && threadIdx.x == 0) a.trace[(bx ? 8 : 0) + i] = wall_clock64(); } while (0)
#else
#define QMV_STAMP(i) do { } while (0)
#endif
    QMV_STAMP(0);
    const uint32_t P = (K + 1) >> 1; // units per column group
    uint32_t pi = 0;
    const uint4* qs_base = qs0;
    const uint32_t* sc_base = (const uint32_t*)sc0;
    const uint32_t n_waves = ((nb2_0_flags >> 20) & 0xFu) + 1, stride = n_waves * 64;
    const uint32_t NB2_0 = nb2_0_flags & 0xFFFFFu, n_parts = (nb2_0_flags >> 24) & 0xF, NB2_1 = nb2_12 & 0xFFFFu, NB2_2 = nb2_12 >> 16;
    uint32_t NB2 = NB2_0, block_begin = 0;
    if (GROUPED && !PAIR) { // (PAIR: two contiguous parts of equal shape, one workgroup per column group of BOTH)
        if ((nb2_0_flags >> 28) & 1) { // contiguous parts: everything from preloaded scalars (see the form above)
            const uint32_t b1 = NB2_0, b2 = b1 + NB2_1;
            if (n_parts > 1 && bx >= b1) pi = 1, block_begin = b1, NB2 = NB2_1;
            if (n_parts > 2 && bx >= b2) pi = 2, block_begin = b2, NB2 = NB2_2;
            qs_base = qs0 + (uint64_t)block_begin * P;
            sc_base = (const uint32_t*)sc0 + (uint64_t)(block_begin >> 1) * P;
        } else {
#pragma unroll
            for (uint32_t t = 1; t < (uint32_t)kMaxQmvParts; t++) {
                const bool take = t < n_parts && bx >= a.parts[t].block_begin;
                pi = take ? t : pi;
                qs_base = take ? a.parts[t].qs : qs_base;
                sc_base = take ? (const uint32_t*)a.parts[t].sc : sc_base;
                NB2 = take ? a.parts[t].NB2 : NB2;
                block_begin = take ? a.parts[t].block_begin : block_begin;
            }
        }
    }
    const uint32_t g = column_group(bx - block_begin, NB2);
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane >> 4, i = lane & 15;
    const uint4* qs = qs_base + (uint64_t)g * P;
    const uint32_t* sc = sc_base + (uint64_t)(g >> 1) * P;
    const uint32_t n_groups = (P + stride * DEPTH - 1) / (stride * DEPTH), p_last = P - 1;
    const bool norm = PROM == 2 || (PRO && ((nb2_0_flags >> 29) & 1) != 0);
    uint32_t p = threadIdx.x;
    // DEPTH items per lane in flight, refilled one by one: item d of the next group is requested into the registers item d of
    // this group has just left (no second register set, and the stream never drains inside a workgroup)
    const uint64_t pair_q = PAIR ? (uint64_t)NB2_0 * P : 0, pair_s = PAIR ? (uint64_t)(NB2_0 >> 1) * P : 0; // part 1 follows part 0 in the arenas
    KonItem<NT, PRO, XV, PAIR> it[DEPTH];
    float pn_pp[4] = {0.f, 0.f, 0.f, 0.f}; // PRENORM
    if constexpr (CONSUME) {
        for (uint32_t z = 0; z < wt->pre_sleep; z++) __builtin_amdgcn_s_sleep(16);
#pragma unroll
        for (int d = 0; d < DEPTH; d++) it[d].load_w(qs, sc, p + d * stride, p_last);
        const uint32_t target = wt->seen[bx] + wt->need;
        if (w == 0) { // one wave polls (bounded: never hang the device; the host sees the flag at its next synchronisation)
            uint32_t spins = 0;
            while ((int32_t)(__hip_atomic_load((const __attribute__((address_space(1))) uint32_t*)wt->cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                if (++spins > 400000u) {
                    if (threadIdx.x == 0) __hip_atomic_store(wt->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                for (uint32_t z = 0; z < wt->poll_sleep; z++) __builtin_amdgcn_s_sleep(1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // the agent-scope x loads below stay behind the poll
        if (n_waves > 1) __syncthreads();
        if (threadIdx.x == 0) wt->seen[bx] = target; // (every wave read it before the barrier)
#pragma unroll
        for (int d = 0; d < DEPTH; d++) it[d].load_x_agent(xa_base, p + d * stride, p_last);
    } else {
        if (PROM == 2) { // the producer's partial sums of squares: BEFORE the weights (loads return in order), folded below while those fly
            const uint32_t n_ssq = K >> 4; // (= a.pro.n_ssq, without the argument-block load)
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t q = lane + 64 * j;
                const float v = (xa_base + K)[min(q, n_ssq - 1)]; // (arm_prenorm lays the partials right behind the vector: no argument-block pointer in front of the first load)
                pn_pp[j] = q < n_ssq ? v : 0.f;
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++) it[d].load(qs, sc, xa_base, xb_base, p + d * stride, p_last, K, pair_q, pair_s);
    }
    // this workgroup's slice of the prologue's side outputs: its inputs are requested now, under the weight stream
    KonTail tail{512.0f, norm, 0.f, K, 0.f, 0.f, 0u, 0u, PROM != 0, PROM == 2, {0.f, 0.f, 0.f, 0.f}};
    if (PROM != 0) { // (PRENORM: xb_base carries the ORIGINAL vector a, a.pro.b the gain: the slices are stored from those)
        tail.slice = (K + n_blocks - 1) / n_blocks;
        tail.k0 = bx * tail.slice;
        const uint32_t ks = min(tail.k0 + i, K - 1);
        tail.a_s = (PROM == 2 ? xb_base : xa_base)[ks];
        tail.b_s = (PROM == 2 ? a.pro.b : xb_base)[ks];
    }
    if (PROM == 2) {
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) tail.pp[j] = pn_pp[j];
    }
    QMV_STAMP(1); // loads issued
    QMV_STAMP(2); // (no separate x phases in this form: same stamp)
    QMV_STAMP(3);
    __builtin_amdgcn_sched_barrier(0);
    if (PROM == 2 && (K >> 4) <= 256 && !(nb2_0_flags >> 31)) // all of the partials are in the registers: the factor now, while the weights fly (else, or with bit 31 — ZGML_HIP_PRENORM_EARLY=0 —, in the tail)
        tail.inv_pre = prenorm_factor(pn_pp, K, a.pro.eps), tail.inv_known = true;
    uint32_t arg_touch = 0; // (argument-block lines into the scalar cache while the loads fly: see the form above)
#if defined(__HIP_DEVICE_COMPILE__)
    {
        const auto kargs = (const __attribute__((address_space(4))) uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();
        constexpr uint32_t kArgDwords = (56 + sizeof(QMVArgs)) / 4;
#pragma unroll
        for (uint32_t o = 16; o < kArgDwords; o += 16) arg_touch |= kargs[o];
    }
#endif
    float pre0 = 0.f, pre_g = 0.f;
    bool have_pre0 = false;
    if (!GROUPED) { // the residual behind the O / down projections, requested under the stream (see the form above)
        const float* const op0 = a.parts[0].epi[0].operand;
        have_pre0 = a.parts[0].n_epi != 0 && op0 != nullptr && op0 != out0;
        pre0 = (have_pre0 ? op0 : xa_base)[have_pre0 ? g * 16 + i : 0];
        const float* const gp = a.next.xg_out ? a.next.gamma : nullptr;
        pre_g = (gp ? gp : xa_base)[gp ? g * 16 + i : 0];
    }
    float ones = 0.f;
    if (PAIR) ones = a.parts[0].epi[2].operand[g * 16 + i]; // the SiLU chain's constant vector, requested under the stream
    f32x2 acc[8], acc_b[PAIR ? 8 : 1];
#pragma unroll
    for (int c = 0; c < 8; c++) acc[c] = f32x2{0.f, 0.f};
#pragma unroll
    for (int c = 0; c < (PAIR ? 8 : 1); c++) acc_b[c] = f32x2{0.f, 0.f};
    float T = 0.f, T_b = 0.f, SS = 0.f;
    auto compute = [&](const KonItem<NT, PRO, XV, PAIR>& item, uint32_t pd) {
        if constexpr (PAIR)
            item.compute(pd, P, K, acc, acc_b, T, T_b, SS);
        else
            item.compute(pd, P, K, acc, acc, T, T_b, SS);
    };
    for (uint32_t gi = 1; gi < n_groups; gi++) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            compute(it[d], p + d * stride);
            __builtin_amdgcn_sched_barrier(0); // (hipcc otherwise moves the refills to the end of the body)
            if constexpr (CONSUME) {
                it[d].load_x_agent(xa_base, p + (DEPTH + d) * stride, p_last);
                it[d].load_w(qs, sc, p + (DEPTH + d) * stride, p_last);
            } else {
                it[d].load(qs, sc, xa_base, xb_base, p + (DEPTH + d) * stride, p_last, K, pair_q, pair_s);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        p += DEPTH * stride;
    }
#pragma unroll
    for (int d = 0; d < DEPTH; d++) compute(it[d], p + d * stride);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"s"(arg_touch));
#endif
    QMV_STAMP(4); // weights streamed
    kon_fold_wave(acc, T, red + w * 16, r, i);
    if constexpr (PAIR) kon_fold_wave(acc_b, T_b, red + (kMaxWaves + 1 + w) * 16, r, i);
    if (PRO) {
        SS = row16_sum(SS);
        SS = kon_fold32(SS, SS);
        SS = kon_fold16(SS, SS);
        if (lane == 0) red[kMaxWaves * 16 + w] = SS;
    }
    if (PROM != 0) tail.eps = a.pro.eps;
    if constexpr (PAIR) {
        __syncthreads();
        if (w == 0) kon_pair_finish(red, a, out0, g, n_waves, tail, ones);
    } else {
        reduce_store<GROUPED, true>(0.f, red, a, pi, out0, g, 0, pre0, have_pre0, n_waves, pub, &tail, pre_g);
    }
    QMV_STAMP(5);
#undef QMV_STAMP
}

template <int DEPTH, int PROM, bool GROUPED, bool XV, bool NT>
__global__ void __launch_bounds__(1024) qmatvec_kon_kernel(QMV_HEAD_PARAMS, QMVArgs a) {
    qmatvec_kon_body<DEPTH, PROM, GROUPED, XV, NT>(qs0, sc0, out0, xa_base, xb_base, in_rs, K, nb2_0_flags, nb2_12, a, blockIdx.x, gridDim.x, nullptr);
}
// gate / up in pairs (8 waves at most: 512-thread launch bound, the two column sets need the registers)
template <int DEPTH, int PROM, bool NT>
__global__ void __launch_bounds__(512, 4) qmatvec_kon_pair_kernel(QMV_HEAD_PARAMS, QMVArgs a) { // (4 waves per SIMD: two workgroups per CU, <= 128 registers)
    qmatvec_kon_body<DEPTH, PROM, true, true, NT, true>(qs0, sc0, out0, xa_base, xb_base, in_rs, K, nb2_0_flags, nb2_12, a, blockIdx.x, gridDim.x, nullptr);
}

// The decode attention and the O projection that consumes it in ONE launch (round 3, Llama-2-7B-class models): workgroups [0, n_attn)
// run the decode-attention body (4 waves each; head-major, then split), the rest the K-on-lanes mat-vec in its consumer form. The
// edge is all-to-all (every column group needs every head), so the hand-off is one counter that every head bumps when its output
// rows are stored write-through — what pays here is not the edge but what overlaps it: the projection's weights (9.4 MB at 4096^2)
// are requested at once and stream while the attention, a 32-workgroup latency chain (record, dynamic words, rope, scores, merges:
// 3.5 us), runs; behind the counter the projection has its x loads, 4 x 32 weights of FMAs per lane and the fold left. All
// workgroups are 256 threads and the kernel is held to 128 registers, so four workgroups per CU are admitted; the launch is only
// built when the whole grid fits three per CU (runtime.hip: fuse_attention_o).
#ifdef ZGML_TRACE // (diagnostics build only: measured slower than two launches; tests load libzgml_hip_trace.so)
struct AttnOArgs {
    const AttnDecodeParams* params;
    float* split_buf;
    uint32_t* split_cnt;
    uint32_t split_min_keys, n_attn, n_sp;
    DecodeHandoff ho; // cnt == nullptr: q / k / v come from the previous launch; out_cnt: the counter the projection waits on
    QmvWait wt;
};
template <int LPK, int DEPTH, bool NT>
__global__ void __launch_bounds__(256, 4) attn_o_kon_kernel(QMV_HEAD_PARAMS, QMVArgs a, AttnOArgs f) {
    if (blockIdx.x < f.n_attn) {
        const uint32_t n_heads = f.ho.n_heads;
        attention_decode_body<LPK, false, 256>(f.params, f.split_buf, f.split_cnt, f.split_min_keys, blockIdx.x % n_heads, blockIdx.x / n_heads, f.n_sp, &f.ho);
    } else {
        qmatvec_kon_body<DEPTH, 0, false, true, NT, false, true>(qs0, sc0, out0, xa_base, xb_base, in_rs, K, nb2_0_flags, nb2_12, a, blockIdx.x - f.n_attn,
                                                                  gridDim.x - f.n_attn, nullptr, &f.wt);
    }
}
#endif

// The q / k / v projection and the decode attention that consumes it in ONE launch (DESIGN.md section 8.0,
// tools/exp/localdep.hip): workgroups [0, n_mv) are the grouped mat-vec's, the rest the attention's (head-major, then
// split). The attention's record fetch, dynamic words, rope tables and KV rows do not depend on the projection and
// overlap it; the edge itself is a per-head-slice counter instead of a kernel boundary. 1024 threads per workgroup (the
// attention's shape); the mat-vec half uses the waves its preloaded head asks for and retires the rest at once.
struct QkvAttnArgs {
    const AttnDecodeParams* params;
    float* split_buf;
    uint32_t* split_cnt;
    uint32_t split_min_keys, n_mv, n_sp;
    uint32_t kvq; // int8 KV caches (the attention half's template variant)
    QmvPublish pub;
    DecodeHandoff ho;
    uint32_t* census; // (the CENSUS instantiation of qkv_attn_kon_kernel only) [0] arrivals, [1] set when a workgroup gave up waiting for the grid
};
template <typename ST, bool Q4, int LPK, bool KVQ, int PROM = 1>
__global__ void __launch_bounds__(1024) qkv_attn_kernel(QMV_HEAD_PARAMS, QMVArgs a, QkvAttnArgs f) {
    constexpr int DEPTH = 1; // K <= 2048 with up to 16 waves: one load step
    if (blockIdx.x < f.n_mv) {
        const uint32_t n_waves = ((nb2_0_flags >> 20) & 0xFu) + 1;
        if (threadIdx.x >= n_waves * 64) return; // (whole waves: they no longer count at the barriers)
        qmatvec_body<ST, false, DEPTH, Q4, PROM, true, true, false>(qs0, sc0, out0, xa_base, xb_base, in_rs, K, nb2_0_flags, nb2_12, a, blockIdx.x, &f.pub);
    } else {
        const uint32_t b = blockIdx.x - f.n_mv, n_heads = f.ho.n_heads; // head-major: the always-active split 0 of every head first
        attention_decode_body<LPK, KVQ>(f.params, f.split_buf, f.split_cnt, f.split_min_keys, b % n_heads, b / n_heads, f.n_sp, &f.ho);
    }
}

// The same for K-on-lanes weights (K > 2048: Llama-2-7B's 768 projection workgroups + one or more per head): 256-thread
// workgroups, both bodies at four waves and within 128 registers, so four workgroups per CU are resident and the whole grid is
// co-resident (fuse_qkv_attention builds the launch only when projection + attention workgroups fit what the occupancy query admits:
// four per CU at <= 128 registers and 96-100 SGPRs, which both the register file and the guide's SGPR admission rule confirm — NO
// margin: the 7B grid is exactly 1024 = 4 x 256; the projection's workgroups carry the lower ids, so should fewer be resident than
// counted the attention's workgroups queue behind them instead of spinning beside them; a wait that gives up is bounded and loud).
// CENSUS = true: the SAME kernel (both bodies compiled in: same registers, same LDS, same launch bounds — the launcher checks the
// register counts agree) whose workgroups, handed a census block, only count themselves and wait (bounded: 200 us) until the
// whole grid has arrived. The planner runs it once per grid size before it builds the fused launch (qkv_attn_kon_census): the
// attention's workgroups spin on workgroups of the same grid, the occupancy query can read one workgroup per CU high near a
// register-file edge (guide: residency and cooperative launch), and the 7B grid is exactly 4 x 256 with no margin.
template <int LPK, int DEPTH, int PROM, bool NT, bool KVQ, bool CENSUS = false>
__global__ void __launch_bounds__(256, 4) qkv_attn_kon_kernel(QMV_HEAD_PARAMS, QMVArgs a, QkvAttnArgs f) {
    if (CENSUS && f.census) {
        if (threadIdx.x == 0) {
            using gu32c = __attribute__((address_space(1))) uint32_t;
            __hip_atomic_fetch_add((gu32c*)f.census, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load((gu32c*)f.census, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > 20000u) { // 200 us of the 100 MHz clock: somebody is not resident
                    __hip_atomic_store((gu32c*)f.census + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        return;
    }
    if (blockIdx.x < f.n_mv) {
        qmatvec_kon_body<DEPTH, PROM, true, true, NT>(qs0, sc0, out0, xa_base, xb_base, in_rs, K, nb2_0_flags, nb2_12, a, blockIdx.x, f.n_mv, &f.pub);
    } else {
        const uint32_t b = blockIdx.x - f.n_mv, n_heads = f.ho.n_heads; // head-major: the always-active split 0 of every head first
        attention_decode_body<LPK, KVQ, 256>(f.params, f.split_buf, f.split_cnt, f.split_min_keys, b % n_heads, b / n_heads, f.n_sp, &f.ho);
    }
}

// ... and the O projection behind the attention in the same launch (third range of workgroups): its weights are requested
// at once, its x (the heads' row stores) is taken over through one counter that every head bumps when its rows are stored.
struct QmvHead { // the leading arguments of a mat-vec launch (QMV_HEAD_PARAMS) as a value
    const uint4* qs0;
    const void* sc0;
    float* out0;
    const float* xa_base;
    const float* xb_base;
    uint32_t in_rs, K, nb2_0_flags, nb2_12;
};
#ifdef ZGML_TRACE // (diagnostics build only: measured slower)
struct QkvAttnOArgs {
    QkvAttnArgs f;
    QmvHead h2;
    QmvWait wt;
    uint32_t n_attn; // attention workgroups (heads x splits)
};
template <typename ST, bool Q4, int LPK>
__global__ void __launch_bounds__(1024) qkv_attn_o_kernel(QMV_HEAD_PARAMS, QMVArgs a, QkvAttnOArgs fo, QMVArgs a2) {
    const QkvAttnArgs& f = fo.f;
    if (blockIdx.x < f.n_mv) {
        const uint32_t n_waves = ((nb2_0_flags >> 20) & 0xFu) + 1;
        if (threadIdx.x >= n_waves * 64) return;
        qmatvec_body<ST, false, 1, Q4, true, true, true, false>(qs0, sc0, out0, xa_base, xb_base, in_rs, K, nb2_0_flags, nb2_12, a, blockIdx.x, &f.pub);
    } else if (blockIdx.x < f.n_mv + fo.n_attn) {
        const uint32_t b = blockIdx.x - f.n_mv, n_heads = f.ho.n_heads;
        attention_decode_body<LPK, false>(f.params, f.split_buf, f.split_cnt, f.split_min_keys, b % n_heads, b / n_heads, f.n_sp, &f.ho);
    } else {
        const QmvHead& h = fo.h2;
        const uint32_t n_waves = ((h.nb2_0_flags >> 20) & 0xFu) + 1;
        if (threadIdx.x >= n_waves * 64) return;
        qmatvec_body<ST, false, 1, Q4, false, false, true, false, true>(h.qs0, h.sc0, h.out0, h.xa_base, h.xb_base, h.in_rs, h.K, h.nb2_0_flags, h.nb2_12, a2,
                                                                       blockIdx.x - f.n_mv - fo.n_attn, nullptr, &fo.wt);
    }
}
#endif

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

// Writes the packed layout from an element source: Src::q(k, n) / Src::scale(k, j).
// One thread per 16-byte item; for the raw source reads are 16-byte coalesced across a group.
template <typename ST, typename Src>
__device__ __forceinline__ void write_packed(const QWeightDev& w, const Src& src) {
    const uint32_t NB = w.N / 32, NB2 = w.N / 16;
    const bool q4 = w.format == QW_Q4;
    const uint32_t U = q4 ? w.KC : 2 * w.KC;
    const uint64_t n_items = (uint64_t)NB2 * U * 16;
    for (uint64_t it = (uint64_t)blockIdx.x * kBlock + threadIdx.x; it < n_items; it += (uint64_t)gridDim.x * kBlock) {
        const uint32_t i = it & 15;
        const uint64_t gu = it >> 4;
        const uint32_t u = gu % U, g = gu / U;
        const uint32_t n = g * 16 + i;
        uint32_t d[4];
        for (int dd = 0; dd < 4; dd++) {
            uint32_t word = 0;
            for (int b = 0; b < 4; b++) {
                if (q4) {
                    const uint32_t k_lo = u * 32 + 8 * dd + b, k_hi = k_lo + 4;
                    const uint32_t lo = k_lo < w.K ? (uint32_t)src.q(k_lo, n) & 15u : 0u;
                    const uint32_t hi = k_hi < w.K ? (uint32_t)src.q(k_hi, n) & 15u : 0u;
                    word |= (lo | (hi << 4)) << (8 * b);
                } else {
                    const uint32_t k = u * 16 + 4 * dd + b;
                    const uint32_t q = k < w.K ? (uint32_t)src.q(k, n) & 255u : 0u;
                    word |= q << (8 * b);
                }
            }
            d[dd] = word;
        }
        ((uint4*)w.qs)[it] = make_uint4(d[0], d[1], d[2], d[3]);
        if ((g & 1) == 0) { // scales are per 32-column block-column: written by the even group
            const uint32_t j = g >> 1;
            const uint64_t ju = (uint64_t)j * U + u;
            if (q4) {
                const uint32_t ka = u * 32 + i, kb = ka + 16;
                const float sa = ka < w.K ? src.scale(ka, j, NB) : 0.f;
                const float sb = kb < w.K ? src.scale(kb, j, NB) : 0.f;
                ((Pair<ST>*)w.sc)[ju * 16 + i] = Pair<ST>{ST(sa), ST(sb)};
            } else {
                const uint32_t k = u * 16 + i;
                ((ST*)w.sc)[ju * 16 + i] = ST(k < w.K ? src.scale(k, j, NB) : 0.f);
            }
        }
    }
}

// the K-on-lanes layout (QW_Q4K, see qmatvec_kon_body): one thread per 16-byte item (g, p)
template <typename Src>
__device__ __forceinline__ void write_packed_kon(const QWeightDev& w, const Src& src) {
    const uint32_t NB = w.N / 32, NB2 = w.N / 16, P = (w.K + 1) / 2;
    const uint64_t n_items = (uint64_t)NB2 * P;
    for (uint64_t it = (uint64_t)blockIdx.x * kBlock + threadIdx.x; it < n_items; it += (uint64_t)gridDim.x * kBlock) {
        const uint32_t pp = it % P, g = it / P;
        uint32_t d[4];
        for (int dd = 0; dd < 4; dd++) {
            const uint32_t k = 2 * pp + (dd >> 1), c0 = 16 * g + 8 * (dd & 1);
            uint32_t word = 0;
            for (int b = 0; b < 4; b++) { // offset-binary; k >= K (odd K): w = 0
                const uint32_t lo = k < w.K ? (uint32_t)(src.q(k, c0 + b) + 8) & 15u : 8u;
                const uint32_t hi = k < w.K ? (uint32_t)(src.q(k, c0 + 4 + b) + 8) & 15u : 8u;
                word |= (lo | (hi << 4)) << (8 * b);
            }
            d[dd] = word;
        }
        ((uint4*)w.qs)[it] = make_uint4(d[0], d[1], d[2], d[3]);
        if ((g & 1) == 0) {
            const uint32_t j = g >> 1, k0 = 2 * pp, k1 = k0 + 1;
            const __half sa = __float2half_rn(src.scale(k0, j, NB)), sb = __float2half_rn(k1 < w.K ? src.scale(k1, j, NB) : 0.f);
            ((uint32_t*)w.sc)[(uint64_t)j * P + pp] = (uint32_t)__half_as_ushort(sa) | ((uint32_t)__half_as_ushort(sb) << 16);
        }
    }
}

struct SynthSrc { // SURVEY §8d generator
    uint32_t N, id;
    bool q4;
    __device__ int q(uint32_t k, uint32_t n) const {
        const uint64_t flat = (uint64_t)k * N + n;
        return q4 ? synth_q4(flat, id) : synth_q8(flat, id);
    }
    __device__ float scale(uint32_t k, uint32_t j, uint32_t NB) const { return synth_scale((uint64_t)k * NB + j, id); }
};

struct RawSrc { // QuantizedWeightUpload as uploaded: int8 [K,N] + one f32 scale per 32 flat elements
    const int8_t* data;
    const float* scales;
    uint32_t N;
    __device__ int q(uint32_t k, uint32_t n) const { return data[(uint64_t)k * N + n]; }
    __device__ float scale(uint32_t k, uint32_t j, uint32_t NB) const { return scales[(uint64_t)k * NB + j]; }
};

struct GgufSrc { // GGUF Q4_0 / Q8_0 blocks over the flat [K,N] order (gguf_loader.zig:118-151)
    const uint8_t* raw;
    uint32_t N;
    bool q4;
    __device__ int q(uint32_t k, uint32_t n) const {
        const uint64_t flat = (uint64_t)k * N + n, blk = flat >> 5;
        const uint32_t j = flat & 31;
        if (q4) {
            const uint8_t b = raw[blk * 18 + 2 + (j >> 1)]; // the reference's interleaved order (F3): elem 2i low, 2i+1 high
            return (int)((j & 1) ? (b >> 4) : (b & 15)) - 8;
        }
        return (int)(int8_t)raw[blk * 34 + 2 + j];
    }
    __device__ float scale(uint32_t k, uint32_t j, uint32_t NB) const {
        const uint8_t* b = raw + ((uint64_t)k * NB + j) * (q4 ? 18 : 34);
        return __half2float(__ushort_as_half((unsigned short)(b[0] | (b[1] << 8))));
    }
};

__global__ void __launch_bounds__(kBlock) pack_gguf_kernel(const uint8_t* __restrict__ raw, QWeightDev w) {
    if (w.format == QW_Q4K)
        write_packed_kon(w, GgufSrc{raw, w.N, true});
    else
        write_packed<__half>(w, GgufSrc{raw, w.N, w.format == QW_Q4}); // GGUF scales are f16: always exact
}

template <typename ST>
__global__ void __launch_bounds__(kBlock) synth_packed_kernel(QWeightDev w, uint32_t id) {
    if (w.format == QW_Q4K)
        write_packed_kon(w, SynthSrc{w.N, id, true});
    else
        write_packed<ST>(w, SynthSrc{w.N, id, w.format == QW_Q4});
}

template <typename ST>
__global__ void __launch_bounds__(kBlock) pack_kernel(const int8_t* __restrict__ data,
                                                      const float* __restrict__ scales, QWeightDev w) {
    if (w.format == QW_Q4K)
        write_packed_kon(w, RawSrc{data, scales, w.N});
    else
        write_packed<ST>(w, RawSrc{data, scales, w.N});
}


// LDS the mat-vec needs: x (K padded to whole units) + the cross-wave reduction area.
size_t qmv_lds_bytes(const QWeightDev& w) {
    const uint32_t U = w.format == QW_Q4 ? w.KC : 2 * w.KC;
    return ((size_t)U * (w.format == QW_Q4 ? 32 : 16) + 4 + kMaxWaves * 16) * sizeof(float);
}

} // namespace

// ── device-side repack (compile_program time) ───────────────────────────────────────────────
void packed_bytes(QWFormat format, uint32_t scale_f16, uint64_t K, uint64_t N, uint64_t* qs_bytes,
                  uint64_t* sc_bytes) {
    const uint64_t NB = N / 32, NB2 = N / 16, KC = (K + 31) / 32;
    if (format == QW_Q4K) { // K on lanes: 16-byte items [NB2][P], f16 scale pairs [NB][P]
        const uint64_t P = (K + 1) / 2;
        *qs_bytes = NB2 * P * 16;
        *sc_bytes = NB * P * 4;
        return;
    }
    const uint64_t U = format == QW_Q4 ? KC : 2 * KC;
    *qs_bytes = NB2 * U * 16 * 16;
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

void launch_pack_gguf(hipStream_t s, const uint8_t* raw_blocks, const QWeightDev& out) {
    pack_gguf_kernel<<<2048, kBlock, 0, s>>>(raw_blocks, out);
}

bool qweight_packable(uint64_t K, uint64_t N, uint64_t bs) {
    if (bs != 32 || N == 0 || K == 0 || N % 32 != 0 || N / 16 >= (1u << 20)) return false; // (the mat-vec head carries N / 16 in 20 bits)
    // x (padded K) must fit the workgroup's LDS next to the reduction area
    const uint64_t KC = (K + 31) / 32;
    return (KC * 32 + 4 + kMaxWaves * 16) * sizeof(float) <= kMaxLds;
}

// Can a mat-vec launch over `w` (K inputs, `total_cols` output columns over all its parts, M rows) take its rmsnorm -> mul(gamma)
// prologue PREPARED by the producing launch (QMV_PRO_PRENORM)? K-on-lanes weights: always (their kernels carry the form);
// n-on-lanes Q4_0 with f16 scales: the x-direct launches under the default cache policy, with one workgroup per 16 inputs
// at least (every workgroup stores a 16-element slice of the absorbed ops' outputs).
bool qmv_prenorm_ok(const QWeightDev& w, uint32_t K, uint64_t total_cols, uint32_t M) {
    if (w.format == QW_Q4K) return true;
    static const bool xd_enabled = !(getenv("ZGML_QMV_XDIRECT") && atoi(getenv("ZGML_QMV_XDIRECT")) == 0);
    static const bool nol = !(getenv("ZGML_HIP_PRENORM_NOL") && atoi(getenv("ZGML_HIP_PRENORM_NOL")) == 0);
    return nol && xd_enabled && w.format == QW_Q4 && w.scale_f16 && !w.stream_nt && M == 1 && K % 16 == 0 && K <= 4096 && total_cols >= K; // (<= 256 partial sums)
}

// Can the gate / up launch over two weights shaped like `w` run as a PAIR launch (one workgroup computes the same 16 columns of both
// and stores silu(gate) * up itself)? K-on-lanes weights: yes; n-on-lanes Q4_0 with f16 scales: the x-direct launches under the
// default cache policy (ZGML_HIP_PAIR_NOL=0 keeps those as a grouped launch + the down projection's MUL prologue).
bool qmv_pair_ok(const QWeightDev& w) {
    if (w.format == QW_Q4K) return true;
    static const bool xd_enabled = !(getenv("ZGML_QMV_XDIRECT") && atoi(getenv("ZGML_QMV_XDIRECT")) == 0);
    static const bool nol = !(getenv("ZGML_HIP_PAIR_NOL") && atoi(getenv("ZGML_HIP_PAIR_NOL")) == 0);
    return nol && xd_enabled && w.format == QW_Q4 && w.scale_f16 && !w.stream_nt;
}

bool qmv_can_group(const QWeightDev& a, const QWeightDev& b) {
    return a.format != QW_RAW && a.format == b.format && a.scale_f16 == b.scale_f16 && a.K == b.K && a.KC == b.KC;
}

namespace {

// Waves per workgroup (measured sweep over 1..16 waves on 576^2 .. 4096x32000, tools/bench_matvec.py
// with ZGML_QMV_WAVES): 4 waves (each streaming K/16 units with the depth-4 pipeline) wins or ties
// everywhere up to K ~ 4096 — narrower workgroups keep more of them co-resident per CU, so one
// workgroup's x staging / reduction tail overlaps another's streaming — and 8 waves win for the long
// K of the down projection (11008). 16-wave workgroups (the first design) lose up to 35 % on
// multi-round grids. Inside the fused decode stream the short-K launches (K <= 2048: SmolLM's 576 and
// 1536) behave differently from the isolated sweep: the launch is a single latency chain, and one
// wave per 4 units (one load step each, up to 16 waves) shortens it: SmolLM-135M decode +3.8 %
// (ZGML_QMV_WAVES_SMALLK sweep: 4 -> 1414, 5 -> 1400, 6 -> 1450, 8 -> 1445, 12+ -> 1470 tok/s);
// for K = 4096 more waves lose in the stream (Llama-2-7B: 8 waves -18 %).
uint32_t qmv_waves(const QWeightDev& w, uint32_t total_blocks = 0) {
    (void)total_blocks;
    const uint32_t U = w.format == QW_Q4 ? w.KC : 2 * w.KC;
    uint32_t waves = cdiv(U, 4);
    static const int small_cap = getenv("ZGML_QMV_WAVES_SMALLK") ? atoi(getenv("ZGML_QMV_WAVES_SMALLK")) : 16;
    uint32_t cap = w.K > 6144 ? 8 : (w.K <= 2048 ? (uint32_t)small_cap : 4); // short K: the launch is one latency chain, more waves shorten it
    static const int env_cap = getenv("ZGML_QMV_WAVES") ? atoi(getenv("ZGML_QMV_WAVES")) : 0;
    if (env_cap > 0) cap = (uint32_t)env_cap;
    if (waves > cap) waves = cap;
    return waves < 1 ? 1 : waves;
}

using KernelFn = void (*)(QMV_HEAD_PARAMS, QMVArgs);

template <typename ST, bool XV, bool Q, bool PRO, bool GRP, bool XD>
KernelFn pick_depth(int depth_sel, bool nt) {
    switch (depth_sel) { // nt only exists at full depth: a weight set beyond the Infinity Cache means K >= 4096
        case 0: return qmatvec_kernel<ST, XV, 1, Q, PRO, GRP, XD, false>;
        case 1: return qmatvec_kernel<ST, XV, 2, Q, PRO, GRP, XD, false>;
        default: return nt ? qmatvec_kernel<ST, XV, 4, Q, PRO, GRP, XD, true> : qmatvec_kernel<ST, XV, 4, Q, PRO, GRP, XD, false>;
    }
}
template <typename ST, bool XV, bool Q, bool XD>
KernelFn pick_mode(bool pro, bool grp, int depth_sel, bool nt) {
    if (pro) return grp ? pick_depth<ST, XV, Q, true, true, XD>(depth_sel, nt) : pick_depth<ST, XV, Q, true, false, XD>(depth_sel, nt);
    return grp ? pick_depth<ST, XV, Q, false, true, XD>(depth_sel, nt) : pick_depth<ST, XV, Q, false, false, XD>(depth_sel, nt);
}
template <typename ST>
KernelFn pick_kernel(bool xvec, bool q4, bool pro, bool grp, int depth_sel, bool xd, bool nt) {
    if (xd) return q4 ? pick_mode<ST, false, true, true>(pro, grp, depth_sel, nt) : pick_mode<ST, false, false, true>(pro, grp, depth_sel, nt);
    if (xvec) return q4 ? pick_mode<ST, true, true, false>(pro, grp, depth_sel, nt) : pick_mode<ST, true, false, false>(pro, grp, depth_sel, nt);
    return q4 ? pick_mode<ST, false, true, false>(pro, grp, depth_sel, nt) : pick_mode<ST, false, false, false>(pro, grp, depth_sel, nt);
}

// `fused` (optional): launch qkv_attn_kernel instead — the grouped mat-vec's workgroups followed by `extra_blocks` of the
// decode attention. Returns false (nothing launched) when the mat-vec is not one of the shapes that kernel is built for.
struct FusedO { // the O projection riding in a fused q/k/v + attention launch (qkv_attn_o_kernel)
    QMVArgs a2;
    QmvHead h2;
    QmvWait wt;
    uint32_t blocks2;
};
bool launch_packed(hipStream_t s, QMVArgs& a, const QWeightDev& w0, uint32_t total_blocks, uint32_t M, bool xvec, const QkvAttnArgs* fused = nullptr,
                   uint32_t extra_blocks = 0, uint32_t d_head = 0, const FusedO* fo = nullptr);
// the leading arguments of a single-matrix, prologue-free, x-direct launch with one load step (what the fused O projection
// needs); false: not that shape
static bool plain_head_depth1(const QMVArgs& a, const QWeightDev& w0, uint32_t blocks, QmvHead& h) {
    static const bool xd_enabled = !(getenv("ZGML_QMV_XDIRECT") && atoi(getenv("ZGML_QMV_XDIRECT")) == 0);
    if (!xd_enabled || a.n_parts != 1 || a.pro.kind != QMV_PRO_NONE || w0.format != QW_Q4 || !w0.scale_f16 || w0.stream_nt) return false;
    const uint32_t waves = qmv_waves(w0, blocks);
    if (cdiv(a.U, waves * 4) != 1 || a.parts[0].NB2 >= (1u << 20)) return false;
    h = QmvHead{a.parts[0].qs, a.parts[0].sc, a.parts[0].out, a.pro.a, a.pro.b, a.in_rs, a.K,
                a.parts[0].NB2 | ((waves - 1) << 20) | (1u << 24) | (a.x_vec ? 1u << 30 : 0u), 0u};
    return true;
}

static bool prenorm_late() { // experiments: ZGML_HIP_PRENORM_EARLY=0 folds the prepared norm's partial sums in the kernel's tail (bit 31 of the head flags)
    static const bool late = getenv("ZGML_HIP_PRENORM_EARLY") && atoi(getenv("ZGML_HIP_PRENORM_EARLY")) == 0;
    return late;
}
// ── the K-on-lanes launches (QW_Q4K) ──
// Waves: 8 for K >= 4096 (4096: one load group of 4 items per lane, every load of the workgroup in flight at once;
// tools/exp/kon.hip sweep, round 3: 4096^2 chain 4.06 us with 8 waves against 4.9 / 4.5 with 4 / 16); short K: one wave per
// 64 k-pairs up to the cap (the launch is one latency chain inside the decode stream, as for the form above).
uint32_t kon_waves(const QWeightDev& w) {
    const uint32_t wave_steps = cdiv((w.K + 1) / 2, 64);
    static const int small_cap = getenv("ZGML_QMV_KON_WAVES_SMALLK") ? atoi(getenv("ZGML_QMV_KON_WAVES_SMALLK")) : 16;
    // (in-decode sweep per launch shape, Llama-2-7B, tools/tune_kon.sh: 4 waves for q/k/v, o, gate/up and the LM head, 8 for
    // the down projection's K = 11008: 765 tok/s with 8 everywhere, 792 with 4, 807 with this rule)
    static const int big_env = getenv("ZGML_QMV_KON_WAVES") ? atoi(getenv("ZGML_QMV_KON_WAVES")) : 0;
    const int big_cap = big_env > 0 ? big_env : (w.K > 6144 ? 8 : 4);
    const uint32_t cap = (uint32_t)std::max(1, std::min(16, w.K <= 2048 ? small_cap : big_cap));
    return std::max(1u, std::min(wave_steps, cap));
}

template <int PROM, bool GRP, bool XV>
KernelFn pick_kon_depth(int depth_sel, bool nt) {
    switch (depth_sel) {
        case 0: return qmatvec_kon_kernel<1, PROM, GRP, XV, false>;
        case 1: return qmatvec_kon_kernel<2, PROM, GRP, XV, false>;
        default:
            if constexpr (PROM == 1) // (an in-kernel prologue runs at most two items per lane: launch_packed_kon)
                return qmatvec_kon_kernel<2, PROM, GRP, XV, false>;
            else
                return nt ? qmatvec_kon_kernel<4, PROM, GRP, XV, true> : qmatvec_kon_kernel<4, PROM, GRP, XV, false>;
    }
}
template <int PROM>
KernelFn pick_kon_mode(bool grp, bool xv, int depth_sel, bool nt) {
    if (xv) return grp ? pick_kon_depth<PROM, true, true>(depth_sel, nt) : pick_kon_depth<PROM, false, true>(depth_sel, nt);
    return grp ? pick_kon_depth<PROM, true, false>(depth_sel, nt) : pick_kon_depth<PROM, false, false>(depth_sel, nt);
}
KernelFn pick_kon(int prom, bool grp, bool xv, int depth_sel, bool nt) {
    if (prom == 2) return pick_kon_mode<2>(grp, xv, depth_sel, nt);
    if (prom == 1) return pick_kon_mode<1>(grp, xv, depth_sel, nt);
    return pick_kon_mode<0>(grp, xv, depth_sel, nt);
}

bool launch_packed_kon(hipStream_t s, QMVArgs& a, const QWeightDev& w0, uint32_t total_blocks, bool xvec, const QkvAttnArgs* fused, uint32_t extra_blocks,
                       uint32_t d_head, const FusedO* fo) {
    if (fo) return false; // (the O projection riding along is only built for the n-on-lanes form; off by default)
    a.x_vec = xvec ? 1 : 0;
    const uint32_t P = (a.K + 1) / 2;
    uint32_t waves = kon_waves(w0);
    int tune_depth = 0; // 0: by the step count
    { // experiments: ZGML_QMV_KON_TUNE="<blocks>x<K>:<waves>,..." overrides the wave count of the launches of that grid and K
        static const char* tune = getenv("ZGML_QMV_KON_TUNE");
        for (const char* q = tune; q && *q;) {
            unsigned b = 0, k = 0, wv = 0, dp = 0;
            const int got = sscanf(q, "%ux%u:%u:%u", &b, &k, &wv, &dp);
            if (got >= 3 && b == total_blocks && k == a.K && wv >= 1 && wv <= 16) {
                waves = wv;
                if (got == 4) tune_depth = (int)dp;
            }
            q = strchr(q, ',');
            if (q) q++;
        }
    }
    static const bool contig_ok = !(getenv("ZGML_QMV_CONTIG") && atoi(getenv("ZGML_QMV_CONTIG")) == 0);
    bool contig = a.n_parts > 1 && a.n_parts <= 3 && contig_ok;
    for (uint32_t t = 1; t < a.n_parts && contig; t++) {
        const QMVPartDev &pv = a.parts[t - 1], &pt = a.parts[t];
        contig = (const char*)pt.qs == (const char*)pv.qs + (size_t)pv.NB2 * P * 16 && (const char*)pt.sc == (const char*)pv.sc + (size_t)(pv.NB2 / 2) * P * 4 &&
                 pt.block_begin == pv.block_begin + pv.NB2;
    }
    if (a.n_parts > 1 && (a.parts[1].NB2 > 0xFFFFu || (a.n_parts > 2 && a.parts[2].NB2 > 0xFFFFu))) contig = false; // 16-bit fields
    const uint32_t nb2_12 = contig ? (a.parts[1].NB2 | (a.n_parts > 2 ? a.parts[2].NB2 << 16 : 0u)) : 0u;
    const size_t lds = ((size_t)kMaxWaves * 16 + kMaxWaves) * sizeof(float);
    const bool pro = a.pro.kind != QMV_PRO_NONE, grp = a.n_parts > 1, nt = w0.stream_nt != 0;
    const int prom = a.pro.kind == QMV_PRO_PRENORM ? 2 : (pro ? 1 : 0);
    if (prom == 2 && a.pro.ssq != a.pro.xg + a.K) { // the kernels find the partial sums right behind the producer's vector (arm_prenorm lays them out so)
        fprintf(stderr, "[zgml_hip] ERROR: a PRENORM launch whose partial sums do not follow its vector: not launched\n");
        return false;
    }
    uint32_t waves_used = waves;
    // an in-kernel prologue doubles the vector loads of an item: two items per lane in flight with twice the waves
    // (four items: the compiler runs out of the 128 registers a 1024-thread launch bound leaves)
    if (prom == 1 && cdiv(P, waves * 64) >= 4) waves_used = std::min<uint32_t>(kMaxWaves, waves * 2);
    const uint32_t n_steps = cdiv(P, waves_used * 64);
    int depth_sel = n_steps >= 4 ? 2 : (n_steps >= 2 ? 1 : 0);
    if (tune_depth == 4) depth_sel = 2;
    if (tune_depth == 2) depth_sel = 1;
    if (prom == 1 && depth_sel >= 2) depth_sel = 1;
    // the kernel's leading arguments: PRENORM streams the producer's vector and keeps the original one for the side outputs
    const float* const head_xa = prom == 2 ? a.pro.xg : a.pro.a;
    const float* const head_xb = prom == 2 ? a.pro.a : a.pro.b;
    const uint32_t flags = a.parts[0].NB2 | ((waves_used - 1) << 20) | (a.n_parts << 24) | (contig ? 1u << 28 : 0u) | (a.pro.kind == QMV_PRO_RMSNORM_MUL ? 1u << 29 : 0u) |
                           (a.x_vec ? 1u << 30 : 0u) | (prenorm_late() ? 1u << 31 : 0u);
    if (fused) { // q / k / v + decode attention in one launch of 256-thread workgroups (qkv_attn_kon_kernel)
        using FusedFn = void (*)(const uint4*, const void*, float*, const float*, const float*, uint32_t, uint32_t, uint32_t, uint32_t, QMVArgs, QkvAttnArgs);
        if (!grp || !contig || !xvec || prom == 1 || (d_head != 64 && d_head != 128)) return false;
        const uint32_t steps4 = cdiv(P, 4 * 64);
        const bool deep = steps4 >= 4; // four items per lane in flight (K >= 2048 with four waves), else two
        const uint32_t flags4 = (flags & ~(0xFu << 20)) | (3u << 20);
        FusedFn ff = nullptr;
#define ZGML_QKV_KON(LPK_, KVQ_)                                                                                                                   \
    ff = prom == 2 ? (deep ? (nt ? qkv_attn_kon_kernel<LPK_, 4, 2, true, KVQ_> : qkv_attn_kon_kernel<LPK_, 4, 2, false, KVQ_>) : qkv_attn_kon_kernel<LPK_, 2, 2, false, KVQ_>) \
                   : (deep ? (nt ? qkv_attn_kon_kernel<LPK_, 4, 0, true, KVQ_> : qkv_attn_kon_kernel<LPK_, 4, 0, false, KVQ_>) : qkv_attn_kon_kernel<LPK_, 2, 0, false, KVQ_>)
        if (d_head == 64) {
            if (fused->kvq) ZGML_QKV_KON(16, true); else ZGML_QKV_KON(16, false);
        } else {
            if (fused->kvq) ZGML_QKV_KON(32, true); else ZGML_QKV_KON(32, false);
        }
#undef ZGML_QKV_KON
        QkvAttnArgs f = *fused;
        f.n_mv = total_blocks;
        hipLaunchKernelGGL(ff, dim3(total_blocks + extra_blocks), dim3(256), lds, s, a.parts[0].qs, a.parts[0].sc, a.parts[0].out, head_xa, head_xb, a.in_rs, a.K, flags4, nb2_12,
                           a, f);
        return true;
    }
    if (a.pair_out) { // gate / up in pairs (arm_pair checked the shapes and the epilogues; the layout conditions are checked here)
        const bool pair_ok = a.n_parts == 2 && contig && xvec && prom != 1 && a.parts[0].NB2 == a.parts[1].NB2 && a.parts[0].NB2 % 2 == 0 && waves_used <= 8;
        if (!pair_ok) {
            fprintf(stderr, "[zgml_hip] ERROR: a gate / up pair launch that does not meet the pair kernel's layout conditions: not launched\n");
            return false;
        }
        using PairFn = void (*)(const uint4*, const void*, float*, const float*, const float*, uint32_t, uint32_t, uint32_t, uint32_t, QMVArgs);
        PairFn pf = nullptr;
        // two items per lane in flight = the bytes of four in a single-matrix launch (deeper: the two column sets no longer fit 128 registers)
        if (prom == 2)
            pf = depth_sel == 0 ? qmatvec_kon_pair_kernel<1, 2, false> : (nt ? qmatvec_kon_pair_kernel<2, 2, true> : qmatvec_kon_pair_kernel<2, 2, false>);
        else
            pf = depth_sel == 0 ? qmatvec_kon_pair_kernel<1, 0, false> : (nt ? qmatvec_kon_pair_kernel<2, 0, true> : qmatvec_kon_pair_kernel<2, 0, false>);
        const size_t lds2 = ((size_t)(2 * kMaxWaves + 1) * 16) * sizeof(float);
        hipLaunchKernelGGL(pf, dim3(a.parts[0].NB2), dim3(waves_used * 64), lds2, s, a.parts[0].qs, a.parts[0].sc, a.parts[0].out, head_xa, head_xb, a.in_rs, a.K, flags, nb2_12, a);
        return true;
    }
    const KernelFn fn = pick_kon(prom, grp, xvec, depth_sel, nt);
    hipLaunchKernelGGL(fn, dim3(total_blocks), dim3(waves_used * 64), lds, s, a.parts[0].qs, a.parts[0].sc, a.parts[0].out, head_xa, head_xb, a.in_rs, a.K, flags, nb2_12, a);
    return true;
}

bool launch_packed(hipStream_t s, QMVArgs& a, const QWeightDev& w0, uint32_t total_blocks, uint32_t M, bool xvec, const QkvAttnArgs* fused,
                   uint32_t extra_blocks, uint32_t d_head, const FusedO* fo) {
#ifdef ZGML_TRACE
    { // diagnostics build only: leave out every mat-vec launch of one grid size (wrong results; the token time then drops by that launch's true cost)
        static const uint32_t skip_blocks = getenv("ZGML_HIP_DEBUG_SKIP_GRID") ? (uint32_t)atoi(getenv("ZGML_HIP_DEBUG_SKIP_GRID")) : 0u;
        if (skip_blocks && total_blocks == skip_blocks) return true;
    }
#endif
    if (w0.format == QW_Q4K) { // (only weights whose every use is an M = 1 mat-vec get this layout: compile_program)
        if (M != 1) return false;
        return launch_packed_kon(s, a, w0, total_blocks, xvec, fused, extra_blocks, d_head, fo);
    }
    const bool q4 = w0.format == QW_Q4;
    a.x_vec = xvec ? 1 : 0;
    // x direct (no LDS staging); the rmsnorm prologue reduces the vector while the weights fly
    static const bool xd_enabled = !(getenv("ZGML_QMV_XDIRECT") && atoi(getenv("ZGML_QMV_XDIRECT")) == 0);
    static const bool xd_norm = !(getenv("ZGML_QMV_XDIRECT_NORM") && atoi(getenv("ZGML_QMV_XDIRECT_NORM")) == 0);
    // (measured: with the rmsnorm prologue the extra per-lane dword loads of x and gamma cost more than the
    // LDS round trip they replace once K is large: Llama-2-7B -5 %, SmolLM-135M +2.7 %)
    static const uint32_t xd_norm_max_k = getenv("ZGML_QMV_XDNORM_MAXK") ? (uint32_t)atoi(getenv("ZGML_QMV_XDNORM_MAXK")) : 2048u;
    const bool xd = xd_enabled && (a.pro.kind != QMV_PRO_RMSNORM_MUL || (xd_norm && a.K <= xd_norm_max_k));
    const bool prenorm = a.pro.kind == QMV_PRO_PRENORM;
    if (prenorm && (!qmv_prenorm_ok(w0, a.K, total_blocks * 16, M) || a.pro.ssq != a.pro.xg + a.K)) { // (arm_prenorm asks the same question and lays the partial sums right behind the vector: never reached)
        fprintf(stderr, "[zgml_hip] ERROR: a PRENORM mat-vec launch outside the shapes its kernel is built for: not launched\n");
        return false;
    }
    uint32_t waves = qmv_waves(w0, total_blocks);
    if (!xd && a.pro.kind != QMV_PRO_NONE) // a staged prologue keeps all of x in the register window: 16 floats per thread
        while (waves < (uint32_t)kMaxWaves && waves * 64 * 4 * kXRegs < a.K) waves++;
    // parts back to back in the weight arenas (and <= 4 of them): the kernel needs no argument-block fetch to find them
    static const bool contig_ok = !(getenv("ZGML_QMV_CONTIG") && atoi(getenv("ZGML_QMV_CONTIG")) == 0);
    bool contig = a.n_parts > 1 && a.n_parts <= 3 && contig_ok;
    const size_t sc_elem = (q4 ? 2 : 1) * (w0.scale_f16 ? 2 : 4);
    for (uint32_t t = 1; t < a.n_parts && contig; t++) {
        const QMVPartDev &pv = a.parts[t - 1], &pt = a.parts[t];
        contig = (const char*)pt.qs == (const char*)pv.qs + (size_t)pv.NB2 * a.U * 256 &&
                 (const char*)pt.sc == (const char*)pv.sc + (size_t)(pv.NB2 / 2) * a.U * 16 * sc_elem &&
                 pt.block_begin == pv.block_begin + pv.NB2;
    }
    dim3 grid(total_blocks, M);
    const size_t lds = xd ? (size_t)kMaxWaves * 16 * sizeof(float) : qmv_lds_bytes(w0);
    const uint32_t n_steps = cdiv(a.U, waves * 4);
    const int depth_sel = n_steps >= 4 ? 2 : (n_steps >= 2 ? 1 : 0); // DEPTH 4 / 2 / 1 (DEPTH 8 for the staged-prologue launches of a 7B model: -9 % tok/s, round 2)
    const bool pro = a.pro.kind == QMV_PRO_MUL || a.pro.kind == QMV_PRO_RMSNORM_MUL, grp = a.n_parts > 1;
    const bool nt = w0.stream_nt != 0;
    // the kernel's leading arguments: PRENORM streams the producer's vector and keeps the original one for the side outputs
    const float* const head_xa = prenorm ? a.pro.xg : a.pro.a;
    const float* const head_xb = prenorm ? a.pro.a : a.pro.b;
    const KernelFn fn = prenorm ? (grp ? (depth_sel == 0 ? qmatvec_prenorm_kernel<1, true> : depth_sel == 1 ? qmatvec_prenorm_kernel<2, true> : qmatvec_prenorm_kernel<4, true>)
                                       : (depth_sel == 0 ? qmatvec_prenorm_kernel<1, false> : depth_sel == 1 ? qmatvec_prenorm_kernel<2, false> : qmatvec_prenorm_kernel<4, false>))
                      : w0.scale_f16 ? pick_kernel<__half>(xvec, q4, pro, grp, depth_sel, xd, nt)
                                     : pick_kernel<float>(xvec, q4, pro, grp, depth_sel, xd, nt);
    if (lds > 64 * 1024) // > 64 KiB of dynamic LDS needs an explicit opt-in per kernel (idempotent)
        hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (a.n_parts > 1 && (a.parts[1].NB2 > 0xFFFFu || (a.n_parts > 2 && a.parts[2].NB2 > 0xFFFFu))) contig = false; // 16-bit fields
    const uint32_t nb2_12 = contig ? (a.parts[1].NB2 | (a.n_parts > 2 ? a.parts[2].NB2 << 16 : 0u)) : 0u;
    if (a.pair_out) { // gate / up in pairs (arm_pair checked the shapes and the epilogues; the layout conditions are checked here)
        const bool pair_ok = qmv_pair_ok(w0) && xd && !pro && !fused && a.n_parts == 2 && contig && M == 1 && a.parts[0].NB2 == a.parts[1].NB2 && a.parts[0].NB2 % 2 == 0;
        if (!pair_ok) {
            fprintf(stderr, "[zgml_hip] ERROR: a gate / up pair launch that does not meet the pair kernel's layout conditions: not launched\n");
            return false;
        }
        const KernelFn pf = prenorm ? (depth_sel == 0 ? qmatvec_pair_kernel<1, 2> : depth_sel == 1 ? qmatvec_pair_kernel<2, 2> : qmatvec_pair_kernel<4, 2>)
                                    : (depth_sel == 0 ? qmatvec_pair_kernel<1, 0> : depth_sel == 1 ? qmatvec_pair_kernel<2, 0> : qmatvec_pair_kernel<4, 0>);
        const size_t lds2 = ((size_t)(2 * kMaxWaves + 1) * 16) * sizeof(float);
        hipLaunchKernelGGL(pf, dim3(a.parts[0].NB2), dim3(waves * 64), lds2, s, a.parts[0].qs, a.parts[0].sc, a.parts[0].out, head_xa, head_xb, a.in_rs, a.K,
                           a.parts[0].NB2 | ((waves - 1) << 20) | (a.n_parts << 24) | (1u << 28) | (a.x_vec ? 1u << 30 : 0u) | (prenorm_late() ? 1u << 31 : 0u), nb2_12, a);
        return true;
    }
    if (fused) {
        if (!(xd && (pro || prenorm) && grp && !nt && q4 && w0.scale_f16 && contig && M == 1 && a.n_parts == 3 && (d_head == 64 || d_head == 128))) return false;
        QkvAttnArgs f = *fused;
        f.n_mv = total_blocks;
        const bool fused_kvq = f.kvq != 0;
#ifdef ZGML_TRACE
        if (fo && depth_sel == 0 && !fused_kvq) { // ... and the O projection behind the attention (diagnostics build only)
            QkvAttnOArgs g{f, fo->h2, fo->wt, extra_blocks};
            const dim3 grid3(total_blocks + extra_blocks + fo->blocks2);
            if (d_head == 64)
                hipLaunchKernelGGL((qkv_attn_o_kernel<__half, true, 16>), grid3, dim3(1024), lds, s, a.parts[0].qs, a.parts[0].sc, a.parts[0].out, a.pro.a, a.pro.b,
                                   a.in_rs, a.K, a.parts[0].NB2 | ((waves - 1) << 20) | (a.n_parts << 24) | (1u << 28) | (a.pro.kind == QMV_PRO_RMSNORM_MUL ? 1u << 29 : 0u) | (a.x_vec ? 1u << 30 : 0u),
                                   nb2_12, a, g, fo->a2);
            else
                hipLaunchKernelGGL((qkv_attn_o_kernel<__half, true, 32>), grid3, dim3(1024), lds, s, a.parts[0].qs, a.parts[0].sc, a.parts[0].out, a.pro.a, a.pro.b,
                                   a.in_rs, a.K, a.parts[0].NB2 | ((waves - 1) << 20) | (a.n_parts << 24) | (1u << 28) | (a.pro.kind == QMV_PRO_RMSNORM_MUL ? 1u << 29 : 0u) | (a.x_vec ? 1u << 30 : 0u),
                                   nb2_12, a, g, fo->a2);
            return true;
        }
#endif
        if (fo) return false; // (the caller asked for the triple: let it fall back as a whole)
        if (depth_sel != 0) return false; // (one load step: K <= 2048)
        using FusedFn = void (*)(const uint4*, const void*, float*, const float*, const float*, uint32_t, uint32_t, uint32_t, uint32_t, QMVArgs, QkvAttnArgs);
        FusedFn ff = nullptr;
        if (d_head == 64)
            ff = prenorm ? (fused_kvq ? qkv_attn_kernel<__half, true, 16, true, 2> : qkv_attn_kernel<__half, true, 16, false, 2>)
                         : (fused_kvq ? qkv_attn_kernel<__half, true, 16, true> : qkv_attn_kernel<__half, true, 16, false>);
        else
            ff = prenorm ? (fused_kvq ? qkv_attn_kernel<__half, true, 32, true, 2> : qkv_attn_kernel<__half, true, 32, false, 2>)
                         : (fused_kvq ? qkv_attn_kernel<__half, true, 32, true> : qkv_attn_kernel<__half, true, 32, false>);
        hipLaunchKernelGGL(ff, dim3(total_blocks + extra_blocks), dim3(1024), lds, s, a.parts[0].qs, a.parts[0].sc, a.parts[0].out, head_xa, head_xb, a.in_rs, a.K,
                           a.parts[0].NB2 | ((waves - 1) << 20) | (a.n_parts << 24) | (1u << 28) | (a.pro.kind == QMV_PRO_RMSNORM_MUL ? 1u << 29 : 0u) |
                               (a.x_vec ? 1u << 30 : 0u) | (prenorm_late() ? 1u << 31 : 0u),
                           nb2_12, a, f);
        return true;
    }
    hipLaunchKernelGGL(fn, grid, dim3(waves * 64), lds, s, a.parts[0].qs, a.parts[0].sc, a.parts[0].out, head_xa, head_xb, a.in_rs, a.K,
                       a.parts[0].NB2 | ((waves - 1) << 20) | (a.n_parts << 24) | (contig ? 1u << 28 : 0u) | (a.pro.kind == QMV_PRO_RMSNORM_MUL ? 1u << 29 : 0u) |
                           (a.x_vec ? 1u << 30 : 0u) | (prenorm_late() ? 1u << 31 : 0u),
                       nb2_12, a);
    return true;
}

} // namespace

uint32_t qmv_max_prologue_k(const QWeightDev& w) { return w.format == QW_Q4K ? UINT32_MAX : kMaxWaves * 64 * 4 * kXRegs; } // widest workgroup; launch_packed widens to fit

void launch_qmatvec_rows(hipStream_t s, const QWeightDev& w, const QMatmulParams& p, bool xvec) {
    QMVArgs a{};
    a.n_parts = 1;
    a.parts[0].qs = (const uint4*)w.qs;
    a.parts[0].sc = w.sc;
    a.parts[0].out = p.dst;
    a.parts[0].NB2 = p.N / 16;
    a.parts[0].block_begin = 0;
    a.parts[0].out_rs = p.dst_rs;
    a.parts[0].n_epi = 0;
    a.pro.kind = QMV_PRO_NONE;
    a.pro.a = p.input;
    a.pro.b = p.input;
    a.M = p.M, a.K = p.K;
    a.U = w.format == QW_Q4 ? w.KC : 2 * w.KC;
    a.in_rs = p.in_rs;
    a.trace = nullptr;
    launch_packed(s, a, w, a.parts[0].NB2, p.M, xvec);
}

static void build_qmv_args(const QmvLaunch& L, QMVArgs& a, uint32_t& blocks, bool& xvec_out) {
    const QWeightDev& w0 = L.parts[0].w;
    a = QMVArgs{};
    a.n_parts = L.n_parts;
    blocks = 0;
    for (uint32_t i = 0; i < L.n_parts; i++) {
        const QmvPart& pt = L.parts[i];
        QMVPartDev& d = a.parts[i];
        d.qs = (const uint4*)pt.w.qs;
        d.sc = pt.w.sc;
        d.out = pt.dst;
        d.NB2 = pt.w.N / 16;
        d.block_begin = blocks;
        d.out_rs = pt.w.N;
        d.n_epi = pt.n_epi;
        for (uint32_t e = 0; e < pt.n_epi; e++) d.epi[e] = pt.epi[e];
        { // the SiLU chain, exactly: NEG, EXP [store], ADD vector, RECIP, MUL by this part's output [store]; M == 1
            const QmvEpiStep* st = pt.epi;
            const bool silu = pt.n_epi == 5 && st[0].op == ZGML_OP_NEG && !st[0].store && st[1].op == ZGML_OP_EXP && st[1].store &&
                              st[2].op == ZGML_OP_ADD && st[2].operand && st[2].operand != pt.dst && st[2].operand != st[1].store && !st[2].store &&
                              st[3].op == ZGML_OP_RECIP && !st[3].store && st[4].op == ZGML_OP_MUL && st[4].operand == pt.dst && st[4].store;
            static const bool silu_on = !(getenv("ZGML_QMV_EPI_SILU") && atoi(getenv("ZGML_QMV_EPI_SILU")) == 0);
            const bool residual = pt.n_epi == 1 && st[0].op == ZGML_OP_ADD && st[0].operand && st[0].operand != pt.dst && st[0].store;
            const bool mulvec = pt.n_epi == 1 && st[0].op == ZGML_OP_MUL && st[0].operand && st[0].operand != pt.dst && st[0].store;
            // (the switch is about the SiLU chain only: the residual form also carries the next launch's rmsnorm, arm_prenorm)
            d.epi_kind = silu ? (silu_on ? kEpiSilu : kEpiGeneric) : (residual ? kEpiResidual : (mulvec ? kEpiMulVec : kEpiGeneric));
        }
        blocks += d.NB2;
    }
    a.pro = L.pro;
    a.next = L.next;
    a.pair_out = L.pair_out;
    if (a.pro.kind == QMV_PRO_NONE) a.pro.b = a.pro.a;
    a.M = 1, a.K = L.K;
    a.U = w0.format == QW_Q4 ? w0.KC : 2 * w0.KC;
    a.in_rs = L.K;
    bool xvec = ((uintptr_t)L.pro.a % 16 == 0) && (L.K % 4 == 0);
    if (L.pro.kind != QMV_PRO_NONE) xvec = xvec && ((uintptr_t)L.pro.b % 16 == 0);
    if (L.pro.kind == QMV_PRO_PRENORM) xvec = xvec && ((uintptr_t)L.pro.xg % 16 == 0);
    a.trace = L.trace;
    xvec_out = xvec;
}

static bool launch_qmv(hipStream_t s, const QmvLaunch& L, const QkvAttnArgs* fused, uint32_t extra_blocks, uint32_t d_head, const FusedO* fo = nullptr) {
    if (L.n_parts == 0) return false;
    QMVArgs a;
    uint32_t blocks = 0;
    bool xvec = false;
    build_qmv_args(L, a, blocks, xvec);
    return launch_packed(s, a, L.parts[0].w, blocks, 1, xvec, fused, extra_blocks, d_head, fo);
}

void launch_qmatvec_fused(hipStream_t s, const QmvLaunch& L) { launch_qmv(s, L, nullptr, 0, 0); }

// the q / k / v projection + the decode attention of its heads in one launch (qkv_attn_kernel). `n_heads` x `n_sp`
// attention workgroups follow the mat-vec's; false: shapes the fused kernel is not built for (launch the two separately)
bool launch_qkv_attention(hipStream_t s, const QmvLaunch& L, const AttnDecodeParams* dev_params, uint32_t n_heads, uint32_t n_kv, uint32_t d_head,
                          const AttnSplit& sp, uint32_t* counters, const uint32_t* idx, uint32_t* seen, uint32_t* timeout, const QmvLaunch* Lo,
                          uint32_t* out_cnt, uint32_t* o_seen, bool kvq) {
    QkvAttnArgs f{};
    f.kvq = kvq ? 1 : 0;
    f.params = dev_params, f.split_buf = sp.buf, f.split_cnt = sp.cnt, f.split_min_keys = sp.min_keys;
    f.n_sp = sp.splits ? sp.splits : 1;
    uint32_t shift = 0;
    while ((16u << shift) < d_head) shift++;
#ifdef ZGML_TRACE // diagnostics build only (tests/handoff_timeout_worker.py loads libzgml_hip_trace.so): one column group never signals
    static const bool drop_publish = getenv("ZGML_HIP_DEBUG_DROP_PUBLISH") && atoi(getenv("ZGML_HIP_DEBUG_DROP_PUBLISH")) != 0;
#else
    constexpr bool drop_publish = false;
#endif
    f.pub = QmvPublish{counters, {0, n_heads, n_heads + n_kv}, shift, drop_publish ? 1u : 0u};
    static const uint32_t poll_sleep = getenv("ZGML_HIP_HANDOFF_SLEEP") ? (uint32_t)atoi(getenv("ZGML_HIP_HANDOFF_SLEEP")) : 2u;
    f.ho = DecodeHandoff{counters, seen, idx, n_heads, d_head / 16, timeout, poll_sleep, nullptr};
#ifdef ZGML_TRACE
    if (Lo) { // the O projection rides along (diagnostics build only)
        FusedO fo;
        bool xvec2 = false;
        build_qmv_args(*Lo, fo.a2, fo.blocks2, xvec2);
        fo.a2.x_vec = xvec2 ? 1 : 0;
        if (!plain_head_depth1(fo.a2, Lo->parts[0].w, fo.blocks2, fo.h2)) return false;
        fo.wt = QmvWait{out_cnt, o_seen, n_heads, timeout};
        f.ho.out_cnt = out_cnt;
        return launch_qmv(s, L, &f, n_heads * f.n_sp, d_head, &fo);
    }
#else
    if (Lo) return false;
#endif
    return launch_qmv(s, L, &f, n_heads * f.n_sp, d_head);
}

// the decode attention of `n_heads` heads + the single-matrix K-on-lanes projection that reads their row stores, one launch
// (attn_o_kon_kernel). `out_cnt`: one zeroed word, `o_seen`: one zeroed word per workgroup of the projection. false: not a shape the
// kernel is built for (nothing launched). attn_o_blocks_per_cu(): what the occupancy query admits per CU (for the planner's guard).
// Is a grid of `grid` 256-thread workgroups of the fused K-on-lanes launch co-resident on this device? Runs the kernel's CENSUS
// instantiation once (synchronises the stream: plan-build time only). 1 = every workgroup saw the whole grid arrive, 0 = some
// workgroup waited in vain (the launch must not be fused), -1 = the census could not run (the caller falls back to the occupancy query).
int qkv_attn_kon_census(hipStream_t s, uint32_t d_head, bool kvq, uint32_t grid) {
    using Fn = void (*)(const uint4*, const void*, float*, const float*, const float*, uint32_t, uint32_t, uint32_t, uint32_t, QMVArgs, QkvAttnArgs);
    const Fn census = d_head == 64 ? (kvq ? (Fn)qkv_attn_kon_kernel<16, 4, 2, true, true, true> : (Fn)qkv_attn_kon_kernel<16, 4, 2, true, false, true>)
                                   : (kvq ? (Fn)qkv_attn_kon_kernel<32, 4, 2, true, true, true> : (Fn)qkv_attn_kon_kernel<32, 4, 2, true, false, true>);
    const Fn product = d_head == 64 ? (kvq ? (Fn)qkv_attn_kon_kernel<16, 4, 2, true, true> : (Fn)qkv_attn_kon_kernel<16, 4, 2, true, false>)
                                    : (kvq ? (Fn)qkv_attn_kon_kernel<32, 4, 2, true, true> : (Fn)qkv_attn_kon_kernel<32, 4, 2, true, false>);
    hipFuncAttributes fa{}, fb{};
    if (hipFuncGetAttributes(&fa, (const void*)census) != hipSuccess || hipFuncGetAttributes(&fb, (const void*)product) != hipSuccess) return -1;
    if (fa.numRegs < fb.numRegs || fa.sharedSizeBytes != fb.sharedSizeBytes) return -1; // (the census must be at least as heavy as what it stands for)
    uint32_t* blk = nullptr;
    if (hipMalloc((void**)&blk, 8) != hipSuccess) return -1;
    int result = -1;
    uint32_t host[2] = {0, 1};
    if (hipMemsetAsync(blk, 0, 8, s) == hipSuccess) {
        QMVArgs a{};
        QkvAttnArgs f{};
        f.census = blk;
        const size_t lds = ((size_t)kMaxWaves * 16 + kMaxWaves) * sizeof(float);
        hipLaunchKernelGGL(census, dim3(grid), dim3(256), lds, s, (const uint4*)nullptr, (const void*)nullptr, (float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0u, 0u, 0u,
                           0u, a, f);
        if (hipMemcpyAsync(host, blk, 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess)
            result = host[0] == grid && host[1] == 0 ? 1 : 0;
    }
    hipFree(blk);
    return result;
}

int qkv_attn_kon_blocks_per_cu(uint32_t d_head, bool kvq) { // what the occupancy query admits per CU for the fused K-on-lanes launch (the deepest variant)
    int nb = 0;
    const void* fn = d_head == 64 ? (kvq ? (const void*)qkv_attn_kon_kernel<16, 4, 2, true, true> : (const void*)qkv_attn_kon_kernel<16, 4, 2, true, false>)
                                  : (kvq ? (const void*)qkv_attn_kon_kernel<32, 4, 2, true, true> : (const void*)qkv_attn_kon_kernel<32, 4, 2, true, false>);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, ((size_t)kMaxWaves * 16 + kMaxWaves) * sizeof(float)) != hipSuccess) return 0;
    return nb;
}
#ifdef ZGML_TRACE // the attention + O projection launch: diagnostics build only (measured slower)
int attn_o_blocks_per_cu(uint32_t d_head) {
    int nb = 0;
    const void* fn = d_head == 64 ? (const void*)attn_o_kon_kernel<16, 4, true> : (const void*)attn_o_kon_kernel<32, 4, true>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, ((size_t)kMaxWaves * 16 + kMaxWaves) * sizeof(float)) != hipSuccess) return 0;
    return nb;
}
bool launch_attention_o(hipStream_t s, const AttnDecodeParams* dev_params, uint32_t n_heads, uint32_t d_head, const AttnSplit& sp, const QmvLaunch& Lo,
                        uint32_t* out_cnt, uint32_t* o_seen, uint32_t* timeout) {
    QMVArgs a;
    uint32_t blocks = 0;
    bool xvec = false;
    build_qmv_args(Lo, a, blocks, xvec);
    const QWeightDev& w0 = Lo.parts[0].w;
    if (w0.format != QW_Q4K || a.n_parts != 1 || a.pro.kind != QMV_PRO_NONE || !xvec || (d_head != 64 && d_head != 128)) return false;
    a.x_vec = 1;
    const uint32_t P = (a.K + 1) / 2, waves = 4; // (256-thread workgroups: the kernel's launch bound)
    const uint32_t n_steps = cdiv(P, waves * 64);
    const int depth = n_steps >= 4 ? 4 : (n_steps >= 2 ? 2 : 1);
    const bool nt = w0.stream_nt != 0;
    AttnOArgs f{};
    f.params = dev_params, f.split_buf = sp.buf, f.split_cnt = sp.cnt, f.split_min_keys = sp.min_keys;
    f.n_sp = sp.splits ? sp.splits : 1;
    f.n_attn = n_heads * f.n_sp;
    f.ho = DecodeHandoff{nullptr, nullptr, nullptr, n_heads, d_head / 16, timeout, 2u, out_cnt};
    static const uint32_t poll_sleep = getenv("ZGML_HIP_ATTN_O_POLL") ? (uint32_t)atoi(getenv("ZGML_HIP_ATTN_O_POLL")) : 10u;
    static const uint32_t pre_sleep = getenv("ZGML_HIP_ATTN_O_PRESLEEP") ? (uint32_t)atoi(getenv("ZGML_HIP_ATTN_O_PRESLEEP")) : 0u;
    f.wt = QmvWait{out_cnt, o_seen, n_heads, timeout, poll_sleep, pre_sleep};
    const uint32_t flags = a.parts[0].NB2 | ((waves - 1) << 20) | (1u << 24) | (1u << 30);
    const size_t lds = ((size_t)kMaxWaves * 16 + kMaxWaves) * sizeof(float);
    using Fn = void (*)(const uint4*, const void*, float*, const float*, const float*, uint32_t, uint32_t, uint32_t, uint32_t, QMVArgs, AttnOArgs);
    Fn fn = nullptr;
#define AO(L) (depth == 4 ? (nt ? (Fn)attn_o_kon_kernel<L, 4, true> : (Fn)attn_o_kon_kernel<L, 4, false>) : depth == 2 ? (Fn)attn_o_kon_kernel<L, 2, false> : (Fn)attn_o_kon_kernel<L, 1, false>)
    fn = d_head == 64 ? AO(16) : AO(32);
#undef AO
    hipLaunchKernelGGL(fn, dim3(f.n_attn + blocks), dim3(256), lds, s, a.parts[0].qs, a.parts[0].sc, a.parts[0].out, a.pro.a, a.pro.b, a.in_rs, a.K, flags, 0u, a, f);
    return true;
}
#endif

void launch_synth_packed(hipStream_t s, const QWeightDev& w, uint32_t matrix_id) {
    if (w.scale_f16)
        synth_packed_kernel<__half><<<2048, kBlock, 0, s>>>(w, matrix_id);
    else
        synth_packed_kernel<float><<<2048, kBlock, 0, s>>>(w, matrix_id);
}

} // namespace zgml
