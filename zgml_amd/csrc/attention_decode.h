// attention_decode.h — the decode-shaped attention of one head with everything that feeds it folded in (device code).
// Included by kernels_generic.hip (the stand-alone launch) and by qmatvec.hip (so that a launch can hold both the q / k / v
// projection's workgroups and the attention's: DESIGN.md section 8.0). Everything lives in an anonymous namespace: each
// translation unit gets its own copy.
#pragma once
#include "kernels.h"

#include <hip/hip_fp16.h>
#include <math.h>

namespace zgml {
namespace {

// Loads / stores through GLOBAL-address-space pointers. A pointer taken from a parameter record is a generic pointer to
// hipcc, and a generic access is a FLAT instruction: it counts on vmcnt AND lgkmcnt, so every `s_waitcnt lgkmcnt(0)` in
// front of a scalar-load result (the records are read with scalar loads, round after round) also waits for every vector
// load issued so far — the phases of a latency-chain kernel serialise. global_load / global_store only count on vmcnt.
typedef float f4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg4(const float* p) {
    const f4v_t v = *(const __attribute__((address_space(1))) f4v_t*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ldg1(const float* p) { return *(const __attribute__((address_space(1))) float*)p; }
__device__ __forceinline__ uint32_t ldgu(const uint32_t* p) { return *(const __attribute__((address_space(1))) uint32_t*)p; }
__device__ __forceinline__ void stg4(float* p, float4 v) { *(__attribute__((address_space(1))) f4v_t*)p = f4v_t{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void stg1(float* p, float v) { *(__attribute__((address_space(1))) float*)p = v; }
__device__ __forceinline__ void stgu(uint32_t* p, uint32_t v) { *(__attribute__((address_space(1))) uint32_t*)p = v; }

constexpr int kAttnBlock = 1024;
constexpr int kAttnUnroll = 4;

// rope of 4 consecutive dims [d0, d0+4) of a head vector (reference.zig:457-478, DeviceOp
// convention: sin at cs + pair + half_d); `own` = x[d0..], `par` = x[(d0 ^ half)..]. Same
// expressions as rope_body so the fused path is bit-identical to the op-by-op one.
__device__ __forceinline__ float4 rope4(float4 own, float4 par, float4 c, float4 s, bool is_hi) {
    if (is_hi) // own = x_hi, par = x_lo: hi = x_hi * c + x_lo * s
        return make_float4(own.x * c.x + par.x * s.x, own.y * c.y + par.y * s.y, own.z * c.z + par.z * s.z,
                           own.w * c.w + par.w * s.w);
    return make_float4(own.x * c.x - par.x * s.x, own.y * c.y - par.y * s.y, own.z * c.z - par.z * s.z,
                       own.w * c.w - par.w * s.w); // lo = x_lo * c - x_hi * s
}

__device__ __forceinline__ float score_of(float dot, float mk, float scale) {
    float score = -INFINITY;
    if (isfinite(mk)) {
        score = dot * scale + mk;
        if (!isfinite(score)) score = -INFINITY;
    }
    return score;
}

// ── decode attention, one workgroup per head (seq_q == 1) ───────────────────────────────────
// At decode lengths this kernel is not bandwidth: a launch keeps only n_heads CUs busy, so what
// it costs is (a) the chain of dependent memory round trips and (b) the instruction stream of ONE
// wave (a CU issues ~1 wave-instruction per 4-5 cycles per SIMD: 1000 instructions are ~2 us).
// So: every load of a phase is issued back to back without branches (a conditional load makes
// hipcc wait for everything before it), reductions inside a key's lane group use DPP instead of
// LDS shuffles, there is no score buffer and a single barrier — each key slot (LPK lanes) runs its
// own online softmax (m, l, acc) over the keys it streams, slots are merged at the end (within
// the wave by shuffles, across waves through LDS) — and waves the context does not need retire at
// the top (a retired wave no longer counts at s_barrier).
// Lane layout: a key is handled by LPK = d_head/4 adjacent lanes holding one float4 of the head
// dimension each (one coalesced row read per key); a wave streams 64/LPK keys per step.

// v + (v of lane ^ 16) and v + (v of lane ^ 32) without the LDS crossbar (ds_bpermute: ~100 cycles and an lgkmcnt wait
// per exchange): gfx950's row swaps. permlane16_swap(v, v) = (rows 0 0 2 2 | rows 1 1 3 3), so their sum (max) is the
// xor-16 exchange's on every lane; permlane32_swap pairs the wave's halves the same way. Same values as the shuffles
// (a + b == b + a exactly).
__device__ __forceinline__ float xor16_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// allreduce (sum) over groups of LPK adjacent lanes; DPP inside a 16-lane row, row swaps above
template <int LPK>
__device__ __forceinline__ float group_sum(float v) {
    if (LPK >= 2) v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
    if (LPK >= 4) v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
    if (LPK >= 8) v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true)); // row_half_mirror
    if (LPK >= 16) v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true)); // row_mirror
    if (LPK >= 32) v = xor16_sum(v);
    if (LPK >= 64) v = xor32_sum(v);
    return v;
}

struct SoftState { // online softmax of one key stream: running max, sum of weights, weighted V (4 dims)
    float m, l;
    float4 acc;
};
// merge stream `o` into `s` (either may be empty: m == -inf)
__device__ __forceinline__ void soft_merge(SoftState& s, float om, float ol, float4 oacc) {
    const float nm = fmaxf(s.m, om);
    const float a = s.m > -INFINITY ? expf(s.m - nm) : 0.0f, b = om > -INFINITY ? expf(om - nm) : 0.0f;
    s.l = s.l * a + ol * b;
    s.acc = make_float4(s.acc.x * a + oacc.x * b, s.acc.y * a + oacc.y * b, s.acc.z * a + oacc.z * b, s.acc.w * a + oacc.w * b);
    s.m = nm;
}

//
// Long contexts (flash-decoding split): the launch has gridDim.y = S workgroups per head. While
// seq_kv < 2 * split_min_keys only workgroup 0 works (the others retire after reading seq_kv) and
// nothing below changes. Above that, n_active = min(S, seq_kv / split_min_keys) workgroups each
// stream a contiguous chunk of the keys, publish their (m, l, acc) with write-through (sc1) stores,
// drain, and add to the head's counter; the last arriver reads all partials with sc1 loads, merges
// them in chunk order (so the result does not depend on arrival order), writes the output and
// re-arms the counter. The guide's fan-in form: sc1 payload + vmcnt(0) drain + agent atomic, every
// consumer load sc1, no fence.
using gu64 = __attribute__((address_space(1))) unsigned long long;
using gu32 = __attribute__((address_space(1))) unsigned int;
__device__ __forceinline__ void split_put(float* p, float a, float b) {
    const unsigned long long v = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
    __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void split_get(const float* p, float& a, float& b) {
    const unsigned long long v = __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = __uint_as_float((uint32_t)v), b = __uint_as_float((uint32_t)(v >> 32));
}

// sum l and acc over the 64 / LPK key slots of a wave (streams already on a common max)
template <int LPK>
__device__ __forceinline__ void slots_sum(SoftState& s) {
#pragma unroll
    for (int off = LPK; off < 16; off <<= 1) {
        s.l += __shfl_xor(s.l, off, 64);
        s.acc.x += __shfl_xor(s.acc.x, off, 64), s.acc.y += __shfl_xor(s.acc.y, off, 64);
        s.acc.z += __shfl_xor(s.acc.z, off, 64), s.acc.w += __shfl_xor(s.acc.w, off, 64);
    }
    if (LPK <= 16) s.l = xor16_sum(s.l), s.acc.x = xor16_sum(s.acc.x), s.acc.y = xor16_sum(s.acc.y), s.acc.z = xor16_sum(s.acc.z), s.acc.w = xor16_sum(s.acc.w);
    if (LPK <= 32) s.l = xor32_sum(s.l), s.acc.x = xor32_sum(s.acc.x), s.acc.y = xor32_sum(s.acc.y), s.acc.z = xor32_sum(s.acc.z), s.acc.w = xor32_sum(s.acc.w);
}
// merge the slots' online-softmax states: common max, one rescale each, sums
template <int LPK>
__device__ __forceinline__ void slots_merge(SoftState& s) {
    float M = s.m;
#pragma unroll
    for (int off = LPK; off < 16; off <<= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
    if (LPK <= 16) M = xor16_max(M);
    if (LPK <= 32) M = xor32_max(M);
    const float f = s.m > -INFINITY ? expf(s.m - M) : 0.0f;
    s.l *= f;
    s.acc = make_float4(s.acc.x * f, s.acc.y * f, s.acc.z * f, s.acc.w * f);
    s.m = M;
    slots_sum<LPK>(s);
}

// One key's (value's) 4 dims as this lane holds them: f32 cache rows, or the quantised-KV form (4 int8 + the
// block scale; QuantizedKVCache, src/quant.zig:645-700). The dot / accumulate expressions of the quantised form
// are those of kvq_attention_kernel, the quantisation is kvq_store_kernel's (= storeColumn), so the fused launch
// writes the same cache bytes and computes the same numbers as the op-by-op plan.
template <bool KVQ>
struct AttnRow {
    float4 v;
};
template <>
struct AttnRow<true> {
    uint32_t w;
    float sc;
};
__device__ __forceinline__ float row_dot(const float4& q, const AttnRow<false>& r) { return q.x * r.v.x + q.y * r.v.y + q.z * r.v.z + q.w * r.v.w; }
__device__ __forceinline__ float row_dot(const float4& q, const AttnRow<true>& r) {
    const uint32_t kw = r.w;
    return (q.x * (float)(int8_t)(kw & 255) + q.y * (float)(int8_t)((kw >> 8) & 255) + q.z * (float)(int8_t)((kw >> 16) & 255) +
            q.w * (float)(int8_t)(kw >> 24)) * r.sc;
}
__device__ __forceinline__ void row_axpy(float4& acc, float wgt, const AttnRow<false>& r) {
    acc.x += wgt * r.v.x, acc.y += wgt * r.v.y, acc.z += wgt * r.v.z, acc.w += wgt * r.v.w;
}
__device__ __forceinline__ void row_axpy(float4& acc, float wgt, const AttnRow<true>& r) {
    const float ws = wgt * r.sc;
    const uint32_t vw = r.w;
    acc.x += ws * (float)(int8_t)(vw & 255), acc.y += ws * (float)(int8_t)((vw >> 8) & 255);
    acc.z += ws * (float)(int8_t)((vw >> 16) & 255), acc.w += ws * (float)(int8_t)(vw >> 24);
}
// storeColumn's arithmetic on the 4 dims of a lane; a block of 32 dims = 8 adjacent lanes
__device__ __forceinline__ AttnRow<true> quantise_block32(float4 v) {
    float mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mx), 0xB1, 0xF, 0xF, true)));  // quad_perm [1,0,3,2]
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mx), 0x4E, 0xF, 0xF, true)));  // quad_perm [2,3,0,1]
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mx), 0x141, 0xF, 0xF, true))); // row_half_mirror
    const float scale = mx > 0.f ? mx / 127.0f : 1.0f, inv = mx > 0.f ? 127.0f / mx : 0.0f;
    auto q8 = [&](float x) { return (uint32_t)(uint8_t)(int8_t)(int)fminf(fmaxf(x * inv, -127.0f), 127.0f); }; // truncates toward zero
    return AttnRow<true>{q8(v.x) | (q8(v.y) << 8) | (q8(v.z) << 16) | (q8(v.w) << 24), scale};
}

// ── quantised KV, 16 dims per lane (round 5) ────────────────────────────────────────────────
// The int8 caches keep a key's d_head bytes contiguous (QuantizedKVCache, src/quant.zig:925-1091: [col][d] int8, then
// [col][d / 32] f32 scales), so a lane can take 16 dims of a key with ONE 16-byte load: a key then occupies d_head / 16 lanes
// instead of d_head / 4, a wave streams 4x the keys per load instruction, and the number of dependent memory round trips of a
// long context drops with it (at Llama-2-7B dimensions a 4-wave workgroup covers 128 keys per round instead of 32). The
// ropes, the storeColumn quantisation and the cache stores stay in the 4-dims-per-lane form above (same bytes in the caches);
// q and the new column are re-laid-out once by lane shuffles, the result goes back through the merge's LDS arrays.
#ifndef ZGML_ATTN_KVQ16
#define ZGML_ATTN_KVQ16 1
#endif
constexpr int kAttnUnrollQ16 = 4;
typedef uint32_t u4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u4v_t ldgu4(const void* p) { return *(const __attribute__((address_space(1))) u4v_t*)p; }
// (float)(int8) of byte B of a dword: one SDWA convert (sign-extended byte select)
template <int B>
__device__ __forceinline__ float s8_to_f32(uint32_t w) {
    float r;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (B == 0) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(r) : "v"(w));
    if constexpr (B == 1) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(r) : "v"(w));
    if constexpr (B == 2) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(r) : "v"(w));
    if constexpr (B == 3) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(r) : "v"(w));
#else
    r = (float)(int8_t)(w >> (8 * B));
#endif
    return r;
}
// The attention workgroups of a fused launch run ONE wave per SIMD, where every instruction costs its 4-5 cycles of issue
// whatever its class: the products go through v_pk_fma_f32 (two per instruction: 6 instructions per 4 int8 instead of 8). One
// dword at a time behind a scheduling barrier — left alone, hipcc converts every row of a step up front and spills.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// sum_e q[e] * k8[e] over the lane's 16 dims: two running sums (even / odd dims), added at the end
__device__ __forceinline__ float dot16(const f32x2_t (&q)[8], const u4v_t& w) {
    f32x2_t s = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const f32x2_t lo = {s8_to_f32<0>(w[c]), s8_to_f32<1>(w[c])}, hi = {s8_to_f32<2>(w[c]), s8_to_f32<3>(w[c])};
        s = __builtin_elementwise_fma(lo, q[2 * c], s);
        s = __builtin_elementwise_fma(hi, q[2 * c + 1], s);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    return s.x + s.y;
}
__device__ __forceinline__ void axpy16(f32x2_t (&acc)[8], float ws, const u4v_t& w) {
    const f32x2_t ws2 = {ws, ws};
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const f32x2_t lo = {s8_to_f32<0>(w[c]), s8_to_f32<1>(w[c])}, hi = {s8_to_f32<2>(w[c]), s8_to_f32<3>(w[c])};
        acc[2 * c] = __builtin_elementwise_fma(lo, ws2, acc[2 * c]);
        acc[2 * c + 1] = __builtin_elementwise_fma(hi, ws2, acc[2 * c + 1]);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
}
// all lanes with the same (lane % L) — the same dims of different key slots — get the sum / max over the wave's 64 / L slots:
// rotations inside a 16-lane row (DPP), row swaps above
template <int L, bool MAX>
__device__ __forceinline__ float slots_all(float v) {
    auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
    if (L <= 1) v = op(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x121, 0xF, 0xF, true))); // row_ror:1
    if (L <= 2) v = op(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x122, 0xF, 0xF, true))); // row_ror:2
    if (L <= 4) v = op(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x124, 0xF, 0xF, true))); // row_ror:4
    if (L <= 8) v = op(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xF, 0xF, true))); // row_ror:8
    if (L <= 16) v = MAX ? xor16_max(v) : xor16_sum(v);
    if (L <= 32) v = MAX ? xor32_max(v) : xor32_sum(v);
    return v;
}

// A launch that also holds the workgroups of the q / k / v projection (qmatvec.hip: fused launch) hands the projections
// over through agent-scope counters instead of a kernel boundary: one counter per 64-column... per head slice of each
// projection, bumped once by every column group (16 columns) that has stored its outputs with write-through stores. The
// attention issues everything that does not depend on the projections first, then waits (bounded) for its head's three
// counters to pass `seen + need`, remembers the new values in its private `seen` words (kv heads have several consumers,
// so the shared counters are never reset) and reads q / k / v with agent-scope loads.
struct DecodeHandoff {
    const uint32_t* cnt; // [n_heads | n_kv | n_kv] monotonic counters
    uint32_t* seen;      // [workgroups of the attention part][3]
    const uint32_t* idx; // [records][3]: the q / k / v counters of each record's head (records need not be in head order)
    uint32_t n_heads;
    uint32_t need;       // column groups per head slice = d_head / 16
    uint32_t* timeout;   // bumped when a wait gives up
    uint32_t poll_sleep; // s_sleep(1) units (64 clocks) between two polls
    uint32_t* out_cnt;   // optional: bumped once per head when its output rows (dst2) are stored write-through — the O
                         // projection's workgroups of the same launch wait for it (qmatvec.hip: QmvWait)
};

// BLOCK: threads of the workgroup that runs the body (1024 stand-alone; 256 inside the launch that also carries the O projection,
// whose workgroups are 4 waves): at most BLOCK / 64 waves work on a chunk of keys, the loop over steps covers the rest
template <int LPK, bool KVQ, int BLOCK = kAttnBlock>
__device__ __forceinline__ void attention_decode_body(const AttnDecodeParams* __restrict__ params, float* split_buf, uint32_t* split_cnt,
                                                      uint32_t split_min_keys, const uint32_t hx, const uint32_t sp_in, const uint32_t n_sp,
                                                      const DecodeHandoff* ho) {
    constexpr int DH = 4 * LPK, HALF = DH / 2;
    // Q16: quantised KV with 16 dims per lane in the key loop (LPKM lanes per key there); KPW4 / LPK stay the mapping of the
    // ropes, the stores, the wave merge and the split merge
    // (d_head >= 128 only: measured on MI355X at position 1900 / short contexts, Llama-2-7B dims 724-738 against 686 tok/s / 868
    // against 843; SmolLM's d_head 64 — 16 waves per workgroup, the whole chunk in one round already — loses 1 % either way)
    constexpr bool Q16 = KVQ && ZGML_ATTN_KVQ16 != 0 && LPK >= 32;
    constexpr int LPKM = Q16 ? DH / 16 : LPK, KPW = 64 / LPKM, KPW4 = 64 / LPK, U = Q16 ? kAttnUnrollQ16 : kAttnUnroll;
    using Row = AttnRow<KVQ>;
    // The head's parameter record in ONE scalar round trip: read through a reference, hipcc fetches it field by field where
    // the fields are used — three dependent rounds of s_load + s_waitcnt before the kernel's first vector load.
    static_assert(sizeof(AttnDecodeParams) == 248, "the bulk fetch below covers exactly 62 dwords");
    AttnDecodeParams P;
    {
        typedef uint32_t u32x16_t __attribute__((ext_vector_type(16)));
        typedef uint32_t u32x8_t __attribute__((ext_vector_type(8)));
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        u32x16_t ra, rb, rc;
        u32x8_t rd;
        u32x4_t re;
        u32x2_t rf;
#if defined(__HIP_DEVICE_COMPILE__)
        const AttnDecodeParams* const rec = params + hx;
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
#ifdef ZGML_TRACE // build with -DZGML_TRACE (ZGML_HIP_ATTN_TRACE=1 then prints the stamps)
    unsigned long long* const trace = P.trace;
#define ATTN_STAMP(i) do { if (trace && threadIdx.x == 0) trace[i] = wall_clock64(); } while (0)
#else
#define ATTN_STAMP(i) do { } while (0)
#endif
    ATTN_STAMP(0);
    __shared__ __attribute__((aligned(8))) float part_ml[2 * (BLOCK / 64)];
    __shared__ float4 part_acc[(BLOCK / 64) * LPK];
    constexpr int COLV = DH / 4 + DH / 32; // Q16: the new K column's words and block scales, then the V column's
    __shared__ __attribute__((aligned(16))) float q16_lds[Q16 ? DH : 4];
    __shared__ __attribute__((aligned(16))) uint32_t col_lds[Q16 ? 2 * COLV : 4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t slot = lane / LPK, li = lane % LPK, d0 = 4 * li, pair = d0 & (HALF - 1);
    const bool is_hi = d0 >= (uint32_t)HALF;
    // ---- phase A: dynamic words (scalar) and everything the ropes need (vector), one round
    // quantised KV: the stores' dynamic words are column indices, the caches are int8 rows + f32 block scales
    const uint32_t seq_kv = KVQ ? min(ldgu(p.dyn_seq_kv), P.kvq_cols) : ldgu(p.dyn_seq_kv), dk = ldgu(P.dyn_k_off), dv = ldgu(P.dyn_v_off);
    const uint32_t d2_off = p.dst2 ? ldgu(p.dyn_dst2_off) : 0; // scalar, with the other dynamic words
    // (a fused launch reads the projections' outputs only after the hand-off below; `ho` is a literal nullptr in the
    // stand-alone kernel, so nothing here is a run-time branch there)
    // (round 3) a hand-off record WITHOUT counters: the projections come from an earlier launch (plain loads, no wait); only its
    // output side is used — the O projection's workgroups of this launch wait for out_cnt
    const bool wait_qkv = ho && ho->cnt;
    float4 q_own, q_par, k_own, k_par, v_new;
    if (!wait_qkv) {
        q_own = ldg4(P.q_src + d0), q_par = ldg4(P.q_src + (d0 ^ HALF));
        k_own = ldg4(P.k_src + d0), k_par = ldg4(P.k_src + (d0 ^ HALF));
        v_new = ldg4(P.v_src + d0);
    }
    const float4 q_c = ldg4(P.q_cs + pair), q_s = ldg4(P.q_cs + HALF + pair);
    const float4 k_c = ldg4(P.k_cs + pair), k_s = ldg4(P.k_cs + HALF + pair);
    ATTN_STAMP(1);
    // this workgroup's keys [k_begin, k_end)
    const uint32_t sp = sp_in;
    uint32_t n_active = 1, k_begin = 0, k_end = seq_kv;
    if (n_sp > 1) {
        n_active = seq_kv / split_min_keys;
        n_active = n_active < 1 ? 1 : (n_active > n_sp ? n_sp : n_active);
        if (sp >= n_active) { // (a fused launch: an idle split still moves its private `seen` words with the counters)
            if (wait_qkv && tid == 0) {
                uint32_t* const seen = ho->seen + (hx * n_sp + sp) * 3;
                seen[0] += ho->need, seen[1] += ho->need, seen[2] += ho->need;
            }
            return;
        }
        const uint32_t chunk = (seq_kv + n_active - 1) / n_active;
        k_begin = min(sp * chunk, seq_kv), k_end = min(k_begin + chunk, seq_kv);
    }
    const uint32_t n_keys = k_end - k_begin;
    // waves the chunk needs: one step of a wave covers KPW * U keys
    // (Q16: one key per slot and wave while waves are left — a slot's key costs 4x the instructions of the 4-dims form, and the
    // rows a wave does not need are skipped by a scalar branch)
    constexpr int KEYS_PER_WAVE = Q16 ? KPW : KPW * U;
    uint32_t NW = (n_keys + KEYS_PER_WAVE - 1) / KEYS_PER_WAVE;
    NW = NW < 1 ? 1 : (NW > (uint32_t)(BLOCK / 64) ? (uint32_t)(BLOCK / 64) : NW);
    if (w >= NW) return;
    const uint32_t keys_per_iter = KPW * NW, step_keys = keys_per_iter * U, last = k_end ? k_end - 1 : 0;
    // ---- phase B: first step's K / V / mask rows (clamped to live rows, unconditional)
    constexpr uint32_t BPC = DH / 32; // quantised form: 32-dim blocks per column (the planner admits block size 32 only)
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
    Row kv[Q16 ? 1 : U], vv[Q16 ? 1 : U];
    float mk[U];
    // Q16: the key loop's own lane mapping (slotm, lim) and ONE set of rows — K rows are re-issued as soon as the scores are
    // formed, V rows as soon as they are accumulated (no second register set: 16-byte rows would not fit twice)
    const uint32_t slotm = lane / LPKM, lim = lane % LPKM;
    // (addresses: wave-uniform base in scalar registers + a 32-bit byte offset per lane, formed from the key index where the load
    // is issued — U rows x 5 arrays of 64-bit pointers carried through the loop are 40 vector registers)
    const char* const kb16 = (const char*)P.k_cache;
    const char* const vb16 = (const char*)P.v_cache;
    const uint32_t sc_base16 = P.kvq_cols * (uint32_t)DH + 4 * (lim / 2); // bytes: the scales start behind the int8 rows
    u4v_t kw[Q16 ? U : 1], vw[Q16 ? U : 1];
    float ks[Q16 ? U : 1], vs[Q16 ? U : 1];
    auto issue_k16 = [&](uint32_t base) { // (clamped to live rows, unconditional: the waits stay counted)
#pragma unroll
        for (int j = 0; j < U; j++) {
            const uint32_t s = min(base + j * keys_per_iter + w * KPW + slotm, last);
            kw[j] = ldgu4(kb16 + (s * (uint32_t)DH + 16 * lim));
            ks[j] = ldg1((const float*)(kb16 + (sc_base16 + s * (4 * BPC))));
            mk[j] = ldg1((const float*)((const char*)p.mask + s * (4 * p.mask_rs))); // (< 4 GB: checked where the launch is planned)
        }
    };
    auto issue_v16 = [&](uint32_t base) {
#pragma unroll
        for (int j = 0; j < U; j++) {
            const uint32_t s = min(base + j * keys_per_iter + w * KPW + slotm, last);
            vw[j] = ldgu4(vb16 + (s * (uint32_t)DH + 16 * lim));
            vs[j] = ldg1((const float*)(vb16 + (sc_base16 + s * (4 * BPC))));
        }
    };
    if constexpr (Q16) {
        issue_k16(k_begin);
        issue_v16(k_begin);
    } else {
#pragma unroll
        for (int j = 0; j < U; j++) {
            const uint32_t s = min(k_begin + j * keys_per_iter + w * KPW + slot, last);
            kv[j] = load_k(s);
            vv[j] = load_v(s);
            mk[j] = ldg1(p.mask + (uint64_t)s * p.mask_rs); // host passes a zero word with stride 0 when there is no mask
        }
    }
    uint32_t target[3] = {0, 0, 0}; // fused launch: the counter values this execution waits for (written back at the end)
    if (wait_qkv) { // ---- hand-off: the q / k / v column groups of this head have been stored (DecodeHandoff)
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
                    if (tid == 0) __hip_atomic_store(ho->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // host-visible word (runtime.hip: handoff_ok)
                    break;
                }
                for (uint32_t z = 0; z < ho->poll_sleep; z++) __builtin_amdgcn_s_sleep(1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // the agent-scope loads of q / k / v below stay behind the poll (compiler ordering; they bypass L1 themselves)
        if (NW > 1) __syncthreads(); // (waves >= NW have retired and do not count)
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
    ATTN_STAMP(2);
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
    ATTN_STAMP(3);
    SoftState st{-INFINITY, 0.f, make_float4(0.f, 0.f, 0.f, 0.f)};
    if constexpr (Q16) {
        // q and the new column in the key loop's mapping, through LDS: the first 4-dims-per-lane group of every wave writes the
        // same values to the same words (every group and every wave hold identical q / columns), each wave reads back behind its
        // own writes (the DS operations of a wave complete in order). The new column stays in LDS: exactly one key slot of one
        // step meets it, behind a wave-uniform branch
        if (slot == 0) {
            *(float4*)&q16_lds[d0] = qv;
            col_lds[li] = k_col.w, col_lds[COLV + li] = v_col.w;
            if ((li & 7) == 0) col_lds[DH / 4 + li / 8] = __float_as_uint(k_col.sc), col_lds[COLV + DH / 4 + li / 8] = __float_as_uint(v_col.sc);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        f32x2_t q16[8], acc16[8];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float4 t4 = *(const float4*)&q16_lds[16 * lim + 4 * c];
            q16[2 * c] = f32x2_t{t4.x, t4.y}, q16[2 * c + 1] = f32x2_t{t4.z, t4.w};
        }
        float m16 = -INFINITY, l16 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) acc16[e] = f32x2_t{0.f, 0.f};
        const uint32_t wu = __builtin_amdgcn_readfirstlane(w); // (scalar: rows wholly behind the chunk's end are skipped by a scalar branch)
        float sc[U], bm;
        auto scores = [&](uint32_t base) {
            bm = -INFINITY;
#pragma unroll
            for (int j = 0; j < U; j++) {
                sc[j] = -INFINITY;
                if (base + j * keys_per_iter + wu * KPW >= k_end) continue;
                const uint32_t t = base + j * keys_per_iter + w * KPW + slotm;
                u4v_t kk = kw[j];
                float kss = ks[j];
                if (__builtin_amdgcn_ballot_w64(t == col_k) != 0 && t == col_k)
                    kk = *(const u4v_t*)&col_lds[4 * lim], kss = __uint_as_float(col_lds[DH / 4 + lim / 2]);
                const float dot = group_sum<LPKM>(dot16(q16, kk) * kss);
                sc[j] = t < k_end ? score_of(dot, mk[j], p.scale) : -INFINITY;
                bm = fmaxf(bm, sc[j]);
            }
        };
        auto update = [&](uint32_t base) {
            const float nm = fmaxf(m16, bm);
            if (nm > -INFINITY) { // per key slot; lanes of a slot agree
                // (__expf = v_exp_f32 of x * log2(e): relative error ~ |x| * 1e-7, three instructions instead of a dozen; the int8
                // caches are compared at 1e-3 of the logit range)
                const float alpha = m16 > -INFINITY ? __expf(m16 - nm) : 0.0f;
                l16 *= alpha;
#pragma unroll
                for (int e = 0; e < 8; e++) acc16[e] *= alpha;
#pragma unroll
                for (int j = 0; j < U; j++) {
                    if (base + j * keys_per_iter + wu * KPW >= k_end) continue;
                    const uint32_t t = base + j * keys_per_iter + w * KPW + slotm;
                    const float wgt = sc[j] > -INFINITY ? __expf(sc[j] - nm) : 0.0f;
                    u4v_t xx = vw[j];
                    float vss = vs[j];
                    if (__builtin_amdgcn_ballot_w64(t == col_v) != 0 && t == col_v)
                        xx = *(const u4v_t*)&col_lds[COLV + 4 * lim], vss = __uint_as_float(col_lds[COLV + DH / 4 + lim / 2]);
                    l16 += wgt;
                    // (no branch on the slot: a dead slot's row contributes 0 * int8 — its bytes are numbers whatever they are, and
                    // its scale, which may be anything, is selected away, not multiplied)
                    axpy16(acc16, sc[j] > -INFINITY ? wgt * vss : 0.0f, xx);
                }
                m16 = nm;
            }
        };
        if (n_keys <= step_keys) { // the usual decode case: everything is already in registers
            if (n_keys) scores(k_begin), update(k_begin);
        } else {
            for (uint32_t base = k_begin; base < k_end; base += step_keys) {
                scores(base);
                issue_k16(base + step_keys); // (the last round re-reads live rows: L2 hits nobody waits for)
                update(base);
                issue_v16(base + step_keys);
            }
        }
        ATTN_STAMP(4);
        // slots of the wave: common max, one rescale, sums over the lanes that hold the same dims; then into the merge's LDS
        // arrays in the 4-dims-per-lane form (lane lim of slot 0 holds four float4 of it)
        const float M = slots_all<LPKM, true>(m16);
        const float f = m16 > -INFINITY ? expf(m16 - M) : 0.0f;
        l16 = slots_all<LPKM, false>(l16 * f);
#pragma unroll
        for (int e = 0; e < 8; e++) acc16[e] = f32x2_t{slots_all<LPKM, false>(acc16[e].x * f), slots_all<LPKM, false>(acc16[e].y * f)};
        if (lane < LPKM) {
#pragma unroll
            for (int c = 0; c < 4; c++) part_acc[w * LPK + 4 * lane + c] = make_float4(acc16[2 * c].x, acc16[2 * c].y, acc16[2 * c + 1].x, acc16[2 * c + 1].y);
            if (lane == 0) part_ml[2 * w] = M, part_ml[2 * w + 1] = l16;
        }
        st.m = M, st.l = l16;
        if (NW == 1) { // a single wave: its own LDS writes, read back in the other mapping (DS operations of a wave complete in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            st.acc = part_acc[w * LPK + li];
        }
    } else {
    // one step: scores of the slot's U keys, then the online-softmax update
    auto step = [&](uint32_t base) {
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
            // prefetch the next step (clamped: the last step re-reads live rows, L2 hits)
            Row kn[U], vn[U];
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
    ATTN_STAMP(4);
    // ---- merge the key slots of the wave, then the waves. Every merge is two-pass (common max first,
    // then ONE rescale per stream and plain sums): no chain of dependent exponentials.
    slots_merge<LPK>(st);
    }
    SoftState r = st;
    if (NW > 1) { // (uniform) a single wave has the workgroup's result already
        if (!Q16 && lane < LPK) {
            part_acc[w * LPK + lane] = st.acc;
            if (lane == 0) part_ml[2 * w] = st.m, part_ml[2 * w + 1] = st.l;
        }
        ATTN_STAMP(5);
        __syncthreads();
        ATTN_STAMP(6);
        if (w != 0) return;
        // wave 0, all 64 lanes: lane ww < 16 fetches wave ww's max for the common max (DPP row reduction);
        // slot g folds waves g, g + KPW, ... — every LDS read is issued up front, the exponentials are independent
        constexpr int MAXW = BLOCK / 64, NPS = MAXW / KPW4 > 0 ? MAXW / KPW4 : 1;
        static_assert(MAXW <= 16, "lanes 0..15 (one DPP row) hold the waves' maxima");
        float M = lane < NW ? part_ml[2 * lane] : -INFINITY;
        float2 ml[NPS];
        float4 pa[NPS];
#pragma unroll
        for (int i = 0; i < NPS; i++) {
            const uint32_t ww = slot + i * KPW4, wc = ww < NW ? ww : 0;
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
            const bool live = slot + i * KPW4 < NW && ml[i].x > -INFINITY;
            const float f = live ? expf(ml[i].x - M) : 0.0f;
            r.l += ml[i].y * f;
            r.acc.x += pa[i].x * f, r.acc.y += pa[i].y * f, r.acc.z += pa[i].z * f, r.acc.w += pa[i].w * f;
        }
        slots_sum<LPK>(r);
    } else {
        ATTN_STAMP(5);
        ATTN_STAMP(6);
    }
    { // wave 0 from here on
        if (wait_qkv && lane == 0) { // every wave of this workgroup has passed the hand-off (merge barrier above / single wave)
            uint32_t* const seen = ho->seen + (hx * n_sp + sp) * 3;
            seen[0] = target[0], seen[1] = target[1], seen[2] = target[2];
        }
        if (n_active > 1) { // publish this chunk; the last arriver merges all of them
            constexpr uint32_t REC = DH + 4; // m, l, pad, pad, acc[DH]
            float* const head_buf = split_buf + (uint64_t)hx * n_sp * REC;
            float* const mine = head_buf + (uint64_t)sp * REC;
            if (lane < LPK) {
                if (lane == 0) split_put(mine, r.m, r.l);
                split_put(mine + 4 + 4 * lane, r.acc.x, r.acc.y);
                split_put(mine + 4 + 4 * lane + 2, r.acc.z, r.acc.w);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the write-through stores have left before the count
            uint32_t old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add((gu32*)(split_cnt + hx), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __shfl(old, 0, 64);
            if (old != n_active - 1) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // compiler only: the loads below stay below the count
            // slot g folds chunks g, g + KPW, ... in chunk order (not arrival order), then the slots merge
            r = SoftState{-INFINITY, 0.f, make_float4(0.f, 0.f, 0.f, 0.f)};
            for (uint32_t c = slot; c < n_active; c += KPW4) {
                const float* rec = head_buf + (uint64_t)c * REC;
                float om, ol;
                float4 oa;
                split_get(rec, om, ol);
                split_get(rec + 4 + 4 * li, oa.x, oa.y);
                split_get(rec + 4 + 4 * li + 2, oa.z, oa.w);
                soft_merge(r, om, ol, oa);
            }
            slots_merge<LPK>(r);
            if (lane == 0) __hip_atomic_store((gu32*)(split_cnt + hx), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-arm
        }
        if (lane < LPK) {
            const float inv_l = r.l > 0.f ? 1.0f / r.l : 0.0f;
            const float o[4] = {r.acc.x * inv_l, r.acc.y * inv_l, r.acc.z * inv_l, r.acc.w * inv_l};
            stg4(p.dst + 4 * lane, make_float4(o[0], o[1], o[2], o[3])); // dst_rs == 1, 16-byte aligned (planner)
            if (p.dst2) {
                const bool through = ho && ho->out_cnt; // consumers in this launch: write-through (agent scope)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float* const d = p.dst2 + (uint64_t)d2_off + (uint64_t)(4 * lane + e) * p.d2_rs;
                    if (through)
                        __hip_atomic_store((__attribute__((address_space(1))) float*)d, o[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        stg1(d, o[e]);
                }
            }
        }
        if (ho && ho->out_cnt) { // the head's rows have left this wave: tell the O projection
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            if (lane == 0) __hip_atomic_fetch_add((gu32*)ho->out_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    ATTN_STAMP(7);
#undef ATTN_STAMP
}

template <int LPK, bool KVQ, int BLOCK = kAttnBlock>
__global__ void __launch_bounds__(BLOCK) attention_decode_kernel(const AttnDecodeParams* __restrict__ params, float* split_buf,
                                                                 uint32_t* split_cnt, uint32_t split_min_keys) {
    attention_decode_body<LPK, KVQ, BLOCK>(params, split_buf, split_cnt, split_min_keys, blockIdx.x, blockIdx.y, gridDim.y, nullptr);
}

} // namespace
} // namespace zgml
