// qmv_common.h — what the quantized mat-vec (qmatvec.hip: M = 1 decode kernels, weight packing) and the quantized tile kernels
// (qmatmul_tiles.hip: M > 1 prefill kernels) share: a few constants, load helpers and the column-group mapping. Everything
// device-side here sits in an anonymous namespace (each translation unit has its own copy; nothing is exported).
#pragma once
#include "kernels.h"

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zgml {

// the M = 1 launch of launch_qmatmul (one matrix, no prologue, no epilogue; M rows through blockIdx.y): qmatvec.hip
void launch_qmatvec_rows(hipStream_t s, const QWeightDev& w, const QMatmulParams& p, bool xvec);

namespace {

constexpr int kBlock = 256;      // helper kernels
constexpr int kMaxWaves = 16;    // mat-vec workgroup: up to 1024 threads

#define ZGML_SDWA "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_"

__device__ __forceinline__ float to_f32(__half h) { return __half2float(h); }
__device__ __forceinline__ float to_f32(float f) { return f; }

// XVEC: x is 16-byte aligned and K % 4 == 0 (decided on the host), so a float4 at any i < K with
// i % 4 == 0 is in range. Loads only here; the out-of-range zeroing happens at commit time so no
// wait sits between the loads.
template <bool XVEC>
__device__ __forceinline__ float4 load_x4(const float* x, uint32_t i, uint32_t K) {
    if (XVEC) return *(const float4*)(x + (i < K ? i : 0));
    const uint32_t last = K - 1; // K >= 1
    return make_float4(x[min(i, last)], x[min(i + 1, last)], x[min(i + 2, last)], x[min(i + 3, last)]);
}

__device__ __forceinline__ float4 zero_tail(float4 v, uint32_t i, uint32_t K) {
    return make_float4(i < K ? v.x : 0.f, i + 1 < K ? v.y : 0.f, i + 2 < K ? v.z : 0.f, i + 3 < K ? v.w : 0.f);
}

// blockIdx.x -> column group. Groups 2j and 2j+1 share the scales of block-column j; the
// dispatcher deals consecutive blocks round-robin over the 8 XCDs, so blocks b and b+8 share an
// XCD: put a pair there (full groups of 16 blocks; the tail maps linearly).
__device__ __forceinline__ uint32_t column_group(uint32_t b, uint32_t NB2) {
    const uint32_t full = NB2 & ~15u;
    if (b >= full) return b;
    const uint32_t j = (b >> 4) * 8 + (b & 7), half = (b >> 3) & 1;
    return 2 * j + half;
}

// Weight loads. NT: non-temporal (`global_load … nt`) — for weight sets larger than the 256 MB
// Infinity Cache every byte is read once per token, and not allocating it in the caches is worth
// ~5 % on the stream (4096^2 Q4_0 4.92 -> 4.67 us); a model that fits (SmolLM-135M) keeps the default
// policy because its weights stay cache-resident from one token to the next (nt: -1.8 % tok/s).
template <bool NT>
__device__ __forceinline__ uint4 wload(const uint4* p) {
    if (NT) {
        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
        const u4v v = __builtin_nontemporal_load((const u4v*)p);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *p;
}

// ── inline-asm helpers ──────────────────────────────────────────────────────────────────────
// The per-weight work is two VALU instructions: an SDWA convert (nibble or byte -> f32) and a
// v_fmac_f32 whose multiplier t[k] comes from lane k of the same 16-lane row through the DPP
// row_newbcast operand. Each 8-weight (Q4) / 4-weight (Q8) dword is ONE asm statement so hipcc
// neither pads the statement boundaries with s_nop nor serialises convert->fmac pairs through a
// single temporary; inside a statement producers and consumers are >= 4 instructions apart and
// two accumulators alternate, so nothing waits on the previous instruction.
// Hazard the assembler does not pad (guide §5.7): a VALU write of the DPP *source* VGPR (t) needs
// 2 wait states before the DPP read — dpp_fence() below provides them once per step.
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


template <typename ST>
struct Pair {
    ST a, b;
};

// all 64 lanes get the sum of the wave's 4 DPP rows, (r0 + r1) + (r2 + r3), by gfx950's row swaps (no LDS round trip)
__device__ __forceinline__ float rows_sum4(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// every lane of a DPP row gets the sum of the row's 16 lanes (quad swaps, half mirror, mirror)
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));
    return v;
}
// workgroup total of one value per thread, the same on every thread, fixed order: lanes, rows, then the (<= 16) waves
// with ONE LDS read per lane and a row fold (a serial loop over the waves is one dependent LDS round trip per wave)
__device__ __forceinline__ float block_total(float v, float* red, uint32_t bdim) {
    v = rows_sum4(row16_sum(v));
    const uint32_t n_waves = bdim >> 6, slot = threadIdx.x & 15;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float t = red[min(slot, n_waves - 1)];
    const float total = row16_sum(slot < n_waves ? t : 0.f);
    __syncthreads();
    return total;
}

inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

constexpr size_t kMaxLds = 160 * 1024;

inline uint32_t device_cus() {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    }
    return (uint32_t)n_cu;
}

} // namespace
} // namespace zgml
