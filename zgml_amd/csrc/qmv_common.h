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
