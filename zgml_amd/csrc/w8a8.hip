// w8a8.hip — the reference's W8A8 mat-vec arm on the device (opt-in: ZGML_HIP_OPT_W8A8).
//
// The reference's CPU executor takes M = 1 quantized matmuls through int8 activations x int8 weights when a weight carries its
// transposed image (src/backend/reference.zig:512-528): quantizeInput (src/quant.zig:604-640) turns x into int8 + one f32 scale per
// 32 values, prepareTransposed (src/quant.zig:560-603) re-quantises the weight per (column, block of 32 k) into [N][K] int8 + [N][K/32]
// f32, gemvRange (src/quant.zig:320-440) forms, per column, sum over blocks of f32(int32 dot) * (s_x * s_w) — blocks ascending, one
// rounding per product, one per add. This file restates exactly that, so the outputs are BIT-IDENTICAL to the reference's W8A8 arm
// (oracle: zo_quantize_input / zo_prepare_transposed / zo_gemv_range; tests/test_hip_w8a8.py):
//   * integers are exact, so the int32 dot of a block may be formed in any order and by any number of lanes (v_dot4_i32_i8);
//   * every f32 operation of the reference is done once, with the same operands, by one lane: max |x| / 127, 127 / max |x|, x * inv
//     truncated toward zero, f32(int) * (s_x * s_w), and the running sum over blocks in ascending order (no FMA anywhere: the build
//     runs with -ffp-contract=off and the expressions below are separate statements).
// Layout (pack_w8a8_kernel, once per weight at compile_program): column group g = 16 columns, step s = two blocks of 32 k:
//   qs: uint4[N / 16][K / 64][64]   lane = 32 p + 16 h + c holds column 16 g + c, k = 64 s + 32 p + 16 h .. + 15
//   sc: float[N / 16][K / 32][16]   the re-quantised weight's scale of (block, column)
// so a wave's 16-byte loads are 1 KiB contiguous. One workgroup owns a column group and all of K (as the exact mat-vecs do);
// its waves split the steps, every wave keeps its loads in flight while the workgroup quantises x (each workgroup redundantly:
// K floats from L2), the block products go to LDS as prod[column][block] and the 16 lanes of wave 0 add them up in order.
// HBM bytes per product: K N (int8) + 4 K N / 32 (scales) + 4 K + 4 N — 18.9 MB at 4096 x 4096 (BASELINE configs[0]'s workload).
#include "kernels.h"

#include <hip/hip_runtime.h>
#include <math.h>

namespace zgml {
namespace {

constexpr int kW8Waves = 8, kW8Block = 64 * kW8Waves, kW8U = 8; // loads in flight per wave

typedef uint32_t u4v_t __attribute__((ext_vector_type(4)));

// prepareTransposed for one (column, block): thread per (n, b); the dequantised value is f32(q) * scale(flat / bs) with the flat
// index k * N + n of the reference's [K][N] layout (scales run along n). Expressions as zo_prepare_transposed.
__global__ void pack_w8a8_kernel(const int8_t* __restrict__ data, const float* __restrict__ scales, uint32_t K, uint32_t N, uint32_t bs,
                                 uint32_t* __restrict__ qs, float* __restrict__ sc) {
    const uint32_t bpr = K / 32;
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint64_t)N * bpr) return;
    const uint32_t n = (uint32_t)(idx % N), b = (uint32_t)(idx / N); // (n fastest: the reads of a warp are contiguous in n)
    float val[32];
    float max_abs = 0.f;
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const uint64_t flat = (uint64_t)(32 * b + k) * N + n;
        val[k] = (float)data[flat] * scales[flat / bs];
        const float a = fabsf(val[k]);
        if (a > max_abs) max_abs = a;
    }
    const float scale = max_abs > 0.f ? max_abs / 127.0f : 1.0f;
    const float inv = max_abs > 0.f ? 127.0f / max_abs : 0.0f;
    const uint32_t g = n / 16, c = n % 16, s = b / 2, p = b % 2;
    sc[((uint64_t)g * bpr + b) * 16 + c] = scale;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        uint32_t w[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            w[d] = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float q = fmaxf(-127.0f, fminf(val[16 * h + 4 * d + e] * inv, 127.0f));
                w[d] |= (uint32_t)(uint8_t)(int8_t)(int)q << (8 * e); // (int): truncation toward zero, as @intFromFloat
            }
        }
        const uint64_t item = ((uint64_t)g * (K / 64) + s) * 64 + 32 * p + 16 * h + c;
        *(u4v_t*)(qs + 4 * item) = u4v_t{w[0], w[1], w[2], w[3]};
    }
}

struct W8A8Args {
    const u4v_t* qs;
    const float* sc;
    const float* x;
    float* dst;
    uint32_t K, N, x_vec;
};

// dynamic LDS: xq int8[K] | xs float[nb4 = K / 32 rounded up to 4] | prod float[16][nb4 + 4]
__global__ void __launch_bounds__(kW8Block) w8a8_matvec_kernel(W8A8Args a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_w8[];
    const uint32_t K = a.K, nb = K / 32, S = K / 64;
    uint32_t* const xq = smem_w8;                   // K / 4 words
    float* const xs = (float*)(smem_w8 + K / 4);     // nb
    const uint32_t nb4 = (nb + 3) / 4 * 4;
    float* const prod = xs + nb4;                    // 16 rows of nb4 + 4 floats (16-byte aligned rows)
    const uint32_t pstride = nb4 + 4;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = blockIdx.x;
    const uint32_t c = lane & 15, h = (lane >> 4) & 1, p = lane >> 5;
    // ---- the wave's first loads (independent of x): weights and, for the lanes that will form the products, their scales
    const u4v_t* const qg = a.qs + (uint64_t)g * S * 64 + lane;
    const float* const sg = a.sc + (uint64_t)g * nb * 16 + c;
    u4v_t wv[kW8U];
    float ws[kW8U];
#pragma unroll
    for (int j = 0; j < kW8U; j++) {
        const uint32_t s = min(w + (uint32_t)j * kW8Waves, S - 1); // (clamped: unconditional loads)
        wv[j] = qg[(uint64_t)s * 64];
        ws[j] = sg[(uint64_t)(2 * s + p) * 16];
    }
    // ---- quantizeInput: 8 consecutive x per thread and round, a block of 32 = 4 adjacent threads
    for (uint32_t k0 = 8 * tid; k0 < K; k0 += 8 * kW8Block) {
        float xv[8];
        if (a.x_vec) { // (uniform) 16-byte aligned input
            const float4 v0 = *(const float4*)(a.x + k0), v1 = *(const float4*)(a.x + k0 + 4);
            xv[0] = v0.x, xv[1] = v0.y, xv[2] = v0.z, xv[3] = v0.w, xv[4] = v1.x, xv[5] = v1.y, xv[6] = v1.z, xv[7] = v1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) xv[e] = a.x[k0 + e];
        }
        float mx = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float t = fabsf(xv[e]);
            if (t > mx) mx = t;
        }
        mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mx), 0xB1, 0xF, 0xF, true))); // quad_perm [1,0,3,2]
        mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mx), 0x4E, 0xF, 0xF, true))); // quad_perm [2,3,0,1]
        const float scale = mx > 0.f ? mx / 127.0f : 1.0f;
        const float inv = mx > 0.f ? 127.0f / mx : 0.0f;
        uint32_t q[2] = {0, 0};
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float t = fmaxf(-127.0f, fminf(xv[e] * inv, 127.0f));
            q[e / 4] |= (uint32_t)(uint8_t)(int8_t)(int)t << (8 * (e % 4));
        }
        xq[k0 / 4] = q[0], xq[k0 / 4 + 1] = q[1];
        if ((tid & 3) == 0) xs[k0 / 32] = scale;
    }
    __syncthreads();
    // ---- block dots: lane (p, h, c) covers 16 k of block 2 s + p for column c; the two halves are added (exact integers), the lane
    // with h == 0 forms the product f32(dot) * (s_x * s_w) — one rounding for the scale product, one for the block product
    for (uint32_t s0 = w; s0 < S; s0 += kW8Waves * kW8U) {
        u4v_t nv[kW8U];
        float ns[kW8U];
        const bool more = s0 + kW8Waves * kW8U < S;
        if (more) {
#pragma unroll
            for (int j = 0; j < kW8U; j++) {
                const uint32_t s = min(s0 + (uint32_t)(j + kW8U) * kW8Waves, S - 1);
                nv[j] = qg[(uint64_t)s * 64];
                ns[j] = sg[(uint64_t)(2 * s + p) * 16];
            }
        }
#pragma unroll
        for (int j = 0; j < kW8U; j++) {
            const uint32_t s = s0 + (uint32_t)j * kW8Waves;
            if (s < S) { // (wave-uniform)
                const uint32_t b = 2 * s + p;
                const u4v_t xw = *(const u4v_t*)(xq + 8 * b + 4 * h);
                int32_t ia = 0;
#pragma unroll
                for (int d = 0; d < 4; d++) ia = __builtin_amdgcn_sdot4((int)wv[j][d], (int)xw[d], ia, false);
                const auto sw = __builtin_amdgcn_permlane16_swap((uint32_t)ia, (uint32_t)ia, false, false); // rows (0 0 2 2 | 1 1 3 3)
                ia = (int32_t)sw[0] + (int32_t)sw[1];
                if (h == 0) {
                    const float combined = xs[b] * ws[j];
                    prod[c * pstride + b] = (float)ia * combined;
                }
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < kW8U; j++) wv[j] = nv[j], ws[j] = ns[j];
        }
    }
    __syncthreads();
    // ---- the reference's running sum, blocks ascending, one lane per column
    if (tid < 16) {
        const float* const pr = prod + tid * pstride;
        float acc = 0.f;
        uint32_t b = 0;
        for (; b + 32 <= nb; b += 32) { // (all reads of a round first: one LDS round trip per 32 blocks, then the ordered adds)
            float4 t[8];
#pragma unroll
            for (int i = 0; i < 8; i++) t[i] = *(const float4*)(pr + b + 4 * i);
#pragma unroll
            for (int i = 0; i < 8; i++) acc = acc + t[i].x, acc = acc + t[i].y, acc = acc + t[i].z, acc = acc + t[i].w;
        }
        for (; b + 4 <= nb; b += 4) {
            const float4 t = *(const float4*)(pr + b);
            acc = acc + t.x, acc = acc + t.y, acc = acc + t.z, acc = acc + t.w;
        }
        for (; b < nb; b++) acc = acc + pr[b];
        a.dst[16 * g + tid] = acc;
    }
}

} // namespace

bool w8a8_applies(uint64_t K, uint64_t N, uint64_t bs) {
    // (LDS: K + 4 K / 32 + 64 (K / 32 + 4) bytes: 52 KB at K = 16384)
    return bs == 32 && K % 64 == 0 && N % 16 == 0 && K >= 64 && K <= 16384 && N >= 16; // K <= 16384 is the reference's own bound (reference.zig:514)
}
void w8a8_packed_bytes(uint64_t K, uint64_t N, uint64_t* qs_bytes, uint64_t* sc_bytes) {
    *qs_bytes = K * N;
    *sc_bytes = (K / 32) * N * sizeof(float);
}
void launch_pack_w8a8(hipStream_t s, const int8_t* raw_data, const float* raw_scales, const QWeightDev& out) {
    const uint64_t items = (uint64_t)out.N * (out.K / 32);
    pack_w8a8_kernel<<<(uint32_t)((items + 255) / 256), 256, 0, s>>>(raw_data, raw_scales, out.K, out.N, out.bs, (uint32_t*)out.qs, (float*)out.sc);
}
void launch_w8a8_matvec(hipStream_t s, const QWeightDev& w, const float* x, float* dst) {
    const uint32_t nb = w.K / 32;
    const size_t nb4 = (nb + 3) / 4 * 4;
    const size_t lds = (size_t)w.K + 4 * nb4 + 4 * 16 * (nb4 + 4);
    W8A8Args a{(const u4v_t*)w.qs, (const float*)w.sc, x, dst, w.K, w.N, ((uintptr_t)x % 16) == 0 ? 1u : 0u};
    w8a8_matvec_kernel<<<w.N / 16, kW8Block, lds, s>>>(a);
}

} // namespace zgml
