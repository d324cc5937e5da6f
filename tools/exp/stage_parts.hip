// What an 8-k-group stage of the M <= 32 shared-A tile kernel costs, part by part (gfx950): 24 MFMAs (16x16x32 bf16, four
// chains) alone, + the B preparation (74 vector instructions), + the A pieces read from LDS, + a workgroup barrier every four
// stages. 512 threads per workgroup (two waves per SIMD), one workgroup per CU. Prints cycles per MFMA per SIMD at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/stage_parts tools/exp/stage_parts.hip && tools/exp/stage_parts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float mfma_f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float xf32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t xu32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8_t as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8_t, v); }
__device__ __forceinline__ uint32_t pack_hi16(float a, float b) { return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u); }
template <int CTRL>
__device__ __forceinline__ float row_bcast_z(int v) { return __int_as_float(__builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true)); }

template <int J, int C>
__device__ __forceinline__ void prep_b(const uint4 (&wq)[C], int sb_lo, int sb_hi, uint4 (&b1)[C], uint4 (&b2)[C]) {
    const int sbits = J >> 1 ? sb_hi : sb_lo;
    constexpr int C0 = 0x150 + 8 * (J & 1);
#pragma unroll
    for (int g = 0; g < C; g++) {
        const uint32_t lo = J == 0 ? wq[g].x : J == 1 ? wq[g].y : J == 2 ? wq[g].z : wq[g].w;
        uint32_t hi = lo >> 4;
        asm volatile("" : "+v"(hi));
#define W_(E, SRC, BYTE) (row_bcast_z<C0 + E>(sbits) * __builtin_amdgcn_cvt_off_f32_i4((int)(((SRC) >> (8 * BYTE)) & 0xFFu)))
        const xf32x2 w01 = {W_(0, lo, 0), W_(1, lo, 1)}, w23 = {W_(2, lo, 2), W_(3, lo, 3)};
        const xf32x2 w45 = {W_(4, hi, 0), W_(5, hi, 1)}, w67 = {W_(6, hi, 2), W_(7, hi, 3)};
#undef W_
        auto top = [](xf32x2 v) { return __builtin_bit_cast(xf32x2, __builtin_bit_cast(xu32x2, v) & 0xFFFF0000u); };
        const xf32x2 l01 = w01 - top(w01), l23 = w23 - top(w23), l45 = w45 - top(w45), l67 = w67 - top(w67);
        b1[g] = make_uint4(pack_hi16(w01.x, w01.y), pack_hi16(w23.x, w23.y), pack_hi16(w45.x, w45.y), pack_hi16(w67.x, w67.y));
        b2[g] = make_uint4(pack_hi16(l01.x, l01.y), pack_hi16(l23.x, l23.y), pack_hi16(l45.x, l45.y), pack_hi16(l67.x, l67.y));
    }
}

// PREP: B pieces prepared per stage (else constant); LDS: A pieces read per stage (else constant); BAR: barrier every 4 stages;
// SCHED: the kernel's sched_group_barrier interleave
template <bool PREP, bool LDS, bool BAR, bool SCHED>
__global__ void __launch_bounds__(512) k(float* out, const uint4* wsrc, int steps) {
    constexpr int C = 2, RT = 2;
    __shared__ uint4 lds[2 * RT * 768];
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t i = threadIdx.x; i < 2 * RT * 768; i += 512) lds[i] = make_uint4(i * 2654435761u, i * 40503u, i ^ 0x3F803F80u, 0x3F803F80u);
    __syncthreads();
    uint4 wq[C] = {wsrc[threadIdx.x], wsrc[threadIdx.x + 512]};
    int sl = 0x3C003C00 + (int)lane, sh = 0x3A003A00 + (int)lane;
    mfma_f4 acc[C][RT], acc2[C][RT];
#pragma unroll
    for (int g = 0; g < C; g++)
#pragma unroll
        for (int t = 0; t < RT; t++) acc[g][t] = acc2[g][t] = mfma_f4{0.f, 0.f, 0.f, 0.f};
    uint4 xa[RT][3];
    struct BP { uint4 b1[C], b2[C]; } p0, p1;
#pragma unroll
    for (int t = 0; t < RT; t++)
#pragma unroll
        for (int p = 0; p < 3; p++) xa[t][p] = lds[(t * 12 + p) * 64 + lane];
    prep_b<0, C>(wq, sl, sh, p0.b1, p0.b2);
    prep_b<1, C>(wq, sl, sh, p1.b1, p1.b2);
    auto read_a = [&](uint32_t buf, int J, int t) {
#pragma unroll
        for (int p = 0; p < 3; p++) xa[t][p] = lds[buf * (RT * 768) + ((t * 4 + J) * 3 + p) * 64 + lane];
    };
    auto mfmas = [&](const BP& o, int t) {
#pragma unroll
        for (int p = 0; p < 3; p++) {
#pragma unroll
            for (int g = 0; g < C; g++) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(xa[t][p]), as_bf16x8(o.b1[g]), acc[g][t], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < C; g++) acc2[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(xa[t][p]), as_bf16x8(o.b2[g]), acc2[g][t], 0, 0, 0);
        }
    };
#define STAGE(CUR, NXT, BUF_N, JN)                                                     \
    do {                                                                               \
        if (PREP) prep_b<JN, C>(wq, sl, sh, NXT.b1, NXT.b2);                           \
        mfmas(CUR, 0);                                                                 \
        if (LDS) read_a(BUF_N, JN, 0);                                                 \
        mfmas(CUR, 1);                                                                 \
        if (LDS) read_a(BUF_N, JN, 1);                                                 \
        if (SCHED) {                                                                   \
            _Pragma("unroll") for (int t_ = 0; t_ < RT; t_++) {                        \
                _Pragma("unroll") for (int k_ = 0; k_ < 6 * C; k_++) {                 \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                 \
                }                                                                      \
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                     \
            }                                                                          \
        }                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                             \
    } while (0)
    uint32_t buf = 0;
    for (int s = 0; s < steps; s++) {
        STAGE(p0, p1, buf, 1);
        STAGE(p1, p0, buf, 2);
        STAGE(p0, p1, buf, 3);
        if (BAR) __syncthreads();
        if (PREP) { wq[0].x += 0x11111111u, wq[1].y ^= wq[0].x; sl ^= 1; }
        STAGE(p1, p0, buf ^ 1, 0);
        buf ^= 1;
    }
#undef STAGE
    float r = 0;
#pragma unroll
    for (int g = 0; g < C; g++)
#pragma unroll
        for (int t = 0; t < RT; t++)
#pragma unroll
            for (int v = 0; v < 4; v++) r += acc[g][t][v] + acc2[g][t][v];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <bool PREP, bool LDS, bool BAR, bool SCHED>
void run(const char* name, float* out, const uint4* w, int wgs) {
    const int steps = 400;
    k<PREP, LDS, BAR, SCHED><<<wgs, 512>>>(out, w, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    k<PREP, LDS, BAR, SCHED><<<wgs, 512>>>(out, w, steps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = 2.0 * steps * 96; // two waves per SIMD, 96 MFMAs per wave and step
    printf("%-58s %3d workgroups: %7.1f us, %5.2f cycles per MFMA per SIMD (%.2f us per step)\n", name, wgs, ms * 1e3, ms * 1e6 * 2.4 / mfma_per_simd, ms * 1e3 / steps);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    uint4* w; hipMalloc(&w, 1024 * 16); hipMemset(w, 0x5a, 1024 * 16);
    for (int wgs : {1, 256}) {
        run<false, false, false, false>("MFMAs only", out, w, wgs);
        run<false, true, false, false>("+ A pieces from LDS", out, w, wgs);
        run<true, false, false, false>("+ B preparation (compiler order)", out, w, wgs);
        run<true, false, false, true>("+ B preparation (1 MFMA : 3 VALU)", out, w, wgs);
        run<true, true, false, true>("+ B preparation + LDS (1 : 3)", out, w, wgs);
        run<true, true, true, true>("+ B preparation + LDS + barrier per step (the kernel's loop)", out, w, wgs);
        run<true, true, true, false>("the same, compiler order", out, w, wgs);
    }
    return 0;
}
