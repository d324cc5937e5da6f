// Experiment: v_mfma_f32_16x16x32_bf16 issue rate on gfx950 — independent accumulators, 1 or 2 waves per SIMD,
// with and without VALU work (a v_perm producing the B operand) in front of each MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    f4 acc[6];
    for (int i = 0; i < 6; i++) acc[i] = f4{0, 0, 0, 0};
    uint4 ua = make_uint4(threadIdx.x, 1, 2, 3), ub = make_uint4(4, threadIdx.x, 6, 7);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 1) { // B operand repacked by 4 v_perm per 6 MFMAs (compiler-visible producer)
                ub.x = __builtin_amdgcn_perm(ub.x, ua.y, 0x07060302u), ub.y = __builtin_amdgcn_perm(ub.y, ua.z, 0x07060302u);
                ub.z = __builtin_amdgcn_perm(ub.z, ua.w, 0x07060302u), ub.w = __builtin_amdgcn_perm(ub.w, ua.x, 0x07060302u);
            }
            const bf8 A = __builtin_bit_cast(bf8, ua), B = __builtin_bit_cast(bf8, ub);
#pragma unroll
            for (int q = 0; q < 6; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc[q], 0, 0, 0);
        }
    }
    float r = 0;
    for (int i = 0; i < 6; i++) r += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
int run(const char* name, float* out, int threads) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, wgs = 256;
    k<MODE><<<wgs, threads>>>(out, 10);
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    k<MODE><<<wgs, threads>>>(out, iters);
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 48; // MFMAs per wave
    const double waves_per_simd = threads / 256.0;
    printf("%-52s %d thr: %.3f ms -> %.1f cyc per MFMA per wave, %.1f cyc per MFMA per SIMD, %.0f TFLOP/s chip\n", name, threads, ms,
           ms * 1e6 / n * 2.4, ms * 1e6 / n * 2.4 / waves_per_simd, n * 16384.0 * (wgs * threads / 64) / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 512 * 4));
    run<0>("16x16x32 bf16, 6 independent accumulators", out, 256);
    run<0>("16x16x32 bf16, 6 independent accumulators", out, 512);
    run<1>("same + 4 v_perm producing B per 6 MFMAs", out, 256);
    run<1>("same + 4 v_perm producing B per 6 MFMAs", out, 512);
    return 0;
}
