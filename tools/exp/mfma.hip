// Experiment: f32 MFMA issue rate on gfx950 and how much independent VALU work fits beside it.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    f16v big = {0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x, v0 = 1.f, v1 = 2.f, v2 = 3.f, v3 = 4.f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 32; j++) {
            if (MODE == 0) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
            } else if (MODE == 1) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
            } else if (MODE == 2 || MODE == 3 || MODE == 5) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
                constexpr int NV = MODE == 2 ? 1 : MODE == 3 ? 2 : 4; // x4 independent VALU
#pragma unroll
                for (int q = 0; q < NV; q++)
                    asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n"
                                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
            } else if (MODE == 4) {
                big = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big, 0, 0, 0);
            } else if (MODE == 6) { // operand produced by a VALU op right before each MFMA
                float t;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(v0));
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(t, b, acc0, 0, 0, 0);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[1] + big[3] + v0 + v1 + v2 + v3;
}

template <int MODE>
int run(const char* name, float* out, int mfma_per_j, double flops_per_mfma) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, wgs = 256; // 4 waves/WG -> 1 wave/SIMD
    k<MODE><<<wgs, 256>>>(out, 10);
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    k<MODE><<<wgs, 256>>>(out, iters);
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 32 * mfma_per_j;
    printf("%-44s %.3f ms -> %.1f ns per MFMA per wave (%.1f cyc @2.4GHz), %.1f TFLOP/s chip\n", name, ms, ms * 1e6 / n,
           ms * 1e6 / n * 2.4, n * flops_per_mfma * 1024 / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 256 * 4));
    run<0>("16x16x4 f32, one dependent chain", out, 1, 2048);
    run<1>("16x16x4 f32, two chains", out, 2, 2048);
    run<2>("one chain + 4 VALU per MFMA", out, 1, 2048);
    run<3>("one chain + 8 VALU per MFMA", out, 1, 2048);
    run<5>("one chain + 16 VALU per MFMA", out, 1, 2048);
    run<4>("32x32x2 f32, one chain", out, 1, 4096);
    run<6>("one chain, A from preceding v_mul", out, 1, 2048);
    return 0;
}
