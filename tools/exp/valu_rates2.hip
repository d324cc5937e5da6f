// SIMD-level issue cost (cycles per wave-instruction with 4 waves per SIMD) of candidate instructions for the tile kernels' B preparation (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/valu_rates2 tools/exp/valu_rates2.hip && tools/exp/valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(X) X X X X X X X X X X X X X X X X
#define KERNEL(NAME, ASM, ...)                                                                                   \
    __global__ void NAME(float* out, int iters, uint32_t seed) {                                                \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;                                 \
        float f0 = a0, f1 = a1, f2_ = a2, f3 = a3;                                                               \
        f2 p0 = {f0, f1}, p1 = {f2_, f3};                                                                        \
        for (int it = 0; it < iters; it++) asm volatile(REP16(ASM) __VA_ARGS__);                                 \
        if (f0 + f1 + p0.x + p1.y + (float)(a0 + a1) == 12345.678f) out[0] = 1;                                  \
    }
KERNEL(k_mul, "v_mul_f32 %0, %2, %3\n v_mul_f32 %1, %3, %2\n", : "=v"(f0), "=v"(f1) : "v"(f2_), "v"(f3))
KERNEL(k_sub, "v_sub_f32 %0, %2, %3\n v_sub_f32 %1, %3, %2\n", : "=v"(f0), "=v"(f1) : "v"(f2_), "v"(f3))
KERNEL(k_fma, "v_fma_f32 %0, %2, %3, %2\n v_fma_f32 %1, %3, %2, %3\n", : "=v"(f0), "=v"(f1) : "v"(f2_), "v"(f3))
KERNEL(k_fmac, "v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %3, %2\n", : "+v"(f0), "+v"(f1) : "v"(f2_), "v"(f3))
KERNEL(k_cvt_ub, "v_cvt_f32_ubyte1 %0, %2\n v_cvt_f32_ubyte2 %1, %2\n", : "=v"(f0), "=v"(f1) : "v"(a2))
KERNEL(k_cvt_i4, "v_cvt_off_f32_i4 %0, %2\n v_cvt_off_f32_i4 %1, %3\n", : "=v"(f0), "=v"(f1) : "v"(a2), "v"(a3))
KERNEL(k_cvt_i4_sdwa, "v_cvt_off_f32_i4_sdwa %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_off_f32_i4_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n", : "=v"(f0), "=v"(f1) : "v"(a2))
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %2, 16, %3\n v_lshl_or_b32 %1, %3, 16, %2\n", : "=v"(a0), "=v"(a1) : "v"(a2), "v"(a3))
KERNEL(k_and_or, "v_and_or_b32 %0, %2, %4, %3\n v_and_or_b32 %1, %3, %4, %2\n", : "=v"(a0), "=v"(a1) : "v"(a2), "v"(a3), "s"(0xffff0000u))
KERNEL(k_alignbit, "v_alignbit_b32 %0, %2, %3, 16\n v_alignbit_b32 %1, %3, %2, 16\n", : "=v"(a0), "=v"(a1) : "v"(a2), "v"(a3))
KERNEL(k_bfi, "v_bfi_b32 %0, %4, %2, %3\n v_bfi_b32 %1, %4, %3, %2\n", : "=v"(a0), "=v"(a1) : "v"(a2), "v"(a3), "s"(0xffff0000u))
KERNEL(k_perm, "v_perm_b32 %0, %2, %3, %4\n v_perm_b32 %1, %3, %2, %4\n", : "=v"(a0), "=v"(a1) : "v"(a2), "v"(a3), "s"(0x07060302u))
KERNEL(k_or, "v_or_b32 %0, %2, %3\n v_or_b32 %1, %3, %2\n", : "=v"(a0), "=v"(a1) : "v"(a2), "v"(a3))
KERNEL(k_lshr, "v_lshrrev_b32 %0, 16, %2\n v_lshrrev_b32 %1, 16, %3\n", : "=v"(a0), "=v"(a1) : "v"(a2), "v"(a3))
KERNEL(k_mul_dpp, "v_mul_f32_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n", : "=v"(f0), "=v"(f1) : "v"(f2_), "v"(f3))
KERNEL(k_cvt_pk_bf16, "v_cvt_pk_bf16_f32 %0, %2, %3\n v_cvt_pk_bf16_f32 %1, %3, %2\n", : "=v"(a0), "=v"(a1) : "v"(f2_), "v"(f3))
KERNEL(k_pk_mul, "v_pk_mul_f32 %0, %1, %1\n v_pk_mul_f32 %0, %1, %1\n", : "=v"(p0) : "v"(p1))
KERNEL(k_pk_fma, "v_pk_fma_f32 %0, %1, %1, %1\n v_pk_fma_f32 %0, %1, %1, %1\n", : "=v"(p0) : "v"(p1))
KERNEL(k_mov_sdwa, "v_mov_b32_sdwa %0, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_mov_b32_sdwa %1, %3 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n", : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3))
KERNEL(k_cvt_f16, "v_cvt_f32_f16 %0, %2\n v_cvt_f32_f16 %1, %3\n", : "=v"(f0), "=v"(f1) : "v"(a2), "v"(a3))
KERNEL(k_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %2, %3\n v_cvt_pkrtz_f16_f32 %1, %3, %2\n", : "=v"(a0), "=v"(a1) : "v"(f2_), "v"(f3))

template <typename K>
void run(const char* name, K kern, float* d) {
    const int iters = 2000;
    for (int threads : {512, 1024}) {
        kern<<<1, threads>>>(d, iters, 1);
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        hipEventRecord(e0);
        kern<<<1, threads>>>(d, iters, 1);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = (double)iters * 32 * (threads / 256);
        printf("%-22s %d waves per SIMD: %.2f ns per wave-instruction on a SIMD (%.2f cycles at 2.4 GHz)\n", name, threads / 256, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    }
}
int main() {
    float* d;
    hipMalloc(&d, 64);
#define R(N) run(#N, N, d)
    R(k_mul); R(k_sub); R(k_fma); R(k_fmac); R(k_cvt_ub); R(k_cvt_i4); R(k_cvt_i4_sdwa); R(k_lshl_or); R(k_and_or); R(k_alignbit); R(k_bfi); R(k_perm);
    R(k_or); R(k_lshr); R(k_mul_dpp); R(k_cvt_pk_bf16); R(k_pk_mul); R(k_pk_fma); R(k_mov_sdwa); R(k_cvt_f16); R(k_pkrtz);
    return 0;
}
