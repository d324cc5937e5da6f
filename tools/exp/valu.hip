// Experiment: issue cost of the candidate per-weight instructions on gfx950 (4 waves/SIMD, all CUs).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, t = 0.5f + threadIdx.x, q = 1.25f;
    float c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    uint32_t w = threadIdx.x * 2654435761u;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { // plain fmac, 4 independent accumulators
            asm volatile(REP64("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(t), "v"(q));
        } else if (MODE == 1) { // dpp row_newbcast fmac
            asm volatile(REP64("v_fmac_f32_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(t), "v"(q));
        } else if (MODE == 2) { // cvt_off sdwa
            asm volatile(REP64("v_cvt_off_f32_i4_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n v_cvt_off_f32_i4_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_off_f32_i4_sdwa %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_off_f32_i4_sdwa %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n")
                         : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3) : "v"(w));
            a0 += c0 + c1 + c2 + c3;
        } else if (MODE == 3) { // cvt_f32_ubyte
            asm volatile(REP64("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4\n")
                         : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3) : "v"(w));
            a0 += c0 + c1 + c2 + c3;
        } else if (MODE == 4) { // pk_fma_f32: 2 fma per instr
            typedef float float2v __attribute__((ext_vector_type(2)));
            float2v p0 = {a0, a1}, p1 = {a2, a3}, tt = {t, t}, qq = {q, q};
            asm volatile(REP64("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n")
                         : "+v"(p0), "+v"(p1) : "v"(tt), "v"(qq));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y;
        } else if (MODE == 5) { // cvt_f32_i32 sdwa sext
            asm volatile(REP64("v_cvt_f32_i32_sdwa %0, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n v_cvt_f32_i32_sdwa %1, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_i32_sdwa %2, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_i32_sdwa %3, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n")
                         : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3) : "v"(w));
            a0 += c0 + c1 + c2 + c3;
        } else if (MODE == 6) { // v_and_or_b32 (magic number build)
            uint32_t r0, r1, r2, r3;
            asm volatile(REP64("v_and_or_b32 %0, %4, 15, %5\n v_and_or_b32 %1, %4, 15, %5\n v_and_or_b32 %2, %4, 15, %5\n v_and_or_b32 %3, %4, 15, %5\n")
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(w), "v"(0x4B000000u));
            a0 += __uint_as_float(r0 ^ r1 ^ r2 ^ r3);
        } else if (MODE == 7) { // mixed: cvt_off sdwa + fmac dpp alternating (as the kernel)
            asm volatile(REP64("v_cvt_off_f32_i4_sdwa %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n v_fmac_f32_dpp %0, %9, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_cvt_off_f32_i4_sdwa %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_fmac_f32_dpp %1, %9, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(w), "v"(t));
        } else if (MODE == 8) { // v_dot2c_f32_bf16
            asm volatile(REP64("v_dot2c_f32_bf16 %0, %4, %5\n v_dot2c_f32_bf16 %1, %4, %5\n v_dot2c_f32_bf16 %2, %4, %5\n v_dot2c_f32_bf16 %3, %4, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w), "v"(w));
        } else if (MODE == 9) { // v_perm_b32
            uint32_t r0, r1, r2, r3;
            asm volatile(REP64("v_perm_b32 %0, %4, %5, %6\n v_perm_b32 %1, %4, %5, %6\n v_perm_b32 %2, %4, %5, %6\n v_perm_b32 %3, %4, %5, %6\n")
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(w), "v"(0x4B000000u), "v"(0x07060500u));
            a0 += __uint_as_float(r0 ^ r1 ^ r2 ^ r3);
        } else if (MODE == 10) { // fmac with SDWA? v_fma_mix / v_mad_mix f16 src: v_fma_mix_f32
            asm volatile(REP64("v_fma_mix_f32 %0, %4, %5, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %5, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %4, %5, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %4, %5, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w), "v"(t));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + c0 + c1 + c2 + c3;
}

template <int MODE>
int run(const char* name, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200, wgs = 256; // 16 waves/WG -> 4 waves/SIMD
    k<MODE><<<wgs, 1024>>>(out, 10);
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    k<MODE><<<wgs, 1024>>>(out, iters);
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x iters x 256 instr
    double instr_per_simd = 4.0 * iters * 256;
    printf("%-28s %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 1024 * 4));
    run<0>("v_fmac_f32", out);
    run<1>("v_fmac_f32_dpp newbcast", out);
    run<2>("v_cvt_off_f32_i4_sdwa", out);
    run<3>("v_cvt_f32_ubyteN", out);
    run<4>("v_pk_fma_f32", out);
    run<5>("v_cvt_f32_i32_sdwa sext", out);
    run<6>("v_and_or_b32", out);
    run<7>("mix cvt_off+fmac_dpp", out);
    run<8>("v_dot2c_f32_bf16", out);
    run<9>("v_perm_b32", out);
    run<10>("v_fma_mix_f32", out);
    return 0;
}
