// Experiment (round 2): issue cost per wave-instruction of the candidate per-weight instruction forms on gfx950,
// by ENCODING class (VOP1/VOP2 e32 vs VOP3 / SDWA / DPP / VOP3P) and by waves per SIMD (1, 2, 4), with the real
// shader clock read in-kernel (s_memtime / s_memrealtime), so the figures are true cycles and not "ns x 2.4".
//   hipcc -O3 --offload-arch=gfx950 tools/exp/valu2.hip -o tools/exp/valu2 && tools/exp/valu2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

struct Stamp { unsigned long long cyc, rt; };

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, Stamp* st, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, t = 0.5f + threadIdx.x, q = 1.25f;
    float c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    uint32_t w = threadIdx.x * 2654435761u, m = 0x0F0F0F0Fu;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { // VOP2 e32 fmac, lane-local multiplier
            asm volatile(REP64("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(t), "v"(q));
        } else if (MODE == 1) { // DPP fmac
            asm volatile(REP64("v_fmac_f32_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(t), "v"(q));
        } else if (MODE == 2) { // SDWA convert
            asm volatile(REP64("v_cvt_off_f32_i4_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n v_cvt_off_f32_i4_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_off_f32_i4_sdwa %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_off_f32_i4_sdwa %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n")
                         : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3) : "v"(w));
        } else if (MODE == 3) { // VOP1 e32 convert of the low nibble
            asm volatile(REP64("v_cvt_off_f32_i4_e32 %0, %4\n v_cvt_off_f32_i4_e32 %1, %4\n v_cvt_off_f32_i4_e32 %2, %4\n v_cvt_off_f32_i4_e32 %3, %4\n")
                         : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3) : "v"(w));
        } else if (MODE == 4) { // VOP1 e32 byte -> f32
            asm volatile(REP64("v_cvt_f32_ubyte0_e32 %0, %4\n v_cvt_f32_ubyte1_e32 %1, %4\n v_cvt_f32_ubyte2_e32 %2, %4\n v_cvt_f32_ubyte3_e32 %3, %4\n")
                         : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3) : "v"(w));
        } else if (MODE == 5) { // VOP2 e32 shift
            uint32_t r0_, r1_, r2_, r3_;
            asm volatile(REP64("v_lshrrev_b32_e32 %0, 4, %4\n v_lshrrev_b32_e32 %1, 8, %4\n v_lshrrev_b32_e32 %2, 12, %4\n v_lshrrev_b32_e32 %3, 16, %4\n")
                         : "=v"(r0_), "=v"(r1_), "=v"(r2_), "=v"(r3_) : "v"(w));
            c0 = __uint_as_float(r0_ ^ r1_ ^ r2_ ^ r3_);
        } else if (MODE == 6) { // VOP2 e32 and with an SGPR mask
            uint32_t r0_, r1_, r2_, r3_;
            asm volatile(REP64("v_and_b32_e32 %0, %5, %4\n v_and_b32_e32 %1, %5, %4\n v_and_b32_e32 %2, %5, %4\n v_and_b32_e32 %3, %5, %4\n")
                         : "=v"(r0_), "=v"(r1_), "=v"(r2_), "=v"(r3_) : "v"(w), "s"(m));
            c0 = __uint_as_float(r0_ ^ r1_ ^ r2_ ^ r3_);
        } else if (MODE == 7) { // VOP3P packed fma
            typedef float float2v __attribute__((ext_vector_type(2)));
            float2v p0 = {a0, a1}, p1 = {a2, a3}, tt = {t, t}, qq = {q, q};
            asm volatile(REP64("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n")
                         : "+v"(p0), "+v"(p1) : "v"(tt), "v"(qq));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y;
        } else if (MODE == 8) { // VOP3 bit-field extract (signed)
            int r0_, r1_, r2_, r3_;
            asm volatile(REP64("v_bfe_i32 %0, %4, 4, 4\n v_bfe_i32 %1, %4, 8, 4\n v_bfe_i32 %2, %4, 12, 4\n v_bfe_i32 %3, %4, 16, 4\n")
                         : "=v"(r0_), "=v"(r1_), "=v"(r2_), "=v"(r3_) : "v"(w));
            c0 = __int_as_float(r0_ ^ r1_ ^ r2_ ^ r3_);
        } else if (MODE == 9) { // fp8 pair -> 2 x f32 (VOP1, word select through op_sel => VOP3 encoding for the high word)
            typedef float float2v __attribute__((ext_vector_type(2)));
            float2v p0, p1;
            asm volatile(REP64("v_cvt_pk_f32_fp8_e32 %0, %2\n v_cvt_pk_f32_fp8_sdwa %1, %2 src0_sel:WORD_1\n v_cvt_pk_f32_fp8_e32 %0, %2\n v_cvt_pk_f32_fp8_sdwa %1, %2 src0_sel:WORD_1\n")
                         : "=v"(p0), "=v"(p1) : "v"(w));
            c0 = p0.x + p0.y + p1.x + p1.y;
        } else if (MODE == 10) { // the current kernel's pair: SDWA convert + DPP fmac
            asm volatile(REP64("v_cvt_off_f32_i4_sdwa %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n v_fmac_f32_dpp %0, %9, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_cvt_off_f32_i4_sdwa %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_fmac_f32_dpp %1, %9, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(w), "v"(t));
        } else if (MODE == 11) { // candidate A (exact, signed): shift e32 + cvt_off e32 + fmac e32, lane-local t
            uint32_t h;
            asm volatile(REP64("v_lshrrev_b32_e32 %6, 4, %7\n v_cvt_off_f32_i4_e32 %4, %7\n v_fmac_f32 %0, %8, %5\n v_cvt_off_f32_i4_e32 %5, %6\n v_fmac_f32 %1, %8, %4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "=&v"(h) : "v"(w), "v"(t));
        } else if (MODE == 12) { // candidate B (unsigned bytes): and + 4 x cvt_ubyte + 4 x fmac
            uint32_t h;
            asm volatile(REP64("v_and_b32_e32 %8, %10, %9\n v_cvt_f32_ubyte0_e32 %4, %8\n v_cvt_f32_ubyte1_e32 %5, %8\n v_cvt_f32_ubyte2_e32 %6, %8\n v_cvt_f32_ubyte3_e32 %7, %8\n v_fmac_f32 %0, %11, %4\n v_fmac_f32 %1, %11, %5\n v_fmac_f32 %2, %11, %6\n v_fmac_f32 %3, %11, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(h) : "v"(w), "s"(m), "v"(t));
        } else if (MODE == 13) { // LDS read of a 16-byte item per lane (the weight operand fetched from an LDS ring)
            extern __shared__ uint4 lds[];
            uint4 v;
            const uint32_t addr = (threadIdx.x & 1023) * 16;
            asm volatile(REP64("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "=v"(v) : "v"(addr));
            c0 = __uint_as_float(v.x ^ v.y ^ v.z ^ v.w);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{t1 - t0, r1 - r0};
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + c0 + c1 + c2 + c3;
}

template <int MODE>
int run(const char* name, float* out, Stamp* st, int instr_per_rep) {
    const int iters = 100, wgs = 256;
    printf("%-34s", name);
    for (int threads : {256, 512, 1024}) { // 1, 2, 4 waves per SIMD
        k<MODE><<<wgs, threads, 16384>>>(out, st, 4);
        CK(hipDeviceSynchronize());
        k<MODE><<<wgs, threads, 16384>>>(out, st, iters);
        CK(hipDeviceSynchronize());
        Stamp h[256];
        CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0;
        for (int i = 0; i < wgs; i++) cyc += h[i].cyc, rt += h[i].rt;
        cyc /= wgs, rt /= wgs;
        const double per_simd = (double)(threads / 256) * iters * 64 * instr_per_rep;
        printf("  %dw/SIMD: %5.2f cyc/instr (clk %.2f GHz)", threads / 256, cyc / per_simd, cyc / (rt * 10.0));
    }
    printf("\n");
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 1024 * 4));
    Stamp* st; CK(hipMalloc(&st, 256 * sizeof(Stamp)));
    run<0>("v_fmac_f32 (VOP2 e32)", out, st, 4);
    run<1>("v_fmac_f32_dpp newbcast", out, st, 4);
    run<2>("v_cvt_off_f32_i4_sdwa", out, st, 4);
    run<3>("v_cvt_off_f32_i4_e32", out, st, 4);
    run<4>("v_cvt_f32_ubyteN_e32", out, st, 4);
    run<5>("v_lshrrev_b32_e32", out, st, 4);
    run<6>("v_and_b32_e32 (sgpr mask)", out, st, 4);
    run<7>("v_pk_fma_f32 (2 fma)", out, st, 4);
    run<8>("v_bfe_i32 (VOP3)", out, st, 4);
    run<9>("v_cvt_pk_f32_fp8 (2 cvt)", out, st, 4);
    run<10>("now: cvt_sdwa + fmac_dpp /weight", out, st, 2);   // per rep 4 instr = 2 weights -> cycles per weight
    run<11>("A: shr+cvt_e32+fmac_e32 /weight", out, st, 2);   // per rep 5 instr = 2 weights
    run<12>("B: and+4ubyte+4fmac /weight", out, st, 4);        // per rep 9 instr = 4 weights
    run<13>("ds_read_b128 + wait (latency)", out, st, 1);
    return 0;
}
