// Experiment (round 3): a Q4_0 mat-vec with K ON LANES — a lane holds 2 consecutive k x 16 columns (one 16-byte item),
// its multiplier t = scale * x is a per-lane scalar, the nibbles become f32 pairs with v_cvt_pk_f32_fp8 (a byte 0x0N is
// the e4m3 value N * 2^-9: denormals and the first binade are linear) and every two weights cost one v_pk_fma_f32.
// Offset-binary nibbles (q = w + 8); the -8 * sum(t) correction is applied per lane before the cross-lane fold.
// Compared against the shipped kernel's ~3.3 VALU instructions per weight (profiles/r02_qmatvec_pmc_series.csv).
//   hipcc -O3 --offload-arch=gfx950 tools/exp/kon.hip -o tools/exp/kon
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

__host__ __device__ inline uint32_t synth_q(uint64_t flat, uint32_t id) { return (uint32_t)((flat * 7 + (flat >> 5) * 3 + id * 5) & 15); }
__host__ __device__ inline float synth_s(uint64_t blk, uint32_t id) { return 0.015625f * (1.0f + (float)((blk + id) % 7) / 8.0f); }

// layout: qs[g][p] (uint4): k = 2p + (dd >> 1), column 16g + 8 (dd & 1) + b (low nibble) / + 4 + b (high nibble), dd = dword, b = byte
//         sc[j][p] (half2): scale(2p, j), scale(2p + 1, j)
__global__ void fill_kernel(uint4* qs, uint32_t* sc, uint32_t K, uint32_t N, uint32_t id) {
    const uint32_t P = K / 2, NB2 = N / 16, NB = N / 32;
    for (uint64_t it = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; it < (uint64_t)NB2 * P; it += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t p = it % P, g = it / P;
        uint32_t d[4];
        for (int dd = 0; dd < 4; dd++) {
            const uint32_t k = 2 * p + (dd >> 1), c0 = 16 * g + 8 * (dd & 1);
            uint32_t word = 0;
            for (int b = 0; b < 4; b++) {
                const uint32_t lo = synth_q((uint64_t)k * N + c0 + b, id), hi = synth_q((uint64_t)k * N + c0 + 4 + b, id);
                word |= (lo | (hi << 4)) << (8 * b);
            }
            d[dd] = word;
        }
        qs[it] = make_uint4(d[0], d[1], d[2], d[3]);
        if ((g & 1) == 0) {
            const uint32_t j = g >> 1;
            const __half a = __float2half(synth_s((uint64_t)(2 * p) * NB + j, id)), b = __float2half(synth_s((uint64_t)(2 * p + 1) * NB + j, id));
            sc[(uint64_t)j * P + p] = (uint32_t)__half_as_ushort(a) | ((uint32_t)__half_as_ushort(b) << 16);
        }
    }
}

template <bool NT>
__device__ __forceinline__ uint4 wload(const uint4* p) {
    if (NT) {
        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
        const u4v v = __builtin_nontemporal_load((const u4v*)p);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *p;
}

__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));
    return v;
}
// (a, b) -> lanes 0-31: a folded over xor 32, lanes 32-63: b folded
__device__ __forceinline__ float fold32(float a, float b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// rows (0, 2): a folded over xor 16, rows (1, 3): b folded
__device__ __forceinline__ float fold16(float a, float b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// NORM: x = rmsnorm(xr) * gamma folded in — the mat-vec is linear in x, so the factor 1/sqrt(mean(x^2) + eps) is applied to the
// finished sums and the sum of squares rides in the same cross-lane fold (every k belongs to exactly one lane of the workgroup)
template <int DEPTH, bool NT, bool NORM, bool ROT, int XMODE = 0, bool GNT = false>
__global__ void __launch_bounds__(1024) kon_kernel(const uint4* __restrict__ qs, const uint32_t* __restrict__ sc, const float* __restrict__ x,
                                                   float* __restrict__ y, uint32_t K, uint32_t n_waves, float post, const float* __restrict__ gamma,
                                                   unsigned long long* stamps) {
    __shared__ float red[16 * 17];
#define STAMP(i) do { if (stamps && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) stamps[(blockIdx.x ? 8 : 0) + i] = __builtin_readcyclecounter(); } while (0)
    STAMP(0);
    const uint32_t g = blockIdx.x, P = K / 2, stride = n_waves * 64;
    const uint4* q = qs + (uint64_t)g * P;
    const uint32_t* s = sc + (uint64_t)(g >> 1) * P;
    const f2* x2 = (const f2*)x;
    const f2* g2 = (const f2*)gamma;
    const uint32_t n_groups = (P + stride * DEPTH - 1) / (stride * DEPTH);
    uint4 wq[DEPTH];
    uint32_t sv[DEPTH];
    f2 xv[DEPTH], gv[DEPTH];
    __shared__ float xl[XMODE == 2 ? 11008 : 4];
    if (XMODE == 2) { // x once per workgroup through LDS (float4 loads), then per-lane ds_read_b64
        for (uint32_t k = threadIdx.x * 4; k < K; k += stride * 4) *(float4*)(xl + k) = *(const float4*)(x + k);
        __syncthreads();
    }
    // every workgroup walks the k-units from a different start (the assignment of units to lanes and steps is free): the
    // workgroups of a launch start together and would otherwise all ask the same L2 channel for the same lines of x at once
    const uint32_t rot = ROT ? (uint32_t)(((uint64_t)blockIdx.x * 2654435761u >> 8) % (P / 64)) * 64 : 0;
    uint32_t p = threadIdx.x;
    auto load = [&](uint4 (&w)[DEPTH], uint32_t (&ss)[DEPTH], f2 (&xx)[DEPTH], f2 (&gg)[DEPTH], uint32_t p0) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            uint32_t pd = min(p0 + d * stride, P - 1) + rot;
            pd = pd >= P ? pd - P : pd;
            if (XMODE == 0) xx[d] = x2[pd];
            if (XMODE == 1) xx[d] = f2{1.0f, 1.0f};
            if (XMODE == 2) xx[d] = ((const f2*)xl)[pd];
            if (NORM) gg[d] = GNT ? __builtin_nontemporal_load(g2 + pd) : g2[pd];
            ss[d] = s[pd];
            w[d] = wload<NT>(q + pd);
        }
    };
    f2 acc[8];
#pragma unroll
    for (int c = 0; c < 8; c++) acc[c] = f2{0.f, 0.f};
    float T = 0.f, SS = 0.f;
    auto compute = [&](const uint4 (&w)[DEPTH], const uint32_t (&ss)[DEPTH], const f2 (&xx)[DEPTH], const f2 (&gg)[DEPTH], uint32_t p0) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const bool ok = p0 + d * stride < P;
            const __half2 h = *(const __half2*)&ss[d];
            float xa = xx[d].x, xb = xx[d].y;
            if (NORM) {
                SS += ok ? xa * xa + xb * xb : 0.f;
                xa *= gg[d].x, xb *= gg[d].y;
            }
            // fp8 bytes are q * 2^-9: fold 512 into t (exact)
            const float t0 = ok ? (__low2float(h) * 512.0f) * xa : 0.f;
            const float t1 = ok ? (__high2float(h) * 512.0f) * xb : 0.f;
            T += t0;
            T += t1;
            const uint32_t dw[4] = {w[d].x, w[d].y, w[d].z, w[d].w};
#pragma unroll
            for (int dd = 0; dd < 4; dd++) {
                const float t = dd < 2 ? t0 : t1;
                const uint32_t lo = dw[dd] & 0x0F0F0F0Fu, hi = (dw[dd] >> 4) & 0x0F0F0F0Fu;
                const int base = 4 * (dd & 1);
                const f2 tt = f2{t, t};
                acc[base + 0] = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)lo, false), tt, acc[base + 0]);
                acc[base + 1] = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)lo, true), tt, acc[base + 1]);
                acc[base + 2] = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)hi, false), tt, acc[base + 2]);
                acc[base + 3] = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)hi, true), tt, acc[base + 3]);
            }
        }
    };
    load(wq, sv, xv, gv, p);
    STAMP(1);
    for (uint32_t gi = 1; gi < n_groups; gi++) {
        uint4 wn[DEPTH];
        uint32_t sn[DEPTH];
        f2 xn[DEPTH], gn[DEPTH];
        load(wn, sn, xn, gn, p + DEPTH * stride);
        compute(wq, sv, xv, gv, p);
#pragma unroll
        for (int d = 0; d < DEPTH; d++) wq[d] = wn[d], sv[d] = sn[d], xv[d] = xn[d], gv[d] = gn[d];
        p += DEPTH * stride;
    }
    compute(wq, sv, xv, gv, p);
    STAMP(2);
    // per lane: w = q - 8  ->  acc -= 8/512 * sum(t')   (mirrors the accumulation chain: all-zero weights give exactly 0)
    const f2 corr = f2{T, T} * (-8.0f / 512.0f);
#pragma unroll
    for (int c = 0; c < 8; c++) acc[c] += corr;
    // fold 64 lanes: column c = 2 * pair + half. After fold32 over (col 2m, 2m + 1) and fold16 over those, register m
    // (m = 0..3) holds in row r the column 4m + {0, 2, 1, 3}[r]
    float a8[8];
#pragma unroll
    for (int m = 0; m < 8; m++) a8[m] = fold32(acc[m].x, acc[m].y);
    float a4[4];
#pragma unroll
    for (int m = 0; m < 4; m++) a4[m] = row16_sum(fold16(a8[2 * m], a8[2 * m + 1]));
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane >> 4, i = lane & 15;
    const float v = (i & 2) ? ((i & 1) ? a4[3] : a4[2]) : ((i & 1) ? a4[1] : a4[0]);
    const uint32_t col = 4 * (i & 3) + ((r & 1) * 2 + (r >> 1));
    if (i < 4) red[w * 17 + col] = v;
    if (NORM) {
        SS = row16_sum(SS);
        SS = fold32(SS, SS);
        SS = fold16(SS, SS);
        if (lane == 0) red[w * 17 + 16] = SS;
    }
    __syncthreads();
    STAMP(3);
    if (threadIdx.x < 16) {
        float sum = 0.f, ssq = 0.f;
        for (uint32_t ww = 0; ww < n_waves; ww++) sum += red[ww * 17 + threadIdx.x];
        if (NORM) {
            for (uint32_t ww = 0; ww < n_waves; ww++) ssq += red[ww * 17 + 16];
            sum *= 1.0f / sqrtf(ssq / (float)K + 1e-5f);
        }
        y[g * 16 + threadIdx.x] = sum * post;
    }
    STAMP(4);
#undef STAMP
}

int main(int argc, char** argv) {
    const uint32_t K = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096;
    const uint64_t qs_bytes = (uint64_t)K * N / 2, sc_bytes = (uint64_t)K * N / 32 * 2;
    int ring = (int)std::min<uint64_t>(64, (700ull << 20) / (qs_bytes + sc_bytes));
    ring &= ~1;
    const bool chain = K == N;
    std::vector<uint4*> qs(ring);
    std::vector<uint32_t*> sc(ring);
    for (int i = 0; i < ring; i++) {
        CK(hipMalloc(&qs[i], qs_bytes));
        CK(hipMalloc(&sc[i], sc_bytes));
        fill_kernel<<<2048, 256>>>(qs[i], sc[i], K, N, i);
    }
    CK(hipDeviceSynchronize());
    std::vector<float> xh(K), gh(K);
    for (uint32_t i = 0; i < K; i++) xh[i] = ((int)(i % 17) - 8) * 0.03125f, gh[i] = 0.5f + (float)(i % 5) * 0.25f;
    float *v0, *v1, *gam;
    unsigned long long* stamps;
    const uint32_t VN = std::max(K, N);
    CK(hipMalloc(&v0, VN * 4));
    CK(hipMalloc(&v1, VN * 4));
    CK(hipMalloc(&gam, K * 4));
    CK(hipMalloc(&stamps, 16 * 8));
    CK(hipMemcpy(gam, gh.data(), K * 4, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    // correctness of every variant on matrix 0 against double-precision sums (plain and with the rmsnorm prologue)
    std::vector<double> ref(N, 0.0), mag(N, 0.0), refn(N, 0.0), magn(N, 0.0);
    double ss = 0;
    for (uint32_t k = 0; k < K; k++) ss += (double)xh[k] * xh[k];
    const double inv = 1.0 / sqrt(ss / K + 1e-5);
    for (uint32_t k = 0; k < K; k++)
        for (uint32_t n = 0; n < N; n++) {
            const double sc_ = (double)__half2float(__float2half(synth_s((uint64_t)k * (N / 32) + n / 32, 0)));
            const double wv = ((double)synth_q((uint64_t)k * N + n, 0) - 8.0) * sc_ * xh[k];
            ref[n] += wv, mag[n] += fabs(wv);
            refn[n] += wv * inv * gh[k], magn[n] += fabs(wv * inv * gh[k]);
        }
    const bool rotate = argc > 3 && atoi(argv[3]) != 0;
    const int xmode = argc > 4 ? atoi(argv[4]) : 0;
    struct Var { const char* name; int depth; bool nt, norm; };
    const Var vars[] = {{"D4 nt", 4, true, false}, {"D2 nt", 2, true, false}, {"D1 nt", 1, true, false}, {"D4 nt norm", 4, true, true}, {"D2 nt norm", 2, true, true},
                        {"D4", 4, false, false},  {"D1", 1, false, false}, {"D1 norm", 1, false, true}};
    auto launch = [&](const Var& v, uint32_t waves, int i, const float* xin, float* yout, float post, unsigned long long* st) {
        const dim3 grid(N / 16), blk(waves * 64);
#define L(D, NTV, NM) if (xmode == 3) kon_kernel<D, NTV, NM, false, 0, true><<<grid, blk, 0, s>>>(qs[i], sc[i], xin, yout, K, waves, post, gam, st); else if (xmode == 1) kon_kernel<D, NTV, NM, false, 1><<<grid, blk, 0, s>>>(qs[i], sc[i], xin, yout, K, waves, post, gam, st); else if (xmode == 2) kon_kernel<D, NTV, NM, false, 2><<<grid, blk, 0, s>>>(qs[i], sc[i], xin, yout, K, waves, post, gam, st); else if (rotate) kon_kernel<D, NTV, NM, true><<<grid, blk, 0, s>>>(qs[i], sc[i], xin, yout, K, waves, post, gam, st); else kon_kernel<D, NTV, NM, false><<<grid, blk, 0, s>>>(qs[i], sc[i], xin, yout, K, waves, post, gam, st)
        if (v.depth == 4 && v.nt && !v.norm) { L(4, true, false); }
        else if (v.depth == 2 && v.nt && !v.norm) { L(2, true, false); }
        else if (v.depth == 1 && v.nt && !v.norm) { L(1, true, false); }
        else if (v.depth == 4 && v.nt && v.norm) { L(4, true, true); }
        else if (v.depth == 2 && v.nt && v.norm) { L(2, true, true); }
        else if (v.depth == 4 && !v.nt && !v.norm) { L(4, false, false); }
        else if (v.depth == 1 && !v.nt && !v.norm) { L(1, false, false); }
        else { L(1, false, true); }
#undef L
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Var& v : vars)
        for (uint32_t waves : {8u}) {
            const uint32_t steps = (K / 2 + waves * 64 - 1) / (waves * 64);
            if (steps > 4u * v.depth || (v.depth > 1 && steps <= (uint32_t)v.depth / 2)) continue; // skip silly combinations
            CK(hipMemcpy(v0, xh.data(), K * 4, hipMemcpyHostToDevice));
            CK(hipMemset(stamps, 0, 128));
            launch(v, waves, 0, v0, v1, 1.0f, stamps);
            CK(hipStreamSynchronize(s));
            std::vector<float> yh(N);
            unsigned long long st[16];
            CK(hipMemcpy(yh.data(), v1, N * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(st, stamps, 128, hipMemcpyDeviceToHost));
            double worst = 0;
            for (uint32_t n = 0; n < N; n++) worst = std::max(worst, fabs(yh[n] - (v.norm ? refn[n] : ref[n])) / ((v.norm ? magn[n] : mag[n]) + 1e-30));
            hipGraph_t gr;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < ring; i++) {
                if (chain)
                    launch(v, waves, i, (i & 1) ? v1 : v0, (i & 1) ? v0 : v1, v.norm ? 1.0f : 0.2f, nullptr);
                else
                    launch(v, waves, i, v0, v1, 1.0f, nullptr);
            }
            CK(hipStreamEndCapture(s, &gr));
            CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
            for (int w = 0; w < 3; w++) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            const int reps = 30;
            for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1000.0 / (reps * ring), bytes = (double)K * N / 32 * 18 + 4.0 * K + 4.0 * N;
            printf("%ux%u rot%d xm%d %-10s waves=%2u %s  %.3f us/launch  %.0f GB/s  frac %.3f  rel.err %.2e  | wg0 cycles: issue %llu stream %llu fold %llu store %llu | last wg: %llu %llu %llu %llu (start +%lld)\n",
                   K, N, (int)rotate, xmode, v.name, waves, chain ? "chain" : "ring ", us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0, worst, st[1] - st[0], st[2] - st[1], st[3] - st[2],
                   st[4] - st[3], st[9] - st[8], st[10] - st[9], st[11] - st[10], st[12] - st[11], (long long)(st[8] - st[0]));
            fflush(stdout);
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(gr));
        }
    return 0;
}
