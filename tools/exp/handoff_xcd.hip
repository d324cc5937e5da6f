// Experiment (round 4): the price of one all-to-all hand-off of a SHORT vector (576 .. 4096 words) among the
// workgroups of ONE XCD (32 CUs, one workgroup each) inside a persistent launch — the per-stage cost of a decoder layer kept
// in one launch on one XCD — against the same hand-off over 64 / 256 workgroups on all XCDs (round 2: tools/exp/handoff.hip,
// 2.9-4.4 us per stage at 4096 words x 256 workgroups).
//
// 256 workgroups are launched (one per CU: 96 KB of LDS each); a workgroup reads its XCC id (s_getreg HW_REG_XCC_ID), and in
// the one-XCD modes only those on XCC `target` take a ticket; tickets 0 .. P-1 play, everybody else exits at once. Per stage a
// participant gathers all N words of the previous stage, derives its N / P words from a checksum over ALL of them (a stale or
// torn word changes every later stage; the host recomputes the chain) and publishes them.
//   T0  8-byte {value, tag = stage + 1} granules, agent-scope (sc1) stores and loads: placement-independent   [guide G16 R2]
//   T3  the same granules with PLAIN stores and sc1 (L1-bypassing) loads: the bytes meet in the XCD's L2. Only meaningful when
//       every participant is on one XCD (checked by construction here); NOT a legal hand-off across XCDs.
// SW sweep waves share the gather. Every spin is bounded (a time-out is reported, never a hang).
//   hipcc -O3 --offload-arch=gfx950 tools/exp/handoff_xcd.hip -o tools/exp/handoff_xcd && tools/exp/handoff_xcd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

using gu32 = __attribute__((address_space(1))) unsigned int;
using gu64 = __attribute__((address_space(1))) unsigned long long;
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int MAXN = 4096;

struct Args {
    unsigned long long* box8; // [4][MAXN] granules
    uint32_t* tickets;        // [8 XCCs] one ticket counter per XCC (zeroed per run)
    uint32_t* census;         // [8] workgroups seen per XCC (diagnostic)
    uint32_t* result;         // [256] final checksum per participant rank
    uint32_t* timeout;
    unsigned long long* ticks; // [2]
    int stages, N, P, one_xcd, target;
};

__device__ __forceinline__ uint32_t mix(uint32_t c, uint32_t n) { return (c ^ (n * 2654435761u)) * 2246822519u + 0x9E3779B9u; }
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xF;
}

template <int T, int SW>
__global__ void __launch_bounds__(256) chain(Args a) {
    extern __shared__ uint32_t xs[]; // [MAXN] + padding to force one workgroup per CU
    __shared__ volatile uint32_t bar, dead_any, rank_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) {
        bar = 0, dead_any = 0;
        const uint32_t x = xcc_id();
        atomicAdd(a.census + x, 1u);
        uint32_t r = 0xFFFFFFFFu;
        if (a.one_xcd) {
            if (x == (uint32_t)a.target) r = atomicAdd(a.tickets + x, 1u);
        } else {
            r = atomicAdd(a.tickets, 1u);
        }
        rank_s = r;
    }
    __syncthreads();
    const uint32_t rank = rank_s;
    if (rank >= (uint32_t)a.P || w >= (uint32_t)SW) return;
    const int N = a.N, PER = N / a.P;
    uint32_t bar_target = 0;
    auto sweep_sync = [&]() {
        if (SW == 1) return;
        bar_target += SW;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (lane == 0) atomicAdd((uint32_t*)&bar, 1u);
        while (bar < bar_target) __builtin_amdgcn_s_sleep(1);
    };
    bool dead = false;
    uint32_t csum = 0;
    if (rank == 0 && tid == 0) a.ticks[0] = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < a.stages; s++) {
        const uint32_t epoch = (uint32_t)s + 1;
        if (s > 0) {
            const int words = N / SW, base = w * words;
            const gu64* g = (const gu64*)(a.box8 + (uint64_t)((s - 1) & 3) * MAXN) + base;
            uint32_t spins = 0;
            // chunks of 8 granules per lane: all 8 loads of a chunk are issued back to back (no data-dependent branch between them),
            // a chunk is re-polled until every lane's tags match
            for (int k0 = 0; k0 < words && !dead; k0 += 8 * 64) {
                for (;;) {
                    unsigned long long x[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int k = k0 + j * 64 + (int)lane;
                        x[j] = __hip_atomic_load(g + (k < words ? k : words - 1), RLX); // sc1: never this CU's L1
                    }
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < 8; j++) ok = ok && (uint32_t)(x[j] >> 32) == (uint32_t)s;
                    if (__all(ok)) {
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const int k = k0 + j * 64 + (int)lane;
                            if (k < words) xs[base + k] = (uint32_t)x[j];
                        }
                        break;
                    }
                    if (++spins > 200000u) {
                        if (lane == 0) *a.timeout = 1 + s;
                        dead = true, dead_any = 1;
                        break;
                    }
                }
            }
            sweep_sync();
            uint32_t c = 0;
            for (int k = lane; k < N; k += 64) c += xs[k] * (uint32_t)(k % 7 + 1);
            for (int off = 32; off; off >>= 1) c += __shfl_xor(c, off, 64);
            csum = c;
            sweep_sync();
            if (dead || dead_any) break;
        }
        if (w == 0) {
            for (int i = lane; i < PER; i += 64) {
                const uint32_t n = rank * PER + i;
                const uint32_t val = mix(csum + (uint32_t)s, n) & 0x7FFFFFFFu;
                const unsigned long long gnl = ((unsigned long long)epoch << 32) | val;
                if (T == 0)
                    __hip_atomic_store((gu64*)(a.box8 + (uint64_t)(s & 3) * MAXN + n), gnl, RLX);
                else
                    *(gu64*)(a.box8 + (uint64_t)(s & 3) * MAXN + n) = gnl; // plain: write-through to THIS XCD's L2 only
            }
        }
    }
    if (tid == 0) {
        a.result[rank] = csum;
        if (rank == 0) a.ticks[1] = __builtin_amdgcn_s_memrealtime();
    }
}

static uint32_t host_chain(int stages, int N) {
    std::vector<uint32_t> x(N);
    uint32_t csum = 0;
    for (int s = 0; s < stages; s++) {
        if (s > 0) {
            uint32_t c = 0;
            for (int k = 0; k < N; k++) c += x[k] * (uint32_t)(k % 7 + 1);
            csum = c;
        }
        for (int n = 0; n < N; n++) x[n] = (((csum + (uint32_t)s) ^ ((uint32_t)n * 2654435761u)) * 2246822519u + 0x9E3779B9u) & 0x7FFFFFFFu;
    }
    return csum;
}

template <int T, int SW>
int run(const char* name, Args a) {
    const uint32_t want = host_chain(a.stages, a.N);
    CK(hipMemset(a.box8, 0, 4ull * MAXN * 8));
    CK(hipMemset(a.tickets, 0, 8 * 4));
    CK(hipMemset(a.census, 0, 8 * 4));
    CK(hipMemset(a.timeout, 0, 4));
    CK(hipMemset(a.result, 0, 256 * 4));
    CK(hipFuncSetAttribute((const void*)chain<T, SW>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    chain<T, SW><<<256, 256, 96 * 1024>>>(a);
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    uint32_t res[256], to, cen[8];
    unsigned long long tk[2];
    CK(hipMemcpy(res, a.result, sizeof(res), hipMemcpyDeviceToHost));
    CK(hipMemcpy(&to, a.timeout, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cen, a.census, 32, hipMemcpyDeviceToHost));
    CK(hipMemcpy(tk, a.ticks, 16, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < a.P; i++) bad += res[i] != want;
    printf("%-30s N=%4d P=%3d SW=%d  %7.3f us/stage (in-kernel %7.3f)  wrong=%d timeout=%u  census %u %u %u %u %u %u %u %u\n", name, a.N, a.P, SW,
           ms * 1e3 / a.stages, (tk[1] - tk[0]) * 0.01 / a.stages, bad, to, cen[0], cen[1], cen[2], cen[3], cen[4], cen[5], cen[6], cen[7]);
    return 0;
}

int main() {
    Args a{};
    a.stages = 400;
    CK(hipMalloc(&a.box8, 4ull * MAXN * 8));
    CK(hipMalloc(&a.tickets, 32));
    CK(hipMalloc(&a.census, 32));
    CK(hipMalloc(&a.result, 256 * 4));
    CK(hipMalloc(&a.timeout, 4));
    CK(hipMalloc(&a.ticks, 16));
    for (int rep = 0; rep < 2; rep++) {
        for (int N : {576, 1536, 4096}) {
            a.N = N;
            a.one_xcd = 1, a.target = 0, a.P = 32;
            if (N % 32) continue;
            run<0, 1>("one XCD, T0 sc1 granules", a);
            run<0, 2>("one XCD, T0 sc1 granules", a);
            run<3, 1>("one XCD, T3 plain st + sc1 ld", a);
            run<3, 2>("one XCD, T3 plain st + sc1 ld", a);
            a.P = 16;
            run<0, 1>("one XCD, T0 sc1 granules", a);
            run<3, 1>("one XCD, T3 plain st + sc1 ld", a);
            a.one_xcd = 0;
            a.P = 32;
            run<0, 1>("any XCD, T0 sc1 granules", a);
            a.P = 64;
            run<0, 1>("any XCD, T0 sc1 granules", a);
            run<0, 2>("any XCD, T0 sc1 granules", a);
            if (N % 256 == 0) {
                a.P = 256;
                run<0, 1>("any XCD, T0 sc1 granules", a);
                run<0, 4>("any XCD, T0 sc1 granules", a);
            }
        }
    }
    return 0;
}
