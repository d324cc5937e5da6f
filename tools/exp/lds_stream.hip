// lds_stream.hip — how fast does a CU stream HBM when its loads land in LDS directly (global_load_lds_dwordx4, gfx950) instead of
// in registers? The question behind the loader-wave / consumer-wave form of the dense f16 M = 32 kernel: three
// register-landing structures stream 4.9-5.1 TB/s, the sink of the same structure 5.6, a plain read-only stream 6.5-7.0. If loader
// waves that ONLY issue LDS-direct loads do not beat the register form at the kernel's geometry (172-688 workgroups, one contiguous
// region per workgroup), the loader / consumer split cannot either.
//   variant 0: registers, DEPTH x 16 B per lane in flight, xor sink            variant 1: LDS-direct, ring of DEPTH KiB per wave,
//   nothing reads the LDS (pure loader)       variant 2: LDS-direct + one consumer wave per loader wave reading the ring (ds_read_b128 + xor)
// build: hipcc --offload-arch=gfx950 -O3 -o lds_stream tools/exp/lds_stream.hip ; run: ./lds_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ void __launch_bounds__(512) stream_regs(const u4v* __restrict__ src, uint64_t items_per_wg, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const u4v* p = src + (uint64_t)blockIdx.x * items_per_wg + lane;
    const uint64_t steps = items_per_wg / 64; // 1 KiB steps of the workgroup's region; wave w takes w, w + nw, ...
    u4v ring[DEPTH];
    u4v acc = {0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        const uint64_t s = w + (uint64_t)d * nw;
        ring[d] = __builtin_nontemporal_load(p + (s < steps ? s : steps - 1) * 64);
    }
    for (uint64_t s0 = w; s0 < steps; s0 += (uint64_t)DEPTH * nw) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const u4v cur = ring[d];
            const uint64_t s = s0 + (uint64_t)(d + DEPTH) * nw;
            ring[d] = __builtin_nontemporal_load(p + (s < steps ? s : steps - 1) * 64);
            acc ^= cur;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x] = 1; // (keeps the loads alive)
}

// LDS-direct: every wave owns DEPTH KiB of LDS; a load instruction lands 64 lanes x 16 B = 1 KiB at M0 + instruction offset.
// CONSUME: wave pairs — even waves load, odd waves read the even wave's ring behind a workgroup barrier per round.
template <int DEPTH, bool CONSUME>
__global__ void __launch_bounds__(512) stream_lds(const u4v* __restrict__ src, uint64_t items_per_wg, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) u4v lds[];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint64_t steps = items_per_wg / 64;
    const u4v* p = src + (uint64_t)blockIdx.x * items_per_wg + lane;
    u4v acc = {0, 0, 0, 0};
    if (!CONSUME) {
        u4v* const ring = lds + (uint64_t)w * DEPTH * 64; // this wave's ring
        for (uint64_t s0 = w; s0 < steps; s0 += (uint64_t)DEPTH * nw) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                const uint64_t s = s0 + (uint64_t)d * nw;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (s < steps ? s : steps - 1) * 64),
                                                 (__attribute__((address_space(3))) void*)(ring + d * 64), 16, 0, 0);
            }
            // the ring is reused every round: wait until the previous round's loads have landed (all but this round's DEPTH)
            __builtin_amdgcn_s_waitcnt(0x0F70 | (DEPTH & 15) | ((DEPTH >> 4) << 14)); // vmcnt(DEPTH), others don't-care
        }
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
        acc = lds[threadIdx.x];
    } else {
        const uint32_t lw = w >> 1, nl = nw >> 1; // loader index, loaders
        u4v* const ring = lds + (uint64_t)lw * 2 * DEPTH * 64; // two half-rings per pair: one being filled, one being read
        const bool loader = (w & 1) == 0;
        uint32_t half = 0;
        if (loader) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                const uint64_t s = lw + (uint64_t)d * nl;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (s < steps ? s : steps - 1) * 64),
                                                 (__attribute__((address_space(3))) void*)(ring + d * 64), 16, 0, 0);
            }
        }
        for (uint64_t s0 = lw; s0 < steps; s0 += (uint64_t)DEPTH * nl) {
            if (loader) { // next round into the other half, then wait for THIS round's loads
#pragma unroll
                for (int d = 0; d < DEPTH; d++) {
                    const uint64_t s = s0 + (uint64_t)(d + DEPTH) * nl;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (s < steps ? s : steps - 1) * 64),
                                                     (__attribute__((address_space(3))) void*)(ring + ((half ^ 1) * DEPTH + d) * 64), 16, 0, 0);
                }
                __builtin_amdgcn_s_waitcnt(0x0F70 | (DEPTH & 15) | ((DEPTH >> 4) << 14));
            }
            __syncthreads(); // the half is complete
            if (!loader) {
#pragma unroll
                for (int d = 0; d < DEPTH; d++) acc ^= ring[(half * DEPTH + d) * 64 + lane];
            }
            __syncthreads(); // the half may be refilled
            half ^= 1;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x] = 1;
}

template <typename F>
static double time_it(F launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main() {
    const uint64_t bytes = 1ull << 30;
    u4v* src;
    uint32_t* out;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&out, 1 << 16));
    CK(hipMemset(src, 1, bytes));
    CK(hipMemset(out, 0, 1 << 16));
    const int wgs[] = {172, 256, 344, 688, 1376, 2048};
    printf("1 GiB read-only, one contiguous region per workgroup, 512 threads; GB/s\n%8s %14s %14s %14s %14s %14s\n", "wgs", "regs d4", "regs d8", "lds d8", "lds d16", "lds d8+consume");
    for (int g : wgs) {
        const uint64_t items = bytes / 16 / g / 64 * 64;
        const double tot = (double)items * 16 * g;
        auto gb = [&](double us) { return tot / us / 1e3; };
        const double a = time_it([&] { stream_regs<4><<<g, 512>>>(src, items, out); }, 10);
        const double b = time_it([&] { stream_regs<8><<<g, 512>>>(src, items, out); }, 10);
        const double c = time_it([&] { stream_lds<8, false><<<g, 512, 8 * 8 * 1024>>>(src, items, out); }, 10);
        CK(hipFuncSetAttribute((const void*)stream_lds<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16 * 1024));
        const double d = time_it([&] { stream_lds<16, false><<<g, 512, 8 * 16 * 1024>>>(src, items, out); }, 10);
        const double e = time_it([&] { stream_lds<8, true><<<g, 512, 4 * 2 * 8 * 1024>>>(src, items, out); }, 10);
        printf("%8d %14.1f %14.1f %14.1f %14.1f %14.1f\n", g, gb(a), gb(b), gb(c), gb(d), gb(e));
    }
    CK(hipGetLastError());
    return 0;
}
