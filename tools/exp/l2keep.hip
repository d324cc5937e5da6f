// Does read-only data stay in an XCD's L2 across a dependent kernel boundary? (If it did, a launch's tail could pull the
// NEXT launch's first weight lines into L2.) Kernel `warm` has workgroup b read 4 KB at W + b * 64 KB; kernel `probe`
// (the next node of the same hipGraph, same grid, so workgroup b runs on the same XCD) times ONE load of that region per
// workgroup with s_memtime. Compared: probe right after warm (same lines), probe of lines nobody touched (cold), and
// probe after warm inside the SAME kernel (the L2-hit reference).
//   hipcc --offload-arch=gfx950 -O3 tools/exp/l2keep.hip -o tools/exp/l2keep && tools/exp/l2keep
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void warm(const uint4* __restrict__ w, uint4* sink) {
    const uint4 v = w[(size_t)blockIdx.x * 4096 + threadIdx.x]; // 256 threads x 16 B = 4 KB at b * 64 KB
    if (v.x == 0x12345678u) sink[0] = v;
}
__global__ void probe(const uint4* __restrict__ w, uint4* sink, unsigned long long* cycles, int same_kernel_warm) {
    if (same_kernel_warm) {
        const uint4 v = w[(size_t)blockIdx.x * 4096 + threadIdx.x];
        if (v.x == 0x12345678u) sink[1] = v;
        __syncthreads();
    }
    if (threadIdx.x < 64) { // one wave, a vector load (a uniform address would become a scalar load through the scalar cache)
        const unsigned long long t0 = __builtin_readcyclecounter();
        const unsigned v = __builtin_nontemporal_load((const unsigned*)&w[(size_t)blockIdx.x * 4096 + 64] + threadIdx.x); // lines of the warmed 4 KB
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the load has returned before the counter is read again
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (v == 0x12345678u) sink[2] = make_uint4(v, v, v, v);
        if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    }
}

int main() {
    const int WGS = 256;
    const size_t bytes = (size_t)WGS * 65536;
    uint4 *w = nullptr, *w2 = nullptr, *sink = nullptr;
    unsigned long long* cyc = nullptr;
    CK(hipMalloc(&w, bytes));
    CK(hipMalloc(&w2, bytes));
    CK(hipMalloc(&sink, 4096));
    CK(hipMalloc(&cyc, WGS * 8));
    CK(hipMemset(w, 1, bytes));
    CK(hipMemset(w2, 1, bytes));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    auto run = [&](const char* what, const uint4* warm_buf, const uint4* probe_buf, int same) -> int {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(warm, dim3(WGS), dim3(256), 0, s, warm_buf, sink);
        hipLaunchKernelGGL(probe, dim3(WGS), dim3(256), 0, s, probe_buf, sink, cyc, same);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        std::vector<unsigned long long> all;
        for (int rep = 0; rep < 20; rep++) {
            CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            std::vector<unsigned long long> h(WGS);
            CK(hipMemcpy(h.data(), cyc, WGS * 8, hipMemcpyDeviceToHost));
            if (rep >= 4) all.insert(all.end(), h.begin(), h.end());
        }
        std::sort(all.begin(), all.end());
        printf("%-58s load latency: median %5llu  p10 %5llu  p90 %5llu counter ticks\n", what, all[all.size() / 2], all[all.size() / 10], all[all.size() * 9 / 10]);
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
        return 0;
    };
    { // reference: lines never touched before (a fresh 16 MB window per repetition, eager launches)
        uint4* big = nullptr;
        const int reps = 24;
        CK(hipMalloc(&big, bytes * reps));
        CK(hipMemset(big, 1, bytes * reps));
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> all;
        for (int rep = 0; rep < reps; rep++) {
            hipLaunchKernelGGL(warm, dim3(WGS), dim3(256), 0, s, w2, sink);
            hipLaunchKernelGGL(probe, dim3(WGS), dim3(256), 0, s, big + (size_t)rep * (bytes / 16), sink, cyc, 0);
            CK(hipStreamSynchronize(s));
            std::vector<unsigned long long> h(WGS);
            CK(hipMemcpy(h.data(), cyc, WGS * 8, hipMemcpyDeviceToHost));
            all.insert(all.end(), h.begin(), h.end());
        }
        std::sort(all.begin(), all.end());
        printf("%-58s load latency: median %5llu  p10 %5llu  p90 %5llu counter ticks\n", "lines never touched since their memset (HBM / MALL)", all[all.size() / 2],
               all[all.size() / 10], all[all.size() * 9 / 10]);
        hipFree(big);
    }
    if (run("same lines, warmed by the PREVIOUS kernel", w, w, 0)) return 1;
    if (run("lines nobody touched (previous kernel warmed another buffer)", w2, w, 0)) return 1;
    if (run("same lines, warmed earlier in the SAME kernel (L2 / L1 hit)", w2, w, 1)) return 1;
    return 0;
}
