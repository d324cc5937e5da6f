// Kernels for tools/exp/aql.cpp (compiled to a raw code object, see the build line there).
#include <hip/hip_runtime.h>
extern "C" __global__ void empty_kernel() {}
// one dependent step: every thread bumps its word (reads what the previous dispatch wrote).
// 256 = the workgroup size aql.cpp dispatches (blockDim would need the hidden kernel arguments).
extern "C" __global__ void bump_kernel(unsigned* p) { p[blockIdx.x * 256u + threadIdx.x] += 1u; }
// the same with agent-scope (sc1) accesses: what a kernel would have to do if the packets carried no fences
extern "C" __global__ void bump_sc1_kernel(unsigned* p) {
    using gu32 = __attribute__((address_space(1))) unsigned;
    gu32* w = (gu32*)p + blockIdx.x * 256u + threadIdx.x;
    const unsigned v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w, v + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every workgroup reads a word another workgroup (another XCD) wrote in the previous dispatch: the cross-XCD visibility
// a fence-less boundary has to provide. p[0..65535] -> q[...] rotated by 257 words, +1
extern "C" __global__ void rotate_kernel(const unsigned* p, unsigned* q) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    q[i] = p[(i + 257u) & 65535u] + 1u;
}
extern "C" __global__ void rotate_sc1_kernel(const unsigned* p, unsigned* q) {
    using gu32 = __attribute__((address_space(1))) unsigned;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    const unsigned v = __hip_atomic_load((const gu32*)p + ((i + 257u) & 65535u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((gu32*)q + i, v + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
