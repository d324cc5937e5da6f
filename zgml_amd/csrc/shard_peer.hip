// shard_peer.hip — the row-shard path's all-gather as PEER STORES over xGMI (ZGML_SHARD_GATHER=peer).
//
// The Llama-2-7B row shard has 4 gather points per layer + the logits: 2-44 KB per point, i.e. pure latency. A stock
// ncclAllGather per point costs a collective launch (proxy hand-shake, several kernels' worth of flags) — SURVEY section 7
// predicts 128 of them per token outweigh the single-GPU token. Here a gather point is ONE small kernel per rank:
//
//   push   the rank's own slice (already in its program buffer) goes to slot `rank` of the point's STAGING area in every peer's
//          fine-grained block (plain stores to peer memory over xGMI), then __threadfence_system(), then one system-scope add on
//          the point's arrival counter in every peer's block;
//   wait   one lane polls the rank's OWN counter until it has seen world - 1 new arrivals (bounded by wall clock: a give-up sets
//          the context's host-visible hand-off word, the step is reported failed — never a hang);
//   land   the peers' slices are copied from the own staging area (system-scope loads: fine-grained memory, never a stale line)
//          into the program buffer, where the next op range reads them as if an all-gather had run.
//
// Staging is what makes this placement- and cache-safe without touching any consumer kernel: peers never write a program
// buffer (coarse-grained hipMalloc memory whose lines this device's L2 may hold), only the fine-grained block.
// Flow control needs no extra message: a rank pushes point i of step t + 1 only after its wait at point i - 1 of step t + 1
// returned, i.e. after every peer pushed that point, which a peer does only after it landed point i of step t (>= 2 points per
// step; attach checks it). Counters are monotonic; each rank keeps its private `seen` word per point.
//
// The greedy token of the N-sharded LM head is gathered as one (max, index) pair per rank instead of 4 x vocab / world bytes of
// logits: first maximum wins (src/nn.zig:122-138) — inside a slice the lowest index, across slices the
// lowest rank, whose indices are lower.
#include "kernels.h"

#include <hip/hip_runtime.h>
#include <math.h>

namespace zgml {
namespace {

using gu32 = __attribute__((address_space(1))) unsigned int;
using gu64 = __attribute__((address_space(1))) unsigned long long;
#define ZGML_SYS __HIP_MEMORY_SCOPE_SYSTEM

constexpr int kPeerBlock = 512;

// The value a peer wait leaves in the context's host-visible hand-off word when it gives up (1 = an in-launch hand-off of a fused
// q / k / v + attention launch: runtime_internal.h handoff_ok tells the two apart — a slow peer must not switch that fusion off)
constexpr uint32_t kHandoffPeer = 2;

// bounded wait of ONE lane for `cnt` to reach `target` (wrap-safe); returns false after `ticks` of the 100 MHz wall clock, or at once
// when the block's POISON word is set: a rank whose wait gave up poisons EVERY rank's block (peer_poison), because its later
// pushes of that step carry slices computed from data that never landed — without the poison a peer that is merely late would find
// all its arrivals present and return a token built from them with no error. The poison is sticky: after any rank
// returned -1 all ranks re-attach (fresh blocks) before they step again.
__device__ __forceinline__ bool peer_wait(const uint32_t* cnt, const uint32_t* poison, uint32_t target, uint64_t ticks, const uint32_t* timeout) {
    // a wait of this step has already given up (the word is host-visible and cleared by the host when it reports the failure): the
    // step is lost, do not wait the bound again at each of its remaining ~130 points
    if (__hip_atomic_load((const gu32*)timeout, __ATOMIC_RELAXED, ZGML_SYS) != 0) return false;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (__hip_atomic_load((const gu32*)poison, __ATOMIC_RELAXED, ZGML_SYS) != 0) return false;
        if ((int32_t)(__hip_atomic_load((const gu32*)cnt, __ATOMIC_RELAXED, ZGML_SYS) - target) >= 0) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > ticks) return false;
        __builtin_amdgcn_s_sleep(8);
    }
}
// one lane per rank: the poison word of every rank's block (own included)
__device__ __forceinline__ void peer_poison(char* const* blocks, uint64_t poison_off, uint32_t world) {
    if (threadIdx.x < world) __hip_atomic_store((gu32*)(blocks[threadIdx.x] + poison_off), 1u, __ATOMIC_RELAXED, ZGML_SYS);
}

__global__ void __launch_bounds__(kPeerBlock) peer_gather_kernel(PeerGatherArgs a) {
    __shared__ uint32_t ok_s;
    const uint32_t tid = threadIdx.x, n2 = a.len / 2; // 8-byte units (attach checks len % 2 == 0 and 8-byte alignment)
    const unsigned long long* const mine = (const unsigned long long*)(a.local + (uint64_t)a.rank * a.len);
    // ---- push: own slice -> slot `rank` of every peer's staging area
    for (uint32_t q = 0; q < a.world; q++) {
        if (q == a.rank) continue;
        unsigned long long* const dst = (unsigned long long*)(a.blocks[q] + a.stage_off) + (uint64_t)a.rank * n2;
        for (uint32_t i = tid; i < n2; i += kPeerBlock) dst[i] = mine[i];
    }
    __threadfence_system(); // every storing thread: its peer stores are visible system-wide before the arrival is counted
    __syncthreads();
    if (tid < a.world && tid != a.rank) __hip_atomic_fetch_add((gu32*)(a.blocks[tid] + a.ctr_off), 1u, __ATOMIC_RELEASE, ZGML_SYS);
    // ---- wait: world - 1 new arrivals on the own counter
    if (tid == 0) {
        const uint32_t target = *a.seen + (a.world - 1);
        const bool ok = peer_wait((const uint32_t*)(a.blocks[a.rank] + a.ctr_off), (const uint32_t*)(a.blocks[a.rank] + a.poison_off), target, a.wait_ticks, a.timeout);
        if (!ok) __hip_atomic_store((gu32*)a.timeout, kHandoffPeer, __ATOMIC_RELAXED, ZGML_SYS);
        *a.seen = target; // (also after a give-up: the step is reported failed, the counters stay in step)
        ok_s = ok ? 1u : 0u;
    }
    __syncthreads();
    __threadfence_system(); // acquire side: the staging loads below are issued after the count was observed
    if (!ok_s) {
        peer_poison(a.blocks, a.poison_off, a.world); // every rank fails this step (and stays failed until all re-attach)
        return;
    }
    // ---- land: the peers' slices, staging -> program buffer
    const unsigned long long* const stage = (const unsigned long long*)(a.blocks[a.rank] + a.stage_off);
    unsigned long long* const full = (unsigned long long*)a.local;
    for (uint32_t q = 0; q < a.world; q++) {
        if (q == a.rank) continue;
        for (uint32_t i = tid; i < n2; i += kPeerBlock)
            full[(uint64_t)q * n2 + i] = __hip_atomic_load((const gu64*)(stage + (uint64_t)q * n2 + i), __ATOMIC_RELAXED, ZGML_SYS);
    }
}

// first maximum of v[0, n): (value, index), ties -> lowest index; NaN never wins (`>` is false), all -inf / NaN -> index 0
__device__ __forceinline__ void block_first_max(const float* v, uint32_t n, float& best, uint32_t& best_i) {
    __shared__ float sv[kPeerBlock];
    __shared__ uint32_t si[kPeerBlock];
    float b = -INFINITY;
    uint32_t bi = 0;
    for (uint32_t i = threadIdx.x; i < n; i += kPeerBlock) { // ascending i per thread: strict > keeps the first
        const float x = v[i];
        if (x > b) b = x, bi = i;
    }
    sv[threadIdx.x] = b, si[threadIdx.x] = bi;
    __syncthreads();
    for (uint32_t s = kPeerBlock / 2; s; s >>= 1) {
        if (threadIdx.x < s) {
            const float o = sv[threadIdx.x + s];
            const uint32_t oi = si[threadIdx.x + s];
            if (o > sv[threadIdx.x] || (o == sv[threadIdx.x] && oi < si[threadIdx.x])) sv[threadIdx.x] = o, si[threadIdx.x] = oi;
        }
        __syncthreads();
    }
    best = sv[0], best_i = si[0];
}

__device__ __forceinline__ unsigned long long pack_pair(float v, uint32_t idx) { return ((unsigned long long)idx << 32) | __float_as_uint(v); }

// ranks in order, strict >: the first maximum over the whole vocabulary
__device__ __forceinline__ int64_t reduce_pairs(const unsigned long long* pairs, uint32_t world, bool sys) {
    float best = -INFINITY;
    uint32_t best_i = 0;
    for (uint32_t q = 0; q < world; q++) {
        const unsigned long long pr = sys ? __hip_atomic_load((const gu64*)(pairs + q), __ATOMIC_RELAXED, ZGML_SYS) : pairs[q];
        const float v = __uint_as_float((uint32_t)pr);
        if (v > best) best = v, best_i = (uint32_t)(pr >> 32);
    }
    return (int64_t)best_i;
}

// peer mode: local (max, index) of the rank's logits slice -> every peer's pair slot -> wait -> the token
__global__ void __launch_bounds__(kPeerBlock) peer_argmax_kernel(PeerArgmaxArgs a) {
    float best;
    uint32_t best_i;
    block_first_max(a.slice, a.len, best, best_i);
    if (threadIdx.x < a.world) { // lane q: the pair into rank q's block (own included), then the arrival
        const unsigned long long pr = pack_pair(best, a.rank * a.len + best_i);
        __hip_atomic_store((gu64*)(a.blocks[threadIdx.x] + a.pairs_off) + a.rank, pr, __ATOMIC_RELAXED, ZGML_SYS);
        __threadfence_system();
        if (threadIdx.x != a.rank) __hip_atomic_fetch_add((gu32*)(a.blocks[threadIdx.x] + a.ctr_off), 1u, __ATOMIC_RELEASE, ZGML_SYS);
    }
    __syncthreads();
    __shared__ uint32_t ok_a;
    if (threadIdx.x == 0) {
        const uint32_t target = *a.seen + (a.world - 1);
        const bool ok = peer_wait((const uint32_t*)(a.blocks[a.rank] + a.ctr_off), (const uint32_t*)(a.blocks[a.rank] + a.poison_off), target, a.wait_ticks, a.timeout);
        if (!ok) __hip_atomic_store((gu32*)a.timeout, kHandoffPeer, __ATOMIC_RELAXED, ZGML_SYS);
        *a.seen = target;
        __threadfence_system();
        *a.out = ok ? reduce_pairs((const unsigned long long*)(a.blocks[a.rank] + a.pairs_off), a.world, true) : (int64_t)-1;
        ok_a = ok ? 1u : 0u;
    }
    __syncthreads();
    if (!ok_a) peer_poison(a.blocks, a.poison_off, a.world);
}

// collective (RCCL) mode: the local pair into slot `rank` of a device array that is then all-gathered, and the reduction behind it
__global__ void __launch_bounds__(kPeerBlock) local_argmax_pair_kernel(const float* slice, uint32_t len, uint32_t rank, unsigned long long* pairs) {
    float best;
    uint32_t best_i;
    block_first_max(slice, len, best, best_i);
    if (threadIdx.x == 0) pairs[rank] = pack_pair(best, rank * len + best_i);
}
__global__ void reduce_pairs_kernel(const unsigned long long* pairs, uint32_t world, int64_t* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = reduce_pairs(pairs, world, false);
}

} // namespace

void launch_peer_gather(hipStream_t s, const PeerGatherArgs& a) { hipLaunchKernelGGL(peer_gather_kernel, dim3(1), dim3(kPeerBlock), 0, s, a); }
void launch_peer_argmax(hipStream_t s, const PeerArgmaxArgs& a) { hipLaunchKernelGGL(peer_argmax_kernel, dim3(1), dim3(kPeerBlock), 0, s, a); }
void launch_local_argmax_pair(hipStream_t s, const float* slice, uint32_t len, uint32_t rank, unsigned long long* pairs) {
    hipLaunchKernelGGL(local_argmax_pair_kernel, dim3(1), dim3(kPeerBlock), 0, s, slice, len, rank, pairs);
}
void launch_reduce_pairs(hipStream_t s, const unsigned long long* pairs, uint32_t world, int64_t* out) {
    hipLaunchKernelGGL(reduce_pairs_kernel, dim3(1), dim3(64), 0, s, pairs, world, out);
}

} // namespace zgml
