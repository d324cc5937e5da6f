// Prefill attention (seq_q >= 16, dense layouts, d_head 64 / 128) on the matrix cores, all in f32.
//
// Reference semantics: src/backend/reference.zig `attention` (score = q.k * scale + mask, softmax over the keys,
// weighted sum of V). attention_rows_kernel spends a workgroup per (query, head): the 32-128 queries of
// a chunk each re-read their head's K / V and pay ~30 VALU instructions per (key, query). Here a wave owns a tile
// of 16 queries and walks over 16-key tiles with `v_mfma_f32_16x16x4_f32` (f32 operands, f32 products, f32 sums: no
// rounding of q, k, p or v that the reference does not do):
//
//   S^T = K Q^T   M = keys, N = queries, k = head dims.  A: lane (key = l & 15, k-slot g = l >> 4), B: lane (query, g).
//                 The k-slot/step -> dim map is free as long as A and B agree: step c = 4 i + j reads dim 16 i + 4 g + j,
//                 so a lane's operands are float4 loads of its key's (query's) row. Result: lane (query = l & 15, g)
//                 holds the scores of keys 4 g + v, v = 0..3 — queries sit on lanes, so the softmax state (m, l) is per lane.
//   O^T = V^T P^T M = dims, N = queries, k = keys.  B: lane (query, g) supplies P[key 4 g + v] for step v — exactly the
//                 register the score tile left there, no transpose. A: lane (row m = l & 15 of dim tile t, g) reads
//                 V[key 4 g + v][dim(t, m)], dim(t, m) = 64 (t >> 2) + 4 m + (t & 3): one float4 load serves 4 dim tiles.
//                 Result: lane (query, g'), register v' of tile t = dim 64 (t >> 2) + 16 g' + 4 v' + (t & 3).
//
// The waves of a workgroup split the key tiles of one (head, query tile) round-robin and merge their (m, l, O) through
// LDS in wave order (deterministic). A key tile whose 16 x 16 mask block is entirely -inf (the causal upper triangle) or
// past seq_kv is skipped before its MFMAs.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "kernels.h"

namespace zgml {
namespace {

__device__ __forceinline__ float tile_score(float dot, float mk, float scale) { // kernels_generic.hip: score_of
    float score = -INFINITY;
    if (isfinite(mk)) {
        score = dot * scale + mk;
        if (!isfinite(score)) score = -INFINITY;
    }
    return score;
}

// all-reduce over the 4 DPP rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48) with gfx950's row swaps: no LDS round
// trip, so nothing here waits on lgkmcnt. permlane16_swap(x, x) = (rows 0 0 2 2 | rows 1 1 3 3), permlane32_swap pairs the halves.
__device__ __forceinline__ float rows_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float v) { // (r0 + r1) + (r2 + r3) on every lane: same order everywhere
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

constexpr int kTileWavesMax = 8;

template <int DH>
__global__ void __launch_bounds__(kTileWavesMax * 64) attention_tiles_kernel(const AttentionParams* __restrict__ params, const float* __restrict__ zero_word,
                                                                             AttnPieceSink sink) {
    constexpr int NI = DH / 16;  // float4 loads per K / Q row slice of a lane = MFMA steps / 4 of the score tile
    constexpr int NH = DH / 64;  // float4 loads per V row slice of a lane; 4 dim tiles each
    constexpr int NT = DH / 16;  // output dim tiles
    extern __shared__ float smem[];
    const AttentionParams& p = params[blockIdx.y];
    const uint32_t q0 = blockIdx.x * 16;
    if (q0 >= p.seq_q) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t r16 = lane & 15, g = lane >> 4;
    const uint32_t seq_kv = *p.dyn_seq_kv;
    const uint32_t n_tiles = (seq_kv + 15) / 16;
    const uint32_t max_w = blockDim.x >> 6;
    const uint32_t NW = n_tiles < 1 ? 1 : (n_tiles > max_w ? max_w : n_tiles);
    if (w >= NW) return;
    const uint32_t last = seq_kv ? seq_kv - 1 : 0;
    const uint32_t qi = min(q0 + r16, p.seq_q - 1); // this lane's query (clamped: the rows of a ragged tile are never stored)
    // global address space: plain global_load (vmcnt only) instead of flat loads, so the waits below stay counted
    using f4 = __attribute__((ext_vector_type(4))) float;
    using gf4 = const __attribute__((address_space(1))) f4;
    using gf1 = const __attribute__((address_space(1))) float;
    gf1* const mask = (gf1*)(p.mask ? p.mask + (uint64_t)qi * p.mask_cs : zero_word);
    const uint32_t mask_rs = p.mask ? p.mask_rs : 0;
    gf1* const kbase = (gf1*)p.k + 4 * g;
    gf1* const vbase = (gf1*)p.v + 4 * r16;

    f4 qv[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) qv[i] = *(gf4*)((gf1*)p.q + (uint64_t)qi * p.q_cs + 16 * i + 4 * g);

    // every load of the loop is unconditional with a clamped row (a load under a branch turns the later counted waits
    // into vmcnt(0)); issue order per tile: K(next) after the score MFMAs, mask(next) after the softmax, V(next) after
    // the PV MFMAs — each lands under the work in front of its first use
    f4 kv[NI], vv[4][NH];
    float mk[4];
    auto load_k = [&](uint32_t tile) {
        const uint32_t key = min(tile * 16 + r16, last);
#pragma unroll
        for (int i = 0; i < NI; i++) kv[i] = *(gf4*)(kbase + (uint64_t)key * p.k_cs + 16 * i);
    };
    auto load_mask = [&](uint32_t tile) {
#pragma unroll
        for (int v = 0; v < 4; v++) mk[v] = mask[(uint64_t)min(tile * 16 + 4 * g + v, last) * mask_rs];
    };
    auto load_v = [&](uint32_t tile) {
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const uint32_t key = min(tile * 16 + 4 * g + v, last);
#pragma unroll
            for (int h = 0; h < NH; h++) vv[v][h] = *(gf4*)(vbase + (uint64_t)key * p.v_cs + 64 * h);
        }
    };

    f4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lsum = 0.f; // m: agreed by the 4 lanes of a query; lsum: this lane's keys only (folded at the end)

    load_k(w), load_mask(w), load_v(w); // (no keys at all: row 0 of the slabs, never used)
    for (uint32_t tile = w; tile < n_tiles; tile += NW) {
        const uint32_t key_base = tile * 16 + 4 * g;
        const uint32_t nxt = min(tile + NW, n_tiles - 1);
        bool any = false;
#pragma unroll
        for (int v = 0; v < 4; v++) any = any || (key_base + v < seq_kv && isfinite(mk[v]));
        const bool live = __builtin_amdgcn_ballot_w64(any) != 0; // wave-uniform: guards MFMAs
        f4 s0 = f4{0.f, 0.f, 0.f, 0.f}, s1 = f4{0.f, 0.f, 0.f, 0.f};
        if (live) {
#pragma unroll
            for (int i = 0; i < NI; i++) { // two accumulators: no MFMA waits for its predecessor
                f4& s = (i & 1) ? s1 : s0;
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[i].x, qv[i].x, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[i].y, qv[i].y, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[i].z, qv[i].z, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[i].w, qv[i].w, s, 0, 0, 0);
            }
        }
        load_k(nxt); // the K registers are free again
        float pw[4] = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            float sc[4];
            float bm = -INFINITY;
#pragma unroll
            for (int v = 0; v < 4; v++) {
                sc[v] = key_base + v < seq_kv ? tile_score(s0[v] + s1[v], mk[v], p.scale) : -INFINITY;
                bm = fmaxf(bm, sc[v]);
            }
            bm = rows_max(bm);
            const float nm = fmaxf(m, bm);
            if (__builtin_amdgcn_ballot_w64(nm != m) != 0) { // some query's maximum moved (x 1 otherwise: skipping is exact)
                const float alpha = m > -INFINITY ? expf(m - nm) : 0.0f;
                lsum *= alpha;
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] *= alpha;
            }
            if (nm > -INFINITY) {
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    pw[v] = sc[v] > -INFINITY ? expf(sc[v] - nm) : 0.0f;
                    lsum += pw[v];
                }
                m = nm;
            }
        }
        load_mask(nxt);
        if (live) {
#pragma unroll
            for (int v = 0; v < 4; v++)
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    acc[4 * h + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[v][h].x, pw[v], acc[4 * h + 0], 0, 0, 0);
                    acc[4 * h + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[v][h].y, pw[v], acc[4 * h + 1], 0, 0, 0);
                    acc[4 * h + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[v][h].z, pw[v], acc[4 * h + 2], 0, 0, 0);
                    acc[4 * h + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[v][h].w, pw[v], acc[4 * h + 3], 0, 0, 0);
                }
        }
        load_v(nxt);
    }
    lsum = rows_sum(lsum);

    // ── merge the waves' streams in wave order, normalise, store ──
    float* const part = smem;                                   // [NW][NT * 4][64]
    float* const pm = smem + (size_t)max_w * NT * 4 * 64;       // [NW][16]
    float* const pl = pm + max_w * 16;                          // [NW][16]
    float* const fac = pl + max_w * 16;                         // [NW][16]
    float* const linv = fac + max_w * 16;                       // [16]
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int v = 0; v < 4; v++) part[((size_t)w * NT * 4 + t * 4 + v) * 64 + lane] = acc[t][v];
    if (lane < 16) pm[w * 16 + lane] = m, pl[w * 16 + lane] = lsum;
    __syncthreads();
    if (tid < 16) {
        float M = -INFINITY;
        for (uint32_t ww = 0; ww < NW; ww++) M = fmaxf(M, pm[ww * 16 + tid]);
        float L = 0.f;
        for (uint32_t ww = 0; ww < NW; ww++) {
            const float mw = pm[ww * 16 + tid];
            const float f = NW == 1 ? 1.0f : (mw > -INFINITY ? expf(mw - M) : 0.0f);
            fac[ww * 16 + tid] = f;
            L += pl[ww * 16 + tid] * f;
        }
        linv[tid] = L > 0.f ? 1.0f / L : 0.0f;
    }
    __syncthreads();
    const uint32_t d2_off = p.dst2 ? *p.dyn_dst2_off : 0;
    for (uint32_t u = tid; u < NH * 4 * 64; u += NW * 64) {
        const uint32_t ln = u & 63, hv = u >> 6, h = hv >> 2, v = hv & 3, qq = ln & 15, gg = ln >> 4;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        for (uint32_t ww = 0; ww < NW; ww++) {
            const float f = fac[ww * 16 + qq];
#pragma unroll
            for (int c = 0; c < 4; c++) o[c] += part[((size_t)ww * NT * 4 + (4 * h + c) * 4 + v) * 64 + ln] * f;
        }
        const float il = linv[qq];
        const uint32_t qo = q0 + qq, d0 = 64 * h + 16 * gg + 4 * v;
        if (qo >= p.seq_q) continue;
        uint32_t hp[4][kAPieces]; // the bf16 pieces of the four values (kernels.h: split_a_pieces) for the piece sink below
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float val = o[c] * il;
            p.dst[(uint64_t)qo * p.dst_cs + (uint64_t)(d0 + c) * p.dst_rs] = val;
            if (p.dst2) p.dst2[(uint64_t)d2_off + (uint64_t)(d0 + c) * p.d2_rs + (uint64_t)qo * p.d2_cs] = val;
            split_a_pieces(val, hp[c]);
        }
        if (sink.ap && p.dst2 && (sink.S & kApF16)) { // ... of an f16-promoted matmul: four halves, round to nearest even
            uint16_t* const dp = sink.ap + a_f16_index(sink.S & ~kApF16, qo, d2_off + d0);
            *(uint2*)dp = make_uint2((uint32_t)__half_as_ushort(__float2half_rn(o[0] * il)) | ((uint32_t)__half_as_ushort(__float2half_rn(o[1] * il)) << 16),
                                     (uint32_t)__half_as_ushort(__float2half_rn(o[2] * il)) | ((uint32_t)__half_as_ushort(__float2half_rn(o[3] * il)) << 16));
        } else if (sink.ap && p.dst2) { // the row store fills row qo, columns d2_off + d0 .. + 3 of the next matmul's input (planner-checked)
            uint16_t* const dp = sink.ap + a_piece_index(sink.S, qo, d2_off + d0);
#pragma unroll
            for (int pc = 0; pc < kAPieces; pc++)
                *(uint2*)(dp + 512 * pc) = make_uint2((hp[0][pc] >> 16) | hp[1][pc], (hp[2][pc] >> 16) | hp[3][pc]);
        }
    }
}

} // namespace

bool attention_tiles_applies(uint32_t max_seq_q, uint32_t d_head) {
    static const bool on = !(getenv("ZGML_HIP_ATTN_TILES") && atoi(getenv("ZGML_HIP_ATTN_TILES")) == 0);
    return on && max_seq_q >= 16 && (d_head == 64 || d_head == 128);
}

// every op dense (q/k/v rows contiguous, 16-byte aligned, strides % 4 == 0 — the planner's `dense`) with this d_head
void launch_attention_tiles(hipStream_t s, const AttentionParams* dev_params, uint32_t n_ops, uint32_t max_seq_q, uint32_t d_head,
                            const float* zero_word, const AttnPieceSink& sink) {
    const uint32_t q_tiles = (max_seq_q + 15) / 16;
    const uint64_t units = (uint64_t)q_tiles * n_ops;
    static const int env_w = getenv("ZGML_HIP_ATTN_TILES_WAVES") ? atoi(getenv("ZGML_HIP_ATTN_TILES_WAVES")) : 0;
    // enough (query tile, head) workgroups to fill the chip: 4 waves each; few: 8, so a long context is spread over more waves
    uint32_t waves = env_w > 0 ? (uint32_t)env_w : (units >= 256 ? 4 : kTileWavesMax);
    waves = waves < 1 ? 1 : (waves > (uint32_t)kTileWavesMax ? (uint32_t)kTileWavesMax : waves);
    const size_t lds = ((size_t)waves * (d_head / 16) * 4 * 64 + (size_t)waves * 16 * 3 + 16) * sizeof(float);
    const dim3 grid(q_tiles, n_ops);
    if (d_head == 128) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attention_tiles_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(attention_tiles_kernel<128>, grid, dim3(waves * 64), lds, s, dev_params, zero_word, sink);
    } else {
        hipLaunchKernelGGL(attention_tiles_kernel<64>, grid, dim3(waves * 64), lds, s, dev_params, zero_word, sink);
    }
}

} // namespace zgml
