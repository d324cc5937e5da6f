// kernels.h — launch interface between the runtime (runtime.hip) and the gfx950 kernels.
// All pointers are device pointers; offsets/strides are in f32 elements unless noted.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/zgml_hip.h"

namespace zgml {

constexpr int kMaxFusedSteps = 8; // Capabilities.hip.max_fused_elementwise_steps

struct FusedStepDev {
    uint32_t op;
    uint32_t swapped;
    const float* secondary; // already offset
};

struct FusedParams {
    float* dst;
    const float* src;
    uint32_t n;
    uint32_t n_steps;
    FusedStepDev steps[kMaxFusedSteps];
};

// a chain of elementwise / fused_elementwise ops over the same n elements (each consumes its predecessor's
// output at the same index) as one launch; every op's output is still stored (step.store)
constexpr int kMaxChainSteps = 12;
struct ChainStepDev {
    uint32_t op;
    uint32_t swapped;
    const float* secondary; // already offset; nullptr for unary ops
    float* store;           // output of the op this step completes (nullptr inside an op's own step list)
};
struct EltChainParams {
    const float* src;
    uint32_t n, n_steps;
    ChainStepDev steps[kMaxChainSteps];
    // optional side output of the chain's final value for an M > 1 quantized matmul over dense rows of `ap_cols` columns
    // (RowChainParams::ap): set by the planner when that matmul follows
    uint16_t* ap = nullptr;
    uint32_t ap_S = 0, ap_cols = 0;
};
void launch_eltwise_chain(hipStream_t s, const EltChainParams& p);

struct RepeatParams {
    float* dst;       // base (dst_offset applied in kernel, as the reference indexes)
    const float* src; // base
    uint32_t n;
    uint32_t src_ne[4], src_strides[4], dst_strides[4];
    uint32_t src_offset, dst_offset;
    uint32_t mode; // 0 = generic index math, 1 = scalar fill, 2 = flat copy, 3 = tile copy
    uint32_t src_n;
};

struct SliceAssignParams {
    float* dst;
    const float* src; // src_offset applied
    uint32_t rows, cols;
    uint32_t dst_row_stride, dst_col_stride, src_row_stride, src_col_stride;
    const uint32_t* dyn_dst_offset; // device word holding the current dst_offset
};

struct RopeParams {
    float* dst;
    const float* src;
    const float* cs;
    uint32_t half_d, seq_len, src_rs, src_cs, cs_cs;
    // optional fused slice_assign of the rotated block (rope k -> KV cache): second destination
    float* dst2;                  // buffer base; nullptr = none
    const uint32_t* dyn_dst2_off; // device word with the current dst_offset
    uint32_t d2_rs, d2_cs;
};

// One record of the batched "movement" launch: a rope (kind 0) or a slice_assign (kind 1).
struct MoveParams {
    uint32_t kind;
    uint32_t n_elems; // threads needed
    union {
        RopeParams rope;
        SliceAssignParams sa;
    };
};

struct AttentionParams {
    float* dst;
    const float* q;
    const float* k;
    const float* v;
    const float* mask; // nullptr when !has_mask
    uint32_t d_head, seq_q;
    const uint32_t* dyn_seq_kv; // device word holding the current seq_kv
    float scale;
    uint32_t q_rs, q_cs, k_rs, k_cs, v_rs, v_cs, mask_rs, mask_cs, dst_rs, dst_cs;
    // optional fused slice_assign of the head output (row store into the concatenated buffer)
    float* dst2;
    const uint32_t* dyn_dst2_off;
    uint32_t d2_rs, d2_cs;
};

// Decode-shaped attention of one head with everything that feeds it folded in (seq_q == 1):
// rope of the head's query, rope of its kv head's new key + K-cache store, V-cache store, the
// attention itself and (via att.dst2) the head-output row store. Heads that share a kv head each
// recompute the new key/value from the projections; `owner` marks the one that writes the cache
// and the k_rot side output. The new column is taken from registers, never re-read from memory.
struct AttnDecodeParams {
    AttentionParams att;      // att.q = q_rot (side output), att.k / att.v = the kv head's slabs
    float* q_rot;             // == att.q, writable
    const float* q_src;       // q projection slice, unit stride
    const float* q_cs;        // cos | sin row of the query rope (sin at + half_d)
    const float* k_src;       // k projection slice, unit stride
    const float* k_cs;        // cos | sin row of the key rope
    const float* v_src;       // v projection slice, unit stride
    float* k_rot;             // the key rope's dense output (owner writes it)
    float* k_cache;           // base of the K cache buffer (store offset is relative to it)
    float* v_cache;
    const uint32_t* dyn_k_off; // current K store offset (elements)
    const uint32_t* dyn_v_off;
    uint32_t k_off, v_off;    // att.k - k_cache, att.v - v_cache (elements)
    uint32_t owner;
    uint32_t max_kv;          // compile-time seq_kv bound: rows [0, max_kv) of the slabs are readable
    // quantised KV caches (kvq_store + attention_kvq folded in): block size (32; 0 = f32 caches) and columns per cache;
    // k_cache / v_cache are then the cache buffers (int8 rows, f32 block scales behind them), the dynamic words columns
    uint32_t kvq_block = 0, kvq_cols = 0;
    unsigned long long* trace; // diagnostics (ZGML_HIP_ATTN_TRACE=1): 8 wall-clock stamps, else nullptr
};

// quantised KV cache (extension ops; layout in include/zgml_hip.h)
struct KvqStoreParams {
    float* cache;                // buffer base
    const float* src;            // src_offset applied, d_head values, unit stride
    const uint32_t* dyn_col;     // device word with the current column
    uint32_t d_head, block_size, n_cols;
};
struct KvqAttentionParams {
    float* dst;                  // dst_off applied
    const float* q;              // q_off applied
    const float* k_cache;        // buffer bases
    const float* v_cache;
    const float* mask;           // mask_off applied; nullptr when !has_mask
    const uint32_t* dyn_seq_kv;
    uint32_t d_head, seq_q, block_size, n_cols, k_col_start, v_col_start, q_cs, dst_cs, mask_rs, mask_cs;
    float scale;
};
void launch_kvq_store_batch(hipStream_t s, const KvqStoreParams* dev_params, uint32_t n_ops);


struct DenseMatmulParams {
    float* dst;       // dst_offset applied
    const float* a;   // a_offset applied
    const void* b;    // b_offset applied; float or __half when b_f16
    uint32_t M, N, K;
    uint32_t a_rs, a_cs, b_rs, b_cs, dst_rs;
    uint32_t b_f16;
};

// dense matmul against an f16-promoted, MFMA-packed B (dense_f16.hip)
struct DenseF16Params {
    float* dst;      // dst_offset applied
    const float* a;  // a_offset applied, a_col_stride == 1
    const void* bp;  // packed half[N/16][ceil(K/32)][64][8]
    uint32_t M, N, K;
    uint32_t a_rs, dst_rs;
    uint32_t stream_nt; // non-temporal weight loads: the program's promoted weights exceed the Infinity Cache
    // M > 1: device block of dense_f16_scratch_bytes(M, K) for the pre-rounded A operand (nullptr: staged form);
    // reuse_a: the block already holds this input (previous launch, same rows)
    void* scratch = nullptr;
    uint32_t reuse_a = 0;
};
uint64_t dense_f16_scratch_bytes(uint32_t M, uint32_t K);
bool dense_f16_a_unpadded(uint32_t M, uint32_t K);
// M > 1: up to dense_f16_max_group() promoted matmuls over the same rows in one launch
bool dense_f16_can_group(const DenseF16Params& a, const DenseF16Params& b);
uint32_t dense_f16_max_group();
void launch_dense_f16_group(hipStream_t s, const DenseF16Params* p, uint32_t n);
bool f16_packable(uint64_t K, uint64_t N);
uint64_t f16_packed_bytes(uint64_t K, uint64_t N);
void launch_pack_f16(hipStream_t s, const float* b, uint32_t b_rs, uint32_t b_cs, uint32_t K, uint32_t N, void* out);
void launch_dense_f16(hipStream_t s, const DenseF16Params& p);

// ── quantized weights on the device ──────────────────────────────────────────
enum QWFormat : uint32_t {
    QW_RAW = 0, // int8 [K*N] + f32 scale per `bs` flat elements (any bs, any N)
    QW_Q4 = 1,  // packed nibbles, lane-tiled (see qmatvec.hip)
    QW_Q8 = 2,  // packed int8, lane-tiled
    QW_Q4K = 3, // packed offset-binary nibbles, K ON LANES (qmatvec.hip: qmatvec_kon_body): M = 1 mat-vecs only, f16 scales
    QW_W8A8 = 4, // the reference's W8A8 arm (w8a8.hip): re-quantised transposed int8 + f32 scale per (column, 32 k); M = 1 only, opt-in
};

struct QWeightDev {
    QWFormat format = QW_RAW;
    uint32_t K = 0, N = 0, bs = 0;
    uint32_t KC = 0;       // K chunks of 32 (padded)
    uint32_t scale_f16 = 0; // packed formats: scales stored as f16 (exact) or f32
    uint32_t stream_nt = 0; // read with non-temporal loads: the program's weights exceed the Infinity Cache
    void* qs = nullptr;     // packed quants or raw int8
    void* sc = nullptr;     // packed scales or raw f32 scales
    uint64_t qs_bytes = 0, sc_bytes = 0;
};

// ── fused mat-vec launch description (M == 1 decode path) ────────────────────────────────────
// One launch = up to 4 weight matrices that read the same input vector (q/k/v, gate/up), an
// optional PROLOGUE that produces that vector on the fly and an optional elementwise EPILOGUE
// per matrix. Every intermediate the unfused program would have written is still stored (by
// workgroup 0 for prologue values, by the owning lanes for epilogue values), so any buffer a
// host could download holds the same numbers as in the one-op-per-launch plan.
constexpr int kMaxQmvParts = 4;
constexpr int kMaxEpiSteps = 6;

struct QmvEpiStep {
    uint32_t op;          // ZGML_OP_*; binary ops read operand[n]
    uint32_t swapped;     // chain value is the right-hand operand
    const float* operand; // indexed by output column n (nullptr for unary)
    float* store;         // optional: write the value after this step to store[n]
};

enum QmvPrologueKind : uint32_t {
    QMV_PRO_NONE = 0,        // x = a
    QMV_PRO_MUL = 1,         // x = a * b                         (elementwise mul feeding the mat-vec)
    QMV_PRO_RMSNORM_MUL = 2, // mid = a / sqrt(mean(a^2) + eps); x = mid * b   (rmsnorm -> mul gamma)
    // RMSNORM_MUL whose inputs the PRODUCING launch has prepared (QmvNextNorm below): `xg` = a * b elementwise and `ssq` =
    // n_ssq partial sums of a^2, so this launch streams ONE vector and scales its finished sums by 1 / sqrt(sum / K + eps)
    // (the mat-vec is linear in x). Only the K-on-lanes kernel consumes this form; a / b still name the original operands
    // (the absorbed ops' own outputs are stored from them).
    QMV_PRO_PRENORM = 3,
};

struct QmvPrologue {
    uint32_t kind = QMV_PRO_NONE;
    float eps = 0.f;
    const float* a = nullptr;
    const float* b = nullptr;
    float* store_mid = nullptr; // RMSNORM_MUL: the bare normalised vector
    float* store_x = nullptr;   // MUL / RMSNORM_MUL: the vector the mat-vec consumes
    const float* xg = nullptr;  // PRENORM: a * b, written by the producing launch
    const float* ssq = nullptr; // PRENORM: n_ssq partial sums of a^2 (one per 16 elements), same producer
    uint32_t n_ssq = 0;
};

// What a launch with a residual-add epilogue (h = y + r) prepares for the rmsnorm -> mul(gamma) -> mat-vec launch that consumes
// h next (runtime.hip: arm_prenorm): xg_out[n] = h[n] * gamma[n] and ssq_out[n / 16] = sum of h^2 over its 16 columns.
struct QmvNextNorm {
    const float* gamma = nullptr;
    float* xg_out = nullptr; // nullptr: nothing to prepare
    float* ssq_out = nullptr;
};

struct QmvPart {
    QWeightDev w;
    float* dst = nullptr; // y (offset applied)
    uint32_t n_epi = 0;
    QmvEpiStep epi[kMaxEpiSteps];
};

struct QmvLaunch {
    uint32_t n_parts = 0;
    QmvPart parts[kMaxQmvParts];
    QmvPrologue pro;
    QmvNextNorm next;
    // gate / up pair (runtime.hip: arm_pair): parts[0] carries the SiLU chain, parts[1] is plain, both K-on-lanes, same shape;
    // the launch also stores silu(parts[0]) * parts[1] here — the product the NEXT mat-vec's MUL prologue would recompute
    float* pair_out = nullptr;
    uint32_t K = 0;
    unsigned long long* trace = nullptr; // diagnostics (ZGML_HIP_QMV_TRACE=1): 8 wall-clock stamps of block 0
};

struct QMatmulParams {
    float* dst;         // dst_offset applied
    const float* input; // input_offset applied
    uint32_t M, N, K;
    uint32_t in_rs, dst_rs;
    uint32_t reuse_split = 0; // M > 1: the scratch already holds this input's A pieces (previous launch, same rows)
};

// generic kernels (kernels_generic.hip)
void launch_elementwise(hipStream_t s, uint32_t op, float* dst, const float* s0, const float* s1, uint32_t n);
void launch_fused_elementwise(hipStream_t s, const FusedParams& p);
void launch_softmax(hipStream_t s, float* dst, const float* src, uint32_t rows, uint32_t cols);
void launch_layernorm(hipStream_t s, float* dst, const float* src, uint32_t rows, uint32_t cols, float eps);
void launch_rmsnorm(hipStream_t s, float* dst, const float* src, uint32_t rows, uint32_t cols, float eps);
// [elementwise add ->] rmsnorm [-> elementwise mul] over dense rows in one launch; every intermediate is
// still stored and the arithmetic (order included) is that of the separate kernels.
struct RowChainParams {
    const float* a0 = nullptr; // pre-add: src row = a0 + a1, stored to add_dst (all nullptr: no add)
    const float* a1 = nullptr;
    float* add_dst = nullptr;
    const float* src = nullptr; // rmsnorm input (== add_dst with a pre-add)
    float* norm_dst = nullptr;
    const float* mul_other = nullptr; // post-mul: mul_dst = norm * mul_other (nullptr: none)
    float* mul_dst = nullptr;
    uint32_t cols = 0;
    float eps = 0.f;
    // Optional side output for an M > 1 quantized matmul that consumes the chain's result (mul_dst, else norm_dst): the
    // rows' exact bf16 pieces in the tile kernels' A-operand layout (qmatmul_tiles.hip: split_a_kernel), written here instead of
    // by a launch of their own. ap_S = the matmul's K steps (cols / 128); only for rows % 16 == 0 and cols % 128 == 0.
    uint16_t* ap = nullptr;
    uint32_t ap_S = 0;
};
// The A operand of the Q4_0 XDL tile kernels (M > 1 qmatmul): x as kAPieces bf16 pieces, x ~ a1 + a2.
// Round 4: TWO pieces by round-to-nearest-even — a1 = bf16(x), a2 = bf16(x - a1) (the difference is exact in
// f32; |x - a1| <= 2^-8 of x's binade, and rounding that 16-bit residual to 8 bits leaves <= 2^-17) — so |x - (a1 + a2)| <=
// 2^-17 |x| PER ELEMENT (bf16 keeps f32's exponent range: no dependence on the row's largest element), and with the weight's
// exact two-piece split all four piece products are computed: |delta| <= 2^-17 sum|x w| = 7.6e-6 sum|x w| against the
// contract's 2e-5 (tests/test_hip_qmatvec.py exercises it with a worst-case vector: measured 4.3e-6). Rounds 2-3 carried three truncation pieces (exact, five of
// six products): one MFMA in five and a third of the A bytes more for accuracy the contract does not ask for.
constexpr int kAPieces = 2;
#if defined(__HIPCC__)
__device__ __forceinline__ void split_a_pieces(float v, uint32_t (&h)[kAPieces]) { // h[p] = piece p as a bf16 in the HIGH half of a dword
    auto rne = [](float f) { // f32 -> bf16 (round to nearest even), as the high half
        const uint32_t u = __float_as_uint(f);
        return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u; // (finite activations: the NaN caveat of the integer form does not arise on this path's inputs being finite)
    };
    float r = v;
#pragma unroll
    for (int p = 0; p < kAPieces; p++) {
        h[p] = rne(r);
        r = r - __uint_as_float(h[p]); // exact
    }
}
#endif
// element (row m, column j) of an activation matrix in the A-operand layout of the XDL tile kernels: index of its bf16 in
// units of 2 bytes for piece 0; piece p follows at + 512 p (64 lanes x 8 bf16 each)
__host__ __device__ inline uint64_t a_piece_index(uint32_t S, uint32_t m, uint32_t j) {
    const uint64_t t = m >> 4, i = m & 15, s = j >> 7, r = (j >> 5) & 3, jj = (j >> 3) & 3, e = j & 7;
    return ((((t * S + s) * 4 + jj) * kAPieces) * 64 + (i + 16 * r)) * 8 + e;
}
// the same for the f16-promoted matmuls' A operand (dense_f16.hip: pack_a_f16_kernel): index of the element's half. A
// producer is told the format by the top bit of its ap_S field: kApF16 | KC (32-k chunks) instead of the 128-k step count.
constexpr uint32_t kApF16 = 0x80000000u;
__host__ __device__ inline uint64_t a_f16_index(uint32_t KC, uint32_t m, uint32_t j) {
    const uint64_t t = m >> 4, i = m & 15, c = j >> 5, r = (j >> 3) & 3, e = j & 7;
    return (((t * KC + c) * 64) + 16 * r + i) * 8 + e;
}
void launch_row_chain(hipStream_t s, const RowChainParams& p, uint32_t rows);
void launch_reduce(hipStream_t s, uint32_t op, float* dst, const float* src, uint32_t n_out, uint32_t reduce_size);
// batched: dev_params = device array of n_ops records; one launch covers all of them
void launch_repeat_batch(hipStream_t s, const RepeatParams* dev_params, uint32_t n_ops, uint32_t max_elems);
void launch_move_batch(hipStream_t s, const MoveParams* dev_params, uint32_t n_ops, uint32_t max_elems);
// all_dense: every op has unit row strides, 16-byte aligned q/k/v rows, d_head a power of two in [4, 256]
// Long-context split of the decode attention (flash decoding): `splits` workgroups per head; partials
// go through `buf` (n_heads * splits * (d_head + 4) floats), `cnt` holds one zeroed arrival counter per
// head. A head splits once seq_kv >= 2 * min_keys. splits <= 1: no split, buf / cnt unused.
struct AttnSplit {
    uint32_t splits = 1, min_keys = 128;
    float* buf = nullptr;
    uint32_t* cnt = nullptr;
};
void launch_attention_decode_batch(hipStream_t s, const AttnDecodeParams* dev_params, uint32_t n_heads, uint32_t d_head,
                                   const AttnSplit& sp, bool kvq = false); // kvq: every head reads quantised caches (block 32)
void launch_kvq_attention_batch(hipStream_t s, const KvqAttentionParams* dev_params, uint32_t n_ops, uint32_t max_seq_q, uint32_t d_head,
                                const AttnSplit& sp = AttnSplit{}); // split applies to seq_q == 1 launches only
// seq_q >= 16, dense layouts, d_head 64 / 128: query tiles on the matrix cores (attention_tiles.hip)
bool attention_tiles_applies(uint32_t max_seq_q, uint32_t d_head);
// A-piece side output of a batched attention launch whose row stores (dst2) fill the dense rows an M > 1 quantized matmul
// reads (the concatenated head outputs in front of the O projection): RowChainParams::ap for the tile kernel's stores
struct AttnPieceSink {
    uint16_t* ap = nullptr;
    uint32_t S = 0, cols = 0;
};
void launch_attention_tiles(hipStream_t s, const AttentionParams* dev_params, uint32_t n_ops, uint32_t max_seq_q, uint32_t d_head,
                            const float* zero_word, const AttnPieceSink& sink = AttnPieceSink{});
void launch_attention_batch(hipStream_t s, const AttentionParams* dev_params, uint32_t n_ops, uint32_t max_seq_q, bool all_dense,
                            uint32_t rows_d_head = 0, const float* zero_word = nullptr, const AttnPieceSink& sink = AttnPieceSink{}); // rows_d_head: the common d_head when every op is dense (else 0)
void launch_dense_matmul(hipStream_t s, const DenseMatmulParams& p);
// `adv` (device-resident decode): the final stage also appends the token — state[0] = token, state[1] += 1 (position),
// tokens[state[2]++] = token — instead of a one-thread launch of its own behind it
struct ArgmaxAdvance {
    uint32_t* state = nullptr;
    int64_t* tokens = nullptr;
    uint32_t cap = 0;
};
void launch_argmax(hipStream_t s, const float* v, uint64_t n, float* scratch_val, int64_t* scratch_idx, int64_t* out,
                   const ArgmaxAdvance& adv = ArgmaxAdvance{});
// Everything LlamaInferencePlan.execute patches on the host per execution (src/llama_inference.zig:405-446; for T > 1
// patch_tokens of zgml_amd/host/llama_decode.cpp), produced on the device as one flat index space: T embedding rows, T
// causal-mask columns, T RoPE rows per layer leaf, the dynamic words (KV store offsets at `pos`, seq_kv = pos + T).
struct ResidentPrepArgs {
    const float *embed, *cos, *sin;
    float *tok_in, *mask;
    float* const* rope_bufs;
    const uint32_t *dyn_kind, *dyn_base, *dyn_stride;
    uint32_t* dyn;
    const uint32_t* state;  // [0] token, [1] position, [2] produced count
    const uint32_t* tokens; // T token ids (decode: &state[0])
    uint32_t d, max_seq, dh, n_rope, n_ops, T;
};
void launch_resident_prep(hipStream_t s, const ResidentPrepArgs& a, uint32_t total); // total = T d + T max_seq + n_rope T 2 dh + n_ops
// The token tail of the device-resident decode loop in ONE launch: the first maximum of v[0, n) (the two stages of launch_argmax:
// the workgroup whose arrival is the last one on `cnt` — one zeroed word, re-armed here — does the second), the advance of the
// device state (`adv`) and, by that same workgroup, the NEXT token's patches (`prep`, T = 1, at the advanced position; skipped once
// the position has reached max_seq). Same (value, index) ordering as launch_argmax: results are identical.
void launch_argmax_tail(hipStream_t s, const float* v, uint64_t n, float* scratch_val, int64_t* scratch_idx, uint32_t* cnt, int64_t* out,
                        const ArgmaxAdvance& adv, const ResidentPrepArgs* prep, uint32_t prep_total);
void launch_copy_f4(hipStream_t s, void* dst, const void* src, uint64_t bytes);
void launch_f32_to_f16(hipStream_t s, void* dst, const float* src, uint64_t n);

// quantized matmul (qmatvec.hip)
// Device-side repack of a QuantizedWeightUpload (raw int8 + f32 scales already on the device)
// into the packed layout. `raw_data`/`raw_scales` are device pointers; `out` must have
// format/K/N/KC/scale_f16 set and qs/sc allocated (packed_bytes() gives the sizes).
// Classification helpers run on the device too: returns bit0 = all values in [-8,7],
// bit1 = all scales exactly representable in f16.
uint32_t classify_qweight(hipStream_t s, const int8_t* raw_data, uint64_t n_elems, const float* raw_scales,
                          uint64_t n_blocks, uint32_t* flag_scratch /* 2 device words */);
// GGUF Q4_0 / Q8_0 blocks (device copy of the file bytes) -> packed layout; out.format selects which.
void launch_pack_gguf(hipStream_t s, const uint8_t* raw_blocks, const QWeightDev& out);
// True when (K, N, block size) can use the packed fast path (else QW_RAW).
bool qweight_packable(uint64_t K, uint64_t N, uint64_t bs);
// W8A8 arm (w8a8.hip; src/backend/reference.zig:512-528): shapes it takes, sizes of its packed image, the pack (prepareTransposed on
// the device, from the raw int8 + f32-scale upload) and the mat-vec (quantizeInput + gemvRange, bit-identical to the reference's)
bool w8a8_applies(uint64_t K, uint64_t N, uint64_t bs);
void w8a8_packed_bytes(uint64_t K, uint64_t N, uint64_t* qs_bytes, uint64_t* sc_bytes);
void launch_pack_w8a8(hipStream_t s, const int8_t* raw_data, const float* raw_scales, const QWeightDev& out);
void launch_w8a8_matvec(hipStream_t s, const QWeightDev& w, const float* x, float* dst);
void packed_bytes(QWFormat format, uint32_t scale_f16, uint64_t K, uint64_t N, uint64_t* qs_bytes,
                  uint64_t* sc_bytes);
void launch_pack_qweight(hipStream_t s, const int8_t* raw_data, const float* raw_scales, const QWeightDev& out);
// Bytes of scratch a qmatmul launch may need behind its `scratch` pointer (A pieces, split-K partial tiles). In addition
// the kQmmScratchHead bytes IN FRONT of that pointer belong to the launchers: fan-in counters of the K-split tile kernel,
// zero at allocation and re-armed by every launch. Whoever allocates the block allocates kQmmScratchHead + bytes,
// zeroes the head and passes base + kQmmScratchHead.
constexpr uint64_t kQmmScratchHead = 16384;  // (round 4: 4096 counter words — the A-stationary f16 kernel keeps one per column group)
constexpr uint64_t kQmmCounterBytes = 4096; // what the Q4_0 tile launchers size their K splits by (unchanged from round 3)
uint64_t qmatmul_scratch_bytes(const QWeightDev& w, uint32_t M);
void launch_qmatmul(hipStream_t s, const QWeightDev& w, const QMatmulParams& p, float* scratch);
// M > 1: up to qmatmul_max_group() quantized matmuls over the same rows in one launch (qmatmul_can_group pairwise)
bool qmatmul_can_group(const QWeightDev& a, const QMatmulParams& pa, const QWeightDev& b, const QMatmulParams& pb);
uint32_t qmatmul_max_group();
void launch_qmatmul_group(hipStream_t s, const QWeightDev* w, const QMatmulParams* p, uint32_t n, float* scratch);
// can a mat-vec launch over `w` consume an rmsnorm -> mul(gamma) prologue prepared by its producer (QMV_PRO_PRENORM)?
bool qmv_prenorm_ok(const QWeightDev& w, uint32_t K, uint64_t total_cols, uint32_t M);
// can a two-matrix launch over weights like `w` compute silu(part 0) * part 1 itself (the gate / up PAIR launch)?
bool qmv_pair_ok(const QWeightDev& w);
// Fused M == 1 launch; all parts must be packed, share K, format and scale type (qmv_can_group).
bool qmv_can_group(const QWeightDev& a, const QWeightDev& b);
// Prologue kinds other than NONE keep the whole input in registers: K <= qmv_max_prologue_k(w).
uint32_t qmv_max_prologue_k(const QWeightDev& w);
void launch_qmatvec_fused(hipStream_t s, const QmvLaunch& L);
// q / k / v projection (grouped, rmsnorm prologue, K <= 2048) + the decode attention of its heads in ONE launch; the
// projection's outputs are handed over through `counters` ([n_heads + 2 n_kv] words; `idx` [n_heads][3] names each record's
// q / k / v counter) with `seen` ([n_heads * splits * 3]) as the consumers' private progress words; counters and seen
// zero-initialised by the caller. false: not fusable, nothing launched.
bool launch_qkv_attention(hipStream_t s, const QmvLaunch& L, const AttnDecodeParams* dev_params, uint32_t n_heads, uint32_t n_kv, uint32_t d_head,
                          const AttnSplit& sp, uint32_t* counters, const uint32_t* idx, uint32_t* seen, uint32_t* timeout,
                          const QmvLaunch* o_proj = nullptr, uint32_t* out_cnt = nullptr, uint32_t* o_seen = nullptr, bool kvq = false);
// (round 3) decode attention + the K-on-lanes projection that reads the heads' row stores in one launch of 256-thread workgroups
// (qmatvec.hip: attn_o_kon_kernel); `out_cnt` one zeroed word, `o_seen` one zeroed word per workgroup of the projection
bool launch_attention_o(hipStream_t s, const AttnDecodeParams* dev_params, uint32_t n_heads, uint32_t d_head, const AttnSplit& sp, const QmvLaunch& o_proj,
                        uint32_t* out_cnt, uint32_t* o_seen, uint32_t* timeout);
int attn_o_blocks_per_cu(uint32_t d_head);
int qkv_attn_kon_census(hipStream_t s, uint32_t d_head, bool kvq, uint32_t grid); // 1 = the whole grid of that launch is co-resident (run once, plan-build time), 0 = not, -1 = unknown
int qkv_attn_kon_blocks_per_cu(uint32_t d_head, bool kvq); // fused q / k / v + attention launch of K-on-lanes weights (256-thread workgroups) // occupancy query of that kernel
// (o_proj: the single-matrix projection that reads the heads' row stores rides in the same launch; out_cnt: one zeroed
// word, o_seen: one zeroed word per workgroup of that projection)
// Deterministic synthetic weights for the roofline micro-benchmark, generated on the device
// directly in the packed layout (SURVEY §8d generator).
void launch_synth_packed(hipStream_t s, const QWeightDev& w, uint32_t matrix_id);

// ── ksplit.hip: the decoder layer of a short-K model (K <= 2048, n-on-lanes Q4_0 weights with f16 scales: SmolLM-135M) in launches
// that end at a K-SPLIT instead of at an all-to-all seam. The O projection is computed by the attention's
// workgroups — head h multiplies its d_head outputs by ITS d_head rows of W_o — and the down projection by the gate / up
// workgroups — the workgroup that holds silu(gate) * up for 32 columns multiplies them by its 32 rows of W_down; each stores a
// PARTIAL output vector. Nothing inside the launch waits for the partials: the launch that consumes the vector next sums them
// in a fixed order in its prologue (a "deferred vector", KsVec), where it has to read the vector anyway, and workgroup 0 of that
// launch stores what the absorbed ops would have stored (the projection's own output and the residual sum). Op stream matched:
// src/models/llama_transformer.zig:192-253; arithmetic of every op: src/backend/reference.zig (qmatmul :530-565, rmsnorm :349-374).
struct KsVec { // x[k] = base[k] + sum_p parts[p][k], p ascending (n_parts == 0: x = base)
    const float* base = nullptr;
    const float* parts = nullptr; // [n_parts][K] dense
    uint32_t n_parts = 0;
    float* sum_dst = nullptr; // the absorbed mat-vec's output buffer: sum of the parts
    float* add_dst = nullptr; // the absorbed residual add's output: base + sum
};
struct KsNormIn { // rmsnorm -> mul(gamma) of a (possibly deferred) vector; gamma == nullptr: no norm (materialise the vector only)
    KsVec x;
    const float* gamma = nullptr;
    float eps = 0.f;
    float* store_mid = nullptr; // x / rms(x)        (the rmsnorm op's output)
    float* store_x = nullptr;   // ... * gamma       (the mul op's output: what the mat-vecs consume)
    uint32_t K = 0;
};
struct KsProjLaunch { // up to 3 mat-vecs over the normalised vector (q / k / v), no epilogues
    KsNormIn in;
    uint32_t n_parts = 0;
    QWeightDev w[3];
    float* dst[3] = {nullptr, nullptr, nullptr};
    unsigned long long* trace = nullptr; // diagnostics build (ZGML_HIP_KS_TRACE=1): 16 wall-clock stamps of workgroup 0
};
struct KsAttnOLaunch { // decode attention of n_heads heads + each head's partial O projection
    const AttnDecodeParams* dev = nullptr;
    uint32_t n_heads = 0, d_head = 0;
    AttnSplit sp;
    bool kvq = false;
    QWeightDev wo;              // K = n_heads * d_head (head h owns rows [h * d_head, (h + 1) * d_head) — checked by the planner)
    float* parts_out = nullptr; // [n_heads][wo.N]
    uint8_t head_of[64] = {0};  // record -> head = its row store's static offset / d_head (the records are not in head order; n_heads <= 64)
    unsigned long long* trace = nullptr;
};
struct KsMlpLaunch { // gate / up (+ SiLU chain + product) + partial down projection
    KsNormIn in;
    QWeightDev gate, up, down;
    float *gate_out = nullptr, *up_out = nullptr, *exp_out = nullptr, *silu_out = nullptr, *prod_out = nullptr;
    const float* ones = nullptr; // the SiLU chain's ADD operand (a vector of ones in the LLaMA plans)
    float* parts_out = nullptr;  // [gate.N / 32][down.N]
    unsigned long long* trace = nullptr;
};
bool ks_weight_ok(const QWeightDev& w);
bool ks_proj_ok(const KsProjLaunch& L);
bool ks_attn_o_ok(const KsAttnOLaunch& L);
bool ks_mlp_ok(const KsMlpLaunch& L);
bool ks_norm_ok(const KsNormIn& in);
uint32_t ks_mlp_parts(const KsMlpLaunch& L); // partial vectors the launch writes (gate.N / 32)
bool ks_layer_a_ok(const KsProjLaunch& PL, const KsAttnOLaunch& AL); // the two as ONE launch (in-launch hand-off of q / k / v)
uint32_t ks_layer_a_proj_wgs(const KsProjLaunch& PL);
bool launch_ks_layer_a(hipStream_t s, const KsProjLaunch& PL, const KsAttnOLaunch& AL, uint32_t n_kv, uint32_t* counters, const uint32_t* idx, uint32_t* seen,
                       uint32_t* timeout);
void launch_ks_proj(hipStream_t s, const KsProjLaunch& L);
void launch_ks_attn_o(hipStream_t s, const KsAttnOLaunch& L);
void launch_ks_mlp(hipStream_t s, const KsMlpLaunch& L);
void launch_ks_norm(hipStream_t s, const KsNormIn& in); // one workgroup: materialises a deferred vector (and its norm, if any)

// ── shard_peer.hip: the row-shard all-gather as peer stores (ZGML_SHARD_GATHER=peer) and the (max, index) gather of the greedy token ──
struct PeerGatherArgs {
    char* const* blocks; // device array [world]: every rank's fine-grained block (own included) as THIS process maps them
    float* local;        // the point's region of this rank's program buffer: world * len floats, own slice already computed
    uint32_t* seen;      // this rank's private count of arrivals already consumed at this point
    uint32_t* timeout;   // the context's host-visible hand-off word
    uint64_t ctr_off, stage_off; // byte offsets of the point's arrival counter / staging area inside a block
    uint64_t wait_ticks;         // give-up bound of the wait, 100 MHz ticks
    uint32_t len, rank, world;
    uint64_t poison_off;         // byte offset of the block's poison word (set in EVERY rank's block by a rank whose wait gave up)
};
struct PeerArgmaxArgs {
    char* const* blocks;
    const float* slice; // this rank's logits slice, len floats
    uint32_t* seen;
    uint32_t* timeout;
    int64_t* out;       // the greedy token (-1 after a give-up)
    uint64_t ctr_off, pairs_off, wait_ticks;
    uint32_t len, rank, world;
    uint64_t poison_off;
};
void launch_peer_gather(hipStream_t s, const PeerGatherArgs& a);
void launch_peer_argmax(hipStream_t s, const PeerArgmaxArgs& a);
void launch_local_argmax_pair(hipStream_t s, const float* slice, uint32_t len, uint32_t rank, unsigned long long* pairs);
void launch_reduce_pairs(hipStream_t s, const unsigned long long* pairs, uint32_t world, int64_t* out);

} // namespace zgml
