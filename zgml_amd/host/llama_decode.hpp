// llama_decode.hpp — host side of the LLaMA decode path above the backend boundary, mirroring the
// reference (Zig is not available here, so this is the compiled host code in C++):
//
//   LlamaConfig                 src/models/llama.zig:34-45
//   LlamaModel (synthetic)      weights in the reference's in-memory form: QuantizedWeight =
//                               int8 [K,N] + f32 scale / 32 flat elements (src/quant.zig:200-256),
//                               as produced from GGUF Q4_0/Q8_0 blocks by
//                               quantizedWeightFromInfo (src/models/gguf_loader.zig:99-154)
//   DecodeProgram               DeviceInference.init for the token_len=1 plan: op stream of
//                               LlamaBlock.forwardCachedMasked (src/models/llama_transformer.zig:192-253)
//                               + LLaMA.forwardCachedMasked (src/models/llama.zig:143-168), lowered
//                               as src/device_inference.zig:61-239 / :665-737 does (DFS node order of
//                               src/graph.zig:1294-1311, views folded into offsets/strides)
//   patchSliceAssignOffset / patchAttentionSeqKV   src/device_inference.zig:242-256
//   LlamaDeviceSession.step     doStep of benchmarks/llama_smollm_bench.zig:290-314
//                               == LlamaInferencePlan.execute steps 1-4 (src/llama_inference.zig:405-466)
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "backend.hpp"

namespace zgml::llama {

struct LlamaConfig {
    uint32_t vocab_size = 0, d_model = 0, n_heads = 0, n_kv_heads = 0, d_ff = 0, n_layers = 0, max_seq_len = 0;
    float rope_base = 10000.0f;
    float rms_norm_eps = 1e-6f;
    bool tied_lm_head = false;
    uint32_t d_head() const { return d_model / n_heads; }
    uint32_t kv_dim() const { return n_kv_heads * d_head(); }
    // Row-shard (config 4): this rank owns columns [rank*N/ws, (rank+1)*N/ws) of every weight and
    // n_heads/ws query heads; 1/1 = unsharded. Only Llama-7B-like shapes shard.
    uint32_t shard_rank = 0, shard_world = 1;
    // Quantised KV cache (src/quant.zig:645-1091, wiring src/llama_inference.zig:277-377):
    // 0 = f32 caches (slice_assign + attention), else the int8 block size (32): one cache buffer per kv
    // head, written by kvq_store and read by attention_kvq (extension DeviceOps, include/zgml_hip.h).
    uint32_t kv_quant_block = 0;
};

LlamaConfig smollm_135m();               // benchmarks/llama_smollm_bench.zig:31-42
LlamaConfig llama2_7b(uint32_t max_seq); // SURVEY §8(a) "L7"
LlamaConfig tiny_test();                 // small GQA config for parity tests

// q4_0_gguf: the Q4_0 model handed over as raw 18-byte GGUF blocks (packed pass-through)
enum class WeightKind : int { q4_0 = 0, q8_0 = 1, f32_dense = 2, q4_0_gguf = 3 };

struct QWeightHost {
    std::vector<int8_t> data;
    std::vector<float> scales;
    std::vector<uint8_t> gguf; // q4_0_gguf: the file blocks (data / scales dropped)
    size_t K = 0, N = 0; // rows, cols
};

struct LlamaModel {
    LlamaConfig cfg;
    WeightKind kind = WeightKind::q4_0;
    std::vector<float> token_embed;             // [vocab, d_model] row per token (col-major [d, vocab])
    std::vector<std::vector<float>> gamma1, gamma2; // per layer [d]
    std::vector<float> gamma_f;
    // per layer: q,k,v,o,gate,up,down; then (untied) the LM head
    std::vector<QWeightHost> qweights;
    // kind == f32_dense: the same weights dequantised to f32 [K, N] (qweights keep only K and N)
    std::vector<std::vector<float>> dense;
    std::vector<float> cos_table, sin_table; // [max_seq][d_head], RoPE.init src/nn.zig:286-311
    size_t lm_head_index() const { return (size_t)cfg.n_layers * 7; }
};

// Deterministic synthetic weights (SURVEY §8d generator, RNG-free). `threads` parallelises the fill.
std::unique_ptr<LlamaModel> make_synthetic_model(const LlamaConfig& cfg, WeightKind kind, int threads);

struct DecodeProgram {
    backend::DeviceProgram program;
    std::vector<std::unique_ptr<std::vector<backend::FusedEwStep>>> steps_storage;
    // host-side leaf tensors patched per step (the plan's token_input / attn_mask / rope leaves)
    std::vector<float> token_input, attn_mask, scalar_one;
    std::vector<std::vector<float>> rope_leaf; // per layer [2*d_head]
    std::vector<std::vector<float>> dead_f32;  // optional f32 master copies (never read by an op)
    std::vector<float> kv_zero;                 // zero initial KV (not uploaded: buffers start zeroed)
    uint32_t token_len = 1; // tokens per execution (1 = decode plan, N = prefill plan)
    uint16_t buf_token_input = 0, buf_attn_mask = 0, buf_logits = 0;
    std::vector<uint16_t> buf_rope, buf_k_cache, buf_v_cache;
    // every KV-cache buffer of the plan, in builder order, with its f32-element count: the consolidated f32 caches
    // (K, V per layer) or, with kv_quant_block, the per-kv-head int8 caches (K, V per head per layer). A prefill plan
    // and the decode plan of the same model list them in the same order, so the hand-off (llama_inference.prefill, then
    // step) is a pairwise copy.
    struct KvBuffer {
        uint16_t buf;
        uint64_t elems;
    };
    std::vector<KvBuffer> kv_buffers;
    std::vector<uint32_t> slice_assign_op_indices, attention_op_indices;
    std::vector<backend::ProgramIO> step_inputs, step_outputs;
    std::vector<float> logits_host;
    // collective points for the row-sharded variant: after op index `op_end` (exclusive), buffer
    // `buf` holds this rank's slice at [rank*len, (rank+1)*len) and must be all-gathered.
    struct GatherPoint {
        uint32_t op_end;
        uint16_t buf;
        uint32_t offset, len_per_rank;
    };
    std::vector<GatherPoint> gather_points;

    void patchSliceAssignOffset(uint32_t pos);
    void patchAttentionSeqKV(uint32_t seq_kv);
};

// include_dead_f32: also register the f32 master copy of every quantized weight as a buffer with
// an initial upload, exactly as DeviceInference does; the backend elides them.
// token_len: 1 builds the decode plan; N > 1 builds the fixed-width prefill plan
// (LlamaInferencePlan with token_len = N, src/llama_inference.zig:405-466; src/llm/device_prefill.zig).
std::unique_ptr<DecodeProgram> build_decode_program(const LlamaModel& model, bool fused_elementwise,
                                                    bool include_dead_f32, uint32_t token_len = 1);

// Patch the plan's host leaves for (token, pos): embedding row, causal mask column, RoPE row
// (llama_smollm_bench.zig:299-309), then the dynamic op fields.
void patch_step(const LlamaModel& model, DecodeProgram& dp, uint32_t token, uint32_t pos);
void patch_tokens(const LlamaModel& model, DecodeProgram& dp, const uint32_t* tokens, uint32_t pos);

struct LlamaDeviceSession {
    const LlamaModel* model = nullptr;
    std::unique_ptr<DecodeProgram> dp;
    backend::Backend be;
    backend::Backend::CompiledHandle handle = nullptr;
    bool init(const LlamaModel& m, const backend::Backend& backend, bool include_dead_f32);
    void deinit();
    // One decode step through the vtable (refresh + execute); returns the logits slice.
    const float* step(uint32_t token, uint32_t pos);
    static uint32_t argmax(const float* logits, uint32_t n); // first max wins, src/nn.zig:122-138 (two vector passes where AVX2 is there)
    static uint32_t argmax_reference(const float* logits, uint32_t n); // the reference's loop itself (tests)
};

} // namespace zgml::llama
