// host_capi.cpp — plain-C exports of the C++ host side for Python harnesses (tests, bench.py).
// The compute backend is passed in as a function table whose signatures are exactly the
// zgml_hip_* entry points (ctx first), so the same session code drives the HIP library or — in
// tests only — the CPU oracle's ctx-taking wrappers.
#include <chrono>
#include <cstring>
#include <functional>

#include "llama_decode.hpp"

using namespace zgml;
using namespace zgml::llama;

extern "C" {

typedef struct zh_config {
    uint32_t vocab_size, d_model, n_heads, n_kv_heads, d_ff, n_layers, max_seq_len;
    float rope_base, rms_norm_eps;
    uint32_t tied_lm_head;
    uint32_t shard_rank, shard_world;
    uint32_t kv_quant_block; // 0 = f32 KV caches, 32 = int8 blocks (quantised KV cache extension ops)
} zh_config;

typedef struct zh_backend_fns { // same signatures as include/zgml_hip.h
    void* ctx;
    void* (*compile_program)(void* ctx, const zgml_device_program* program);
    void (*refresh_program)(void* ctx, void* handle, const zgml_device_op* ops, uint64_t n_ops);
    void (*execute_program)(void* ctx, void* handle, const zgml_program_io* inputs, uint64_t n_inputs,
                            const zgml_program_io* outputs, uint64_t n_outputs);
    void (*free_program)(void* ctx, void* handle);
} zh_backend_fns;

typedef struct zh_gather_point {
    uint32_t op_end;
    uint16_t buf;
    uint16_t _pad;
    uint32_t offset, len_per_rank;
} zh_gather_point;

struct zh_model {
    std::unique_ptr<LlamaModel> model;
    std::unique_ptr<DecodeProgram> dp;
    std::vector<zgml_qweight_upload> qw_storage;
    zgml_device_program flat;
};

struct zh_session {
    zh_model* m;
    zh_backend_fns fns;
    void* handle;
    // the adapter's per-token refresh (zig/backend_hip.zig: refreshProgram): when set, a step derives (slice_pos, seq_kv) from the
    // patched ops exactly as the reference does for its wgpu backend (stepDynamicStateFromOps, src/backend/program.zig:7466-7490)
    // and hands over those two numbers instead of the whole op list
    int (*refresh_dynamic)(void* ctx, void* handle, uint32_t slice_pos, uint32_t seq_kv) = nullptr;
};

void zh_preset(const char* name, uint32_t max_seq, zh_config* out) {
    LlamaConfig c = !strcmp(name, "smollm-135m") ? smollm_135m() : !strcmp(name, "llama2-7b") ? llama2_7b(max_seq ? max_seq : 2048) : tiny_test();
    if (max_seq) c.max_seq_len = max_seq;
    *out = {c.vocab_size, c.d_model, c.n_heads, c.n_kv_heads, c.d_ff, c.n_layers, c.max_seq_len, c.rope_base, c.rms_norm_eps,
            c.tied_lm_head ? 1u : 0u, 0, 1, 0};
}

zh_model* zh_model_create_ex(const zh_config* cfg, int weight_kind, int fused_elementwise, int include_dead_f32, int threads,
                             uint32_t token_len);
zh_model* zh_model_create(const zh_config* cfg, int weight_kind, int fused_elementwise, int include_dead_f32, int threads) {
    return zh_model_create_ex(cfg, weight_kind, fused_elementwise, include_dead_f32, threads, 1);
}

zh_model* zh_model_create_ex(const zh_config* cfg, int weight_kind, int fused_elementwise, int include_dead_f32, int threads,
                             uint32_t token_len) {
    LlamaConfig c;
    c.vocab_size = cfg->vocab_size, c.d_model = cfg->d_model, c.n_heads = cfg->n_heads, c.n_kv_heads = cfg->n_kv_heads;
    c.d_ff = cfg->d_ff, c.n_layers = cfg->n_layers, c.max_seq_len = cfg->max_seq_len, c.rope_base = cfg->rope_base;
    c.rms_norm_eps = cfg->rms_norm_eps, c.tied_lm_head = cfg->tied_lm_head != 0;
    c.shard_rank = cfg->shard_rank, c.shard_world = cfg->shard_world ? cfg->shard_world : 1;
    c.kv_quant_block = cfg->kv_quant_block;
    if (c.kv_quant_block && (c.d_head() % c.kv_quant_block || c.kv_quant_block % 4 || ((uint64_t)c.max_seq_len * c.d_head()) % 4)) return nullptr;
    if (c.n_heads == 0 || c.d_model % c.n_heads || c.n_heads % c.n_kv_heads) return nullptr;
    if (c.shard_world > 1 && (c.tied_lm_head || c.n_kv_heads % c.shard_world || (c.d_model / c.shard_world) % 32 ||
                              (c.d_ff / c.shard_world) % 32 || (c.vocab_size / c.shard_world) % 32 || c.d_ff % c.shard_world ||
                              c.vocab_size % c.shard_world))
        return nullptr; // shard slices must stay whole 32-column scale blocks
    if (((WeightKind)weight_kind == WeightKind::f32_dense || (WeightKind)weight_kind == WeightKind::q4_0_gguf) && c.shard_world > 1)
        return nullptr; // shard slices are cut from the int8 form
    if (token_len > 1 && c.shard_world > 1) return nullptr; // the row-shard driver is decode-only
    auto* m = new zh_model();
    m->model = make_synthetic_model(c, (WeightKind)weight_kind, threads);
    m->dp = build_decode_program(*m->model, fused_elementwise != 0, include_dead_f32 != 0, token_len ? token_len : 1);
    m->flat = m->dp->program.view(m->qw_storage);
    return m;
}

void zh_model_free(zh_model* m) { delete m; }

const zgml_device_program* zh_model_program(zh_model* m) { return &m->flat; }

void zh_model_patch(zh_model* m, uint32_t token, uint32_t pos) { patch_step(*m->model, *m->dp, token, pos); }
void zh_model_patch_tokens(zh_model* m, const uint32_t* tokens, uint32_t pos) { patch_tokens(*m->model, *m->dp, tokens, pos); }
uint32_t zh_model_token_len(zh_model* m) { return m->dp->token_len; }

// One execution of a token_len = N plan (prefill chunk): patch, refresh, execute; logits of the
// last position are left in the model's host buffer. Returns its greedy token.
int64_t zh_session_prefill(struct zh_session* s, const uint32_t* tokens, uint32_t pos, float* logits_out);
void zh_session_set_refresh_dynamic(struct zh_session* s, int (*fn)(void*, void*, uint32_t, uint32_t));

const zgml_program_io* zh_model_step_inputs(zh_model* m, uint64_t* n) {
    *n = m->dp->step_inputs.size();
    return m->dp->step_inputs.data();
}
const zgml_program_io* zh_model_step_outputs(zh_model* m, uint64_t* n) {
    *n = m->dp->step_outputs.size();
    return m->dp->step_outputs.data();
}
const float* zh_model_logits(zh_model* m) { return m->dp->logits_host.data(); }
const float* zh_model_token_embed(zh_model* m) { return m->model->token_embed.data(); }
const float* zh_model_rope_table(zh_model* m, int sin_table) {
    return sin_table ? m->model->sin_table.data() : m->model->cos_table.data();
}
uint16_t zh_model_buf(zh_model* m, int which, uint32_t layer) {
    switch (which) {
        case 0: return m->dp->buf_token_input;
        case 1: return m->dp->buf_attn_mask;
        case 2: return m->dp->buf_logits;
        case 3: return m->dp->buf_rope[layer];
        case 4: return m->dp->buf_k_cache[layer];
        case 5: return m->dp->buf_v_cache[layer];
        default: return 0;
    }
}
// KV-cache buffers of the plan in builder order (DecodeProgram::kv_buffers): ids into `bufs`, f32-element counts into `elems`.
uint64_t zh_model_kv_buffers(zh_model* m, uint16_t* bufs, uint64_t* elems, uint64_t cap) {
    const auto& k = m->dp->kv_buffers;
    for (uint64_t i = 0; i < k.size() && i < cap; i++) bufs[i] = k[i].buf, elems[i] = k[i].elems;
    return k.size();
}
uint64_t zh_model_gather_points(zh_model* m, zh_gather_point* out, uint64_t cap) {
    const auto& g = m->dp->gather_points;
    for (uint64_t i = 0; i < g.size() && i < cap; i++) out[i] = {g[i].op_end, g[i].buf, 0, g[i].offset, g[i].len_per_rank};
    return g.size();
}
uint64_t zh_model_quant_bytes(zh_model* m, uint64_t* n_weights) { // Q4_0/Q8_0 file-format bytes of the quantized weights
    uint64_t elems = 0;
    for (const auto& q : m->model->qweights) elems += q.K * q.N;
    if (n_weights) *n_weights = elems;
    if (m->model->kind == WeightKind::f32_dense) return elems * 2; // f16-promoted bytes
    return elems / 32 * (m->model->kind == WeightKind::q8_0 ? 34 : 18);
}

zh_session* zh_session_create(zh_model* m, const zh_backend_fns* fns) {
    void* h = fns->compile_program(fns->ctx, &m->flat);
    if (!h) return nullptr;
    return new zh_session{m, *fns, h};
}

void zh_session_free(zh_session* s) {
    if (!s) return;
    s->fns.free_program(s->fns.ctx, s->handle);
    delete s;
}

void* zh_session_handle(zh_session* s) { return s->handle; }

void zh_session_set_refresh_dynamic(zh_session* s, int (*fn)(void*, void*, uint32_t, uint32_t)) { s->refresh_dynamic = fn; }

// stepDynamicStateFromOps (src/backend/program.zig:7466-7490): the first dynamic slice_assign gives slice_pos, the first attention
// seq_kv; false when the op list has neither (then the whole list is refreshed)
static bool step_dynamic_state(const std::vector<zgml::backend::DeviceOp>& ops, uint32_t& slice_pos, uint32_t& seq_kv) {
    bool has_sa = false, has_att = false;
    for (const zgml::backend::DeviceOp& op : ops) {
        if (op.kind == ZGML_DOP_SLICE_ASSIGN && !has_sa && op.u.slice_assign.patch_stride != 0 && op.u.slice_assign.dst_offset >= op.u.slice_assign.dst_base_offset) {
            slice_pos = (op.u.slice_assign.dst_offset - op.u.slice_assign.dst_base_offset) / op.u.slice_assign.patch_stride;
            has_sa = true;
        } else if (op.kind == ZGML_DOP_KVQ_STORE && !has_sa && op.u.kvq_store.patch_stride != 0 && op.u.kvq_store.col >= op.u.kvq_store.col_base) { // (extension op: same role)
            slice_pos = (op.u.kvq_store.col - op.u.kvq_store.col_base) / op.u.kvq_store.patch_stride;
            has_sa = true;
        } else if (op.kind == ZGML_DOP_ATTENTION && !has_att) {
            seq_kv = op.u.attention.seq_kv, has_att = true;
        } else if (op.kind == ZGML_DOP_ATTENTION_KVQ && !has_att) {
            seq_kv = op.u.attention_kvq.seq_kv, has_att = true;
        }
        if (has_sa && has_att) break;
    }
    return has_sa || has_att;
}
static void session_refresh(zh_session* s, DecodeProgram& dp) {
    uint32_t slice_pos = 0, seq_kv = 0;
    if (s->refresh_dynamic && step_dynamic_state(dp.program.ops, slice_pos, seq_kv) && s->refresh_dynamic(s->fns.ctx, s->handle, slice_pos, seq_kv) == 0) return;
    s->fns.refresh_program(s->fns.ctx, s->handle, dp.program.ops.data(), dp.program.ops.size());
}

// doStep (llama_smollm_bench.zig:290-314): patch, refresh, execute. Returns the greedy token.
int64_t zh_session_step(zh_session* s, uint32_t token, uint32_t pos, float* logits_out) {
    DecodeProgram& dp = *s->m->dp;
    patch_step(*s->m->model, dp, token, pos);
    session_refresh(s, dp);
    s->fns.execute_program(s->fns.ctx, s->handle, dp.step_inputs.data(), dp.step_inputs.size(), dp.step_outputs.data(),
                           dp.step_outputs.size());
    if (logits_out) std::memcpy(logits_out, dp.logits_host.data(), dp.logits_host.size() * sizeof(float));
    return LlamaDeviceSession::argmax(dp.logits_host.data(), (uint32_t)dp.logits_host.size());
}

int64_t zh_session_prefill(zh_session* s, const uint32_t* tokens, uint32_t pos, float* logits_out) {
    DecodeProgram& dp = *s->m->dp;
    patch_tokens(*s->m->model, dp, tokens, pos);
    session_refresh(s, dp);
    s->fns.execute_program(s->fns.ctx, s->handle, dp.step_inputs.data(), dp.step_inputs.size(), dp.step_outputs.data(),
                           dp.step_outputs.size());
    if (logits_out) std::memcpy(logits_out, dp.logits_host.data(), dp.logits_host.size() * sizeof(float));
    return LlamaDeviceSession::argmax(dp.logits_host.data(), (uint32_t)dp.logits_host.size());
}

// the session's greedy pick and the reference's own loop (tests compare them; first maximum wins, src/nn.zig:122-138)
uint32_t zh_argmax(const float* logits, uint32_t n) { return LlamaDeviceSession::argmax(logits, n); }
uint32_t zh_argmax_reference(const float* logits, uint32_t n) { return LlamaDeviceSession::argmax_reference(logits, n); }

// Greedy decode through the vtable contract (host logits every step): feeds `first_token` at
// `start_pos`, then its own argmax, for n_steps; writes the produced tokens; returns seconds.
double zh_session_decode(zh_session* s, uint32_t first_token, uint32_t start_pos, uint32_t n_steps, int64_t* tokens_out) {
    uint32_t tok = first_token;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < n_steps; i++) {
        const int64_t next = zh_session_step(s, tok, start_pos + i, nullptr);
        if (tokens_out) tokens_out[i] = next;
        tok = (uint32_t)next;
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

} // extern "C"
