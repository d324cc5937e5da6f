// llama_decode.cpp — see llama_decode.hpp for the reference map.
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include "llama_decode.hpp"

#include <cstdlib>

#include <algorithm>
#include <functional>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>

namespace zgml::llama {

using backend::DeviceOp;
using backend::FusedEwStep;
using backend::ProgramIO;

LlamaConfig smollm_135m() {
    LlamaConfig c;
    c.vocab_size = 49152, c.d_model = 576, c.n_heads = 9, c.n_kv_heads = 3, c.d_ff = 1536, c.n_layers = 30;
    c.max_seq_len = 2048, c.rope_base = 10000.0f, c.rms_norm_eps = 1e-5f, c.tied_lm_head = true;
    return c;
}

LlamaConfig llama2_7b(uint32_t max_seq) {
    LlamaConfig c;
    c.vocab_size = 32000, c.d_model = 4096, c.n_heads = 32, c.n_kv_heads = 32, c.d_ff = 11008, c.n_layers = 32;
    c.max_seq_len = max_seq, c.rope_base = 10000.0f, c.rms_norm_eps = 1e-5f, c.tied_lm_head = false;
    return c;
}

LlamaConfig tiny_test() {
    LlamaConfig c;
    c.vocab_size = 512, c.d_model = 128, c.n_heads = 4, c.n_kv_heads = 2, c.d_ff = 256, c.n_layers = 2;
    c.max_seq_len = 64, c.rope_base = 10000.0f, c.rms_norm_eps = 1e-5f, c.tied_lm_head = true;
    return c;
}

// ── synthetic weights ───────────────────────────────────────────────────────────────────────
namespace {

// f32 -> f16 -> f32 round trip so block scales are exactly what a GGUF file can hold
float f16_round(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = x & 0x80000000u;
    int32_t exp = (int32_t)((x >> 23) & 0xFF) - 127;
    uint32_t man = x & 0x7FFFFF;
    if (exp < -14 || exp > 15) return f; // generator never leaves the normal f16 range
    const uint32_t rem = man & 0x1FFF;
    man &= ~0x1FFFu;
    if (rem > 0x1000 || (rem == 0x1000 && (man & 0x2000))) {
        man += 0x2000;
        if (man & 0x800000) {
            man = 0;
            exp++;
        }
    }
    x = sign | ((uint32_t)(exp + 127) << 23) | man;
    float r;
    std::memcpy(&r, &x, 4);
    return r;
}

// bits of an f16-exact f32 in the normal range (the generator's scales): truncation is exact
inline uint16_t f32_to_f16_bits(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t exp = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
    return (uint16_t)(sign | ((uint32_t)exp << 10) | ((x & 0x7FFFFF) >> 13));
}

// element (k, n_global) of weight `id` with full width N_full; Q4_0: stored nibble - 8 (values
// in [-8,7], what quantizedWeightFromInfo's .q4_0 arm yields), Q8_0: int8.
inline int8_t synth_q(uint64_t flat, uint32_t id, WeightKind kind) {
    if (kind != WeightKind::q8_0) // dense and gguf-block models are the Q4_0 model in another form
        return (int8_t)((int)((flat * 7 + (flat >> 5) * 3 + (uint64_t)id * 5) & 15) - 8);
    return (int8_t)((int)((flat * 13 + (uint64_t)id * 29) % 255) - 127);
}
// block scale: small enough that a 30-layer random-ish stack stays finite; exact in f16
inline float synth_scale(uint64_t block, uint32_t id, WeightKind kind) {
    const float base = kind != WeightKind::q8_0 ? 1.0f / 512.0f : 1.0f / 8192.0f;
    return f16_round(base * (1.0f + (float)((block + id) % 7) * 0.125f));
}

void parallel_rows(size_t n, int threads, const std::function<void(size_t, size_t)>& fn) {
    threads = std::max(1, threads);
    if (threads == 1 || n < 64) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> ts;
    const size_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        const size_t b = t * chunk, e = std::min(n, b + chunk);
        if (b < e) ts.emplace_back(fn, b, e);
    }
    for (auto& t : ts) t.join();
}

// columns [n0, n0 + N_loc) of the K x N_full synthetic weight `id`
QWeightHost make_qweight(size_t K, size_t N_full, size_t n0, size_t N_loc, uint32_t id, WeightKind kind, int threads) {
    QWeightHost w;
    w.K = K, w.N = N_loc;
    w.data.resize(K * N_loc);
    w.scales.resize((K * N_loc + 31) / 32);
    const size_t nb_full = N_full / 32, nb_loc = N_loc / 32, b0 = n0 / 32;
    parallel_rows(K, threads, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; k++) {
            int8_t* row = w.data.data() + k * N_loc;
            const uint64_t base = (uint64_t)k * N_full + n0;
            for (size_t n = 0; n < N_loc; n++) row[n] = synth_q(base + n, id, kind);
            for (size_t b = 0; b < nb_loc; b++) w.scales[k * nb_loc + b] = synth_scale((uint64_t)k * nb_full + b0 + b, id, kind);
        }
    });
    return w;
}

} // namespace

std::unique_ptr<LlamaModel> make_synthetic_model(const LlamaConfig& cfg, WeightKind kind, int threads) {
    auto m = std::make_unique<LlamaModel>();
    m->cfg = cfg;
    m->kind = kind;
    const uint32_t d = cfg.d_model, dh = cfg.d_head(), kvd = cfg.kv_dim(), ff = cfg.d_ff, ws = cfg.shard_world,
                   r = cfg.shard_rank;
    // token embedding: LLaMA.init formula (src/models/llama.zig:86-91)
    m->token_embed.resize((size_t)cfg.vocab_size * d);
    const float scale = 1.0f / std::sqrt((float)d);
    parallel_rows(cfg.vocab_size, threads, [&](size_t v0, size_t v1) {
        for (size_t i = v0 * d; i < v1 * d; i++) {
            const float fi = (float)i;
            m->token_embed[i] = scale * std::sin(fi * 0.1f + 0.3f) * std::cos(fi * 0.07f + 0.5f);
        }
    });
    // norm gains: the reference initialises them to 1; use a mild deterministic pattern so the
    // repeat+mul ops are exercised with non-trivial data
    auto gamma = [&](uint32_t seed) {
        std::vector<float> g(d);
        for (uint32_t i = 0; i < d; i++) g[i] = 1.0f + 0.03125f * (float)((int)((i + seed) % 9) - 4);
        return g;
    };
    for (uint32_t l = 0; l < cfg.n_layers; l++) {
        m->gamma1.push_back(gamma(2 * l));
        m->gamma2.push_back(gamma(2 * l + 1));
    }
    m->gamma_f = gamma(977);
    // quantized projection weights, [K, N] with N = output features (sharded along N)
    auto shard = [&](size_t N) { return std::pair<size_t, size_t>{(size_t)r * (N / ws), N / ws}; };
    for (uint32_t l = 0; l < cfg.n_layers; l++) {
        const uint32_t id = l * 8;
        const struct {
            size_t K, N;
        } shapes[7] = {{d, d}, {d, kvd}, {d, kvd}, {d, d}, {d, ff}, {d, ff}, {ff, d}};
        for (int j = 0; j < 7; j++) {
            auto [n0, nl] = shard(shapes[j].N);
            m->qweights.push_back(make_qweight(shapes[j].K, shapes[j].N, n0, nl, id + j, kind, threads));
        }
    }
    if (!cfg.tied_lm_head) {
        auto [n0, nl] = shard(cfg.vocab_size);
        m->qweights.push_back(make_qweight(d, cfg.vocab_size, n0, nl, cfg.n_layers * 8, kind, threads));
    }
    if (kind == WeightKind::q4_0_gguf) { // the same model as 18-byte file blocks (gguf_loader.zig:118-141, interleaved nibbles)
        for (auto& q : m->qweights) {
            const size_t nb = q.K * q.N / 32;
            q.gguf.resize(nb * 18);
            parallel_rows(nb, threads, [&](size_t b0, size_t b1) {
                for (size_t b = b0; b < b1; b++) {
                    uint8_t* blk = q.gguf.data() + b * 18;
                    const uint16_t h = f32_to_f16_bits(q.scales[b]);
                    blk[0] = (uint8_t)(h & 0xFF), blk[1] = (uint8_t)(h >> 8);
                    for (int j = 0; j < 16; j++)
                        blk[2 + j] = (uint8_t)(((q.data[b * 32 + 2 * j] + 8) & 15) | (((q.data[b * 32 + 2 * j + 1] + 8) & 15) << 4));
                }
            });
            std::vector<int8_t>().swap(q.data);
            std::vector<float>().swap(q.scales);
        }
    }
    if (kind == WeightKind::f32_dense) { // the same model, dequantised: f32 [K, N] per weight; int8 form dropped
        for (auto& q : m->qweights) {
            std::vector<float> w(q.K * q.N);
            parallel_rows(q.K, threads, [&](size_t k0, size_t k1) {
                for (size_t i = k0 * q.N; i < k1 * q.N; i++) w[i] = (float)q.data[i] * q.scales[i / 32];
            });
            m->dense.push_back(std::move(w));
            std::vector<int8_t>().swap(q.data);
            std::vector<float>().swap(q.scales);
        }
    }
    // RoPE tables (src/nn.zig:297-310), f32 math
    m->cos_table.resize((size_t)cfg.max_seq_len * dh);
    m->sin_table.resize((size_t)cfg.max_seq_len * dh);
    for (uint32_t pos = 0; pos < cfg.max_seq_len; pos++)
        for (uint32_t i = 0; i < dh / 2; i++) {
            const float freq = (float)pos / std::pow(cfg.rope_base, (float)(2 * i) / (float)dh);
            const float c = std::cos(freq), s = std::sin(freq);
            m->cos_table[(size_t)pos * dh + i] = m->cos_table[(size_t)pos * dh + i + dh / 2] = c;
            m->sin_table[(size_t)pos * dh + i] = m->sin_table[(size_t)pos * dh + i + dh / 2] = s;
        }
    return m;
}

// ── DeviceInference lowering of the decode plan ─────────────────────────────────────────────
void DecodeProgram::patchSliceAssignOffset(uint32_t pos) {
    for (uint32_t idx : slice_assign_op_indices) {
        auto& op = program.ops[idx];
        if (op.kind == ZGML_DOP_KVQ_STORE) { // quantised cache: the dynamic field is the column
            op.u.kvq_store.col = op.u.kvq_store.col_base + pos * op.u.kvq_store.patch_stride;
            continue;
        }
        auto& sa = op.u.slice_assign;
        if (sa.patch_stride == 0) continue;
        sa.dst_offset = sa.dst_base_offset + pos * sa.patch_stride;
    }
}

void DecodeProgram::patchAttentionSeqKV(uint32_t seq_kv) {
    for (uint32_t idx : attention_op_indices) {
        auto& op = program.ops[idx];
        if (op.kind == ZGML_DOP_ATTENTION_KVQ)
            op.u.attention_kvq.seq_kv = seq_kv;
        else
            op.u.attention.seq_kv = seq_kv;
    }
}

namespace {

struct Builder {
    DecodeProgram& dp;
    explicit Builder(DecodeProgram& d) : dp(d) {}
    uint16_t buffer(uint64_t elems) {
        dp.program.buffer_sizes.push_back(std::max<uint64_t>(elems, 1));
        return (uint16_t)(dp.program.buffer_sizes.size() - 1);
    }
    uint16_t leaf(const std::vector<float>& data) { // BufferMap.ensure for an op == .none tensor
        const uint16_t b = buffer(data.size());
        if (!data.empty()) {
            ProgramIO io{};
            io.buf_idx = b, io.offset = 0, io.host_ptr = (void*)data.data(), io.size = (uint32_t)(data.size() * sizeof(float));
            dp.program.initial_uploads.push_back(io);
        }
        return b;
    }
    void op(const DeviceOp& o) { dp.program.ops.push_back(o); }
    // x.matMul(false, w, false) of one projection: qmatmul against the quantized weight table, or —
    // dense model — a `matmul` whose B is the [K, N] f32 weight leaf (N contiguous, same flat k*N+n
    // order as the quantized form; a backend may promote it to f16, src/backend/wgpu.zig:1071-1104)
    std::vector<uint16_t> dense_buf; // weight index -> buffer id (dense models)
    void proj(uint16_t dst, uint16_t input, uint16_t widx, uint32_t M, uint32_t N, uint32_t K, uint32_t in_off, uint32_t in_rs,
              uint32_t dst_off, uint32_t dst_rs) {
        if (dense_buf.empty()) {
            op(DeviceOp::qmatmul(dst, input, widx, M, N, K, in_off, in_rs, dst_off, dst_rs));
            return;
        }
        backend::MatMulGeometry g{};
        g.M = M, g.N = N, g.K = K, g.a_row_stride = in_rs, g.a_col_stride = 1, g.b_row_stride = N, g.b_col_stride = 1;
        g.a_offset = in_off, g.b_offset = 0, g.dst_offset = dst_off, g.dst_row_stride = dst_rs;
        op(DeviceOp::matmul(dst, input, dense_buf[widx], g));
    }
    // rmsnorm -> repeat(gamma) -> mul  (applyRmsNorm, llama_transformer.zig:120-127)
    uint16_t rms_norm(uint16_t x, uint16_t gamma, uint32_t d, float eps, uint32_t T) {
        const uint16_t bare = buffer((uint64_t)d * T), rep = buffer((uint64_t)d * T), out = buffer((uint64_t)d * T);
        op(DeviceOp::rmsnorm(bare, x, T, d, eps));
        const uint32_t sne[4] = {d, 1, 1, 1}, sst[4] = {1, d, d, d}, dne[4] = {d, T, 1, 1}, dst_[4] = {1, d, d * T, d * T};
        op(DeviceOp::repeat(rep, gamma, d * T, sne, dne, sst, dst_)); // gamma.repeatLike(bare): [d] -> [d, T]
        op(DeviceOp::elementwise(ZGML_OP_MUL, out, bare, rep, d * T));
        return out;
    }
    const FusedEwStep* steps(std::initializer_list<FusedEwStep> s) {
        dp.steps_storage.push_back(std::make_unique<std::vector<FusedEwStep>>(s));
        return dp.steps_storage.back()->data();
    }
};

} // namespace

std::unique_ptr<DecodeProgram> build_decode_program(const LlamaModel& model, bool fused_elementwise,
                                                    bool include_dead_f32, uint32_t token_len) {
    const LlamaConfig& c = model.cfg;
    auto dpp = std::make_unique<DecodeProgram>();
    DecodeProgram& dp = *dpp;
    Builder b(dp);
    const uint32_t d = c.d_model, dh = c.d_head(), ff = c.d_ff, S = c.max_seq_len, ws = c.shard_world, r = c.shard_rank;
    const uint32_t H = c.n_heads, KV = c.n_kv_heads, n_rep = H / KV;
    const uint32_t H_loc = H / ws, KV_loc = KV / ws, h0 = r * H_loc, kv0 = r * KV_loc;
    const uint32_t d_loc = d / ws, kvd_loc = c.kv_dim() / ws, ff_loc = ff / ws;
    const float attn_scale = 1.0f / std::sqrt((float)dh);
    // rehearsal switch (tests): mark the gather points also at world size 1 (token_len 1, untied head), so a single
    // GPU exercises every collective call of the N > 1 path as a 1-rank in-place all-gather
    const bool rehearse = ws == 1 && token_len <= 1 && !c.tied_lm_head && getenv("ZGML_HOST_SHARD_POINTS_WORLD1") &&
                          atoi(getenv("ZGML_HOST_SHARD_POINTS_WORLD1")) != 0;
    const bool sharded = ws > 1 || rehearse;
    const uint32_t T = token_len ? token_len : 1; // tokens per execution: 1 = decode plan, N = prefill plan
    dp.token_len = T;

    // quantized weight table (all borrowed from the model), or dense f32 weight leaves
    for (const auto& qw : model.qweights) {
        if (model.kind == WeightKind::q4_0_gguf)
            dp.program.qweights.push_back(backend::QuantizedWeightUpload::ggufBlocks(qw.gguf.data(), qw.gguf.size(), qw.K, qw.N));
        else if (model.kind != WeightKind::f32_dense)
            dp.program.qweights.push_back({qw.data.data(), qw.data.size(), qw.scales.data(), qw.scales.size(), qw.K, qw.N, 32});
    }
    if (model.kind == WeightKind::f32_dense)
        for (const auto& w : model.dense) b.dense_buf.push_back(b.leaf(w));

    dp.token_input.assign((size_t)d * T, 0.f);
    dp.attn_mask.assign((size_t)S * T, -std::numeric_limits<float>::infinity()); // [max_seq, T], column per query
    for (uint32_t j = 0; j < T; j++)
        for (uint32_t s2 = 0; s2 <= j; s2++) dp.attn_mask[(size_t)j * S + s2] = 0.f;
    dp.scalar_one.assign(1, 1.0f);
    dp.buf_token_input = b.leaf(dp.token_input);
    dp.buf_attn_mask = b.leaf(dp.attn_mask);

    uint16_t x = dp.buf_token_input;
    for (uint32_t l = 0; l < c.n_layers; l++) {
        const uint16_t wq = l * 7 + 0, wk = l * 7 + 1, wv = l * 7 + 2, wo = l * 7 + 3, wg = l * 7 + 4, wu = l * 7 + 5, wd = l * 7 + 6;
        const uint16_t g1 = b.leaf(model.gamma1[l]), g2 = b.leaf(model.gamma2[l]);
        dp.rope_leaf.emplace_back((size_t)2 * dh * T, 0.f);
        const uint16_t rope_cs = b.leaf(dp.rope_leaf.back());
        dp.buf_rope.push_back(rope_cs);
        // consolidated KV caches [d_head, max_seq * n_kv_heads] (this rank's kv heads), zero-initialised:
        // the backend zero-fills every buffer at compile, so no upload is recorded for them
        const uint16_t k_cache = b.buffer((uint64_t)dh * S * KV_loc), v_cache = b.buffer((uint64_t)dh * S * KV_loc);
        dp.buf_k_cache.push_back(k_cache);
        dp.buf_v_cache.push_back(v_cache);
        if (!c.kv_quant_block) {
            dp.kv_buffers.push_back({k_cache, (uint64_t)dh * S * KV_loc});
            dp.kv_buffers.push_back({v_cache, (uint64_t)dh * S * KV_loc});
        }
        if (include_dead_f32 && model.kind != WeightKind::f32_dense) { // f32 master copies: buffers + uploads, never referenced by an op (F8)
            for (int j = 0; j < 7; j++) {
                const auto& qw = model.qweights[l * 7 + j];
                dp.dead_f32.emplace_back(qw.K * qw.N);
                auto& f = dp.dead_f32.back();
                for (size_t i = 0; i < f.size(); i++) f[i] = (float)qw.data[i] * qw.scales[i / 32];
                b.leaf(f);
            }
        }
        const uint16_t attn_buf = b.buffer((uint64_t)d * T);

        // ---- attention half; DFS post-order of the graph rooted at `output` (graph.zig:1294-1311):
        // the sliceAssignRows chain is visited from its last link, so head H-1 comes first and the
        // row stores unwind in ascending order afterwards.
        const uint16_t norm1 = b.rms_norm(x, g1, d, c.rms_norm_eps, T);
        const uint16_t q_proj = b.buffer((uint64_t)d_loc * T), k_proj = b.buffer((uint64_t)kvd_loc * T),
                       v_proj = b.buffer((uint64_t)kvd_loc * T);
        bool q_done = false;
        std::vector<char> kv_done(KV_loc, 0);
        bool k_proj_done = false, v_proj_done = false;
        std::vector<uint16_t> attn_out(H_loc);
        const uint32_t kvq = c.kv_quant_block; // quantised KV: one int8 cache buffer per kv head (K and V)
        std::vector<uint16_t> kq_cache(KV_loc), vq_cache(KV_loc);
        if (kvq)
            for (uint32_t j = 0; j < KV_loc; j++) {
                const uint64_t elems = (uint64_t)S * dh / 4 + (uint64_t)S * (dh / kvq);
                kq_cache[j] = b.buffer(elems), vq_cache[j] = b.buffer(elems);
                dp.kv_buffers.push_back({kq_cache[j], elems});
                dp.kv_buffers.push_back({vq_cache[j], elems});
            }
        for (int hl = (int)H_loc - 1; hl >= 0; hl--) {
            const uint32_t h = h0 + hl, kvh = h / n_rep, kvl = kvh - kv0;
            if (!q_done) {
                b.proj(q_proj, norm1, wq, T, d_loc, d, 0, d, 0, d_loc);
                q_done = true;
            }
            const uint16_t q_rot = b.buffer((uint64_t)dh * T);
            b.op(DeviceOp::rope(q_rot, q_proj, rope_cs, dh / 2, T, hl * dh, 0, 0, 1, d_loc, 2 * dh));
            if (!kv_done[kvl]) {
                kv_done[kvl] = 1;
                if (!k_proj_done) {
                    b.proj(k_proj, norm1, wk, T, kvd_loc, d, 0, d, 0, kvd_loc);
                    k_proj_done = true;
                }
                const uint16_t k_rot = b.buffer((uint64_t)dh * T);
                b.op(DeviceOp::rope(k_rot, k_proj, rope_cs, dh / 2, T, kvl * dh, 0, 0, 1, kvd_loc, 2 * dh));
                const uint32_t slab = kvl * S * dh; // k_cache.sliceColumns(kv_h*max_seq, ...)
                if (kvq) { // k_cache.storeColumn(pos + j, k_rot[:, j]) (llama_inference.zig:300-320)
                    for (uint32_t j = 0; j < T; j++) {
                        dp.slice_assign_op_indices.push_back((uint32_t)dp.program.ops.size());
                        b.op(DeviceOp::kvq_store({kq_cache[kvl], k_rot, dh, kvq, S, j * dh, j, j, 1}));
                    }
                } else {
                    dp.slice_assign_op_indices.push_back((uint32_t)dp.program.ops.size());
                    b.op(DeviceOp::slice_assign(k_cache, k_rot, dh, T, slab, slab, 1, dh, 0, 1, dh, dh));
                }
                if (!v_proj_done) {
                    b.proj(v_proj, norm1, wv, T, kvd_loc, d, 0, d, 0, kvd_loc);
                    v_proj_done = true;
                }
                if (kvq) {
                    for (uint32_t j = 0; j < T; j++) {
                        dp.slice_assign_op_indices.push_back((uint32_t)dp.program.ops.size());
                        b.op(DeviceOp::kvq_store({vq_cache[kvl], v_proj, dh, kvq, S, j * kvd_loc + kvl * dh, j, j, 1}));
                    }
                } else {
                    dp.slice_assign_op_indices.push_back((uint32_t)dp.program.ops.size());
                    b.op(DeviceOp::slice_assign(v_cache, v_proj, dh, T, slab, slab, 1, dh, kvl * dh, 1, kvd_loc, dh));
                }
            }
            attn_out[hl] = b.buffer((uint64_t)dh * T);
            if (kvq) { // attentionQuantized over the kv head's caches
                zgml_op_attention_kvq a{};
                a.dst = attn_out[hl], a.q = q_rot, a.k = kq_cache[kvl], a.v = vq_cache[kvl], a.mask = dp.buf_attn_mask, a.has_mask = 1;
                a.d_head = dh, a.seq_q = T, a.seq_kv = S, a.scale = attn_scale, a.block_size = kvq, a.n_cols = S;
                a.k_col_start = 0, a.v_col_start = 0, a.q_off = 0, a.q_cs = dh, a.dst_off = 0, a.dst_cs = dh;
                a.mask_off = 0, a.mask_rs = 1, a.mask_cs = S;
                dp.attention_op_indices.push_back((uint32_t)dp.program.ops.size());
                b.op(DeviceOp::attention_kvq(a));
                continue;
            }
            zgml_op_attention a{};
            a.dst = attn_out[hl], a.q = q_rot, a.k = k_cache, a.v = v_cache, a.mask = dp.buf_attn_mask, a.has_mask = 1;
            a.d_head = dh, a.seq_q = T, a.seq_kv = S, a.scale = attn_scale;
            a.q_off = 0, a.k_off = kvl * S * dh, a.v_off = kvl * S * dh, a.mask_off = 0, a.dst_off = 0;
            a.q_rs = 1, a.q_cs = dh, a.k_rs = 1, a.k_cs = dh, a.v_rs = 1, a.v_cs = dh;
            a.mask_rs = 1, a.mask_cs = S, a.dst_rs = 1, a.dst_cs = dh;
            dp.attention_op_indices.push_back((uint32_t)dp.program.ops.size());
            b.op(DeviceOp::attention(a));
        }
        for (uint32_t hl = 0; hl < H_loc; hl++) { // sliceAssignRows(attn_out, h*d_head): patch_stride 0
            const uint32_t row = (h0 + hl) * dh;
            dp.slice_assign_op_indices.push_back((uint32_t)dp.program.ops.size());
            b.op(DeviceOp::slice_assign(attn_buf, attn_out[hl], dh, T, 0, row, 1, d, 0, 1, dh, 0));
        }
        if (sharded) dp.gather_points.push_back({(uint32_t)dp.program.ops.size(), attn_buf, 0, d_loc});
        const uint16_t attn_proj = b.buffer((uint64_t)d * T), after_attn = b.buffer((uint64_t)d * T);
        // sharded (T == 1 only): the rank's slice of the full vector; unsharded: dense [T, d]
        b.proj(attn_proj, attn_buf, wo, T, d_loc, d, 0, d, r * d_loc, sharded ? d_loc : d);
        if (sharded) dp.gather_points.push_back({(uint32_t)dp.program.ops.size(), attn_proj, 0, d_loc});
        b.op(DeviceOp::elementwise(ZGML_OP_ADD, after_attn, x, attn_proj, d * T));

        // ---- SwiGLU half (swigluFfn, llama_transformer.zig:129-133; silu = nn.zig:38-44)
        const uint16_t norm2 = b.rms_norm(after_attn, g2, d, c.rms_norm_eps, T);
        const uint32_t nf = ff_loc * T; // elements of one [d_ff, T] activation
        const uint16_t gate = b.buffer(nf), up = b.buffer(nf), silu = b.buffer(nf), act = b.buffer((uint64_t)ff * T);
        const uint16_t one = b.leaf(dp.scalar_one), one_rep = b.buffer(nf);
        b.proj(gate, norm2, wg, T, ff_loc, d, 0, d, 0, ff_loc);
        const uint32_t ne1[4] = {1, 1, 1, 1}, st1[4] = {1, 1, 1, 1}, nef[4] = {ff_loc, T, 1, 1}, stf[4] = {1, ff_loc, nf, nf};
        if (fused_elementwise) {
            const uint16_t exp_neg = b.buffer(nf);
            b.op(DeviceOp::fused_elementwise(b.steps({{ZGML_OP_NEG, 0, 0, 0, 0}, {ZGML_OP_EXP, 0, 0, 0, 0}}), 2, nf, exp_neg, gate));
            b.op(DeviceOp::repeat(one_rep, one, nf, ne1, nef, st1, stf));
            b.op(DeviceOp::fused_elementwise(
                b.steps({{ZGML_OP_ADD, 0, 0, one_rep, 0}, {ZGML_OP_RECIP, 0, 0, 0, 0}, {ZGML_OP_MUL, 1, 0, gate, 0}}), 3, nf,
                silu, exp_neg));
        } else { // appendElementwiseChainOps (device_inference.zig:373-397)
            const uint16_t t_neg = b.buffer(nf), t_exp = b.buffer(nf), t_add = b.buffer(nf), t_rec = b.buffer(nf);
            b.op(DeviceOp::elementwise(ZGML_OP_NEG, t_neg, gate, gate, nf));
            b.op(DeviceOp::elementwise(ZGML_OP_EXP, t_exp, t_neg, t_neg, nf));
            b.op(DeviceOp::repeat(one_rep, one, nf, ne1, nef, st1, stf));
            b.op(DeviceOp::elementwise(ZGML_OP_ADD, t_add, t_exp, one_rep, nf));
            b.op(DeviceOp::elementwise(ZGML_OP_RECIP, t_rec, t_add, t_add, nf));
            b.op(DeviceOp::elementwise(ZGML_OP_MUL, silu, gate, t_rec, nf));
        }
        b.proj(up, norm2, wu, T, ff_loc, d, 0, d, 0, ff_loc);
        b.op(DeviceOp::elementwise(ZGML_OP_MUL, act, silu, up, nf, r * ff_loc, 0, 0));
        if (sharded) dp.gather_points.push_back({(uint32_t)dp.program.ops.size(), act, 0, ff_loc});
        const uint16_t down = b.buffer((uint64_t)d * T), out = b.buffer((uint64_t)d * T);
        b.proj(down, act, wd, T, d_loc, ff, 0, ff, r * d_loc, sharded ? d_loc : d);
        if (sharded) dp.gather_points.push_back({(uint32_t)dp.program.ops.size(), down, 0, d_loc});
        b.op(DeviceOp::elementwise(ZGML_OP_ADD, out, after_attn, down, d * T));
        x = out;
    }

    // final norm + LM head (llama.zig:158-165)
    const uint16_t gf = b.leaf(model.gamma_f);
    const uint16_t normf = b.rms_norm(x, gf, d, c.rms_norm_eps, T);
    dp.buf_logits = b.buffer((uint64_t)c.vocab_size * T);
    if (c.tied_lm_head) { // x.matMul(false, token_embed, true): dense f32, B K-contiguous
        const uint16_t embed = b.leaf(model.token_embed);
        backend::MatMulGeometry g{};
        g.M = T, g.N = c.vocab_size, g.K = d, g.a_row_stride = d, g.a_col_stride = 1, g.b_row_stride = 1, g.b_col_stride = d;
        g.a_offset = 0, g.b_offset = 0, g.dst_offset = 0, g.dst_row_stride = c.vocab_size;
        b.op(DeviceOp::matmul(dp.buf_logits, normf, embed, g));
    } else {
        const uint32_t v_loc = c.vocab_size / ws;
        b.proj(dp.buf_logits, normf, (uint16_t)model.lm_head_index(), T, v_loc, d, 0, d, r * v_loc,
                               sharded ? v_loc : c.vocab_size);
        if (sharded) dp.gather_points.push_back({(uint32_t)dp.program.ops.size(), dp.buf_logits, 0, v_loc});
    }
    dp.program.n_buffers = (uint16_t)dp.program.buffer_sizes.size();

    // per-step I/O (device_inference.zig:159-177): token_input, attn_mask, every layer's rope leaf
    auto io = [](uint16_t buf, std::vector<float>& v) {
        ProgramIO p{};
        p.buf_idx = buf, p.offset = 0, p.host_ptr = v.data(), p.size = (uint32_t)(v.size() * sizeof(float));
        return p;
    };
    dp.step_inputs.push_back(io(dp.buf_token_input, dp.token_input));
    dp.step_inputs.push_back(io(dp.buf_attn_mask, dp.attn_mask));
    for (uint32_t l = 0; l < c.n_layers; l++) dp.step_inputs.push_back(io(dp.buf_rope[l], dp.rope_leaf[l]));
    // Last-column logits: [vocab, T] column-major -> last col at offset (T-1)*vocab (llama_inference.zig:463-465)
    dp.logits_host.assign(c.vocab_size, 0.f);
    backend::ProgramIO out_io = io(dp.buf_logits, dp.logits_host);
    out_io.offset = (uint32_t)((size_t)(T - 1) * c.vocab_size * sizeof(float));
    dp.step_outputs.push_back(out_io);
    return dpp;
}

void patch_step(const LlamaModel& model, DecodeProgram& dp, uint32_t token, uint32_t pos) { patch_tokens(model, dp, &token, pos); }

// LlamaInferencePlan.execute steps 1-4 for token_len tokens at positions [pos, pos + T)
void patch_tokens(const LlamaModel& model, DecodeProgram& dp, const uint32_t* tokens, uint32_t pos) {
    const LlamaConfig& c = model.cfg;
    const uint32_t d = c.d_model, dh = c.d_head(), S = c.max_seq_len, T = dp.token_len;
    const float ninf = -std::numeric_limits<float>::infinity();
    for (uint32_t j = 0; j < T; j++) {
        std::memcpy(dp.token_input.data() + (size_t)j * d, model.token_embed.data() + (size_t)tokens[j] * d, d * sizeof(float));
        float* col = dp.attn_mask.data() + (size_t)j * S;
        const uint32_t valid_upto = pos + j + 1;
        std::fill(col, col + valid_upto, 0.f);
        if (valid_upto < S) std::fill(col + valid_upto, col + S, ninf);
    }
    for (auto& leaf : dp.rope_leaf)
        for (uint32_t j = 0; j < T; j++) {
            std::memcpy(leaf.data() + (size_t)j * 2 * dh, model.cos_table.data() + (size_t)(pos + j) * dh, dh * sizeof(float));
            std::memcpy(leaf.data() + (size_t)j * 2 * dh + dh, model.sin_table.data() + (size_t)(pos + j) * dh, dh * sizeof(float));
        }
    dp.patchSliceAssignOffset(pos);
    dp.patchAttentionSeqKV(pos + T);
}

bool LlamaDeviceSession::init(const LlamaModel& m, const backend::Backend& backend, bool include_dead_f32) {
    model = &m;
    be = backend;
    const bool fuse = be.capabilities.fused_elementwise &&
                      (!be.capabilities.max_fused_elementwise_steps || *be.capabilities.max_fused_elementwise_steps >= 3);
    dp = build_decode_program(m, fuse, include_dead_f32, 1);
    if (!be.supportsProgram(dp->program)) return false; // error.UnsupportedDeviceOp
    handle = be.compileProgram(dp->program);
    return handle != nullptr; // error.CompileFailed
}

void LlamaDeviceSession::deinit() {
    if (handle) be.freeProgram(handle);
    handle = nullptr;
    dp.reset();
}

const float* LlamaDeviceSession::step(uint32_t token, uint32_t pos) {
    patch_step(*model, *dp, token, pos);
    be.refreshProgram(handle, dp->program.ops.data(), dp->program.ops.size());
    be.executeProgram(handle, dp->step_inputs.data(), dp->step_inputs.size(), dp->step_outputs.data(), dp->step_outputs.size());
    return dp->logits_host.data();
}

// first maximum wins (src/nn.zig:122-138): `logits[c] > best` starting from logits[0], so a NaN never wins and a NaN in front
// keeps index 0
static uint32_t argmax_scalar(const float* logits, uint32_t n) {
    uint32_t best = 0;
    float best_val = logits[0];
    for (uint32_t c = 1; c < n; c++)
        if (logits[c] > best_val) {
            best_val = logits[c];
            best = c;
        }
    return best;
}
#if defined(__x86_64__)
// The same index from two vector passes: the maximum (any NaN anywhere -> the scalar loop decides), then the first element equal
// to it. The scalar loop's compare-and-keep chain costs 35-70 us per 49152 logits — more than everything else the host does
// per token of the drop-in path together; this is ~10 us.
__attribute__((target("avx2"))) static uint32_t argmax_avx2(const float* x, uint32_t n) {
    __m256 m = _mm256_loadu_ps(x);
    __m256 unord = _mm256_cmp_ps(m, m, _CMP_UNORD_Q);
    uint32_t i = 8;
    for (; i + 8 <= n; i += 8) {
        const __m256 v = _mm256_loadu_ps(x + i);
        unord = _mm256_or_ps(unord, _mm256_cmp_ps(v, v, _CMP_UNORD_Q));
        m = _mm256_max_ps(m, v);
    }
    if (_mm256_movemask_ps(unord)) return argmax_scalar(x, n);
    float t[8];
    _mm256_storeu_ps(t, m);
    float mx = t[0];
    for (int k = 1; k < 8; k++)
        if (t[k] > mx) mx = t[k];
    for (; i < n; i++) {
        if (x[i] != x[i]) return argmax_scalar(x, n);
        if (x[i] > mx) mx = x[i];
    }
    const __m256 mv = _mm256_set1_ps(mx);
    for (i = 0; i + 8 <= n; i += 8) {
        const int k = _mm256_movemask_ps(_mm256_cmp_ps(_mm256_loadu_ps(x + i), mv, _CMP_EQ_OQ));
        if (k) return i + (uint32_t)__builtin_ctz((unsigned)k);
    }
    for (; i < n; i++)
        if (x[i] == mx) return i;
    return 0;
}
#endif
uint32_t LlamaDeviceSession::argmax(const float* logits, uint32_t n) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n >= 64) return argmax_avx2(logits, n);
#endif
    return argmax_scalar(logits, n);
}
uint32_t LlamaDeviceSession::argmax_reference(const float* logits, uint32_t n) { return argmax_scalar(logits, n); }

} // namespace zgml::llama
