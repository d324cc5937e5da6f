// runtime.hip — the C ABI of include/zgml_hip.h: context, program compile / refresh / execute,
// host dense-matmul override, profiling counters and the measurement extensions.
//
// Execution model (MI355X-first, not a translation of the Metal/WGPU backends):
//   * one HIP stream per context; a compiled program is a fixed launch list captured once into a
//     hipGraph and replayed per token (decode is launch-bound: ~1.7k DeviceOps per SmolLM token);
//   * the two per-step dynamic fields of the reference (slice_assign.dst_offset and
//     attention.seq_kv, src/device_inference.zig:242-256) live in a device-resident word per op
//     ("dyn block") that kernels dereference, so refresh_program never re-records the graph;
//   * per-step inputs are packed into one pinned staging buffer, moved with ONE H2D copy and
//     scattered on the device; outputs are gathered the same way (the reference's per-token
//     traffic is 32 small uploads + one logits download, src/llama_inference.zig:405-466);
//   * quantized weights are re-packed on the device at compile time (qmatvec.hip) and buffers no
//     op references (the dead f32 master copies of quantized weights) are neither
//     allocated nor uploaded.
// There is no CPU fallback anywhere: every DeviceOp kind has a kernel.
#define ZGML_RUNTIME_MAIN
#include "runtime_internal.h"

namespace zgml_rt {
std::string g_create_error;
}

namespace {


// ── Capabilities.hip ────────────────────────────────────────────────────────────────────────
void fill_caps(zgml_capabilities* c) {
    memset(c, 0, sizeof(*c));
    c->compiled_programs = 1;
    c->host_visible_program_memory = 0; // discrete HBM: no per-op CPU fallback is possible
    c->dense_matmul_f32 = 1;
    c->dense_matmul_f16 = 1;
    c->qmatmul = 1;
    c->fused_elementwise = 1;
    c->max_fused_elementwise_steps_has = 1;
    c->max_fused_elementwise_steps = kMaxFusedSteps;
    c->f16_weight_promotion = 1; // opt-in via ZGML_HIP_OPT_F16_DENSE_WEIGHTS
    c->dynamic_program_refresh = 1;
    c->prefill_attention = 1;
    c->decode_attention = 1;
    c->quantized_kv = 1; // extension ops kvq_store / attention_kvq (include/zgml_hip.h)
    c->command_buffer_execution = 1;
    c->attention_supported = 1;
    c->attention_max_seq_kv_has = 0; // online softmax over key tiles: no score-buffer cap
    c->attention_max_d_head_has = 1;
    c->attention_max_d_head = 512;
}

// packed-GGUF pass-through (include/zgml_hip.h): 0 = reference form, 1 = Q4_0 blocks, 2 = Q8_0 blocks
int gguf_form(const zgml_qweight_upload& qw) {
    if (qw.scales || qw.scales_len || qw.block_size != 32 || !qw.data) return 0;
    const uint64_t n = qw.rows * qw.cols;
    if (!n || n % 32 || qw.cols % 32 || !qweight_packable(qw.rows, qw.cols, 32)) return 0;
    if (qw.data_len == n / 32 * ZGML_QW_GGUF_Q4_0_BLOCK_BYTES) return 1;
    if (qw.data_len == n / 32 * ZGML_QW_GGUF_Q8_0_BLOCK_BYTES) return 2;
    return 0;
}

bool elementwise_op_ok(uint32_t op) { return op >= ZGML_OP_ADD && op <= ZGML_OP_GELU; }

// buffer ids an op touches (opBuffersValid, src/backend.zig:303-325)
void op_buffers(const zgml_device_op& op, std::vector<uint16_t>& out) {
    switch (op.kind) {
        case ZGML_DOP_ELEMENTWISE:
            out.insert(out.end(), {op.u.elementwise.dst, op.u.elementwise.src0, op.u.elementwise.src1});
            break;
        case ZGML_DOP_MATMUL: out.insert(out.end(), {op.u.matmul.dst, op.u.matmul.a, op.u.matmul.b}); break;
        case ZGML_DOP_QMATMUL: out.insert(out.end(), {op.u.qmatmul.dst, op.u.qmatmul.input}); break;
        case ZGML_DOP_SOFTMAX:
        case ZGML_DOP_LAYERNORM:
        case ZGML_DOP_RMSNORM: out.insert(out.end(), {op.u.softmax.dst, op.u.softmax.src}); break;
        case ZGML_DOP_REDUCE: out.insert(out.end(), {op.u.reduce.dst, op.u.reduce.src}); break;
        case ZGML_DOP_REPEAT: out.insert(out.end(), {op.u.repeat.dst, op.u.repeat.src}); break;
        case ZGML_DOP_SLICE_ASSIGN: out.insert(out.end(), {op.u.slice_assign.dst, op.u.slice_assign.src}); break;
        case ZGML_DOP_ROPE: out.insert(out.end(), {op.u.rope.dst, op.u.rope.src, op.u.rope.cos_sin}); break;
        case ZGML_DOP_ATTENTION:
            out.insert(out.end(), {op.u.attention.dst, op.u.attention.q, op.u.attention.k, op.u.attention.v,
                                   op.u.attention.mask});
            break;
        case ZGML_DOP_FUSED_ELEMENTWISE: {
            const auto& fe = op.u.fused_elementwise;
            out.insert(out.end(), {fe.dst, fe.src});
            for (uint32_t s = 0; s < fe.n_steps; s++)
                if (fe.steps[s].op == ZGML_OP_ADD || fe.steps[s].op == ZGML_OP_MUL) out.push_back(fe.steps[s].secondary_buf);
            break;
        }
        case ZGML_DOP_KVQ_STORE: out.insert(out.end(), {op.u.kvq_store.cache, op.u.kvq_store.src}); break;
        case ZGML_DOP_ATTENTION_KVQ:
            out.insert(out.end(), {op.u.attention_kvq.dst, op.u.attention_kvq.q, op.u.attention_kvq.k, op.u.attention_kvq.v,
                                   op.u.attention_kvq.mask});
            break;
        default: break;
    }
}

// DeviceProgram.isSupportedBy(Capabilities.hip), src/backend.zig:277-297
bool program_supported(const zgml_device_program* pr) {
    if (!pr) return false;
    if ((uint64_t)pr->n_buffers != pr->n_buffer_sizes) return false;
    std::vector<uint16_t> ids;
    for (uint64_t i = 0; i < pr->n_ops; i++) {
        const zgml_device_op& op = pr->ops[i];
        if (op.kind == ZGML_DOP_FUSED_ELEMENTWISE && op.u.fused_elementwise.n_steps && !op.u.fused_elementwise.steps) return false;
        if (op.kind == ZGML_DOP_FUSED_ELEMENTWISE && op.u.fused_elementwise.n_steps > (uint32_t)kMaxFusedSteps) return false;
        ids.clear(); // buffer ids first: the per-kind checks below index buffer_sizes with them
        op_buffers(op, ids);
        for (uint16_t id : ids)
            if ((uint64_t)id >= pr->n_buffer_sizes) return false;
        switch (op.kind) {
            case ZGML_DOP_ELEMENTWISE:
                if (!elementwise_op_ok(op.u.elementwise.op)) return false;
                break;
            case ZGML_DOP_MATMUL:
            case ZGML_DOP_SOFTMAX:
            case ZGML_DOP_LAYERNORM:
            case ZGML_DOP_RMSNORM:
            case ZGML_DOP_REPEAT:
            case ZGML_DOP_SLICE_ASSIGN:
            case ZGML_DOP_ROPE: break;
            case ZGML_DOP_QMATMUL: {
                const auto& q = op.u.qmatmul;
                if ((uint64_t)q.weight_idx >= pr->n_qweights) return false;
                const zgml_qweight_upload& qw = pr->qweights[q.weight_idx];
                if (qw.block_size == 0) return false;
                if (qw.rows != q.K || qw.cols != q.N) return false;
                const uint64_t n_elems = (uint64_t)q.K * q.N;
                const uint64_t n_blocks = (n_elems + qw.block_size - 1) / qw.block_size;
                if (gguf_form(qw)) break; // packed-GGUF pass-through, validated by gguf_form()
                if (qw.data_len < n_elems || qw.scales_len < n_blocks) return false;
                break;
            }
            case ZGML_DOP_REDUCE:
                if (op.u.reduce.op != ZGML_OP_SUM && op.u.reduce.op != ZGML_OP_MAX) return false;
                break;
            case ZGML_DOP_ATTENTION:
                if (op.u.attention.d_head > 512) return false;
                break;
            case ZGML_DOP_KVQ_STORE: { // extension ops: quantised KV cache
                const auto& st = op.u.kvq_store;
                if (!st.block_size || st.block_size % 4 || st.d_head % st.block_size || ((uint64_t)st.n_cols * st.d_head) % 4) return false;
                if ((uint64_t)st.n_cols * st.d_head / 4 + (uint64_t)st.n_cols * (st.d_head / st.block_size) > pr->buffer_sizes[st.cache]) return false;
                break;
            }
            case ZGML_DOP_ATTENTION_KVQ: {
                const auto& a = op.u.attention_kvq;
                if (!a.block_size || a.block_size % 4 || a.d_head % a.block_size || ((uint64_t)a.n_cols * a.d_head) % 4) return false;
                if (a.d_head < 16 || a.d_head > 256 || (a.d_head & (a.d_head - 1))) return false; // kernel instances: 16..256, power of two
                if (a.q_off % 4 || a.q_cs % 4 || a.dst_off % 4 || a.dst_cs % 4) return false;    // float4 access to q / dst
                const uint64_t need = (uint64_t)a.n_cols * a.d_head / 4 + (uint64_t)a.n_cols * (a.d_head / a.block_size);
                if (need > pr->buffer_sizes[a.k] || need > pr->buffer_sizes[a.v]) return false;
                break;
            }
            case ZGML_DOP_FUSED_ELEMENTWISE: {
                const auto& fe = op.u.fused_elementwise;
                if (fe.n_steps > (uint32_t)kMaxFusedSteps) return false;
                for (uint32_t s = 0; s < fe.n_steps; s++)
                    if (!elementwise_op_ok(fe.steps[s].op)) return false;
                break;
            }
            default: return false;
        }
    }
    return true;
}

// ── small device helpers ────────────────────────────────────────────────────────────────────
// (blockIdx.y strides over a row: a prefill chunk's token rows are one transfer of 512 KB — one workgroup walking that alone took
// 200 us from device memory and 680 us from mapped host memory)
__global__ void scatter_words_kernel(const IoTableDev* table, const uint32_t* stage) {
    const IoTableDev e = table[blockIdx.x];
    uint32_t* dst = (uint32_t*)e.dev;
    const uint32_t* src = stage + e.stage_off_words;
    for (uint32_t i = blockIdx.y * blockDim.x + threadIdx.x; i < e.n_words; i += blockDim.x * gridDim.y) dst[i] = src[i];
}
// (blockIdx.y strides over a row: the logits of a decode step are one row of ~50k words)
__global__ void gather_words_wide_kernel(const IoTableDev* table, uint32_t* stage) {
    const IoTableDev e = table[blockIdx.x];
    const uint32_t* src = (const uint32_t*)e.dev;
    uint32_t* dst = stage + e.stage_off_words;
    for (uint32_t i = blockIdx.y * blockDim.x + threadIdx.x; i < e.n_words; i += blockDim.x * gridDim.y) dst[i] = src[i];
}

uint32_t io_grid_y(uint32_t max_row_words) { return std::max<uint32_t>(1, std::min<uint32_t>(64, max_row_words / 1024)); } // workgroups per transfer row

void free_resident_graph(zgml_hip_program* p); // below (zgml_resident is defined there)

// Diagnostics: ZGML_HIP_GRAPH_DUMP=<dir> writes <dir>/<tag>.dot (hipGraphDebugDotPrint) and prints the node-type
// histogram of every graph the runtime instantiates (how the rocprofv3 crash inside hipGraphLaunch of the per-token
// graph was narrowed down: DESIGN.md section 5).
void dump_graph(hipGraph_t g, const char* tag) {
    const char* dir = getenv("ZGML_HIP_GRAPH_DUMP");
    if (!dir || !g) return;
    size_t n = 0;
    if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) return;
    std::vector<hipGraphNode_t> nodes(n);
    if (n && hipGraphGetNodes(g, nodes.data(), &n) != hipSuccess) return;
    std::map<int, size_t> hist;
    size_t max_shmem = 0, n_kernel = 0;
    for (hipGraphNode_t nd : nodes) {
        hipGraphNodeType t;
        if (hipGraphNodeGetType(nd, &t) != hipSuccess) continue;
        hist[(int)t]++;
        if (t == hipGraphNodeTypeKernel) {
            hipKernelNodeParams kp{};
            if (hipGraphKernelNodeGetParams(nd, &kp) == hipSuccess) max_shmem = std::max<size_t>(max_shmem, kp.sharedMemBytes), n_kernel++;
        }
    }
    fprintf(stderr, "[zgml_hip] graph %s: %zu nodes;", tag, n);
    for (auto& kv : hist) fprintf(stderr, " type%d=%zu", kv.first, kv.second);
    fprintf(stderr, " (kernel=%d memcpy=%d memset=%d host=%d empty=%d event_record=%d wait_event=%d); max dynamic LDS %zu B over %zu kernel nodes\n",
            (int)hipGraphNodeTypeKernel, (int)hipGraphNodeTypeMemcpy, (int)hipGraphNodeTypeMemset, (int)hipGraphNodeTypeHost,
            (int)hipGraphNodeTypeEmpty, (int)hipGraphNodeTypeEventRecord, (int)hipGraphNodeTypeWaitEvent, max_shmem, n_kernel);
    const std::string path = std::string(dir) + "/" + tag + ".dot";
    hipGraphDebugDotPrint(g, path.c_str(), 0);
}

// Drop every captured graph of the program: both bake the plan's kernel nodes and the device parameter arrays
// build_plan() is about to free, so a plan rebuild must never leave one behind (the resident graph included).
void free_io_graph(zgml_hip_program* p) {
    if (p->io_graph_exec) hipGraphExecDestroy(p->io_graph_exec);
    if (p->io_graph) hipGraphDestroy(p->io_graph);
    p->io_graph_exec = nullptr, p->io_graph = nullptr;
}

void free_graph(zgml_hip_program* p) {
    free_io_graph(p);
    if (p->graph_exec) hipGraphExecDestroy(p->graph_exec);
    if (p->graph) hipGraphDestroy(p->graph);
    p->graph_exec = nullptr;
    p->graph = nullptr;
    if (p->graph_tail_exec) hipGraphExecDestroy(p->graph_tail_exec);
    if (p->graph_tail) hipGraphDestroy(p->graph_tail);
    p->graph_tail_exec = nullptr, p->graph_tail = nullptr;
    if (p->shard_graph_exec) hipGraphExecDestroy(p->shard_graph_exec);
    if (p->shard_graph) hipGraphDestroy(p->shard_graph);
    p->shard_graph_exec = nullptr, p->shard_graph = nullptr;
    free_resident_graph(p);
}

bool ensure_stage(zgml_hip_program* p, uint64_t bytes) {
    if (bytes <= p->stage_cap) return true;
    zgml_hip_ctx* ctx = p->ctx;
    hipStreamSynchronize(ctx->stream);
    free_io_graph(p); // (its kernels hold the buffers' addresses)
    if (p->stage_host) hipHostFree(p->stage_host);
    if (p->stage_out_host) hipHostFree(p->stage_out_host);
    if (p->stage_dev) hipFree(p->stage_dev);
    p->stage_host = p->stage_out_host = p->stage_dev = nullptr;
    uint64_t cap = 1 << 16;
    while (cap < bytes) cap <<= 1;
    if (!CTX_CHECK(ctx, hipHostMalloc(&p->stage_host, cap, hipHostMallocMapped))) return false;
    if (!CTX_CHECK(ctx, hipHostMalloc(&p->stage_out_host, cap, hipHostMallocMapped))) return false;
    if (!CTX_CHECK(ctx, hipMalloc(&p->stage_dev, cap))) return false;
    p->stage_cap = cap;
    return true;
}

// (re)build the cached transfer table when the descriptor list changed
bool prepare_io(zgml_hip_program* p, IoPlan& plan, const zgml_program_io* ios, uint64_t n) {
    if (&plan == &p->in_plan && p->hoist_ok)
        for (uint64_t i = 0; i < n; i++)
            if (ios[i].buf_idx < p->hoist_guard.size() && p->hoist_guard[ios[i].buf_idx]) { // the host writes what a hoisted repeat read or wrote
                p->hoist_ok = false;
                p->plan_dirty = true;
                break;
            }
    bool same = plan.entries.size() == n;
    for (uint64_t i = 0; same && i < n; i++)
        same = plan.entries[i] == IoEntry{ios[i].buf_idx, ios[i].offset, ios[i].size};
    if (same) return true;
    zgml_hip_ctx* ctx = p->ctx;
    hipStreamSynchronize(ctx->stream);
    free_io_graph(p); // (captured for the tables that are about to change)
    plan.entries.clear();
    plan.word_aligned = true;
    plan.total_words = 0, plan.max_row_words = 0;
    if (plan.table_dev) hipFree(plan.table_dev);
    plan.table_dev = nullptr;
    std::vector<IoTableDev> table;
    for (uint64_t i = 0; i < n; i++) {
        const zgml_program_io& io = ios[i];
        if (io.buf_idx >= p->bufs.size()) {
            ctx->fail("program I/O names buffer " + std::to_string(io.buf_idx) + " which does not exist");
            return false;
        }
        if (!p->bufs[io.buf_idx]) {
            ctx->fail("program I/O names buffer " + std::to_string(io.buf_idx) +
                      " which has no f32 image on the device: no op references it and it was elided at compile"
                      " time (ZGML_HIP_OPT_SKIP_DEAD_UPLOADS=0 keeps it), or it is a matmul weight promoted to f16");
            return false;
        }
        if ((uint64_t)io.offset + io.size > p->sizes[io.buf_idx] * sizeof(float)) {
            ctx->fail("program I/O out of range for buffer " + std::to_string(io.buf_idx));
            return false;
        }
        plan.entries.push_back({io.buf_idx, io.offset, io.size});
        if ((io.offset & 3) || (io.size & 3)) plan.word_aligned = false;
        table.push_back({(float*)((char*)p->bufs[io.buf_idx] + io.offset), plan.total_words, io.size / 4});
        plan.max_row_words = std::max<uint32_t>(plan.max_row_words, io.size / 4);
        plan.total_words += (io.size + 3) / 4;
    }
    // the input table carries one more row: the program's dynamic words (one per op), so that a refresh's changes ride in the same
    // staging copy and scatter launch as the inputs instead of a transfer of their own (upload_inputs)
    const bool with_dyn = &plan == &p->in_plan && plan.word_aligned && !table.empty() && p->dyn_dev && !p->ops.empty();
    if (with_dyn) table.push_back({(float*)p->dyn_dev, plan.total_words, (uint32_t)p->ops.size()}), plan.max_row_words = std::max<uint32_t>(plan.max_row_words, (uint32_t)p->ops.size());
    if (plan.word_aligned && !table.empty()) {
        if (!CTX_CHECK(ctx, hipMalloc((void**)&plan.table_dev, table.size() * sizeof(IoTableDev)))) return false;
        if (!CTX_CHECK(ctx, h2d_sync(ctx->stream, plan.table_dev, table.data(), table.size() * sizeof(IoTableDev)))) return false;
    }
    plan.dyn_row = with_dyn;
    return ensure_stage(p, ((uint64_t)plan.total_words + (with_dyn ? p->ops.size() : 0)) * 4);
}

// ── plan building ───────────────────────────────────────────────────────────────────────────
// Serial mode: one launch per DeviceOp in program order. Batched mode (ZGML_HIP_OPT_FUSION, the
// default): ops are grouped by dependency level (schedule.h); inside a level every rope /
// slice_assign / attention / repeat goes into ONE launch of its kind (parameters in a device
// array, blockIdx.y = op), the rest launch one by one.
float* buf_at(zgml_hip_program* p, uint16_t idx, uint64_t off) { return p->bufs[idx] + off; }

template <typename T>
const T* upload_params(zgml_hip_program* p, const std::vector<T>& v) {
    void* d = nullptr;
    if (hipMalloc(&d, v.size() * sizeof(T)) != hipSuccess) {
        p->ctx->fail("plan: parameter array allocation failed");
        return nullptr;
    }
    if (h2d_sync(p->ctx->stream, d, v.data(), v.size() * sizeof(T)) != hipSuccess) p->ctx->fail("plan: parameter array upload failed");
    p->param_blobs.push_back(d);
    return (const T*)d;
}

// Split policy of one decode-attention launch (kernels.h AttnSplit): up to 16 workgroups per head, one
// per `min_keys` keys of the longest context the plan allows, the whole launch within ~256 workgroups.
// ZGML_HIP_OPT_ATTN_SPLIT_MIN_KEYS (or the ZGML_HIP_ATTN_SPLIT_MIN_KEYS environment variable) moves the
// threshold (>= 32, 0 = off); ZGML_HIP_ATTN_SPLIT caps the workgroups per head (<= 1 = off).
// `default_min_keys`: 128 — except where a workgroup covers few keys per memory round trip: the 4-wave form of the f32-KV decode
// attention at d_head 128 (stand-alone and inside the K-on-lanes fused launch) streams 32 keys per round, so a head waits four
// dependent rounds per 128 keys; splitting from 64 keys on gave Llama-2-7B +4 % over positions 200..328 (760 -> 791 tok/s) at no
// cost at positions 4..132 and 1900 (profiles/r05_split_sweep_7b.txt; 32 is worse again: the merge costs more than a round).
AttnSplit attn_split_for(zgml_hip_program* p, uint32_t n_heads, uint32_t d_head, uint32_t max_kv, int default_min_keys = 128) {
    static const int want = getenv("ZGML_HIP_ATTN_SPLIT") ? atoi(getenv("ZGML_HIP_ATTN_SPLIT")) : 16;
    static const int min_keys_env = getenv("ZGML_HIP_ATTN_SPLIT_MIN_KEYS") ? atoi(getenv("ZGML_HIP_ATTN_SPLIT_MIN_KEYS")) : -1;
    AttnSplit sp;
    const int64_t min_keys = p->ctx->opt_attn_split_min_keys >= 0 ? p->ctx->opt_attn_split_min_keys : (min_keys_env >= 0 ? min_keys_env : default_min_keys);
    if (min_keys == 0 || want <= 1) return sp;
    sp.min_keys = (uint32_t)std::min<int64_t>(std::max<int64_t>(32, min_keys), 1 << 30);
    uint32_t S = std::min<uint32_t>((uint32_t)std::max(want, 1), max_kv / sp.min_keys);
    static const uint32_t wg_cap = getenv("ZGML_HIP_ATTN_SPLIT_WGS") ? (uint32_t)atoi(getenv("ZGML_HIP_ATTN_SPLIT_WGS")) : 256u;
    S = std::min(S, std::max(1u, wg_cap / std::max(n_heads, 1u))); // about one (1024-thread) workgroup per CU: idle ones still cost dispatch
    if (S <= 1) return sp;
    const uint64_t need = (uint64_t)n_heads * S * (d_head + 4);
    if (need > p->split_buf_floats) { // earlier launches keep their (smaller) block: it stays in param_blobs
        void* d = nullptr;
        if (hipMalloc(&d, need * sizeof(float)) != hipSuccess) return sp;
        p->param_blobs.push_back(d);
        p->split_buf = (float*)d, p->split_buf_floats = need;
    }
    if (n_heads > p->split_cnt_words) {
        void* d = nullptr;
        const size_t bytes = ((size_t)n_heads * 4 + 255) / 256 * 256;
        if (hipMalloc(&d, bytes) != hipSuccess || memset_sync(p->ctx->stream, d, 0, bytes) != hipSuccess) return sp;
        p->param_blobs.push_back(d);
        p->split_cnt = (uint32_t*)d, p->split_cnt_words = n_heads;
    }
    sp.splits = S, sp.buf = p->split_buf, sp.cnt = p->split_cnt;
    return sp;
}

RepeatParams make_repeat(zgml_hip_program* p, const zgml_op_repeat& r) {
    RepeatParams rp{};
    rp.dst = p->bufs[r.dst];
    rp.src = p->bufs[r.src];
    rp.n = r.n;
    for (int d = 0; d < 4; d++) {
        rp.src_ne[d] = r.src_ne[d] ? r.src_ne[d] : 1;
        rp.src_strides[d] = r.src_strides[d];
        rp.dst_strides[d] = r.dst_strides[d] ? r.dst_strides[d] : 1;
    }
    rp.src_offset = r.src_offset, rp.dst_offset = r.dst_offset;
    const uint64_t src_n = (uint64_t)r.src_ne[0] * r.src_ne[1] * r.src_ne[2] * r.src_ne[3];
    rp.src_n = (uint32_t)src_n;
    // the reference's fast paths (reference.zig:401-419), in its order of precedence
    if (src_n == 1)
        rp.mode = 1;
    else if (src_n >= r.n)
        rp.mode = 2;
    else if (r.n % src_n == 0 && r.src_strides[0] == 1 && (r.src_ne[1] <= 1 || r.src_strides[1] == r.src_ne[0]) &&
             (r.src_ne[2] <= 1 || r.src_strides[2] == r.src_ne[0] * r.src_ne[1]) &&
             (r.src_ne[3] <= 1 || r.src_strides[3] == r.src_ne[0] * r.src_ne[1] * r.src_ne[2]))
        rp.mode = 3;
    else
        rp.mode = 0;
    return rp;
}

SliceAssignParams make_slice_assign(zgml_hip_program* p, const zgml_op_slice_assign& sa, size_t op_index) {
    SliceAssignParams sp{};
    sp.dst = p->bufs[sa.dst];
    sp.src = buf_at(p, sa.src, sa.src_offset);
    sp.rows = sa.rows, sp.cols = sa.cols;
    sp.dst_row_stride = sa.dst_row_stride, sp.dst_col_stride = sa.dst_col_stride;
    sp.src_row_stride = sa.src_row_stride, sp.src_col_stride = sa.src_col_stride;
    sp.dyn_dst_offset = p->dyn_dev + op_index;
    return sp;
}

RopeParams make_rope(zgml_hip_program* p, const zgml_op_rope& r) {
    RopeParams rp{};
    rp.dst = buf_at(p, r.dst, r.dst_off);
    rp.src = buf_at(p, r.src, r.src_off);
    rp.cs = buf_at(p, r.cos_sin, r.cs_off);
    rp.half_d = r.half_d, rp.seq_len = r.seq_len, rp.src_rs = r.src_rs, rp.src_cs = r.src_cs, rp.cs_cs = r.cs_cs;
    rp.dst2 = nullptr, rp.dyn_dst2_off = nullptr, rp.d2_rs = rp.d2_cs = 0;
    return rp;
}

AttentionParams make_attention(zgml_hip_program* p, const zgml_op_attention& a, size_t op_index) {
    AttentionParams ap{};
    ap.dst = buf_at(p, a.dst, a.dst_off);
    ap.q = buf_at(p, a.q, a.q_off);
    ap.k = buf_at(p, a.k, a.k_off);
    ap.v = buf_at(p, a.v, a.v_off);
    ap.mask = a.has_mask ? buf_at(p, a.mask, a.mask_off) : nullptr;
    ap.d_head = a.d_head, ap.seq_q = a.seq_q;
    ap.dyn_seq_kv = p->dyn_dev + op_index;
    ap.scale = a.scale;
    ap.q_rs = a.q_rs, ap.q_cs = a.q_cs, ap.k_rs = a.k_rs, ap.k_cs = a.k_cs, ap.v_rs = a.v_rs, ap.v_cs = a.v_cs;
    ap.mask_rs = a.mask_rs, ap.mask_cs = a.mask_cs, ap.dst_rs = a.dst_rs, ap.dst_cs = a.dst_cs;
    ap.dst2 = nullptr, ap.dyn_dst2_off = nullptr, ap.d2_rs = ap.d2_cs = 0;
    return ap;
}

// launch for a single op of a kind that is never batched
bool make_single(zgml_hip_program* p, size_t i, Launch& L) {
    const zgml_device_op& op = p->ops[i];
    L.kind = op.kind, L.n_ops = 1, L.op_lo = L.op_hi = (uint32_t)i;
    switch (op.kind) {
        case ZGML_DOP_ELEMENTWISE: {
            const auto e = op.u.elementwise;
            float* dst = buf_at(p, e.dst, e.dst_offset);
            const float* s0 = buf_at(p, e.src0, e.src0_offset);
            const float* s1 = buf_at(p, e.src1, e.src1_offset);
            L.run = [=](hipStream_t s) { launch_elementwise(s, e.op, dst, s0, s1, e.n); };
            return true;
        }
        case ZGML_DOP_MATMUL: {
            const auto m = op.u.matmul;
            DenseMatmulParams dp{};
            dp.dst = buf_at(p, m.dst, m.geom.dst_offset);
            dp.a = buf_at(p, m.a, m.geom.a_offset);
            dp.b = buf_at(p, m.b, m.geom.b_offset);
            dp.M = (uint32_t)m.geom.M, dp.N = (uint32_t)m.geom.N, dp.K = (uint32_t)m.geom.K;
            dp.a_rs = (uint32_t)m.geom.a_row_stride, dp.a_cs = (uint32_t)m.geom.a_col_stride;
            dp.b_rs = (uint32_t)m.geom.b_row_stride, dp.b_cs = (uint32_t)m.geom.b_col_stride;
            dp.dst_rs = (uint32_t)m.geom.dst_row_stride;
            dp.b_f16 = 0;
            if (m.b < p->f16_weights.size() && p->f16_weights[m.b]) { // promoted at compile time
                DenseF16Params fp{dp.dst, dp.a, p->f16_weights[m.b], dp.M, dp.N, dp.K, dp.a_rs, dp.dst_rs, p->f16_stream_nt ? 1u : 0u};
                if (p->scratch && dense_f16_scratch_bytes(dp.M, dp.K)) { // same adjacency rule as the quantized split below
                    const uint64_t pos = p->plan.size();
                    fp.scratch = p->scratch;
                    fp.reuse_a = p->split_pos + 1 == pos && p->split_input == dp.a && p->split_M == dp.M && p->split_K == dp.K &&
                                 p->split_in_rs == dp.a_rs && p->split_kind == 2;
                    // ... or the launch that produced the rows wrote the operand itself (as for the quantized split below)
                    static const bool fuse_pack = !(getenv("ZGML_HIP_FUSE_SPLIT") && atoi(getenv("ZGML_HIP_FUSE_SPLIT")) == 0);
                    if (!fp.reuse_a && fuse_pack && dense_f16_a_unpadded(dp.M, dp.K) && dp.a_rs == dp.K) {
                        for (size_t back = p->plan.size(); back-- > 0;) {
                            const Launch& prev = p->plan[back];
                            const bool fits = prev.hook && prev.hook->out == dp.a &&
                                              (prev.hook->rows ? prev.hook->rows == dp.M && prev.hook->cols == dp.K : (uint64_t)dp.M * dp.K == prev.hook->n);
                            if (fits) {
                                *prev.hook->ap = (uint16_t*)p->scratch, *prev.hook->ap_S = kApF16 | (dp.K / 32);
                                if (prev.hook->ap_cols) *prev.hook->ap_cols = dp.K;
                                fp.reuse_a = 1;
                                break;
                            }
                            if (prev.kind == ZGML_DOP_QMATMUL || prev.kind == ZGML_DOP_MATMUL) break; // may own the scratch
                        }
                    }
                    p->split_pos = pos, p->split_input = dp.a, p->split_M = dp.M, p->split_K = dp.K, p->split_in_rs = dp.a_rs, p->split_kind = 2;
                    bool joins = fp.reuse_a && p->f16_group && p->f16_group->size() < dense_f16_max_group();
                    for (size_t t = 0; joins && t < p->f16_group->size(); t++) joins = dense_f16_can_group((*p->f16_group)[t], fp);
                    if (joins) { // same rows, same K (q/k/v, gate/up): one launch
                        p->f16_group->push_back(fp);
                        Launch& prev = p->plan.back();
                        prev.n_ops++, prev.op_hi = (uint32_t)i;
                        p->split_pos = pos - 1;
                        return false;
                    }
                    auto group = std::make_shared<std::vector<DenseF16Params>>();
                    group->push_back(fp);
                    p->f16_group = group;
                    L.run = [=](hipStream_t s) { launch_dense_f16_group(s, group->data(), (uint32_t)group->size()); };
                    return true;
                }
                L.run = [=](hipStream_t s) { launch_dense_f16(s, fp); };
                return true;
            }
            L.run = [=](hipStream_t s) { launch_dense_matmul(s, dp); };
            return true;
        }
        case ZGML_DOP_QMATMUL: {
            const auto q = op.u.qmatmul;
            QMatmulParams qp{};
            qp.dst = buf_at(p, q.dst, q.dst_offset);
            qp.input = buf_at(p, q.input, q.input_offset);
            qp.M = q.M, qp.N = q.N, qp.K = q.K;
            qp.in_rs = q.input_row_stride ? q.input_row_stride : q.K;
            qp.dst_rs = q.dst_row_stride ? q.dst_row_stride : q.N;
            const QWeightDev w = p->qweights[q.weight_idx];
            float* scratch = p->scratch;
            // q/k/v and gate/up read the same rows: the launch right after one that split the same input
            // (adjacent in the plan, so nothing rewrote the rows or the scratch in between) reuses its A pieces
            const uint64_t pos = p->plan.size();
            const bool splits = scratch && qmatmul_scratch_bytes(w, qp.M) != 0;
            qp.reuse_split = splits && p->split_pos + 1 == pos && p->split_input == qp.input && p->split_M == qp.M && p->split_K == qp.K &&
                             p->split_in_rs == qp.in_rs && p->split_kind == 1;
            // ... or finds them written by the launch that produced the rows (a row chain right in front of it, no other
            // splitting launch in between): that launch is armed and this one skips its split_a_kernel launch
            static const bool fuse_split = !(getenv("ZGML_HIP_FUSE_SPLIT") && atoi(getenv("ZGML_HIP_FUSE_SPLIT")) == 0);
            if (splits && !qp.reuse_split && fuse_split && qp.M % 16 == 0 && qp.K % 128 == 0 && qp.in_rs == qp.K) {
                for (size_t back = p->plan.size(); back-- > 0;) {
                    const Launch& prev = p->plan[back];
                    const bool fits = prev.hook && prev.hook->out == qp.input &&
                                      (prev.hook->rows ? prev.hook->rows == qp.M && prev.hook->cols == qp.K : (uint64_t)qp.M * qp.K == prev.hook->n);
                    if (fits) {
                        *prev.hook->ap = (uint16_t*)scratch, *prev.hook->ap_S = qp.K / 128;
                        if (prev.hook->ap_cols) *prev.hook->ap_cols = qp.K;
                        qp.reuse_split = 1;
                        if (getenv("ZGML_HIP_DEBUG_PLAN")) fprintf(stderr, "[zgml_hip] A pieces of op %u (M %u, K %u) written by the launch of kind %u at plan[%zu]\n", (unsigned)i, qp.M, qp.K, prev.kind, back);
                        break;
                    }
                    if (prev.kind == ZGML_DOP_QMATMUL || prev.kind == ZGML_DOP_MATMUL) break; // may own the scratch
                }
            }
            if (splits)
                p->split_pos = pos, p->split_input = qp.input, p->split_M = qp.M, p->split_K = qp.K, p->split_in_rs = qp.in_rs, p->split_kind = 1;
            // ... and joins its launch when the kernel can take another part (same rows, same K: q/k/v, gate/up)
            bool joins = qp.reuse_split && p->qmm_group && p->qmm_group->size() < qmatmul_max_group();
            for (size_t t = 0; joins && t < p->qmm_group->size(); t++) joins = qmatmul_can_group((*p->qmm_group)[t].first, (*p->qmm_group)[t].second, w, qp);
            if (joins) {
                p->qmm_group->push_back({w, qp});
                Launch& prev = p->plan.back();
                prev.n_ops++, prev.op_hi = (uint32_t)i;
                p->split_pos = pos - 1; // the merged launch still sits at the previous plan position
                return false;
            }
            auto group = std::make_shared<std::vector<std::pair<QWeightDev, QMatmulParams>>>();
            group->push_back({w, qp});
            p->qmm_group = splits ? group : nullptr;
            L.run = [=](hipStream_t s) {
                QWeightDev ws[4];
                QMatmulParams ps[4];
                const uint32_t n = (uint32_t)group->size();
                for (uint32_t t = 0; t < n; t++) ws[t] = (*group)[t].first, ps[t] = (*group)[t].second;
                launch_qmatmul_group(s, ws, ps, n, scratch);
            };
            return true;
        }
        case ZGML_DOP_SOFTMAX:
        case ZGML_DOP_LAYERNORM:
        case ZGML_DOP_RMSNORM: {
            const auto r = op.u.rmsnorm;
            float* dst = buf_at(p, r.dst, r.dst_offset);
            const float* src = buf_at(p, r.src, r.src_offset);
            const uint32_t kind = op.kind;
            L.run = [=](hipStream_t s) {
                if (kind == ZGML_DOP_SOFTMAX)
                    launch_softmax(s, dst, src, r.rows, r.cols);
                else if (kind == ZGML_DOP_LAYERNORM)
                    launch_layernorm(s, dst, src, r.rows, r.cols, r.eps);
                else
                    launch_rmsnorm(s, dst, src, r.rows, r.cols, r.eps);
            };
            return true;
        }
        case ZGML_DOP_REDUCE: {
            const auto r = op.u.reduce;
            float* dst = buf_at(p, r.dst, r.dst_offset);
            const float* src = buf_at(p, r.src, r.src_offset);
            L.run = [=](hipStream_t s) { launch_reduce(s, r.op, dst, src, r.n_out, r.reduce_size); };
            return true;
        }
        case ZGML_DOP_FUSED_ELEMENTWISE: {
            const auto fe = op.u.fused_elementwise;
            FusedParams fp{};
            fp.dst = buf_at(p, fe.dst, fe.dst_offset);
            fp.src = buf_at(p, fe.src, fe.src_offset);
            fp.n = fe.n;
            fp.n_steps = fe.n_steps;
            for (uint32_t s = 0; s < fe.n_steps; s++) {
                fp.steps[s].op = fe.steps[s].op;
                fp.steps[s].swapped = fe.steps[s].is_swapped;
                const bool bin = fe.steps[s].op == ZGML_OP_ADD || fe.steps[s].op == ZGML_OP_MUL;
                fp.steps[s].secondary = bin ? buf_at(p, fe.steps[s].secondary_buf, fe.steps[s].secondary_offset) : nullptr;
            }
            L.run = [=](hipStream_t s) { launch_fused_elementwise(s, fp); };
            return true;
        }
        default: return false;
    }
}

// One schedulable item: an op, optionally with the slice_assign that was folded into it
// (rope -> KV store, attention -> row store).
struct PlanItem {
    uint32_t op;
    int store = -1;
};

// the batchable kinds of one group of mutually independent items -> one "movement" launch (ropes
// and slice_assigns together), one attention launch, one repeat launch; the rest one by one
void emit_batches(zgml_hip_program* p, const std::vector<PlanItem>& group) {
    std::vector<RepeatParams> reps, hoisted;
    uint32_t hoisted_max = 0;
    std::vector<MoveParams> moves;
    std::vector<AttentionParams> atts;
    std::vector<int64_t> att_store_off; // per attention: the static offset of its folded row store (else -1)
    std::vector<KvqStoreParams> kstores;
    std::map<uint32_t, std::vector<KvqAttentionParams>> katts; // by d_head
    uint32_t kst_lo = UINT32_MAX, kst_hi = 0, kat_lo = UINT32_MAX, kat_hi = 0, kat_max_q = 0;
    uint32_t rep_max = 0, move_max = 0, att_max = 0, n_rep = 0, n_move = 0, n_att = 0;
    uint32_t lo[3] = {UINT32_MAX, UINT32_MAX, UINT32_MAX}, hi[3] = {0, 0, 0};
    auto track = [&](int k, const PlanItem& it) {
        lo[k] = std::min(lo[k], it.op);
        hi[k] = std::max(hi[k], it.store >= 0 ? std::max(it.op, (uint32_t)it.store) : it.op);
    };
    for (const PlanItem& it : group) {
        const uint32_t i = it.op;
        const zgml_device_op& op = p->ops[i];
        switch (op.kind) {
            case ZGML_DOP_REPEAT:
                if (i < p->hoist_op.size() && p->hoist_op[i]) { // constant: runs once, now (below)
                    hoisted.push_back(make_repeat(p, op.u.repeat));
                    hoisted_max = std::max(hoisted_max, op.u.repeat.n);
                    break;
                }
                reps.push_back(make_repeat(p, op.u.repeat));
                rep_max = std::max(rep_max, op.u.repeat.n);
                n_rep++;
                track(0, it);
                break;
            case ZGML_DOP_SLICE_ASSIGN: {
                MoveParams m{};
                m.kind = 1;
                m.n_elems = op.u.slice_assign.rows * op.u.slice_assign.cols;
                m.sa = make_slice_assign(p, op.u.slice_assign, i);
                moves.push_back(m);
                move_max = std::max(move_max, m.n_elems);
                n_move++;
                track(1, it);
                break;
            }
            case ZGML_DOP_ROPE: {
                MoveParams m{};
                m.kind = 0;
                m.n_elems = op.u.rope.half_d * op.u.rope.seq_len;
                m.rope = make_rope(p, op.u.rope);
                if (it.store >= 0) {
                    const auto& sa = p->ops[it.store].u.slice_assign;
                    m.rope.dst2 = p->bufs[sa.dst];
                    m.rope.dyn_dst2_off = p->dyn_dev + it.store;
                    m.rope.d2_rs = sa.dst_row_stride, m.rope.d2_cs = sa.dst_col_stride;
                    n_move++;
                }
                moves.push_back(m);
                move_max = std::max(move_max, m.n_elems);
                n_move++;
                track(1, it);
                break;
            }
            case ZGML_DOP_ATTENTION: {
                AttentionParams ap = make_attention(p, op.u.attention, i);
                if (it.store >= 0) {
                    const auto& sa = p->ops[it.store].u.slice_assign;
                    ap.dst2 = p->bufs[sa.dst];
                    ap.dyn_dst2_off = p->dyn_dev + it.store;
                    ap.d2_rs = sa.dst_row_stride, ap.d2_cs = sa.dst_col_stride;
                    n_att++;
                    att_store_off.push_back(sa.patch_stride == 0 ? (int64_t)sa.dst_offset : -1); // -1: moves with the position
                } else {
                    att_store_off.push_back(-1);
                }
                atts.push_back(ap);
                att_max = std::max(att_max, op.u.attention.seq_q);
                n_att++;
                track(2, it);
                break;
            }
            case ZGML_DOP_KVQ_STORE: {
                const auto& st = op.u.kvq_store;
                kstores.push_back({p->bufs[st.cache], buf_at(p, st.src, st.src_offset), p->dyn_dev + i, st.d_head, st.block_size, st.n_cols});
                kst_lo = std::min(kst_lo, i), kst_hi = std::max(kst_hi, i);
                break;
            }
            case ZGML_DOP_ATTENTION_KVQ: {
                const auto& a = op.u.attention_kvq;
                KvqAttentionParams kp{};
                kp.dst = buf_at(p, a.dst, a.dst_off), kp.q = buf_at(p, a.q, a.q_off);
                kp.k_cache = p->bufs[a.k], kp.v_cache = p->bufs[a.v];
                kp.mask = a.has_mask ? buf_at(p, a.mask, a.mask_off) : nullptr;
                kp.dyn_seq_kv = p->dyn_dev + i;
                kp.d_head = a.d_head, kp.seq_q = a.seq_q, kp.block_size = a.block_size, kp.n_cols = a.n_cols;
                kp.k_col_start = a.k_col_start, kp.v_col_start = a.v_col_start, kp.q_cs = a.q_cs, kp.dst_cs = a.dst_cs;
                kp.mask_rs = a.mask_rs, kp.mask_cs = a.mask_cs, kp.scale = a.scale;
                katts[a.d_head].push_back(kp);
                kat_lo = std::min(kat_lo, i), kat_hi = std::max(kat_hi, i), kat_max_q = std::max(kat_max_q, a.seq_q);
                break;
            }
            default: {
                Launch L;
                if (make_single(p, i, L)) p->plan.push_back(std::move(L));
            }
        }
    }
    if (!kstores.empty()) {
        const KvqStoreParams* d = upload_params(p, kstores);
        const uint32_t n = (uint32_t)kstores.size();
        p->plan.push_back({ZGML_DOP_KVQ_STORE, n, kst_lo, kst_hi, [=](hipStream_t s) { launch_kvq_store_batch(s, d, n); }});
    }
    for (auto& kv : katts) {
        const KvqAttentionParams* d = upload_params(p, kv.second);
        const uint32_t n = (uint32_t)kv.second.size(), dh = kv.first, mq = kat_max_q;
        AttnSplit sp; // decode launches: long contexts split a head's keys over several workgroups
        if (mq == 1) {
            uint32_t max_kv = 0;
            for (const auto& a : kv.second) max_kv = std::max(max_kv, a.n_cols - std::min(a.n_cols, std::max(a.k_col_start, a.v_col_start)));
            sp = attn_split_for(p, n, dh, max_kv);
        }
        p->plan.push_back({ZGML_DOP_ATTENTION_KVQ, n, kat_lo, kat_hi, [=](hipStream_t s) { launch_kvq_attention_batch(s, d, n, mq, dh, sp); }});
    }
    if (!hoisted.empty()) { // (build_plan runs outside any capture, on the context stream: ordered before the plan's first execution)
        const RepeatParams* d = upload_params(p, hoisted);
        launch_repeat_batch(p->ctx->stream, d, (uint32_t)hoisted.size(), hoisted_max);
    }
    if (!reps.empty()) {
        const RepeatParams* d = upload_params(p, reps);
        const uint32_t n = (uint32_t)reps.size(), mx = rep_max;
        p->plan.push_back({ZGML_DOP_REPEAT, n_rep, lo[0], hi[0], [=](hipStream_t s) { launch_repeat_batch(s, d, n, mx); }});
    }
    if (!moves.empty()) {
        const MoveParams* d = upload_params(p, moves);
        const uint32_t n = (uint32_t)moves.size(), mx = move_max;
        p->plan.push_back({ZGML_DOP_SLICE_ASSIGN, n_move, lo[1], hi[1], [=](hipStream_t s) { launch_move_batch(s, d, n, mx); }});
    }
    if (!atts.empty()) {
        bool dense = true;
        for (const AttentionParams& a : atts) {
            const uint32_t dh = a.d_head;
            dense = dense && a.q_rs == 1 && a.k_rs == 1 && a.v_rs == 1 && dh >= 4 && dh <= 256 && (dh & (dh - 1)) == 0 &&
                    (a.q_cs % 4) == 0 && (a.k_cs % 4) == 0 && (a.v_cs % 4) == 0 && ((uintptr_t)a.q % 16) == 0 &&
                    ((uintptr_t)a.k % 16) == 0 && ((uintptr_t)a.v % 16) == 0;
        }
        uint32_t rows_dh = dense ? atts[0].d_head : 0; // the streaming kernel: one d_head >= 8 per launch
        for (const AttentionParams& a : atts)
            if (a.d_head != rows_dh || a.d_head < 8) rows_dh = 0;
        const float* zero = p->zero_word;
        const AttentionParams* d = upload_params(p, atts);
        const uint32_t n = (uint32_t)atts.size(), mx = att_max;
        auto sink = std::make_shared<AttnPieceSink>(); // armed by the matmul that reads the heads' row stores, if one follows
        Launch L{ZGML_DOP_ATTENTION, n_att, lo[2], hi[2], [=](hipStream_t s) { launch_attention_batch(s, d, n, mx, dense, rows_dh, zero, *sink); }};
        // every head stores its rows into one dense [seq_q x cols] matrix (d2_rs == 1, a common row stride) and the tile
        // kernel will run: that matrix may be the next quantized matmul's input
        bool one_matrix = dense && rows_dh && zero && attention_tiles_applies(mx, rows_dh) && atts[0].dst2 && atts[0].d2_rs == 1;
        for (const AttentionParams& a : atts)
            one_matrix = one_matrix && a.dst2 == atts[0].dst2 && a.d2_rs == 1 && a.d2_cs == atts[0].d2_cs && a.seq_q == mx;
        std::vector<int64_t> offs = att_store_off;
        std::sort(offs.begin(), offs.end());
        for (size_t h = 0; h < offs.size(); h++) one_matrix = one_matrix && offs[h] == (int64_t)(h * rows_dh); // head h's columns, statically
        if (one_matrix && (uint64_t)rows_dh * n == atts[0].d2_cs) // the heads tile the whole row
            L.hook = std::make_shared<SplitHook>(SplitHook{atts[0].dst2, mx, atts[0].d2_cs, 0, &sink->ap, &sink->S, &sink->cols});
        p->plan.push_back(std::move(L));
    }
}

void free_param_blobs(zgml_hip_program* p) {
    for (void* d : p->param_blobs) hipFree(d);
    p->param_blobs.clear();
}

// ── macro-op fusion around the quantized mat-vecs (M == 1) ───────────────────────────────────
// PROLOGUE  : the elementwise mul (optionally preceded by rmsnorm) that produces a mat-vec's input
//             is computed inside the mat-vec while it stages x (each workgroup recomputes it from
//             L2-resident vectors; workgroup 0 stores the intermediates);
// EPILOGUE  : elementwise / fused_elementwise ops that consume the mat-vec output column-wise are
//             applied by the 16 lanes that own the outputs (residual add, the SiLU chain, ...);
// GROUPING  : mat-vecs of one dependency level that read the same vector (q/k/v, gate/up) share
//             one launch.
// Legality is decided from access spans only (no LLaMA-specific pattern): a macro-op is placed at
// its last member; a member X may be delayed past a non-member C only if X and C do not conflict.
struct Macro {
    std::vector<uint32_t> members; // op indices, ascending
    uint32_t position = 0;         // index of the last member
    bool qmv = false;
    int store = -1;                // rope / attention anchors: the slice_assign folded into them
    uint32_t anchor = 0;           // the qmatmul op (or the rope / attention op)
    QmvPrologue pro;
    uint64_t pro_sig[3] = {0, 0, 0}; // (kind, a, b) identity for grouping
    bool owns_prologue = false;
    uint32_t n_epi = 0;
    QmvEpiStep epi[kMaxEpiSteps];
    OpAccess access;
    // decode attention of one kv group: rope k + K store + V store, and per head rope q +
    // attention (+ row store), all in one launch record per head
    struct Head {
        uint32_t rq, att;
        int row_store;
    };
    std::vector<Head> heads;
    uint32_t rk = 0, sk = 0, sv = 0;
    bool kvq = false; // the same block over quantised caches: sk / sv are kvq_store ops, the heads' `att` attention_kvq ops
    // [add ->] rmsnorm [-> mul] over dense rows (any row count): one launch (RowChainParams)
    bool chain = false;
    int chain_add = -1, chain_mul = -1;
    // elementwise / fused_elementwise ops feeding each other index by index: one launch (EltChainParams)
    bool elt_chain = false;
    EltChainParams elt{};
};

bool ops_conflict(const OpAccess& x, const OpAccess& c) {
    for (const Span& w : x.writes) {
        for (const Span& r : c.reads)
            if (spans_overlap(w, r)) return true;
        for (const Span& w2 : c.writes)
            if (spans_overlap(w, w2)) return true;
    }
    for (const Span& r : x.reads)
        for (const Span& w2 : c.writes)
            if (spans_overlap(r, w2)) return true;
    return false;
}

struct ExactSpan {
    uint16_t buf;
    uint64_t off, n;
    bool operator==(const ExactSpan& o) const { return buf == o.buf && off == o.off && n == o.n; }
};

bool barrier_between(const std::vector<uint64_t>& barriers, uint32_t lo, uint32_t hi) { // a barrier b with lo < b <= hi
    for (uint64_t b : barriers)
        if (b > lo && b <= hi) return true;
    return false;
}

void add_access(OpAccess& dst, const OpAccess& src) {
    dst.reads.insert(dst.reads.end(), src.reads.begin(), src.reads.end());
    dst.writes.insert(dst.writes.end(), src.writes.begin(), src.writes.end());
}

bool anchor_ok(zgml_hip_program* p, uint32_t i) {
    const zgml_device_op& op = p->ops[i];
    if (op.kind != ZGML_DOP_QMATMUL || op.u.qmatmul.M != 1) return false;
    const QWeightDev& w = p->qweights[op.u.qmatmul.weight_idx];
    return w.format != QW_RAW && w.format != QW_W8A8 && (op.u.qmatmul.input_offset % 4) == 0; // (W8A8: the reference's arm, launched as it is — no prologue / epilogue fusion)
}

// A grouped q / k / v projection launch directly followed by the decode-attention launch of exactly its heads becomes ONE
// launch (qmatvec.hip: qkv_attn_kernel): the projection's outputs reach the attention through
// per-head-slice counters instead of a kernel boundary, and everything the attention can do without them overlaps the
// projection. Only for the shapes that kernel is built for (short K, Q4_0 with f16 scales, d_head 64 / 128, f32 KV).
void fuse_qkv_attention(zgml_hip_program* p) {
    static const bool on = !(getenv("ZGML_HIP_FUSE_QKV_ATTN") && atoi(getenv("ZGML_HIP_FUSE_QKV_ATTN")) == 0);
    if (!on || p->ctx->fuse_qkv_off || !p->ctx->handoff_flag_dev) return; // (off for good once a hand-off wait has timed out in this context)
    // Residency: the attention's workgroups spin on counters the projection's workgroups of the SAME grid bump, and HIP
    // promises neither dispatch order nor co-residency. The launch is only built when the whole grid (1024-thread workgroups)
    // fits the device at ONE workgroup per CU — the occupancy query may say two, but it reads one high near a register-file
    // edge (guide: residency and cooperative launch) and a stranded producer would mean a time-out, not a slowdown; the
    // attention's split count shrinks to fit, and if even one split per head does not fit the two launches stay apart.
    // ZGML_HIP_OPT_FUSE_RESIDENT_WGS overrides the capacity (0 refuses every fusion).
    const uint64_t resident_cap = p->ctx->opt_fuse_resident_wgs >= 0 ? (uint64_t)p->ctx->opt_fuse_resident_wgs : (uint64_t)std::max(p->ctx->n_cu, 1);
    for (size_t i = 0; i + 1 < p->plan.size(); i++) {
        const auto qd = p->plan[i].qmv_desc;
        const auto ad = p->plan[i + 1].adec_desc;
        if (!qd || !ad) continue;
        const QmvLaunch& L = *qd;
        { // a declared barrier (zgml_hip_program_set_barriers: a collective of the caller) between the two launches keeps them apart
            const uint64_t lo_b = std::min(p->plan[i].op_lo, p->plan[i + 1].op_lo), hi_b = std::max(p->plan[i].op_hi, p->plan[i + 1].op_hi);
            bool cut = false;
            for (uint64_t b : p->barriers) cut = cut || (b > lo_b && b <= hi_b);
            if (cut) continue;
        }
        const bool kon = L.parts[0].w.format == QW_Q4K && L.K > 2048; // the 256-thread form of the launch (qkv_attn_kon_kernel)
        static const bool kon_on = !(getenv("ZGML_HIP_FUSE_QKV_ATTN_KON") && atoi(getenv("ZGML_HIP_FUSE_QKV_ATTN_KON")) == 0);
        if (L.n_parts != 3 || (L.K > 2048 && !(kon && kon_on)) || L.pro.kind == QMV_PRO_NONE || (ad->dh != 64 && ad->dh != 128) || L.trace) continue;
        if (kon && L.pro.kind != QMV_PRO_PRENORM) continue; // (an in-kernel rmsnorm prologue runs eight waves: launch_packed_kon)
        const bool kvq = ad->kvq;
        bool ok = true;
        for (uint32_t t = 0; t < 3; t++) ok = ok && L.parts[t].n_epi == 0 && (L.parts[t].w.format == QW_Q4 || L.parts[t].w.format == QW_Q4K) && L.parts[t].w.scale_f16;
        const uint32_t nh = ad->nh, dh = ad->dh;
        ok = ok && (uint64_t)nh * dh == L.parts[0].w.N && L.parts[1].w.N == L.parts[2].w.N && L.parts[1].w.N % dh == 0;
        const uint32_t n_kv = ok ? (uint32_t)(L.parts[1].w.N / dh) : 0;
        ok = ok && n_kv != 0 && nh % n_kv == 0;
        std::vector<uint32_t> idx(3 * (size_t)nh);
        std::vector<char> head_seen(nh, 0);
        for (uint32_t r = 0; ok && r < nh; r++) { // the records are not in head order: each one's head from its pointers
            const AttnDecodeParams& a = ad->host[r];
            const ptrdiff_t qo = a.q_src - L.parts[0].dst, ko = a.k_src - L.parts[1].dst, vo = a.v_src - L.parts[2].dst;
            ok = qo >= 0 && qo % dh == 0 && (uint64_t)qo < (uint64_t)nh * dh && ko >= 0 && ko % dh == 0 && (uint64_t)ko < (uint64_t)n_kv * dh && vo == ko &&
                 a.kvq_block == (kvq ? 32u : 0u);
            if (!ok) break;
            const uint32_t h = (uint32_t)(qo / dh), kvh = (uint32_t)(ko / dh);
            ok = !head_seen[h] && kvh == h / (nh / n_kv);
            head_seen[h] = 1;
            idx[3 * r] = h, idx[3 * r + 1] = nh + kvh, idx[3 * r + 2] = nh + n_kv + kvh;
        }
        if (!ok) continue;
        uint32_t n_sp = ad->sp.splits ? ad->sp.splits : 1;
        uint64_t n_mv = 0;
        for (uint32_t t = 0; t < 3; t++) n_mv += L.parts[t].w.N / 16;
        uint64_t cap = resident_cap;
        if (kon && p->ctx->opt_fuse_resident_wgs < 0) {
            // 256-thread workgroups: what the occupancy query admits (four per CU at 127 registers: 1024; the 7B launch is 768 projection
            // workgroups + 32 heads x up to 8 splits = 1024). No margin here, unlike the 1024-thread form: with 4 instead of 8 splits
            // the launch LOSES at long contexts (position 1900: 590 against 632 tok/s unfused; with 8: 644), and the projection's
            // workgroups have the lower ids — should fewer be resident than the query says, the attention's workgroups queue behind
            // them instead of spinning beside them (no overlap then, nothing worse); a stranded wait is still bounded and loud.
            const int per_cu = qkv_attn_kon_blocks_per_cu(ad->dh, kvq);
            static const int eighths = getenv("ZGML_HIP_FUSE_KON_CAP_EIGHTHS") ? atoi(getenv("ZGML_HIP_FUSE_KON_CAP_EIGHTHS")) : 8; // (experiments)
            cap = per_cu > 0 ? (uint64_t)per_cu * (uint64_t)std::max(p->ctx->n_cu, 1) * (uint64_t)eighths / 8 : 0;
        }
        if (n_mv + nh > cap) continue; // not even one attention workgroup per head beside the projection's: two launches
        // (the 256-thread form only with ALL the splits the stand-alone attention would use: with fewer it loses at long contexts —
        // 7B dimensions with 4 of 8: 590 against 632 tok/s at position 1900 — so a wider model keeps its two launches)
        if (kon && p->ctx->opt_fuse_resident_wgs < 0 && (cap - n_mv) / nh < n_sp) continue;
        n_sp = (uint32_t)std::min<uint64_t>(n_sp, (cap - n_mv) / nh);
        for (uint32_t t = 0; kon && t < 3; t++) ok = ok && L.parts[t].w.format == QW_Q4K;
        if (!ok) continue;
        if (kon && p->ctx->opt_fuse_resident_wgs < 0) {
            // ... and the grid it comes to is CENSUSED once per context and size: the kernel's own census instantiation counts its
            // workgroups and each waits (bounded) for all of them — co-residency measured, not inferred
            const uint32_t grid = (uint32_t)(n_mv + (uint64_t)nh * n_sp);
            const uint64_t key = ((uint64_t)ad->dh << 40) | ((uint64_t)(kvq ? 1 : 0) << 32) | grid;
            auto it = p->ctx->census.find(key);
            if (it == p->ctx->census.end()) {
                static const bool census_on = !(getenv("ZGML_HIP_FUSE_CENSUS") && atoi(getenv("ZGML_HIP_FUSE_CENSUS")) == 0);
                const int r = census_on ? qkv_attn_kon_census(p->ctx->stream, ad->dh, kvq, grid) : -1;
                it = p->ctx->census.emplace(key, r).first;
                if (r == 0) fprintf(stderr, "[zgml_hip] the fused q/k/v + attention launch of %u workgroups is NOT co-resident on this device (census): two launches instead\n", grid);
            }
            if (it->second == 0) continue;
        }
        // ... and the single-matrix projection that reads exactly the heads' row stores (the O projection) rides along:
        // no prologue, K = n_heads * d_head, every head's rows stored at its static offset h * d_head of that input
        std::shared_ptr<QmvLaunch> od = i + 2 < p->plan.size() ? p->plan[i + 2].qmv_desc : nullptr;
        // (measured: 93 launches per SmolLM-135M token instead of 123, parity green, and SLOWER — 1714-1734 against 1772 tok/s:
        // this edge is all-to-all (every column group of the projection needs every head), its hand-off costs more than the
        // boundary it replaces. Off unless ZGML_HIP_FUSE_QKV_ATTN_O=1.)
#ifdef ZGML_TRACE // (diagnostics build only)
        static const bool with_o = getenv("ZGML_HIP_FUSE_QKV_ATTN_O") && atoi(getenv("ZGML_HIP_FUSE_QKV_ATTN_O")) != 0;
#else
        constexpr bool with_o = false;
#endif
        bool o_ok = with_o && !kvq && od && od->n_parts == 1 && od->pro.kind == QMV_PRO_NONE && od->K == nh * dh && od->K <= 2048 && !od->trace &&
                    od->parts[0].w.format == QW_Q4 && od->parts[0].w.scale_f16;
        for (uint32_t r = 0; o_ok && r < nh; r++) {
            const AttnDecodeParams& a = ad->host[r];
            o_ok = a.att.dst2 == od->pro.a && a.att.d2_rs == 1 && a.att.dyn_dst2_off >= p->dyn_dev && a.att.dyn_dst2_off < p->dyn_dev + p->ops.size();
            if (!o_ok) break;
            const zgml_device_op& so = p->ops[(size_t)(a.att.dyn_dst2_off - p->dyn_dev)];
            o_ok = so.kind == ZGML_DOP_SLICE_ASSIGN && so.u.slice_assign.patch_stride == 0 && so.u.slice_assign.dst_offset == idx[3 * r] * dh;
        }
        if (kon) o_ok = false;
        if (o_ok && n_mv + (uint64_t)nh * n_sp + od->parts[0].w.N / 16 > resident_cap) o_ok = false;
        const uint32_t o_blocks = o_ok ? (uint32_t)(od->parts[0].w.N / 16) : 0;
        const size_t n_cnt = 32 * ((size_t)nh + 2 * n_kv + 1); // one counter per 128 bytes (the last: the heads' outputs)
        const size_t words = n_cnt + (size_t)nh * n_sp * 3 + 1 + 3 * (size_t)nh + o_blocks;
        uint32_t* block = nullptr;
        if (hipMalloc((void**)&block, words * 4) != hipSuccess) continue;
        if (memset_sync(p->ctx->stream, block, 0, words * 4) != hipSuccess) {
            hipFree(block);
            continue;
        }
        p->fuse_owned.push_back(block); // (freed by the next build_plan, after its stream sync)
        uint32_t *counters = block, *seen = block + n_cnt, *timeout = p->ctx->handoff_flag_dev, *idx_dev = seen + (size_t)nh * n_sp * 3 + 1;
        if (h2d_sync(p->ctx->stream, idx_dev, idx.data(), idx.size() * 4) != hipSuccess) continue;
        const AttnDecodeParams* d = ad->dev;
        AttnSplit sp = ad->sp;
        sp.splits = n_sp; // (possibly fewer than the stand-alone launch would use: the residency guard above)
        uint32_t *out_cnt = counters + 32 * ((size_t)nh + 2 * n_kv), *o_seen = idx_dev + 3 * (size_t)nh;
        uint32_t n_ops = p->plan[i].n_ops + p->plan[i + 1].n_ops, lo = std::min(p->plan[i].op_lo, p->plan[i + 1].op_lo),
                 hi = std::max(p->plan[i].op_hi, p->plan[i + 1].op_hi);
        if (o_ok) n_ops += p->plan[i + 2].n_ops, lo = std::min(lo, p->plan[i + 2].op_lo), hi = std::max(hi, p->plan[i + 2].op_hi);
        const QmvLaunch Lo = o_ok ? *od : QmvLaunch{};
        Launch F{ZGML_DOP_QMATMUL, n_ops, lo, hi, [=](hipStream_t s) {
                     if (o_ok && launch_qkv_attention(s, L, d, nh, n_kv, dh, sp, counters, idx_dev, seen, timeout, &Lo, out_cnt, o_seen)) return;
                     if (!launch_qkv_attention(s, L, d, nh, n_kv, dh, sp, counters, idx_dev, seen, timeout, nullptr, nullptr, nullptr, kvq)) {
                         launch_qmatvec_fused(s, L);
                         launch_attention_decode_batch(s, d, nh, dh, sp, kvq);
                     }
                     if (o_ok) launch_qmatvec_fused(s, Lo);
                 }};
        p->plan[i] = std::move(F);
        p->plan.erase(p->plan.begin() + (ptrdiff_t)i + 1, p->plan.begin() + (ptrdiff_t)i + (o_ok ? 3 : 2));
    }
}

// The decode-attention launch directly followed by the single-matrix K-on-lanes projection that reads exactly its heads' row
// stores (the O projection of a Llama-2-7B-class model) becomes ONE launch of 256-thread workgroups (qmatvec.hip:
// attn_o_kon_kernel): the projection's weights stream while the attention's latency chain runs, the projection's workgroups wait
// (bounded) for one counter every head bumps. The waiting workgroups spin on workgroups of the same grid, so — like the q/k/v +
// attention launch above — it is only built when the WHOLE grid is resident: the occupancy query minus one workgroup per CU
// (the query reads one high near a register-file edge), the attention's split count shrinks to fit, otherwise two launches.
#ifdef ZGML_TRACE // diagnostics build only: the kernel is not in the product library
void fuse_attention_o(zgml_hip_program* p) {
    // MEASURED SLOWER, off unless ZGML_HIP_FUSE_ATTN_O=1 (parity green: tests/test_hip_fused_qkv.py): Llama-2-7B 800 -> 749 tok/s
    // (766 when the projection's weight loads are delayed until the attention is about done; polling interval and split count
    // make no difference). The projection's stream does hide under the attention, but the attention's dependent round trips run
    // under 256 workgroups' worth of memory traffic and the all-to-all edge (write-through rows, counter, agent-scope x loads by
    // 256 workgroups) costs more than the boundary it replaces — the round-2 finding for every all-to-all edge, now also with
    // 9.4 MB of prefetch credit on the other side of the scale.
    static const bool on = getenv("ZGML_HIP_FUSE_ATTN_O") && atoi(getenv("ZGML_HIP_FUSE_ATTN_O")) != 0;
    if (!on || p->ctx->fuse_qkv_off || !p->ctx->handoff_flag_dev || p->ctx->opt_fuse_resident_wgs == 0) return;
    for (size_t i = 0; i + 1 < p->plan.size(); i++) {
        const auto ad = p->plan[i].adec_desc;
        const auto od = p->plan[i + 1].qmv_desc;
        if (!ad || !od || ad->kvq || (ad->dh != 64 && ad->dh != 128)) continue;
        { // a declared barrier between the two launches keeps them apart
            const uint64_t lo_b = std::min(p->plan[i].op_lo, p->plan[i + 1].op_lo), hi_b = std::max(p->plan[i].op_hi, p->plan[i + 1].op_hi);
            bool cut = false;
            for (uint64_t b : p->barriers) cut = cut || (b > lo_b && b <= hi_b);
            if (cut) continue;
        }
        const uint32_t nh = ad->nh, dh = ad->dh;
        bool ok = od->n_parts == 1 && od->pro.kind == QMV_PRO_NONE && od->K == nh * dh && !od->trace && od->parts[0].w.format == QW_Q4K &&
                  ((uintptr_t)od->pro.a % 16 == 0) && (od->K % 4 == 0);
        std::vector<char> seen_head(nh, 0);
        for (uint32_t r = 0; ok && r < nh; r++) { // every head's rows land at a static offset h * d_head of exactly the projection's input
            const AttnDecodeParams& a = ad->host[r];
            ok = a.att.dst2 == od->pro.a && a.att.d2_rs == 1 && a.att.dyn_dst2_off >= p->dyn_dev && a.att.dyn_dst2_off < p->dyn_dev + p->ops.size();
            if (!ok) break;
            const zgml_device_op& so = p->ops[(size_t)(a.att.dyn_dst2_off - p->dyn_dev)];
            ok = so.kind == ZGML_DOP_SLICE_ASSIGN && so.u.slice_assign.patch_stride == 0 && so.u.slice_assign.dst_offset % dh == 0 &&
                 so.u.slice_assign.dst_offset / dh < nh && !seen_head[so.u.slice_assign.dst_offset / dh];
            if (ok) seen_head[so.u.slice_assign.dst_offset / dh] = 1;
        }
        if (!ok) continue;
        const uint32_t o_blocks = (uint32_t)(od->parts[0].w.N / 16);
        const int bpc = attn_o_blocks_per_cu(dh);
        const uint64_t cap = (uint64_t)std::max(bpc - 1, 1) * (uint64_t)std::max(p->ctx->n_cu, 1);
        if (o_blocks + nh > cap) continue;
        uint32_t n_sp = ad->sp.splits ? ad->sp.splits : 1;
        n_sp = (uint32_t)std::min<uint64_t>(n_sp, (cap - o_blocks) / nh);
        const size_t words = 32 + o_blocks; // the counter on a line of its own, then one `seen` word per projection workgroup
        uint32_t* block = nullptr;
        if (hipMalloc((void**)&block, words * 4) != hipSuccess) continue;
        if (memset_sync(p->ctx->stream, block, 0, words * 4) != hipSuccess) {
            hipFree(block);
            continue;
        }
        p->fuse_owned.push_back(block);
        uint32_t *out_cnt = block, *o_seen = block + 32, *timeout = p->ctx->handoff_flag_dev;
        const AttnDecodeParams* d = ad->dev;
        AttnSplit sp = ad->sp;
        sp.splits = n_sp;
        const AttnSplit sp_plain = ad->sp;
        const QmvLaunch Lo = *od; // (after arm_prenorm: the residual epilogue's side outputs ride along)
        const uint32_t n_ops = p->plan[i].n_ops + p->plan[i + 1].n_ops, lo = std::min(p->plan[i].op_lo, p->plan[i + 1].op_lo),
                       hi = std::max(p->plan[i].op_hi, p->plan[i + 1].op_hi);
        Launch F{ZGML_DOP_ATTENTION, n_ops, lo, hi, [=](hipStream_t s) {
                     if (!launch_attention_o(s, d, nh, dh, sp, Lo, out_cnt, o_seen, timeout)) {
                         launch_attention_decode_batch(s, d, nh, dh, sp_plain, false);
                         launch_qmatvec_fused(s, Lo);
                     }
                 }};
        p->plan[i] = std::move(F);
        p->plan.erase(p->plan.begin() + (ptrdiff_t)i + 1);
    }
}
#endif

// The decoder layer of a short-K model as launches that end at a K-split (ksplit.hip): in the plan built above a
// layer is [q/k/v (rmsnorm prologue)] [decode attention] [O (+ residual)] [gate / up (rmsnorm prologue, SiLU chain)] [down (product
// prologue, + residual)]. Here
//   [q/k/v] [attention] [O]   ->  [ks_proj] [ks_attn_o]   (each head's workgroup adds its partial of the O projection)
//   [gate / up] [down]        ->  [ks_mlp]                (each 32-column workgroup adds its partial of the down projection)
// and the O / down projections' outputs and residual sums become DEFERRED vectors (kernels.h: KsVec): the next launch of the plan
// sums the partials in its prologue and its workgroup 0 stores the absorbed ops' buffers. When that next launch is not one that can
// (anything but a ks_proj / ks_mlp / the final rmsnorm -> mul row chain reading exactly that vector), a one-workgroup launch
// materialises the vector at once. Conditions are checked on spans and pointers, not on a model name: n-on-lanes Q4_0 weights with
// f16 scales (ks_weight_ok), K <= 2048, d_head 64 / 128, residual and norm outputs that do not alias their inputs.
void fuse_ksplit(zgml_hip_program* p) {
    p->has_deferred = false;
    if (!p->ksplit || p->ksplit_off || !p->barriers.empty()) return;
    auto overlap = [](const float* a, size_t na, const float* b, size_t nb) { return a && b && a < b + nb && b < a + na; };
    auto residual_epi = [](const QmvPart& pt) {
        return pt.n_epi == 1 && pt.epi[0].op == ZGML_OP_ADD && pt.epi[0].operand && pt.epi[0].operand != pt.dst && pt.epi[0].store && pt.epi[0].store != pt.dst;
    };
    // outputs the consumer's workgroup 0 stores late must not overlap anything the other workgroups still read
    auto norm_in_safe = [&](const KsNormIn& in) {
        const size_t K = in.K;
        const float* outs[4] = {in.x.sum_dst, in.x.add_dst, in.store_mid, in.store_x};
        for (const float* o : outs) {
            if (!o) continue;
            if (overlap(o, K, in.x.base, K) || overlap(o, K, in.gamma, K) || overlap(o, K, in.x.parts, K * in.x.n_parts)) return false;
            for (const float* o2 : outs)
                if (o2 && o2 != o && overlap(o, K, o2, K)) return false;
        }
        return true;
    };
    std::vector<Launch> out;
    KsVec pend{};
    uint32_t pend_K = 0;
    bool have_pend = false;
    uint32_t pend_lo = 0, pend_hi = 0;
    auto flush = [&]() { // nobody took the deferred vector: materialise it now
        if (!have_pend) return;
        KsNormIn in;
        in.x = pend, in.K = pend_K;
        Launch M{ZGML_DOP_ELEMENTWISE, 0, pend_lo, pend_hi, [in](hipStream_t s) { launch_ks_norm(s, in); }};
        M.tag = "ks-materialise";
        out.push_back(std::move(M));
        have_pend = false;
    };
    float *parts_o = nullptr, *parts_d = nullptr; // one block each for the whole program: producer and consumer follow each other in stream order
    size_t parts_o_n = 0, parts_d_n = 0;
    auto grow_parts = [&](float*& blk, size_t& have, size_t want) -> bool {
        if (have >= want) return true;
        float* nb = nullptr;
        if (hipMalloc((void**)&nb, want * sizeof(float)) != hipSuccess || memset_sync(p->ctx->stream, nb, 0, want * sizeof(float)) != hipSuccess) {
            if (nb) hipFree(nb);
            return false;
        }
        p->fuse_owned.push_back(nb); // (an earlier, smaller block stays owned too: launches already emitted point into it)
        blk = nb, have = want;
        return true;
    };
    std::vector<Launch>& plan = p->plan;
    size_t i = 0;
    uint32_t n_fused = 0;
    auto new_trace = [&](const char* what) -> unsigned long long* { // diagnostics build + ZGML_HIP_KS_TRACE=1
        static const bool want = getenv("ZGML_HIP_KS_TRACE") && atoi(getenv("ZGML_HIP_KS_TRACE"));
        unsigned long long* t = nullptr;
        if (!want || p->ks_traces.size() >= 24 || hipHostMalloc((void**)&t, 32 * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess) return nullptr;
        memset(t, 0, 32 * sizeof(unsigned long long));
        p->ks_traces.push_back({t, what});
        return t;
    };
    while (i < plan.size()) {
        // ---- [q/k/v] [attention] [O]
        if (i + 2 < plan.size() && plan[i].qmv_desc && plan[i + 1].adec_desc && plan[i + 2].qmv_desc) {
            const QmvLaunch &Q = *plan[i].qmv_desc, &O = *plan[i + 2].qmv_desc;
            const AdecDesc& A = *plan[i + 1].adec_desc;
            bool ok = Q.n_parts == 3 && Q.pro.kind == QMV_PRO_RMSNORM_MUL && !Q.trace && !O.trace && O.n_parts == 1 && O.pro.kind == QMV_PRO_NONE && residual_epi(O.parts[0]) &&
                      !O.next.xg_out && (A.dh == 64 || A.dh == 128) && O.K == A.nh * A.dh && ks_weight_ok(O.parts[0].w);
            for (uint32_t t = 0; ok && t < 3; t++) ok = Q.parts[t].n_epi == 0 && ks_weight_ok(Q.parts[t].w);
            std::vector<char> seen_head(A.nh, 0);
            uint8_t head_of[64] = {0};
            ok = ok && A.nh <= 64;
            for (uint32_t r = 0; ok && r < A.nh; r++) { // every head row-stores into the O projection's input at h * d_head, every h once
                const AttnDecodeParams& a = A.host[r];
                ok = a.att.dst2 == O.pro.a && a.att.d2_rs == 1 && a.att.dyn_dst2_off >= p->dyn_dev && a.att.dyn_dst2_off < p->dyn_dev + p->ops.size() && a.att.dst_rs == 1 &&
                     ((uintptr_t)a.att.dst2 % 16) == 0;
                if (!ok) break;
                const zgml_device_op& so = p->ops[(size_t)(a.att.dyn_dst2_off - p->dyn_dev)];
                ok = so.kind == ZGML_DOP_SLICE_ASSIGN && so.u.slice_assign.patch_stride == 0 && so.u.slice_assign.dst_offset % A.dh == 0 &&
                     so.u.slice_assign.dst_offset / A.dh < A.nh && !seen_head[so.u.slice_assign.dst_offset / A.dh];
                if (ok) seen_head[so.u.slice_assign.dst_offset / A.dh] = 1, head_of[r] = (uint8_t)(so.u.slice_assign.dst_offset / A.dh);
            }
            KsProjLaunch PL;
            KsAttnOLaunch AL;
            if (ok) {
                PL.in.K = Q.K, PL.in.gamma = Q.pro.b, PL.in.eps = Q.pro.eps, PL.in.store_mid = Q.pro.store_mid, PL.in.store_x = Q.pro.store_x;
                if (have_pend && pend.add_dst == Q.pro.a && pend_K == Q.K)
                    PL.in.x = pend;
                else
                    PL.in.x.base = Q.pro.a;
                PL.n_parts = 3;
                for (uint32_t t = 0; t < 3; t++) PL.w[t] = Q.parts[t].w, PL.dst[t] = Q.parts[t].dst;
                const size_t N = O.parts[0].w.N;
                ok = ks_proj_ok(PL) && norm_in_safe(PL.in) && grow_parts(parts_o, parts_o_n, (size_t)A.nh * N);
                PL.trace = new_trace("ks-proj"), AL.trace = new_trace("ks-attn-o");
                memcpy(AL.head_of, head_of, sizeof head_of);
                AL.dev = A.dev, AL.n_heads = A.nh, AL.d_head = A.dh, AL.sp = A.sp, AL.kvq = A.kvq, AL.wo = O.parts[0].w, AL.parts_out = parts_o;
                ok = ok && ks_attn_o_ok(AL);
            }
            if (ok) {
                const bool took = PL.in.x.n_parts != 0;
                if (!took) flush();
                have_pend = false;
                // ... as ONE launch when the whole grid is resident (the attention's workgroups spin on the projection's: one 768-thread
                // workgroup per CU is what is counted on; the split count shrinks to fit) and the context's hand-offs have not timed out
                static const bool fuse_a = !(getenv("ZGML_HIP_KSPLIT_FUSE_A") && atoi(getenv("ZGML_HIP_KSPLIT_FUSE_A")) == 0);
                bool fused = false;
                if (fuse_a && !p->ctx->fuse_qkv_off && p->ctx->handoff_flag_dev && ks_layer_a_ok(PL, AL)) {
                    const uint32_t nh = A.nh, dh = A.dh;
                    const uint32_t n_kv = (uint32_t)(Q.parts[1].w.N / dh);
                    std::vector<uint32_t> idx(3 * (size_t)nh);
                    bool hk = n_kv != 0 && Q.parts[1].w.N % dh == 0 && Q.parts[2].w.N == Q.parts[1].w.N && (uint64_t)nh * dh == Q.parts[0].w.N && nh % n_kv == 0;
                    for (uint32_t r = 0; hk && r < nh; r++) { // each record's q / k / v counters from its pointers into the projections' outputs
                        const AttnDecodeParams& ar = A.host[r];
                        const ptrdiff_t qo = ar.q_src - Q.parts[0].dst, ko = ar.k_src - Q.parts[1].dst, vo = ar.v_src - Q.parts[2].dst;
                        hk = qo >= 0 && qo % dh == 0 && (uint64_t)qo < (uint64_t)nh * dh && ko >= 0 && ko % dh == 0 && (uint64_t)ko < (uint64_t)n_kv * dh && vo == ko;
                        if (hk) idx[3 * r] = (uint32_t)(qo / dh), idx[3 * r + 1] = nh + (uint32_t)(ko / dh), idx[3 * r + 2] = nh + n_kv + (uint32_t)(ko / dh);
                    }
                    const uint64_t cap = p->ctx->opt_fuse_resident_wgs >= 0 ? (uint64_t)p->ctx->opt_fuse_resident_wgs : (uint64_t)std::max(p->ctx->n_cu, 1);
                    const uint32_t n_pw = ks_layer_a_proj_wgs(PL);
                    uint32_t n_sp = AL.sp.splits ? AL.sp.splits : 1;
                    if (hk && n_pw + nh <= cap) {
                        n_sp = (uint32_t)std::min<uint64_t>(n_sp, (cap - n_pw) / nh);
                        const size_t n_cnt = 32 * ((size_t)nh + 2 * n_kv), words = n_cnt + (size_t)nh * n_sp * 3 + 3 * (size_t)nh;
                        uint32_t* block = nullptr;
                        if (hipMalloc((void**)&block, words * 4) == hipSuccess) {
                            p->fuse_owned.push_back(block);
                            uint32_t *counters = block, *seen = block + n_cnt, *idx_dev = seen + (size_t)nh * n_sp * 3;
                            if (memset_sync(p->ctx->stream, block, 0, words * 4) == hipSuccess && h2d_sync(p->ctx->stream, idx_dev, idx.data(), idx.size() * 4) == hipSuccess) {
                                KsAttnOLaunch AF = AL;
                                AF.sp.splits = n_sp;
                                uint32_t* const timeout = p->ctx->handoff_flag_dev;
                                Launch LA{ZGML_DOP_QMATMUL, plan[i].n_ops + plan[i + 1].n_ops + plan[i + 2].n_ops, std::min({plan[i].op_lo, plan[i + 1].op_lo, plan[i + 2].op_lo}),
                                          std::max({plan[i].op_hi, plan[i + 1].op_hi, plan[i + 2].op_hi}), [PL, AF, AL, n_kv, counters, idx_dev, seen, timeout](hipStream_t s) {
                                              if (!launch_ks_layer_a(s, PL, AF, n_kv, counters, idx_dev, seen, timeout)) {
                                                  launch_ks_proj(s, PL);
                                                  launch_ks_attn_o(s, AL);
                                              }
                                          }};
                                LA.tag = took ? "ks-layer-a (sums the deferred vector): q / k / v + attention + partial O projection"
                                              : "ks-layer-a: q / k / v + attention + partial O projection";
                                out.push_back(std::move(LA));
                                fused = true;
                            }
                        }
                    }
                }
                if (!fused) {
                    Launch L1{ZGML_DOP_QMATMUL, plan[i].n_ops, plan[i].op_lo, plan[i].op_hi, [PL](hipStream_t s) { launch_ks_proj(s, PL); }};
                    L1.tag = took ? "ks-proj (sums the deferred vector)" : "ks-proj";
                    Launch L2{ZGML_DOP_ATTENTION, plan[i + 1].n_ops + plan[i + 2].n_ops, std::min(plan[i + 1].op_lo, plan[i + 2].op_lo), std::max(plan[i + 1].op_hi, plan[i + 2].op_hi),
                              [AL](hipStream_t s) { launch_ks_attn_o(s, AL); }};
                    L2.tag = "ks-attention + partial O projection";
                    out.push_back(std::move(L1));
                    out.push_back(std::move(L2));
                }
                pend = KsVec{O.parts[0].epi[0].operand, parts_o, A.nh, O.parts[0].dst, O.parts[0].epi[0].store};
                pend_K = (uint32_t)O.parts[0].w.N, have_pend = true, pend_lo = plan[i + 2].op_lo, pend_hi = plan[i + 2].op_hi;
                i += 3, n_fused++;
                continue;
            }
        }
        // ---- [gate / up] [down]
        if (i + 1 < plan.size() && plan[i].qmv_desc && plan[i + 1].qmv_desc) {
            const QmvLaunch &G = *plan[i].qmv_desc, &D = *plan[i + 1].qmv_desc;
            const QmvPart &ga = G.parts[0], &up = G.parts[1];
            const QmvEpiStep* st = ga.epi;
            bool ok = G.n_parts == 2 && G.pro.kind == QMV_PRO_RMSNORM_MUL && !G.pair_out && !G.trace && !D.trace && D.n_parts == 1 && D.pro.kind == QMV_PRO_MUL && D.pro.store_x &&
                      residual_epi(D.parts[0]) && !D.next.xg_out;
            // the SiLU chain exactly as build_qmv_args recognises it (NEG, EXP [store], ADD vector, RECIP, MUL by the gate [store])
            ok = ok && ga.n_epi == 5 && st[0].op == ZGML_OP_NEG && !st[0].store && st[1].op == ZGML_OP_EXP && st[1].store && st[2].op == ZGML_OP_ADD && st[2].operand &&
                 st[2].operand != ga.dst && st[2].operand != st[1].store && !st[2].store && st[3].op == ZGML_OP_RECIP && !st[3].store && st[4].op == ZGML_OP_MUL &&
                 st[4].operand == ga.dst && st[4].store && up.n_epi == 0;
            ok = ok && ((D.pro.a == st[4].store && D.pro.b == up.dst) || (D.pro.b == st[4].store && D.pro.a == up.dst)) && D.K == ga.w.N;
            KsMlpLaunch ML;
            if (ok) {
                ML.in.K = G.K, ML.in.gamma = G.pro.b, ML.in.eps = G.pro.eps, ML.in.store_mid = G.pro.store_mid, ML.in.store_x = G.pro.store_x;
                if (have_pend && pend.add_dst == G.pro.a && pend_K == G.K)
                    ML.in.x = pend;
                else
                    ML.in.x.base = G.pro.a;
                ML.gate = ga.w, ML.up = up.w, ML.down = D.parts[0].w;
                ML.gate_out = ga.dst, ML.up_out = up.dst, ML.exp_out = st[1].store, ML.silu_out = st[4].store, ML.prod_out = D.pro.store_x, ML.ones = st[2].operand;
                ok = grow_parts(parts_d, parts_d_n, (size_t)ks_mlp_parts(ML) * ML.down.N);
                ML.parts_out = parts_d, ML.trace = new_trace("ks-mlp");
                ok = ok && ks_mlp_ok(ML) && norm_in_safe(ML.in);
            }
            if (ok) {
                const bool took = ML.in.x.n_parts != 0;
                if (!took) flush();
                have_pend = false;
                Launch L{ZGML_DOP_QMATMUL, plan[i].n_ops + plan[i + 1].n_ops, std::min(plan[i].op_lo, plan[i + 1].op_lo), std::max(plan[i].op_hi, plan[i + 1].op_hi),
                         [ML](hipStream_t s) { launch_ks_mlp(s, ML); }};
                L.tag = took ? "ks-mlp (sums the deferred vector): gate / up + partial down projection" : "ks-mlp: gate / up + partial down projection";
                out.push_back(std::move(L));
                pend = KsVec{D.parts[0].epi[0].operand, parts_d, ks_mlp_parts(ML), D.parts[0].dst, D.parts[0].epi[0].store};
                pend_K = (uint32_t)ML.down.N, have_pend = true, pend_lo = plan[i + 1].op_lo, pend_hi = plan[i + 1].op_hi;
                i += 2, n_fused++;
                continue;
            }
        }
        // ---- the deferred vector's rmsnorm -> mul row chain (the final norm in front of the LM head): one workgroup does both
        if (have_pend && plan[i].rc_desc && plan[i].rc_rows == 1) {
            const RowChainParams& rc = *plan[i].rc_desc;
            KsNormIn in;
            in.x = pend, in.K = pend_K, in.gamma = rc.mul_other, in.eps = rc.eps, in.store_mid = rc.norm_dst, in.store_x = rc.mul_dst;
            if (!rc.a0 && !rc.ap && rc.src == pend.add_dst && rc.mul_other && rc.mul_dst && rc.cols == pend_K && ks_norm_ok(in) && norm_in_safe(in)) {
                Launch L{ZGML_DOP_RMSNORM, plan[i].n_ops, std::min(plan[i].op_lo, pend_lo), std::max(plan[i].op_hi, pend_hi), [in](hipStream_t s) { launch_ks_norm(s, in); }};
                L.tag = "ks-norm (sums the deferred vector): rmsnorm -> mul";
                out.push_back(std::move(L));
                have_pend = false;
                i++;
                continue;
            }
        }
        flush();
        out.push_back(std::move(plan[i]));
        i++;
    }
    flush();
    if (n_fused) {
        plan = std::move(out);
        p->has_deferred = true;
    }
}

// A mat-vec launch with a residual-add epilogue (h = y + r: the O / down projections) that is DIRECTLY followed by the launch
// whose rmsnorm -> mul(gamma) prologue consumes h prepares that prologue (kernels.h: QmvNextNorm / QMV_PRO_PRENORM): it also
// stores h * gamma and, per 16 columns, the sum of h^2. The consumer then streams one vector instead of two — x (16 KB at
// K = 4096) stays in a CU's L1 across the workgroups it hosts, x and gamma together do not (tools/exp/kon.hip: +0.5 us at
// 4096^2, +0.8 at 4096 x 11008, +1.8 at 4096 x 32000) — and needs no sum over x. Consumers: K-on-lanes launches (QW_Q4K) and, since
// round 4, the x-direct n-on-lanes launches of short-K models (qmv_prenorm_ok: SmolLM-135M's q/k/v and gate/up, whose in-kernel
// prologue — second vector, sum of squares, barrier — cost 1.1 us of a 2.6 us launch by the stamps).
void arm_prenorm(zgml_hip_program* p) {
    static const bool on = !(getenv("ZGML_HIP_PRENORM") && atoi(getenv("ZGML_HIP_PRENORM")) == 0);
    if (!on) return;
    for (size_t i = 1; i < p->plan.size(); i++) {
        const auto C = p->plan[i].qmv_desc, P = p->plan[i - 1].qmv_desc;
        if (!C || !P || C->pro.kind != QMV_PRO_RMSNORM_MUL) continue;
        bool ok = true;
        uint64_t cols = 0;
        for (uint32_t t = 0; t < C->n_parts; t++) cols += C->parts[t].w.N;
        for (uint32_t t = 0; t < C->n_parts; t++) ok = ok && C->parts[t].w.format == C->parts[0].w.format && qmv_prenorm_ok(C->parts[t].w, C->K, cols, 1);
        const QmvPart& pp = P->parts[0];
        ok = ok && P->n_parts == 1 && pp.w.format != QW_RAW && pp.w.N == C->K && pp.n_epi == 1 && pp.epi[0].op == ZGML_OP_ADD && pp.epi[0].operand &&
             pp.epi[0].operand != pp.dst && pp.epi[0].store == C->pro.a && C->K % 16 == 0 && !P->next.xg_out;
        if (!ok) continue;
        { // a declared barrier between the two launches keeps them independent (zgml_hip_program_set_barriers)
            const uint64_t lo_b = std::min(p->plan[i - 1].op_lo, p->plan[i].op_lo), hi_b = std::max(p->plan[i - 1].op_hi, p->plan[i].op_hi);
            bool cut = false;
            for (uint64_t b : p->barriers) cut = cut || (b > lo_b && b <= hi_b);
            if (cut) continue;
        }
        const size_t need = ((size_t)C->K + C->K / 16) * sizeof(float);
        if (p->prenorm_bytes < need) { // one block for the whole program: the pairs follow each other in stream order
            float* blk = nullptr;
            if (hipMalloc((void**)&blk, need) != hipSuccess || memset_sync(p->ctx->stream, blk, 0, need) != hipSuccess) {
                if (blk) hipFree(blk);
                continue;
            }
            p->owned.push_back(blk);
            p->prenorm_buf = blk, p->prenorm_bytes = need;
        }
        float *xg = p->prenorm_buf, *ssq = p->prenorm_buf + C->K;
        P->next = QmvNextNorm{C->pro.b, xg, ssq};
        C->pro.kind = QMV_PRO_PRENORM;
        C->pro.xg = xg, C->pro.ssq = ssq, C->pro.n_ssq = C->K / 16;
    }
}

// gate / up -> down: the grouped launch {gate with the SiLU chain, up} directly followed by the mat-vec whose MUL prologue
// multiplies exactly silu(gate) and up becomes a PAIR launch (qmatvec.hip: qmatvec_kon_pair_kernel): one workgroup computes
// the same 16 columns of both matrices (x is loaded once) and stores the product itself; the down projection then streams
// one vector with no prologue (x = a * b per lane meant two 44 KB vectors per workgroup at Llama-2-7B's d_ff). K-on-lanes
// weights, and since round 4 the x-direct n-on-lanes launches of short-K models (qmv_pair_ok: SmolLM-135M, whose down projection
// spent 0.5 us more than the O projection in front of its first load on the second vector).
void arm_pair(zgml_hip_program* p) {
    // (ZGML_QMV_EPI_SILU=0 asks for the SiLU chain through the generic step interpreter: the pair launch IS a fused SiLU epilogue)
    static const bool on = !(getenv("ZGML_HIP_PAIR") && atoi(getenv("ZGML_HIP_PAIR")) == 0) && !(getenv("ZGML_QMV_EPI_SILU") && atoi(getenv("ZGML_QMV_EPI_SILU")) == 0);
    if (!on) return;
    for (size_t i = 1; i < p->plan.size(); i++) {
        const auto D = p->plan[i].qmv_desc, G = p->plan[i - 1].qmv_desc;
        if (!D || !G || D->pro.kind != QMV_PRO_MUL || !D->pro.store_x || G->n_parts != 2 || G->pair_out) continue;
        const QmvPart &ga = G->parts[0], &up = G->parts[1];
        const QmvEpiStep* st = ga.epi;
        // the SiLU chain exactly as build_qmv_args recognises it (NEG, EXP [store], ADD vector, RECIP, MUL by the gate [store])
        bool ok = ga.n_epi == 5 && st[0].op == ZGML_OP_NEG && !st[0].store && st[1].op == ZGML_OP_EXP && st[1].store && st[2].op == ZGML_OP_ADD &&
                  st[2].operand && st[2].operand != ga.dst && st[2].operand != st[1].store && !st[2].store && st[3].op == ZGML_OP_RECIP && !st[3].store &&
                  st[4].op == ZGML_OP_MUL && st[4].operand == ga.dst && st[4].store && up.n_epi == 0;
        ok = ok && ga.w.format == up.w.format && qmv_pair_ok(ga.w) && qmv_pair_ok(up.w) && ga.w.N == up.w.N && ga.w.K == up.w.K && (ga.w.N / 16) % 2 == 0 && ga.w.N == D->K &&
             (G->pro.kind == QMV_PRO_NONE || G->pro.kind == QMV_PRO_PRENORM) && D->n_parts == 1;
        // the product's operands: silu(gate) and up, in either order (an f32 product does not depend on it)
        ok = ok && ((D->pro.a == st[4].store && D->pro.b == up.dst) || (D->pro.b == st[4].store && D->pro.a == up.dst));
        // (n-on-lanes form under a prepared norm: every workgroup stores a 16-element slice of the absorbed ops' outputs, and a pair
        // launch has one workgroup per column group of ONE matrix: it needs N >= K of them)
        ok = ok && (ga.w.format == QW_Q4K || G->pro.kind != QMV_PRO_PRENORM || ga.w.N >= G->K);
        // the two weights back to back in the arenas (the pair kernel finds part 1 from part 0)
        ok = ok && (const char*)up.w.qs == (const char*)ga.w.qs + ga.w.qs_bytes && (const char*)up.w.sc == (const char*)ga.w.sc + ga.w.sc_bytes;
        ok = ok && ((uintptr_t)G->pro.a % 16 == 0) && (G->K % 4 == 0) && (G->pro.kind != QMV_PRO_PRENORM || (uintptr_t)G->pro.xg % 16 == 0);
        if (!ok) continue;
        { // a declared barrier between the two launches keeps them independent (zgml_hip_program_set_barriers)
            const uint64_t lo_b = std::min(p->plan[i - 1].op_lo, p->plan[i].op_lo), hi_b = std::max(p->plan[i - 1].op_hi, p->plan[i].op_hi);
            bool cut = false;
            for (uint64_t b : p->barriers) cut = cut || (b > lo_b && b <= hi_b);
            if (cut) continue;
        }
        G->pair_out = D->pro.store_x;
        QmvPrologue plain;
        plain.a = D->pro.store_x;
        D->pro = plain;
    }
}

void build_fused_plan(zgml_hip_program* p) {
    const auto& ops = p->ops;
    const size_t n = ops.size();
    const Schedule& s0 = p->sched; // per-op access spans
    { // constant repeats (see zgml_hip_program::hoist_op)
        static const bool hoist_on = !(getenv("ZGML_HIP_HOIST_REPEAT") && atoi(getenv("ZGML_HIP_HOIST_REPEAT")) == 0);
        p->hoist_op.assign(n, 0);
        p->hoist_guard.assign(p->bufs.size(), 0);
        if (hoist_on && p->hoist_ok && p->barriers.empty()) {
            std::vector<uint32_t> writers(p->bufs.size(), 0);
            for (size_t i = 0; i < n; i++) {
                std::vector<uint16_t> seen;
                for (const Span& w : s0.access[i].writes)
                    if (w.buf < writers.size() && std::find(seen.begin(), seen.end(), w.buf) == seen.end()) writers[w.buf]++, seen.push_back(w.buf);
            }
            for (size_t i = 0; i < n; i++) {
                if (ops[i].kind != ZGML_DOP_REPEAT) continue;
                const auto& r = ops[i].u.repeat;
                if (r.src >= writers.size() || r.dst >= writers.size() || r.src == r.dst || writers[r.src] != 0 || writers[r.dst] != 1) continue;
                p->hoist_op[i] = 1;
                p->hoist_guard[r.src] = p->hoist_guard[r.dst] = 1;
            }
        }
    }
    std::vector<int> owner(n, -1);  // op -> macro id that absorbed it
    std::vector<Macro> macros;

    auto exact_dst = [&](uint32_t i, ExactSpan& out) -> bool { // contiguous vector outputs only
        const zgml_device_op& o = ops[i];
        switch (o.kind) {
            case ZGML_DOP_ELEMENTWISE: out = {o.u.elementwise.dst, o.u.elementwise.dst_offset, o.u.elementwise.n}; return true;
            case ZGML_DOP_FUSED_ELEMENTWISE:
                out = {o.u.fused_elementwise.dst, o.u.fused_elementwise.dst_offset, o.u.fused_elementwise.n};
                return true;
            case ZGML_DOP_RMSNORM:
                if (o.u.rmsnorm.rows != 1) return false;
                out = {o.u.rmsnorm.dst, o.u.rmsnorm.dst_offset, o.u.rmsnorm.cols};
                return true;
            case ZGML_DOP_QMATMUL:
                if (o.u.qmatmul.M != 1) return false;
                out = {o.u.qmatmul.dst, o.u.qmatmul.dst_offset, o.u.qmatmul.N};
                return true;
            default: return false;
        }
    };
    auto last_writer = [&](const ExactSpan& sp, uint32_t before) -> int { // latest op < before writing into sp
        const Span q{sp.buf, sp.off, sp.off + sp.n};
        for (int j = (int)before - 1; j >= 0; j--)
            for (const Span& w : s0.access[j].writes)
                if (spans_overlap(w, q)) return j;
        return -1;
    };
    auto readers_until_overwrite = [&](const ExactSpan& sp, uint32_t after, std::vector<uint32_t>& out) {
        const Span q{sp.buf, sp.off, sp.off + sp.n};
        for (uint32_t j = after + 1; j < n; j++) {
            for (const Span& r : s0.access[j].reads)
                if (spans_overlap(r, q)) {
                    out.push_back(j);
                    break;
                }
            for (const Span& w : s0.access[j].writes)
                if (spans_overlap(w, q)) return;
        }
    };
    // members may be delayed to `last` only if they conflict with no non-member in between
    auto delay_legal = [&](const std::vector<uint32_t>& members, uint32_t last) -> bool {
        for (uint32_t x : members)
            for (uint32_t c = x + 1; c < last; c++) {
                if (std::find(members.begin(), members.end(), c) != members.end()) continue;
                if (ops_conflict(s0.access[x], s0.access[c])) return false;
            }
        return !members.empty() && !barrier_between(p->barriers, members.front(), last);
    };

    // ---- prologues: one absorbed producer chain per input vector, shared by all its consumers
    struct ProInfo {
        QmvPrologue pro;
        std::vector<uint32_t> absorbed;  // P (and Q)
        std::vector<uint32_t> consumers; // qmatmul ops
        OpAccess reads;                  // what a non-owner consumer reads instead of P.dst
    };
    std::vector<ProInfo> pros;
    std::vector<int> pro_of(n, -1);
    for (uint32_t i = 0; i < n; i++) {
        if (!anchor_ok(p, i) || pro_of[i] >= 0) continue;
        const auto& q = ops[i].u.qmatmul;
        const ExactSpan in{q.input, q.input_offset, q.K};
        const int P = last_writer(in, i);
        if (P < 0 || owner[P] >= 0) continue;
        const zgml_device_op& po = ops[P];
        if (po.kind != ZGML_DOP_ELEMENTWISE || po.u.elementwise.op != ZGML_OP_MUL) continue;
        const auto& e = po.u.elementwise;
        if (!(ExactSpan{e.dst, e.dst_offset, e.n} == in)) continue;
        if ((e.dst_offset % 4) || (e.src0_offset % 4) || (e.src1_offset % 4)) continue;
        std::vector<uint32_t> readers;
        readers_until_overwrite(in, (uint32_t)P, readers);
        bool ok = !readers.empty();
        for (uint32_t r : readers) {
            if (!anchor_ok(p, r)) ok = false;
            else {
                const auto& rq = ops[r].u.qmatmul;
                if (!(ExactSpan{rq.input, rq.input_offset, rq.K} == in)) ok = false;
                if (rq.K > qmv_max_prologue_k(p->qweights[rq.weight_idx])) ok = false;
            }
        }
        if (!ok) continue;
        ProInfo info;
        info.absorbed = {(uint32_t)P};
        info.consumers = readers;
        info.pro.kind = QMV_PRO_MUL;
        info.pro.a = buf_at(p, e.src0, e.src0_offset);
        info.pro.b = buf_at(p, e.src1, e.src1_offset);
        info.pro.store_x = buf_at(p, e.dst, e.dst_offset);
        info.reads.reads = s0.access[P].reads;
        // rmsnorm feeding one side of the mul?
        for (int side = 0; side < 2; side++) {
            const ExactSpan sp = side == 0 ? ExactSpan{e.src0, e.src0_offset, e.n} : ExactSpan{e.src1, e.src1_offset, e.n};
            const int Q = last_writer(sp, (uint32_t)P);
            if (Q < 0 || owner[Q] >= 0 || ops[Q].kind != ZGML_DOP_RMSNORM) continue;
            const auto& rn = ops[Q].u.rmsnorm;
            ExactSpan qd{};
            if (!exact_dst((uint32_t)Q, qd) || !(qd == sp) || (rn.src_offset % 4) || (rn.dst_offset % 4)) continue;
            std::vector<uint32_t> qreaders;
            readers_until_overwrite(sp, (uint32_t)Q, qreaders);
            if (qreaders.size() != 1 || qreaders[0] != (uint32_t)P) continue;
            info.absorbed = {(uint32_t)Q, (uint32_t)P};
            info.pro.kind = QMV_PRO_RMSNORM_MUL;
            info.pro.eps = rn.eps;
            info.pro.a = buf_at(p, rn.src, rn.src_offset);
            info.pro.b = side == 0 ? buf_at(p, e.src1, e.src1_offset) : buf_at(p, e.src0, e.src0_offset);
            info.pro.store_mid = buf_at(p, rn.dst, rn.dst_offset);
            info.reads.reads = s0.access[Q].reads;
            info.reads.reads.push_back(side == 0 ? s0.access[P].reads[1] : s0.access[P].reads[0]);
            break;
        }
        // the absorbed ops are delayed to the first consumer, and their inputs must survive until the last
        std::vector<uint32_t> mem = info.absorbed;
        if (!delay_legal(mem, info.consumers.front())) continue;
        // every consumer recomputes the prologue from its inputs, so each input must keep its value
        // from the op that originally read it (Q for the rmsnorm source, P for the rest) up to the
        // last consumer
        bool inputs_live = true;
        auto live = [&](const Span& r, uint32_t from) {
            for (uint32_t c = from + 1; c < info.consumers.back() && inputs_live; c++) {
                if (std::find(info.absorbed.begin(), info.absorbed.end(), c) != info.absorbed.end()) continue;
                for (const Span& w : s0.access[c].writes)
                    if (spans_overlap(r, w)) inputs_live = false;
            }
        };
        if (info.pro.kind == QMV_PRO_RMSNORM_MUL) {
            live(info.reads.reads[0], info.absorbed.front()); // x, read by the rmsnorm
            live(info.reads.reads[1], (uint32_t)P);            // gamma (repeat output), read by the mul
        } else {
            for (const Span& r : info.reads.reads) live(r, (uint32_t)P);
        }
        if (!inputs_live || barrier_between(p->barriers, info.absorbed.front(), info.consumers.back())) continue;
        const int id = (int)pros.size();
        for (uint32_t c : info.consumers) pro_of[c] = id;
        for (uint32_t x : info.absorbed) owner[x] = -2; // claimed; macro id assigned below
        pros.push_back(std::move(info));
    }

    // ---- one macro per mat-vec anchor: (prologue) + anchor + epilogue chain
    std::vector<char> in_macro(n, 0);
    for (uint32_t i = 0; i < n; i++) {
        if (!anchor_ok(p, i)) continue;
        Macro m;
        m.qmv = true;
        m.anchor = i;
        const auto& q = ops[i].u.qmatmul;
        m.pro.kind = QMV_PRO_NONE;
        m.pro.a = buf_at(p, q.input, q.input_offset);
        OpAccess acc = s0.access[i];
        if (pro_of[i] >= 0) {
            const ProInfo& info = pros[pro_of[i]];
            m.pro = info.pro;
            m.owns_prologue = info.consumers.front() == i;
            if (m.owns_prologue) {
                for (uint32_t x : info.absorbed) {
                    m.members.push_back(x);
                    add_access(acc, s0.access[x]);
                }
            } else { // recompute without storing: read the prologue's inputs instead of its output
                m.pro.store_mid = nullptr;
                m.pro.store_x = nullptr;
                acc.reads = info.reads.reads;
            }
        }
        m.pro_sig[0] = m.pro.kind, m.pro_sig[1] = (uint64_t)(uintptr_t)m.pro.a, m.pro_sig[2] = (uint64_t)(uintptr_t)m.pro.b;
        m.members.push_back(i);
        // epilogue chain
        ExactSpan cur{q.dst, q.dst_offset, q.N};
        uint32_t cur_idx = i;
        for (;;) {
            std::vector<uint32_t> readers;
            readers_until_overwrite(cur, cur_idx, readers);
            int B = -1;
            for (uint32_t r : readers)
                if (!in_macro[r] && owner[r] == -1 &&
                    (ops[r].kind == ZGML_DOP_ELEMENTWISE || ops[r].kind == ZGML_DOP_FUSED_ELEMENTWISE)) {
                    B = (int)r;
                    break;
                }
            if (B < 0) break;
            const zgml_device_op& bo = ops[B];
            QmvEpiStep steps[kMaxEpiSteps];
            uint32_t ns = 0;
            ExactSpan bdst{};
            bool ok = exact_dst((uint32_t)B, bdst) && bdst.n == cur.n;
            if (ok && bo.kind == ZGML_DOP_ELEMENTWISE) {
                const auto& e = bo.u.elementwise;
                const bool binary = e.op == ZGML_OP_ADD || e.op == ZGML_OP_MUL;
                const ExactSpan s0sp{e.src0, e.src0_offset, e.n}, s1sp{e.src1, e.src1_offset, e.n};
                if (s0sp == cur)
                    steps[ns++] = {e.op, 0, binary ? buf_at(p, e.src1, e.src1_offset) : nullptr, nullptr};
                else if (binary && s1sp == cur)
                    steps[ns++] = {e.op, 1, buf_at(p, e.src0, e.src0_offset), nullptr};
                else
                    ok = false;
            } else if (ok) {
                const auto& f = bo.u.fused_elementwise;
                ok = ExactSpan{f.src, f.src_offset, f.n} == cur && f.n_steps <= (uint32_t)kMaxEpiSteps;
                for (uint32_t t = 0; ok && t < f.n_steps; t++) {
                    const bool binary = f.steps[t].op == ZGML_OP_ADD || f.steps[t].op == ZGML_OP_MUL;
                    steps[ns++] = {f.steps[t].op, f.steps[t].is_swapped,
                                   binary ? buf_at(p, f.steps[t].secondary_buf, f.steps[t].secondary_offset) : nullptr, nullptr};
                }
            }
            if (!ok || ns == 0 || m.n_epi + ns > (uint32_t)kMaxEpiSteps) break;
            std::vector<uint32_t> trial = m.members;
            trial.push_back((uint32_t)B);
            if (!delay_legal(trial, (uint32_t)B)) break;
            steps[ns - 1].store = buf_at(p, bdst.buf, bdst.off);
            for (uint32_t t = 0; t < ns; t++) m.epi[m.n_epi++] = steps[t];
            m.members.push_back((uint32_t)B);
            in_macro[B] = 1;
            add_access(acc, s0.access[B]);
            cur = bdst;
            cur_idx = (uint32_t)B;
        }
        if (m.members.size() > 1 && !delay_legal(m.members, m.members.back())) { // absorbed prologue + epilogue together
            // fall back to the bare anchor (keeps correctness trivially)
            for (uint32_t x : m.members)
                if (x != i) in_macro[x] = 0;
            Macro plain;
            plain.qmv = true, plain.anchor = i, plain.members = {i};
            plain.pro.kind = QMV_PRO_NONE, plain.pro.a = buf_at(p, q.input, q.input_offset);
            plain.pro_sig[0] = 0, plain.pro_sig[1] = (uint64_t)(uintptr_t)plain.pro.a;
            plain.access = s0.access[i];
            m = plain;
            if (pro_of[i] >= 0) { // the prologue ops must then run on their own
                for (uint32_t x : pros[pro_of[i]].absorbed) owner[x] = -1;
                for (uint32_t c : pros[pro_of[i]].consumers) pro_of[c] = -1;
            }
        } else {
            m.access = acc;
        }
        std::sort(m.members.begin(), m.members.end());
        m.position = m.members.back();
        for (uint32_t x : m.members) in_macro[x] = 1;
        macros.push_back(std::move(m));
    }
    // consumers whose prologue was revoked after their macro was built: rebuild plainly
    for (Macro& m : macros)
        if (m.qmv && m.pro.kind != QMV_PRO_NONE && pro_of[m.anchor] < 0) {
            const auto& q = ops[m.anchor].u.qmatmul;
            std::vector<uint32_t> keep;
            for (uint32_t x : m.members)
                if (x >= m.anchor) keep.push_back(x);
            for (uint32_t x : m.members)
                if (x < m.anchor) in_macro[x] = 0;
            m.members = keep;
            m.pro = QmvPrologue{};
            m.pro.a = buf_at(p, q.input, q.input_offset);
            m.pro_sig[0] = 0, m.pro_sig[1] = (uint64_t)(uintptr_t)m.pro.a, m.pro_sig[2] = 0;
            m.access = OpAccess{};
            for (uint32_t x : m.members) add_access(m.access, s0.access[x]);
        }
    // ---- decode attention (seq_q == 1): per kv group {rope k, K store, V store} + per head
    // {rope q, attention, row store} become one launch record per head (AttnDecodeParams). The new
    // K/V column is recomputed by every head of the group from the projections and never re-read
    // from the cache, so the heads of a group need no ordering among themselves.
    {
        auto last_writer_span = [&](const Span& q, uint32_t before) -> int {
            for (int j = (int)before - 1; j >= 0; j--)
                for (const Span& w : s0.access[j].writes)
                    if (spans_overlap(w, q)) return j;
            return -1;
        };
        auto pow2 = [](uint32_t v) { return v && (v & (v - 1)) == 0; };
        auto aligned4 = [&](uint16_t buf, uint64_t off) { return ((uintptr_t)buf_at(p, buf, off) % 16) == 0; };
        struct Cand {
            uint32_t rq, att, rk, sk, sv;
            int row_store;
        };
        std::map<uint32_t, std::vector<Cand>> groups; // by K store op
        static const bool enabled = !(getenv("ZGML_HIP_ATTN_DECODE") && atoi(getenv("ZGML_HIP_ATTN_DECODE")) == 0);
        for (uint32_t i = 0; enabled && i < n; i++) {
            if (in_macro[i] || ops[i].kind != ZGML_DOP_ATTENTION) continue;
            const auto& t = ops[i].u.attention;
            const uint32_t dh = t.d_head;
            if (t.seq_q != 1 || !pow2(dh) || dh < 8 || dh > 256 || t.q_rs != 1 || t.k_rs != 1 || t.v_rs != 1 || t.dst_rs != 1 ||
                t.k_cs % 4 || t.v_cs % 4 || t.seq_kv == 0 || !aligned4(t.q, t.q_off) || !aligned4(t.k, t.k_off) ||
                !aligned4(t.v, t.v_off) || !aligned4(t.dst, t.dst_off))
                continue;
            // query rope
            const int rq = last_writer(ExactSpan{t.q, t.q_off, dh}, i);
            if (rq < 0 || in_macro[rq] || ops[rq].kind != ZGML_DOP_ROPE) continue;
            const auto& q = ops[rq].u.rope;
            if (q.dst != t.q || q.dst_off != t.q_off || 2 * q.half_d != dh || q.seq_len != 1 || q.src_rs != 1 ||
                !aligned4(q.src, q.src_off) || !aligned4(q.cos_sin, q.cs_off))
                continue;
            // K store (dynamic column) fed by the key rope
            const int sk = last_writer_span(Span{t.k, t.k_off, t.k_off + (uint64_t)(t.seq_kv - 1) * t.k_cs + dh}, i);
            if (sk < 0 || in_macro[sk] || ops[sk].kind != ZGML_DOP_SLICE_ASSIGN) continue;
            const auto& ks = ops[sk].u.slice_assign;
            if (ks.dst != t.k || !ks.patch_stride || ks.patch_stride != t.k_cs || ks.dst_base_offset != t.k_off || ks.rows != dh ||
                ks.cols != 1 || ks.dst_row_stride != 1 || ks.src_row_stride != 1)
                continue;
            const int rk = last_writer(ExactSpan{ks.src, ks.src_offset, dh}, sk);
            if (rk < 0 || in_macro[rk] || ops[rk].kind != ZGML_DOP_ROPE) continue;
            const auto& kr = ops[rk].u.rope;
            if (kr.dst != ks.src || kr.dst_off != ks.src_offset || 2 * kr.half_d != dh || kr.seq_len != 1 || kr.src_rs != 1 ||
                !aligned4(kr.src, kr.src_off) || !aligned4(kr.cos_sin, kr.cs_off) || !aligned4(kr.dst, kr.dst_off))
                continue;
            // V store
            const int sv = last_writer_span(Span{t.v, t.v_off, t.v_off + (uint64_t)(t.seq_kv - 1) * t.v_cs + dh}, i);
            if (sv < 0 || in_macro[sv] || ops[sv].kind != ZGML_DOP_SLICE_ASSIGN) continue;
            const auto& vs = ops[sv].u.slice_assign;
            if (vs.dst != t.v || !vs.patch_stride || vs.patch_stride != t.v_cs || vs.dst_base_offset != t.v_off || vs.rows != dh ||
                vs.cols != 1 || vs.dst_row_stride != 1 || vs.src_row_stride != 1 || !aligned4(vs.src, vs.src_offset))
                continue;
            // the slabs must hold seq_kv (compile-time bound) whole columns: speculative reads stay inside
            if (t.k_off + (uint64_t)(t.seq_kv - 1) * t.k_cs + dh > p->sizes[t.k] || t.v_off + (uint64_t)(t.seq_kv - 1) * t.v_cs + dh > p->sizes[t.v])
                continue;
            // optional row store of the dense head output
            int row_store = -1;
            if (t.dst_cs == dh) {
                std::vector<uint32_t> readers;
                readers_until_overwrite(ExactSpan{t.dst, t.dst_off, dh}, i, readers);
                for (uint32_t r : readers) {
                    if (in_macro[r] || ops[r].kind != ZGML_DOP_SLICE_ASSIGN) continue;
                    const auto& sa = ops[r].u.slice_assign;
                    if (sa.src != t.dst || sa.src_offset != t.dst_off || sa.rows != dh || sa.cols != 1 || sa.src_row_stride != 1) continue;
                    row_store = (int)r;
                    break;
                }
            }
            groups[(uint32_t)sk].push_back({(uint32_t)rq, i, (uint32_t)rk, (uint32_t)sk, (uint32_t)sv, row_store});
        }
        for (auto& kv : groups) {
            const std::vector<Cand>& hs = kv.second;
            bool same = true;
            for (const Cand& c : hs) same = same && c.rk == hs[0].rk && c.sv == hs[0].sv;
            if (!same) continue;
            Macro m;
            m.rk = hs[0].rk, m.sk = hs[0].sk, m.sv = hs[0].sv;
            m.members = {m.rk, m.sk, m.sv};
            for (const Cand& c : hs) {
                m.members.push_back(c.rq);
                m.members.push_back(c.att);
                if (c.row_store >= 0) m.members.push_back((uint32_t)c.row_store);
                m.heads.push_back({c.rq, c.att, c.row_store});
            }
            std::sort(m.members.begin(), m.members.end());
            if (std::adjacent_find(m.members.begin(), m.members.end()) != m.members.end()) continue; // an op claimed twice
            m.position = m.members.back();
            if (!delay_legal(m.members, m.position)) continue;
            for (uint32_t x : m.members) {
                add_access(m.access, s0.access[x]);
                in_macro[x] = 1;
            }
            m.anchor = hs[0].att;
            macros.push_back(std::move(m));
        }
    }

    // ---- the same fold over quantised KV caches (extension ops): per kv head {rope k, kvq_store K, kvq_store V}
    // + per head {rope q, attention_kvq, row store}. The fused kernel quantises the new column exactly as
    // storeColumn does and uses the quantised values, so cache bytes and outputs equal the op-by-op plan's.
    {
        auto whole_writer = [&](uint16_t buf, uint32_t before) -> int { // latest op < before writing anywhere into buf
            const Span q{buf, 0, p->sizes[buf]};
            for (int j = (int)before - 1; j >= 0; j--)
                for (const Span& w : s0.access[j].writes)
                    if (spans_overlap(w, q)) return j;
            return -1;
        };
        auto aligned4 = [&](uint16_t buf, uint64_t off) { return ((uintptr_t)buf_at(p, buf, off) % 16) == 0; };
        struct Cand {
            uint32_t rq, att, rk, sk, sv;
            int row_store;
        };
        std::map<uint32_t, std::vector<Cand>> groups; // by K store op
        static const bool enabled = !(getenv("ZGML_HIP_ATTN_DECODE") && atoi(getenv("ZGML_HIP_ATTN_DECODE")) == 0) &&
                                    !(getenv("ZGML_HIP_ATTN_DECODE_KVQ") && atoi(getenv("ZGML_HIP_ATTN_DECODE_KVQ")) == 0);
        for (uint32_t i = 0; enabled && i < n; i++) {
            if (in_macro[i] || ops[i].kind != ZGML_DOP_ATTENTION_KVQ) continue;
            const auto& t = ops[i].u.attention_kvq;
            const uint32_t dh = t.d_head;
            if (t.seq_q != 1 || t.block_size != 32 || (dh != 32 && dh != 64 && dh != 128 && dh != 256) || t.k_col_start || t.v_col_start ||
                t.seq_kv == 0 || t.n_cols == 0 || !aligned4(t.q, t.q_off) || !aligned4(t.dst, t.dst_off) || t.k == t.v)
                continue;
            const uint64_t cache_elems = (uint64_t)t.n_cols * dh / 4 + (uint64_t)t.n_cols * (dh / 32);
            if (cache_elems > p->sizes[t.k] || cache_elems > p->sizes[t.v]) continue;
            // (the key loop addresses cache rows, scales and mask words as a uniform base + a 32-bit byte offset: attention_decode.h)
            if (cache_elems * 4 >= (1ull << 32) || (t.has_mask && (uint64_t)t.n_cols * t.mask_rs * 4 >= (1ull << 32))) continue;
            const int rq = last_writer(ExactSpan{t.q, t.q_off, dh}, i);
            if (rq < 0 || in_macro[rq] || ops[rq].kind != ZGML_DOP_ROPE) continue;
            const auto& q = ops[rq].u.rope;
            if (q.dst != t.q || q.dst_off != t.q_off || 2 * q.half_d != dh || q.seq_len != 1 || q.src_rs != 1 ||
                !aligned4(q.src, q.src_off) || !aligned4(q.cos_sin, q.cs_off))
                continue;
            const int sk = whole_writer(t.k, i), sv = whole_writer(t.v, i);
            if (sk < 0 || sv < 0 || in_macro[sk] || in_macro[sv] || ops[sk].kind != ZGML_DOP_KVQ_STORE || ops[sv].kind != ZGML_DOP_KVQ_STORE) continue;
            const auto &ks = ops[sk].u.kvq_store, &vs = ops[sv].u.kvq_store;
            if (ks.cache != t.k || vs.cache != t.v || ks.d_head != dh || vs.d_head != dh || ks.block_size != 32 || vs.block_size != 32 ||
                ks.n_cols != t.n_cols || vs.n_cols != t.n_cols || !aligned4(vs.src, vs.src_offset))
                continue;
            const int rk = last_writer(ExactSpan{ks.src, ks.src_offset, dh}, sk);
            if (rk < 0 || in_macro[rk] || ops[rk].kind != ZGML_DOP_ROPE) continue;
            const auto& kr = ops[rk].u.rope;
            if (kr.dst != ks.src || kr.dst_off != ks.src_offset || 2 * kr.half_d != dh || kr.seq_len != 1 || kr.src_rs != 1 ||
                !aligned4(kr.src, kr.src_off) || !aligned4(kr.cos_sin, kr.cs_off) || !aligned4(kr.dst, kr.dst_off))
                continue;
            int row_store = -1;
            if (t.dst_cs == dh) {
                std::vector<uint32_t> readers;
                readers_until_overwrite(ExactSpan{t.dst, t.dst_off, dh}, i, readers);
                for (uint32_t r : readers) {
                    if (in_macro[r] || ops[r].kind != ZGML_DOP_SLICE_ASSIGN) continue;
                    const auto& sa = ops[r].u.slice_assign;
                    if (sa.src != t.dst || sa.src_offset != t.dst_off || sa.rows != dh || sa.cols != 1 || sa.src_row_stride != 1) continue;
                    row_store = (int)r;
                    break;
                }
            }
            groups[(uint32_t)sk].push_back({(uint32_t)rq, i, (uint32_t)rk, (uint32_t)sk, (uint32_t)sv, row_store});
        }
        for (auto& kv : groups) {
            const std::vector<Cand>& hs = kv.second;
            bool same = true;
            for (const Cand& c : hs) same = same && c.rk == hs[0].rk && c.sv == hs[0].sv;
            if (!same) continue;
            Macro m;
            m.kvq = true;
            m.rk = hs[0].rk, m.sk = hs[0].sk, m.sv = hs[0].sv;
            m.members = {m.rk, m.sk, m.sv};
            for (const Cand& c : hs) {
                m.members.push_back(c.rq);
                m.members.push_back(c.att);
                if (c.row_store >= 0) m.members.push_back((uint32_t)c.row_store);
                m.heads.push_back({c.rq, c.att, c.row_store});
            }
            std::sort(m.members.begin(), m.members.end());
            if (std::adjacent_find(m.members.begin(), m.members.end()) != m.members.end()) continue;
            m.position = m.members.back();
            if (!delay_legal(m.members, m.position)) continue;
            for (uint32_t x : m.members) {
                add_access(m.access, s0.access[x]);
                in_macro[x] = 1;
            }
            m.anchor = hs[0].att;
            macros.push_back(std::move(m));
        }
    }

    // rope -> slice_assign (K into the cache) and attention -> slice_assign (head output into the
    // concatenated buffer): the copy of the anchor's dense output is done by the anchor itself
    for (uint32_t i = 0; i < n; i++) {
        if (in_macro[i] || (ops[i].kind != ZGML_DOP_ROPE && ops[i].kind != ZGML_DOP_ATTENTION)) continue;
        ExactSpan out{};
        uint32_t rows, cols, srs, scs;
        if (ops[i].kind == ZGML_DOP_ROPE) {
            const auto& r = ops[i].u.rope;
            rows = 2 * r.half_d, cols = r.seq_len, srs = 1, scs = rows;
            out = {r.dst, r.dst_off, (uint64_t)rows * cols};
        } else {
            const auto& t = ops[i].u.attention;
            rows = t.d_head, cols = t.seq_q, srs = t.dst_rs, scs = t.dst_cs;
            if (srs != 1 || scs != rows) continue; // dense head output only
            out = {t.dst, t.dst_off, (uint64_t)rows * cols};
        }
        std::vector<uint32_t> readers;
        readers_until_overwrite(out, i, readers);
        for (uint32_t r : readers) {
            if (in_macro[r] || ops[r].kind != ZGML_DOP_SLICE_ASSIGN) continue;
            const auto& sa = ops[r].u.slice_assign;
            if (sa.src != out.buf || sa.src_offset != out.off || sa.rows != rows || sa.cols != cols ||
                sa.src_row_stride != srs || sa.src_col_stride != scs)
                continue;
            if (!delay_legal({i, r}, r)) continue;
            Macro m;
            m.members = {i, r};
            m.position = r;
            m.anchor = i;
            m.store = (int)r;
            m.access = s0.access[i];
            add_access(m.access, s0.access[r]);
            in_macro[i] = in_macro[r] = 1;
            macros.push_back(std::move(m));
            break;
        }
    }
    // [elementwise add ->] rmsnorm [-> elementwise mul] on the same dense rows: one launch instead of three
    // (the M > 1 / prefill form of what the mat-vec prologue does at M = 1; every intermediate is still stored)
    {
        static const bool enabled = !(getenv("ZGML_HIP_ROW_CHAIN") && atoi(getenv("ZGML_HIP_ROW_CHAIN")) == 0);
        auto free_op = [&](int j) { return j >= 0 && !in_macro[j] && owner[j] == -1; };
        for (uint32_t i = 0; enabled && i < n; i++) {
            if (!free_op((int)i) || ops[i].kind != ZGML_DOP_RMSNORM) continue;
            const auto& rn = ops[i].u.rmsnorm;
            const uint64_t cnt = (uint64_t)rn.rows * rn.cols;
            if (!cnt) continue;
            const ExactSpan S{rn.src, rn.src_offset, cnt}, D{rn.dst, rn.dst_offset, cnt};
            int A = last_writer(S, i), P = -1;
            if (free_op(A) && ops[A].kind == ZGML_DOP_ELEMENTWISE && ops[A].u.elementwise.op == ZGML_OP_ADD) {
                const auto& e = ops[A].u.elementwise;
                if (!(ExactSpan{e.dst, e.dst_offset, e.n} == S)) A = -1;
            } else {
                A = -1;
            }
            std::vector<uint32_t> readers;
            readers_until_overwrite(D, i, readers);
            if (readers.size() == 1 && free_op((int)readers[0]) && ops[readers[0]].kind == ZGML_DOP_ELEMENTWISE &&
                ops[readers[0]].u.elementwise.op == ZGML_OP_MUL) {
                const auto& e = ops[readers[0]].u.elementwise;
                const bool s0 = ExactSpan{e.src0, e.src0_offset, e.n} == D, s1 = ExactSpan{e.src1, e.src1_offset, e.n} == D;
                if (e.n == cnt && (s0 != s1)) P = (int)readers[0];
            }
            // widest legal chain first
            const std::vector<std::vector<int>> tries = {{A, (int)i, P}, {(int)i, P}, {A, (int)i}};
            for (const auto& t : tries) {
                std::vector<uint32_t> mem;
                for (int x : t)
                    if (x >= 0) mem.push_back((uint32_t)x);
                if (mem.size() < 2 || std::find(t.begin(), t.end(), -1) != t.end()) continue;
                if (!delay_legal(mem, mem.back())) continue;
                Macro m;
                m.chain = true;
                m.members = mem;
                m.position = mem.back();
                m.anchor = i;
                m.chain_add = t.front() == (int)i ? -1 : t.front();
                m.chain_mul = t.back() == (int)i ? -1 : t.back();
                for (uint32_t x : mem) {
                    add_access(m.access, s0.access[x]);
                    in_macro[x] = 1;
                }
                macros.push_back(std::move(m));
                break;
            }
        }
    }
    // elementwise chains (the SiLU chain and its product with the up projection at M > 1, ...): ops over the
    // same n elements where each consumes its predecessor's output at the same index become one launch
    {
        static const bool enabled = !(getenv("ZGML_HIP_ELT_CHAIN") && atoi(getenv("ZGML_HIP_ELT_CHAIN")) == 0);
        auto free_op = [&](int j) { return j >= 0 && !in_macro[j] && owner[j] == -1; };
        auto is_elt = [&](uint32_t j) { return ops[j].kind == ZGML_DOP_ELEMENTWISE || ops[j].kind == ZGML_DOP_FUSED_ELEMENTWISE; };
        // append op j's steps; `cur` = the chain value's span (nullptr for the first op, which sets src)
        auto append = [&](EltChainParams& c, uint32_t j, const ExactSpan* cur, ExactSpan& out) -> bool {
            const zgml_device_op& o = ops[j];
            if (o.kind == ZGML_DOP_ELEMENTWISE) {
                const auto& e = o.u.elementwise;
                const bool binary = e.op == ZGML_OP_ADD || e.op == ZGML_OP_MUL;
                const ExactSpan a{e.src0, e.src0_offset, e.n}, b{e.src1, e.src1_offset, e.n};
                if (c.n_steps + 1 > (uint32_t)kMaxChainSteps) return false;
                ChainStepDev st{e.op, 0, nullptr, buf_at(p, e.dst, e.dst_offset)};
                if (!cur) {
                    c.src = buf_at(p, e.src0, e.src0_offset), c.n = e.n;
                    st.secondary = binary ? buf_at(p, e.src1, e.src1_offset) : nullptr;
                } else if (a == *cur && !(binary && b == *cur)) {
                    st.secondary = binary ? buf_at(p, e.src1, e.src1_offset) : nullptr;
                } else if (binary && b == *cur && !(a == *cur)) {
                    st.swapped = 1, st.secondary = buf_at(p, e.src0, e.src0_offset);
                } else {
                    return false;
                }
                c.steps[c.n_steps++] = st;
                out = {e.dst, e.dst_offset, e.n};
                return true;
            }
            const auto& f = o.u.fused_elementwise;
            if (f.n_steps == 0 || c.n_steps + f.n_steps > (uint32_t)kMaxChainSteps) return false;
            if (!cur)
                c.src = buf_at(p, f.src, f.src_offset), c.n = f.n;
            else if (!(ExactSpan{f.src, f.src_offset, f.n} == *cur))
                return false;
            for (uint32_t t = 0; t < f.n_steps; t++) {
                const bool binary = f.steps[t].op == ZGML_OP_ADD || f.steps[t].op == ZGML_OP_MUL;
                c.steps[c.n_steps++] = {f.steps[t].op, f.steps[t].is_swapped,
                                        binary ? buf_at(p, f.steps[t].secondary_buf, f.steps[t].secondary_offset) : nullptr, nullptr};
            }
            c.steps[c.n_steps - 1].store = buf_at(p, f.dst, f.dst_offset);
            out = {f.dst, f.dst_offset, f.n};
            return true;
        };
        for (uint32_t i = 0; enabled && i < n; i++) {
            if (!free_op((int)i) || !is_elt(i)) continue;
            Macro m;
            ExactSpan cur{};
            if (!append(m.elt, i, nullptr, cur)) continue;
            m.members = {i};
            for (;;) {
                std::vector<uint32_t> readers;
                readers_until_overwrite(cur, m.members.back(), readers);
                bool grown = false;
                for (uint32_t r : readers) {
                    if (!free_op((int)r) || !is_elt(r) || std::find(m.members.begin(), m.members.end(), r) != m.members.end()) continue;
                    EltChainParams trial_p = m.elt;
                    ExactSpan nxt{};
                    if (!append(trial_p, r, &cur, nxt) || nxt.n != cur.n) continue;
                    std::vector<uint32_t> trial = m.members;
                    trial.push_back(r);
                    if (!delay_legal(trial, r)) continue;
                    m.elt = trial_p, m.members = trial, cur = nxt, grown = true;
                    break;
                }
                if (!grown) break;
            }
            if (m.members.size() < 2) continue;
            m.elt_chain = true;
            m.position = m.members.back();
            m.anchor = i;
            for (uint32_t x : m.members) {
                add_access(m.access, s0.access[x]);
                in_macro[x] = 1;
            }
            macros.push_back(std::move(m));
        }
    }
    // everything else is a singleton
    for (uint32_t i = 0; i < n; i++) {
        if (in_macro[i]) continue;
        Macro m;
        m.members = {i};
        m.position = i;
        m.access = s0.access[i];
        macros.push_back(std::move(m));
    }
    std::sort(macros.begin(), macros.end(), [](const Macro& a, const Macro& b) { return a.position < b.position; });

    std::vector<OpAccess> access(macros.size());
    std::vector<uint64_t> position(macros.size());
    for (size_t i = 0; i < macros.size(); i++) access[i] = macros[i].access, position[i] = macros[i].position;
    std::vector<uint32_t> level;
    std::vector<std::vector<uint32_t>> levels;
    levels_from_access(access, position, p->barriers, level, levels);

    for (const auto& lv : levels) {
        std::vector<PlanItem> plain_ops;
        std::vector<Launch> chains; // the items of a level are mutually independent: their order is free
        std::vector<const Macro*> qmvs;
        std::map<uint32_t, std::vector<AttnDecodeParams>> adec_by_dh; // one launch per head size
        uint32_t adec_lo = UINT32_MAX, adec_hi = 0, adec_ops = 0;
        for (uint32_t mi : lv) {
            if (!macros[mi].heads.empty() && macros[mi].kvq) {
                const Macro& m = macros[mi];
                const auto& kr = ops[m.rk].u.rope;
                const auto& ks = ops[m.sk].u.kvq_store;
                const auto& vs = ops[m.sv].u.kvq_store;
                bool first_head = true;
                for (const Macro::Head& h : m.heads) {
                    const auto& t = ops[h.att].u.attention_kvq;
                    const auto& qr = ops[h.rq].u.rope;
                    AttnDecodeParams a{};
                    a.att.dst = buf_at(p, t.dst, t.dst_off), a.att.q = buf_at(p, t.q, t.q_off);
                    a.att.mask = t.has_mask ? buf_at(p, t.mask, t.mask_off) : p->zero_word;
                    a.att.mask_rs = t.has_mask ? t.mask_rs : 0, a.att.mask_cs = t.has_mask ? t.mask_cs : 0;
                    a.att.d_head = t.d_head, a.att.seq_q = 1, a.att.dyn_seq_kv = p->dyn_dev + h.att, a.att.scale = t.scale;
                    a.att.q_rs = 1, a.att.q_cs = t.q_cs, a.att.dst_rs = 1, a.att.dst_cs = t.dst_cs;
                    if (h.row_store >= 0) {
                        const auto& sa = ops[h.row_store].u.slice_assign;
                        a.att.dst2 = p->bufs[sa.dst];
                        a.att.dyn_dst2_off = p->dyn_dev + h.row_store;
                        a.att.d2_rs = sa.dst_row_stride, a.att.d2_cs = sa.dst_col_stride;
                    }
                    a.q_rot = buf_at(p, t.q, t.q_off);
                    a.q_src = buf_at(p, qr.src, qr.src_off), a.q_cs = buf_at(p, qr.cos_sin, qr.cs_off);
                    a.k_src = buf_at(p, kr.src, kr.src_off), a.k_cs = buf_at(p, kr.cos_sin, kr.cs_off);
                    a.v_src = buf_at(p, vs.src, vs.src_offset);
                    a.k_rot = buf_at(p, kr.dst, kr.dst_off);
                    a.k_cache = p->bufs[ks.cache], a.v_cache = p->bufs[vs.cache];
                    a.dyn_k_off = p->dyn_dev + m.sk, a.dyn_v_off = p->dyn_dev + m.sv; // column indices
                    a.owner = first_head ? 1 : 0;
                    a.max_kv = t.n_cols;
                    a.kvq_block = 32, a.kvq_cols = t.n_cols;
                    static const bool want_trace_q = getenv("ZGML_HIP_ATTN_TRACE") && atoi(getenv("ZGML_HIP_ATTN_TRACE"));
                    if (want_trace_q && first_head) { // (diagnostics build: stamps of the int8-KV launches too)
                        unsigned long long* tq = nullptr;
                        if (hipHostMalloc((void**)&tq, 8 * sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess) {
                            memset(tq, 0, 8 * sizeof(unsigned long long));
                            a.trace = tq;
                            p->attn_traces.push_back(tq);
                        }
                    }
                    first_head = false;
                    adec_by_dh[t.d_head | 0x10000u].push_back(a);
                }
                adec_lo = std::min(adec_lo, m.members.front()), adec_hi = std::max(adec_hi, m.members.back());
                adec_ops += (uint32_t)m.members.size();
                continue;
            }
            if (!macros[mi].heads.empty()) {
                const Macro& m = macros[mi];
                const auto& kr = ops[m.rk].u.rope;
                const auto& ks = ops[m.sk].u.slice_assign;
                const auto& vs = ops[m.sv].u.slice_assign;
                bool first_head = true;
                for (const Macro::Head& h : m.heads) {
                    const auto& t = ops[h.att].u.attention;
                    const auto& qr = ops[h.rq].u.rope;
                    AttnDecodeParams a{};
                    a.att = make_attention(p, t, h.att);
                    if (!a.att.mask) a.att.mask = p->zero_word, a.att.mask_rs = 0, a.att.mask_cs = 0;
                    if (h.row_store >= 0) {
                        const auto& sa = ops[h.row_store].u.slice_assign;
                        a.att.dst2 = p->bufs[sa.dst];
                        a.att.dyn_dst2_off = p->dyn_dev + h.row_store;
                        a.att.d2_rs = sa.dst_row_stride, a.att.d2_cs = sa.dst_col_stride;
                    }
                    a.q_rot = buf_at(p, t.q, t.q_off);
                    a.q_src = buf_at(p, qr.src, qr.src_off);
                    a.q_cs = buf_at(p, qr.cos_sin, qr.cs_off);
                    a.k_src = buf_at(p, kr.src, kr.src_off);
                    a.k_cs = buf_at(p, kr.cos_sin, kr.cs_off);
                    a.v_src = buf_at(p, vs.src, vs.src_offset);
                    a.k_rot = buf_at(p, kr.dst, kr.dst_off);
                    a.k_cache = p->bufs[ks.dst], a.v_cache = p->bufs[vs.dst];
                    a.dyn_k_off = p->dyn_dev + m.sk, a.dyn_v_off = p->dyn_dev + m.sv;
                    a.k_off = t.k_off, a.v_off = t.v_off;
                    a.owner = first_head ? 1 : 0;
                    a.max_kv = p->sched.bounds[h.att].max_seq_kv ? p->sched.bounds[h.att].max_seq_kv : t.seq_kv;
                    static const bool want_trace = getenv("ZGML_HIP_ATTN_TRACE") && atoi(getenv("ZGML_HIP_ATTN_TRACE"));
                    if (want_trace && first_head) { // one record per launch (the group's owner head)
                        unsigned long long* t = nullptr;
                        if (hipHostMalloc((void**)&t, 8 * sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess) {
                            memset(t, 0, 8 * sizeof(unsigned long long));
                            a.trace = t;
                            p->attn_traces.push_back(t);
                        }
                    }
                    first_head = false;
                    adec_by_dh[t.d_head].push_back(a);
                }
                adec_lo = std::min(adec_lo, m.members.front()), adec_hi = std::max(adec_hi, m.members.back());
                adec_ops += (uint32_t)m.members.size();
                continue;
            }
            if (macros[mi].chain) {
                const Macro& m = macros[mi];
                const auto& rn = ops[m.anchor].u.rmsnorm;
                RowChainParams rc;
                rc.src = buf_at(p, rn.src, rn.src_offset), rc.norm_dst = buf_at(p, rn.dst, rn.dst_offset), rc.cols = rn.cols, rc.eps = rn.eps;
                if (m.chain_add >= 0) {
                    const auto& e = ops[m.chain_add].u.elementwise;
                    rc.a0 = buf_at(p, e.src0, e.src0_offset), rc.a1 = buf_at(p, e.src1, e.src1_offset), rc.add_dst = buf_at(p, e.dst, e.dst_offset);
                }
                if (m.chain_mul >= 0) {
                    const auto& e = ops[m.chain_mul].u.elementwise;
                    const bool norm_is_s0 = e.src0 == rn.dst && e.src0_offset == rn.dst_offset;
                    rc.mul_other = norm_is_s0 ? buf_at(p, e.src1, e.src1_offset) : buf_at(p, e.src0, e.src0_offset);
                    rc.mul_dst = buf_at(p, e.dst, e.dst_offset);
                }
                const uint32_t rows = rn.rows;
                auto rcp = std::make_shared<RowChainParams>(rc); // shared: a later matmul may arm its A-piece output
                Launch L{ZGML_DOP_RMSNORM, (uint32_t)m.members.size(), m.members.front(), m.members.back(),
                         [=](hipStream_t s) { launch_row_chain(s, *rcp, rows); }};
                L.hook = std::make_shared<SplitHook>(SplitHook{rcp->mul_dst ? rcp->mul_dst : rcp->norm_dst, rows, rcp->cols, 0, &rcp->ap, &rcp->ap_S, nullptr});
                L.rc_desc = rcp, L.rc_rows = rows;
                chains.push_back(std::move(L));
                continue;
            }
            if (macros[mi].elt_chain) {
                const Macro& m = macros[mi];
                auto ecp = std::make_shared<EltChainParams>(m.elt); // shared: a later matmul may arm its A-piece output
                Launch L{ZGML_DOP_FUSED_ELEMENTWISE, (uint32_t)m.members.size(), m.members.front(), m.members.back(),
                         [=](hipStream_t s) { launch_eltwise_chain(s, *ecp); }};
                if (ecp->n_steps && ecp->steps[ecp->n_steps - 1].store)
                    L.hook = std::make_shared<SplitHook>(SplitHook{ecp->steps[ecp->n_steps - 1].store, 0, 0, ecp->n, &ecp->ap, &ecp->ap_S, &ecp->ap_cols});
                chains.push_back(std::move(L));
                continue;
            }
            if (macros[mi].qmv)
                qmvs.push_back(&macros[mi]);
            else if (macros[mi].store >= 0)
                plain_ops.push_back({macros[mi].anchor, macros[mi].store});
            else
                plain_ops.push_back({macros[mi].members[0], -1});
        }
        emit_batches(p, plain_ops);
        for (Launch& c : chains) p->plan.push_back(std::move(c));
        for (auto& kv : adec_by_dh) {
            const AttnDecodeParams* d = upload_params(p, kv.second);
            const uint32_t nh = (uint32_t)kv.second.size(), dh = kv.first & 0xFFFFu;
            const bool kvq = (kv.first & 0x10000u) != 0; // quantised-KV heads launch on their own
            uint32_t max_kv = 0;
            for (const auto& a : kv.second) max_kv = std::max(max_kv, a.max_kv);
            const AttnSplit sp = attn_split_for(p, nh, dh, max_kv, dh == 128 && !kvq ? 64 : 128);
            Launch AL{ZGML_DOP_ATTENTION, adec_ops, adec_lo, adec_hi, [=](hipStream_t s) { launch_attention_decode_batch(s, d, nh, dh, sp, kvq); }};
            AL.adec_desc = std::make_shared<AdecDesc>(AdecDesc{kv.second, d, nh, dh, sp, kvq});
            p->plan.push_back(std::move(AL));
            adec_ops = 0; // profile accounting: ops counted once
        }
        // group mat-vecs that stage the same vector
        std::vector<char> used(qmvs.size(), 0);
        for (size_t i = 0; i < qmvs.size(); i++) {
            if (used[i]) continue;
            QmvLaunch L;
            std::vector<const Macro*> grp;
            const Macro* first = qmvs[i];
            const QWeightDev& w0 = p->qweights[ops[first->anchor].u.qmatmul.weight_idx];
            for (size_t j = i; j < qmvs.size() && grp.size() < (size_t)kMaxQmvParts; j++) {
                if (used[j]) continue;
                const Macro* c = qmvs[j];
                const QWeightDev& wj = p->qweights[ops[c->anchor].u.qmatmul.weight_idx];
                if (c->pro_sig[0] != first->pro_sig[0] || c->pro_sig[1] != first->pro_sig[1] || c->pro_sig[2] != first->pro_sig[2]) continue;
                if (!qmv_can_group(w0, wj)) continue;
                used[j] = 1;
                grp.push_back(c);
            }
            L.n_parts = (uint32_t)grp.size();
            L.K = w0.K;
            L.pro = first->pro;
            L.pro.store_mid = nullptr, L.pro.store_x = nullptr;
            uint32_t lo = UINT32_MAX, hi = 0, n_ops = 0;
            for (size_t k = 0; k < grp.size(); k++) {
                const Macro* c = grp[k];
                const auto& q = ops[c->anchor].u.qmatmul;
                L.parts[k].w = p->qweights[q.weight_idx];
                L.parts[k].dst = buf_at(p, q.dst, q.dst_offset);
                L.parts[k].n_epi = c->n_epi;
                for (uint32_t e = 0; e < c->n_epi; e++) L.parts[k].epi[e] = c->epi[e];
                if (c->owns_prologue) L.pro.store_mid = c->pro.store_mid, L.pro.store_x = c->pro.store_x;
                lo = std::min(lo, c->members.front());
                hi = std::max(hi, c->members.back());
                n_ops += (uint32_t)c->members.size();
            }
            static const bool want_qmv_trace = getenv("ZGML_HIP_QMV_TRACE") && atoi(getenv("ZGML_HIP_QMV_TRACE"));
            if (want_qmv_trace) {
                unsigned long long* t = nullptr;
                if (hipHostMalloc((void**)&t, 16 * sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess) {
                    memset(t, 0, 16 * sizeof(unsigned long long));
                    L.trace = t;
                    p->qmv_traces.push_back({t, L.n_parts, L.pro.kind, (uint32_t)w0.K, (uint32_t)w0.N}); // (kind as planned: arm_prenorm / arm_pair may still rewrite it)
                }
            }
            auto desc = std::make_shared<QmvLaunch>(L); // shared with the launch: arm_prenorm may still rewrite it
            Launch QL{ZGML_DOP_QMATMUL, n_ops, lo, hi, [desc](hipStream_t s) { launch_qmatvec_fused(s, *desc); }};
            QL.qmv_desc = desc;
            p->plan.push_back(std::move(QL));
        }
    }
    fuse_ksplit(p);
    arm_prenorm(p);
    arm_pair(p);
    fuse_qkv_attention(p);
#ifdef ZGML_TRACE
    fuse_attention_o(p);
#endif
}

void build_plan(zgml_hip_program* p) {
    { // a rebuild synchronises the stream and may launch (hoisted repeats): never inside a caller's stream capture (ADVICE r03 — a
      // first upload to a hoist-guarded buffer, or a fusion time-out, can mark the plan dirty between two enqueue calls)
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(p->ctx->stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
            p->ctx->fail("build_plan: the plan has to be rebuilt, but the context's stream is being captured — run one step outside the capture first");
            return;
        }
    }
    hipStreamSynchronize(p->ctx->stream); // the previous plan's parameter arrays may still be in use
    p->plan.clear();
    for (size_t i = 0; i < p->ops.size(); i++) { // (refresh_program refuses this already; a launch must never be skipped silently)
        const zgml_device_op& op = p->ops[i];
        if (op.kind == ZGML_DOP_QMATMUL && op.u.qmatmul.M != 1 && op.u.qmatmul.weight_idx < p->qweights.size() &&
            p->qweights[op.u.qmatmul.weight_idx].format == QW_Q4K)
            p->ctx->fail("build_plan: op " + std::to_string(i) + " is an M > 1 qmatmul over a weight packed for M = 1 mat-vecs (K-on-lanes layout)");
    }
    free_param_blobs(p);
    for (void* d : p->fuse_owned) hipFree(d); // counters / seen / idx of the previous plan's fused launches
    p->fuse_owned.clear();
    p->fuse_epoch = p->ctx->fuse_epoch;
    p->split_buf = nullptr, p->split_cnt = nullptr, p->split_buf_floats = 0, p->split_cnt_words = 0; // lived in the blobs
    p->split_pos = UINT64_MAX - 1, p->split_input = nullptr, p->qmm_group = nullptr, p->f16_group = nullptr;
    p->plan_batched = p->ctx->opt_fusion && p->batching_safe;
    if (p->plan_batched) {
        p->sched = build_schedule(p->ops, p->sizes, p->barriers, &p->seq_kv_bound);
        if (!dynamic_fields_in_bounds(p->sched, p->ops)) p->plan_batched = false;
    }
    if (p->plan_batched) {
        build_fused_plan(p);
    } else {
        p->hoist_op.clear(); // op by op: every repeat runs
        for (uint32_t i = 0; i < p->ops.size(); i++) emit_batches(p, {PlanItem{i, -1}});
    }
    p->plan_dirty = false;
    if (getenv("ZGML_HIP_DEBUG_PLAN")) {
        uint64_t by_kind[ZGML_DOP_COUNT] = {0}, ops_by_kind[ZGML_DOP_COUNT] = {0};
        for (const auto& L : p->plan)
            if (L.kind < ZGML_DOP_COUNT) by_kind[L.kind]++, ops_by_kind[L.kind] += L.n_ops;
        fprintf(stderr, "[zgml_hip] plan: %zu launches for %zu ops (batched=%d):", p->plan.size(), p->ops.size(), (int)p->plan_batched);
        for (int k = 0; k < ZGML_DOP_COUNT; k++)
            if (by_kind[k]) fprintf(stderr, " kind%d=%llu(%llu ops)", k, (unsigned long long)by_kind[k], (unsigned long long)ops_by_kind[k]);
        fprintf(stderr, "\n");
    }
}

void set_dyn_from_ops(zgml_hip_program* p) {
    for (size_t i = 0; i < p->ops.size(); i++) {
        uint32_t v = 0;
        if (p->ops[i].kind == ZGML_DOP_SLICE_ASSIGN) v = p->ops[i].u.slice_assign.dst_offset;
        if (p->ops[i].kind == ZGML_DOP_ATTENTION) v = p->ops[i].u.attention.seq_kv;
        if (p->ops[i].kind == ZGML_DOP_KVQ_STORE) v = p->ops[i].u.kvq_store.col;
        if (p->ops[i].kind == ZGML_DOP_ATTENTION_KVQ) v = p->ops[i].u.attention_kvq.seq_kv;
        if (p->dyn_host[i] != v) {
            p->dyn_host[i] = v;
            p->dyn_dirty = true;
        }
    }
}

// copy ops (and their fused steps) into program-owned storage
void note_seq_kv_bounds(zgml_hip_program* p) {
    p->seq_kv_bound.resize(p->ops.size(), 0);
    for (size_t i = 0; i < p->ops.size(); i++) {
        uint32_t v = 0;
        if (p->ops[i].kind == ZGML_DOP_ATTENTION) v = p->ops[i].u.attention.seq_kv;
        if (p->ops[i].kind == ZGML_DOP_ATTENTION_KVQ) v = p->ops[i].u.attention_kvq.seq_kv;
        p->seq_kv_bound[i] = std::max(p->seq_kv_bound[i], v);
    }
}

void own_ops(zgml_hip_program* p, const zgml_device_op* ops, uint64_t n_ops) {
    // a static refresh may change an op's kind: bounds of ops that are no longer the same attention start over
    if (p->seq_kv_bound.size() == n_ops && p->ops.size() == n_ops)
        for (uint64_t i = 0; i < n_ops; i++)
            if (p->ops[i].kind != ops[i].kind) p->seq_kv_bound[i] = 0;
    p->ops.assign(ops, ops + n_ops);
    p->dyn_ops.clear(); // the ops with a dynamic field (what a per-token refresh touches: zgml_hip_refresh_dynamic)
    for (uint64_t i = 0; i < n_ops; i++)
        if ((ops[i].kind == ZGML_DOP_SLICE_ASSIGN && ops[i].u.slice_assign.patch_stride != 0) || ops[i].kind == ZGML_DOP_ATTENTION ||
            (ops[i].kind == ZGML_DOP_KVQ_STORE && ops[i].u.kvq_store.patch_stride != 0) || ops[i].kind == ZGML_DOP_ATTENTION_KVQ)
            p->dyn_ops.push_back((uint32_t)i);
    p->steps.assign(n_ops, {});
    for (uint64_t i = 0; i < n_ops; i++) {
        if (ops[i].kind == ZGML_DOP_FUSED_ELEMENTWISE) {
            const auto& fe = ops[i].u.fused_elementwise;
            p->steps[i].assign(fe.steps, fe.steps + fe.n_steps);
            p->ops[i].u.fused_elementwise.steps = p->steps[i].data();
        }
    }
    note_seq_kv_bounds(p);
}

// true when the static part of two ops is identical (dynamic fields and step pointers ignored). No copies: the bytes of the
// ACTIVE union member (the rest may be uninitialised padding) are compared around the member's dynamic field — this runs once per
// op per token on the drop-in path (refresh_program is called before every execute: src/device_inference.zig:260-263).
bool same_static(const zgml_device_op& a, const zgml_device_op& b) {
    if (a.kind != b.kind) return false;
    auto except = [](const void* pa, const void* pb, size_t len, size_t off, size_t flen) { // equal but for [off, off + flen)
        const char *x = (const char*)pa, *y = (const char*)pb;
        return memcmp(x, y, off) == 0 && memcmp(x + off + flen, y + off + flen, len - off - flen) == 0;
    };
    switch (a.kind) {
        case ZGML_DOP_SLICE_ASSIGN:
            return except(&a.u.slice_assign, &b.u.slice_assign, sizeof(a.u.slice_assign), offsetof(zgml_op_slice_assign, dst_offset), sizeof(uint32_t));
        case ZGML_DOP_ATTENTION: return except(&a.u.attention, &b.u.attention, sizeof(a.u.attention), offsetof(zgml_op_attention, seq_kv), sizeof(uint32_t));
        case ZGML_DOP_KVQ_STORE: return except(&a.u.kvq_store, &b.u.kvq_store, sizeof(a.u.kvq_store), offsetof(zgml_op_kvq_store, col), sizeof(uint32_t));
        case ZGML_DOP_ATTENTION_KVQ:
            return except(&a.u.attention_kvq, &b.u.attention_kvq, sizeof(a.u.attention_kvq), offsetof(zgml_op_attention_kvq, seq_kv), sizeof(uint32_t));
        case ZGML_DOP_FUSED_ELEMENTWISE: {
            const auto &fa = a.u.fused_elementwise, &fb = b.u.fused_elementwise;
            if (fa.n_steps != fb.n_steps) return false;
            for (uint32_t s = 0; s < fa.n_steps; s++)
                if (fa.steps[s].op != fb.steps[s].op || fa.steps[s].is_swapped != fb.steps[s].is_swapped ||
                    fa.steps[s].secondary_buf != fb.steps[s].secondary_buf || fa.steps[s].secondary_offset != fb.steps[s].secondary_offset)
                    return false;
            return except(&fa, &fb, sizeof(fa), offsetof(zgml_op_fused_elementwise, steps), sizeof(fa.steps));
        }
        case ZGML_DOP_ELEMENTWISE: return memcmp(&a.u.elementwise, &b.u.elementwise, sizeof(a.u.elementwise)) == 0;
        case ZGML_DOP_MATMUL: return memcmp(&a.u.matmul, &b.u.matmul, sizeof(a.u.matmul)) == 0;
        case ZGML_DOP_QMATMUL: return memcmp(&a.u.qmatmul, &b.u.qmatmul, sizeof(a.u.qmatmul)) == 0;
        case ZGML_DOP_SOFTMAX: return memcmp(&a.u.softmax, &b.u.softmax, sizeof(a.u.softmax)) == 0;
        case ZGML_DOP_LAYERNORM: return memcmp(&a.u.layernorm, &b.u.layernorm, sizeof(a.u.layernorm)) == 0;
        case ZGML_DOP_RMSNORM: return memcmp(&a.u.rmsnorm, &b.u.rmsnorm, sizeof(a.u.rmsnorm)) == 0;
        case ZGML_DOP_REDUCE: return memcmp(&a.u.reduce, &b.u.reduce, sizeof(a.u.reduce)) == 0;
        case ZGML_DOP_REPEAT: return memcmp(&a.u.repeat, &b.u.repeat, sizeof(a.u.repeat)) == 0;
        case ZGML_DOP_ROPE: return memcmp(&a.u.rope, &b.u.rope, sizeof(a.u.rope)) == 0;
        default: return false;
    }
}

void run_plan(zgml_hip_program* p, hipStream_t s, size_t first, size_t count) {
#ifdef ZGML_TRACE
    // diagnostics build only: ZGML_HIP_SKIP_KINDS=<bitmask of DeviceOp tags> drops those launches (timing ablation
    // only; results are garbage)
    static const unsigned skip = getenv("ZGML_HIP_SKIP_KINDS") ? (unsigned)strtoul(getenv("ZGML_HIP_SKIP_KINDS"), nullptr, 0) : 0u;
    // ZGML_HIP_SKIP_MOD="<period>:<bitmask>" drops launch i >= 1 when bit ((i-1) % period) is set
    static unsigned mod_period = 0, mod_mask = 0;
    static bool mod_init = false;
    if (!mod_init) {
        mod_init = true;
        if (const char* e = getenv("ZGML_HIP_SKIP_MOD")) sscanf(e, "%u:%x", &mod_period, &mod_mask);
    }
    for (size_t i = first; i < first + count && i < p->plan.size(); i++) {
        if (skip & (1u << p->plan[i].kind)) continue;
        if (mod_period && i >= 1 && (mod_mask & (1u << ((i - 1) % mod_period)))) continue;
        p->plan[i].run(s);
    }
#else
    for (size_t i = first; i < first + count && i < p->plan.size(); i++) p->plan[i].run(s);
#endif
}

void flush_dyn(zgml_hip_program* p) {
    if (!p->dyn_dirty || p->ops.empty()) return;
    hipMemcpyAsync(p->dyn_dev, p->dyn_host, p->ops.size() * sizeof(uint32_t), hipMemcpyHostToDevice, p->ctx->stream);
    p->dyn_dirty = false;
}

// enqueue the whole program on the context stream (graph replay when enabled)
void enqueue(zgml_hip_program* p) {
    zgml_hip_ctx* ctx = p->ctx;
    if (p->plan_dirty || p->fuse_epoch != p->ctx->fuse_epoch) {
        free_graph(p);
        build_plan(p);
    }
    flush_dyn(p);
    if (ctx->opt_profile) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (auto& L : p->plan) {
            hipEventRecord(e0, ctx->stream);
            L.run(ctx->stream);
            hipEventRecord(e1, ctx->stream);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (L.kind < ZGML_DOP_COUNT) p->profile.time_ns[L.kind] += (uint64_t)(ms * 1e6);
            L.prof_ns += (uint64_t)(ms * 1e6);
            L.prof_calls++;
        }
        hipEventDestroy(e0);
        hipEventDestroy(e1);
        if (getenv("ZGML_HIP_DEBUG_PLAN") && atoi(getenv("ZGML_HIP_DEBUG_PLAN")) >= 2 && p->plan[0].prof_calls == 8) {
            // per-launch table after 8 profiled executions (first 40 launches: one layer and a bit)
            for (size_t i = 0; i < p->plan.size() && i < 40; i++)
                fprintf(stderr, "[zgml_hip] launch %3zu kind %2u ops %4u [%u..%u]  %.2f us\n", i, p->plan[i].kind, p->plan[i].n_ops,
                        p->plan[i].op_lo, p->plan[i].op_hi, p->plan[i].prof_ns / 1e3 / p->plan[i].prof_calls);
        }
        return;
    }
    if (ctx->opt_graph && !p->plan.empty()) {
        if (!p->graph_exec) {
            // two graphs (zgml_hip_program::graph_tail): the head holds the first sixth of the launches (at least 8: its device
            // time has to cover the host's submission of the tail), short plans stay one graph
            static const int split_env = getenv("ZGML_HIP_GRAPH_SPLIT") ? atoi(getenv("ZGML_HIP_GRAPH_SPLIT")) : -1;
            size_t head = p->plan.size() >= 48 ? std::max<size_t>(8, p->plan.size() / 6) : p->plan.size();
            if (split_env == 0) head = p->plan.size();
            if (split_env > 0) head = std::min<size_t>((size_t)split_env, p->plan.size());
            auto capture = [&](size_t first, size_t count, hipGraph_t* g_out, hipGraphExec_t* e_out, const char* tag) {
                hipGraph_t g = nullptr;
                if (!CTX_CHECK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal))) return false;
                run_plan(p, ctx->stream, first, count);
                if (!CTX_CHECK(ctx, hipStreamEndCapture(ctx->stream, &g)) || !g) return false;
                dump_graph(g, tag);
                hipGraphExec_t ge = nullptr;
                if (!CTX_CHECK(ctx, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0))) {
                    hipGraphDestroy(g);
                    return false;
                }
                *g_out = g, *e_out = ge;
                return true;
            };
            if (capture(0, head, &p->graph, &p->graph_exec, "program") && head < p->plan.size() &&
                !capture(head, p->plan.size() - head, &p->graph_tail, &p->graph_tail_exec, "program-tail")) {
                hipGraphExecDestroy(p->graph_exec), hipGraphDestroy(p->graph); // (all or nothing: eager below)
                p->graph_exec = nullptr, p->graph = nullptr;
            }
        }
        if (p->graph_exec) {
            CTX_CHECK(ctx, hipGraphLaunch(p->graph_exec, ctx->stream));
            if (p->graph_tail_exec) CTX_CHECK(ctx, hipGraphLaunch(p->graph_tail_exec, ctx->stream));
            return;
        }
    }
    run_plan(p, ctx->stream, 0, p->plan.size());
}

uint64_t now_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
               std::chrono::steady_clock::now().time_since_epoch())
        .count();
}

bool grow(zgml_hip_ctx* ctx, float** ptr, uint64_t* cap, uint64_t elems) {
    if (elems <= *cap) return true;
    if (*ptr) hipFree(*ptr);
    *ptr = nullptr;
    *cap = 0;
    if (!CTX_CHECK(ctx, hipMalloc((void**)ptr, elems * sizeof(float)))) return false;
    *cap = elems;
    return true;
}

// ── device-resident LLaMA decode (extension; see include/zgml_hip.h) ─────────────────────────
} // namespace
struct zgml_resident {
    float *embed = nullptr, *cos = nullptr, *sin = nullptr;
    uint32_t vocab = 0, d = 0, max_seq = 0, dh = 0, n_rope = 0;
    float *tok_in = nullptr, *mask = nullptr, *logits = nullptr;
    float** rope_bufs = nullptr;   // device array of n_rope pointers
    uint32_t* dyn_kind = nullptr;  // per op: 0 static, 1 slice_assign(base,stride), 2 attention
    uint32_t* dyn_base = nullptr;
    uint32_t* dyn_stride = nullptr;
    uint32_t* state = nullptr;     // [0] token, [1] pos, [2] produced count
    uint32_t token_len = 1;        // T of the plan (token_input holds T rows): 1 = decode, > 1 = prefill chunk
    uint32_t* tok_dev = nullptr;   // prefill: the chunk's T token ids (the only per-chunk host -> device traffic)
    int64_t* tokens = nullptr;     // produced tokens (device)
    uint32_t tokens_cap = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph_multi = nullptr; // `multi_n` consecutive tokens as one graph (ZGML_HIP_RESIDENT_TOKENS_PER_GRAPH)
    hipGraphExec_t graph_multi_exec = nullptr;
    uint32_t multi_n = 0;
};
using Resident = zgml_resident;
namespace {

void free_resident_graph(zgml_hip_program* p) {
    Resident* r = p->resident;
    if (!r) return;
    if (r->graph_exec) hipGraphExecDestroy(r->graph_exec);
    if (r->graph) hipGraphDestroy(r->graph);
    r->graph_exec = nullptr, r->graph = nullptr;
    if (r->graph_multi_exec) hipGraphExecDestroy(r->graph_multi_exec);
    if (r->graph_multi) hipGraphDestroy(r->graph_multi);
    r->graph_multi_exec = nullptr, r->graph_multi = nullptr, r->multi_n = 0;
}

void free_resident(zgml_hip_program* p) {
    Resident* r = p->resident;
    if (!r) return;
    free_resident_graph(p);
    hipFree(r->embed);
    hipFree(r->cos);
    hipFree(r->sin);
    hipFree(r->rope_bufs);
    hipFree(r->dyn_kind);
    hipFree(r->dyn_base);
    hipFree(r->dyn_stride);
    hipFree(r->state);
    hipFree(r->tok_dev);
    hipFree(r->tokens);
    delete r;
    p->resident = nullptr;
}

} // namespace

// ════════════════════════════════ C ABI ════════════════════════════════

extern "C" {

zgml_hip_ctx* zgml_hip_create(int device_ordinal) {
    g_create_error.clear();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        g_create_error = std::string("no HIP device: ") + (e == hipSuccess ? "count is 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device_ordinal < 0 || device_ordinal >= n) {
        g_create_error = "device ordinal out of range";
        return nullptr;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) {
        g_create_error = "hipGetDeviceProperties failed";
        return nullptr;
    }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        g_create_error = std::string("device is ") + prop.gcnArchName + ", this backend is built for gfx950 only";
        return nullptr;
    }
    if (hipSetDevice(device_ordinal) != hipSuccess) {
        g_create_error = "hipSetDevice failed";
        return nullptr;
    }
    zgml_hip_ctx* ctx = new zgml_hip_ctx();
    ctx->device = device_ordinal;
    // environment overrides of the option defaults (profilers: ZGML_HIP_GRAPH=0 traces eager launches)
    if (const char* e = getenv("ZGML_HIP_GRAPH")) ctx->opt_graph = atoi(e) != 0;
    if (const char* e = getenv("ZGML_HIP_FUSION")) ctx->opt_fusion = atoi(e) != 0;
    if (const char* e = getenv("ZGML_HIP_KSPLIT")) ctx->opt_ksplit = atoi(e) != 0;
    if (const char* e = getenv("ZGML_HIP_W8A8")) ctx->opt_w8a8 = atoi(e) != 0;
    if (const char* e = getenv("ZGML_HIP_HOST_PROF")) ctx->host_prof = atoi(e) != 0;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        g_create_error = "hipStreamCreate failed";
        delete ctx;
        return nullptr;
    }
    hipMalloc((void**)&ctx->arg_val, 256 * sizeof(float));
    hipMalloc((void**)&ctx->arg_idx, 256 * sizeof(int64_t));
    hipMalloc((void**)&ctx->arg_cnt, 128);
    hipMemset(ctx->arg_cnt, 0, 128);
    hipMalloc((void**)&ctx->arg_out, sizeof(int64_t));
    hipHostMalloc((void**)&ctx->arg_out_host, sizeof(int64_t), hipHostMallocDefault);
    ctx->n_cu = prop.multiProcessorCount;
    // the hand-off flag of the fused launches: pinned + mapped; without it no fusion is built (fuse_qkv_attention)
    if (hipHostMalloc((void**)&ctx->handoff_flag, 64, hipHostMallocMapped) == hipSuccess) {
        memset(ctx->handoff_flag, 0, 64);
        if (hipHostGetDevicePointer((void**)&ctx->handoff_flag_dev, ctx->handoff_flag, 0) != hipSuccess) ctx->handoff_flag_dev = nullptr;
    }
    return ctx;
}

void zgml_hip_shard_destroy(zgml_hip_ctx* ctx);
void zgml_hip_destroy(zgml_hip_ctx* ctx) {
    if (ctx && ctx->host_prof && ctx->prof_calls[1]) {
        static const char* const names[4] = {"refresh", "upload (pack + H2D + scatter)", "enqueue (dynamic words + graph launch)", "download (gather + D2H + wait + unpack)"};
        fprintf(stderr, "[zgml_hip] host time per call of the vtable path's phases:\n");
        for (int k = 0; k < 4; k++)
            if (ctx->prof_calls[k]) fprintf(stderr, "  %-44s %8.2f us x %llu\n", names[k], ctx->prof_ns[k] / 1e3 / ctx->prof_calls[k], (unsigned long long)ctx->prof_calls[k]);
    }
    if (!ctx) return;
    zgml_hip_shard_destroy(ctx);
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    ctx->drop_b_cache();
    hipFree(ctx->mm_a);
    hipFree(ctx->mm_b);
    hipFree(ctx->mm_c);
    hipFree(ctx->arg_val);
    hipFree(ctx->arg_idx);
    hipFree(ctx->arg_cnt);
    hipFree(ctx->arg_out);
    hipHostFree(ctx->arg_out_host);
    if (ctx->handoff_flag) hipHostFree(ctx->handoff_flag);
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* zgml_hip_last_error(const zgml_hip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void zgml_hip_clear_error(zgml_hip_ctx* ctx) {
    if (ctx) ctx->err.clear();
}

void zgml_hip_capabilities(zgml_capabilities* out) {
    if (out) fill_caps(out);
}

int zgml_hip_program_supported(const zgml_device_program* program) { return program_supported(program) ? 1 : 0; }

int zgml_hip_set_option(zgml_hip_ctx* ctx, int option, int64_t value) {
    if (!ctx) return -1;
    switch (option) {
        case ZGML_HIP_OPT_FUSION: ctx->opt_fusion = value != 0; return 0;
        case ZGML_HIP_OPT_GRAPH: ctx->opt_graph = value != 0; return 0;
        case ZGML_HIP_OPT_PROFILE: ctx->opt_profile = value != 0; return 0;
        case ZGML_HIP_OPT_SKIP_DEAD_UPLOADS: ctx->opt_skip_dead = value != 0; return 0;
        case ZGML_HIP_OPT_F16_DENSE_WEIGHTS: ctx->opt_f16_dense = value != 0; return 0;
        case ZGML_HIP_OPT_ATTN_SPLIT_MIN_KEYS: ctx->opt_attn_split_min_keys = value < 0 ? -1 : value; return 0;
        case ZGML_HIP_OPT_FUSE_RESIDENT_WGS:
            ctx->opt_fuse_resident_wgs = value < 0 ? -1 : value;
            ctx->fuse_epoch++; // existing programs rebuild their plans under the new capacity
            return 0;
        case ZGML_HIP_OPT_KSPLIT: ctx->opt_ksplit = value != 0; return 0; // (latched per program at compile_program)
        case ZGML_HIP_OPT_W8A8: ctx->opt_w8a8 = value != 0; return 0;     // (decides the weights' device format at compile_program)
        case ZGML_HIP_OPT_DENSE_WEIGHT_CACHE:
            ctx->b_cache_cap = value > 0 ? (uint64_t)value : 0;
            if (!ctx->b_cache_cap) ctx->drop_b_cache();
            return 0;
        default: return -1;
    }
}

int zgml_hip_dense_matmul_f32(zgml_hip_ctx* ctx, float* dst, uint64_t dst_len, const float* a, uint64_t a_len,
                              const float* b, uint64_t b_len, const zgml_matmul_geom* g) {
    if (!ctx || !g || !dst || !a || !b) return 0;
    if (g->M == 0 || g->N == 0) return 1;
    // spans actually touched (the caller's slices may be larger)
    const uint64_t a_span = g->a_offset + (g->M - 1) * g->a_row_stride + (g->K ? (g->K - 1) * g->a_col_stride : 0) + 1;
    const uint64_t b_span = g->b_offset + (g->K ? (g->K - 1) * g->b_row_stride : 0) + (g->N - 1) * g->b_col_stride + 1;
    const uint64_t c_span = g->dst_offset + (g->M - 1) * g->dst_row_stride + g->N;
    if (a_span > a_len || b_span > b_len || c_span > dst_len) return 0; // caller falls back
    if (g->M > UINT32_MAX || g->N > UINT32_MAX || g->K > UINT32_MAX) return 0;
    hipSetDevice(ctx->device);
    if (!grow(ctx, &ctx->mm_a, &ctx->mm_a_cap, a_span) || !grow(ctx, &ctx->mm_b, &ctx->mm_b_cap, b_span) ||
        !grow(ctx, &ctx->mm_c, &ctx->mm_c_cap, c_span))
        return 0;
    hipStream_t s = ctx->stream;
    if (!CTX_CHECK(ctx, hipMemcpyAsync(ctx->mm_a, a, a_span * 4, hipMemcpyHostToDevice, s))) return 0;
    const float* b_dev = ctx->mm_b;
    bool b_cached = false;
    if (ctx->b_cache_cap && b_span * 4 <= ctx->b_cache_cap) { // weight cache: B stays on the device across calls
        auto it = ctx->b_cache.find(b);
        if (it != ctx->b_cache.end() && it->second.span >= b_span) {
            b_dev = it->second.dev, b_cached = true;
            ctx->b_cache_hits++;
        } else {
            if (it != ctx->b_cache.end()) { // same pointer, larger span now: replace
                ctx->b_cache_bytes -= it->second.span * 4;
                hipFree(it->second.dev);
                ctx->b_cache.erase(it);
            }
            if (ctx->b_cache_bytes + b_span * 4 > ctx->b_cache_cap) ctx->drop_b_cache(); // simplest policy: start over
            float* d = nullptr;
            if (hipMalloc((void**)&d, b_span * 4) == hipSuccess) {
                if (!CTX_CHECK(ctx, hipMemcpyAsync(d, b, b_span * 4, hipMemcpyHostToDevice, s))) {
                    hipFree(d);
                    return 0;
                }
                ctx->b_cache[b] = {d, b_span};
                ctx->b_cache_bytes += b_span * 4;
                b_dev = d, b_cached = true;
            }
            ctx->b_cache_misses++;
        }
    }
    if (!b_cached && !CTX_CHECK(ctx, hipMemcpyAsync(ctx->mm_b, b, b_span * 4, hipMemcpyHostToDevice, s))) return 0;
    // rows of dst may be strided: keep the untouched gaps as the caller has them
    if (g->dst_row_stride != g->N || g->dst_offset != 0)
        if (!CTX_CHECK(ctx, hipMemcpyAsync(ctx->mm_c, dst, c_span * 4, hipMemcpyHostToDevice, s))) return 0;
    DenseMatmulParams dp{};
    dp.dst = ctx->mm_c + g->dst_offset;
    dp.a = ctx->mm_a + g->a_offset;
    dp.b = b_dev + g->b_offset;
    dp.M = (uint32_t)g->M, dp.N = (uint32_t)g->N, dp.K = (uint32_t)g->K;
    dp.a_rs = (uint32_t)g->a_row_stride, dp.a_cs = (uint32_t)g->a_col_stride;
    dp.b_rs = (uint32_t)g->b_row_stride, dp.b_cs = (uint32_t)g->b_col_stride;
    dp.dst_rs = (uint32_t)g->dst_row_stride;
    launch_dense_matmul(s, dp);
    if (!CTX_CHECK(ctx, hipMemcpyAsync(dst, ctx->mm_c, c_span * 4, hipMemcpyDeviceToHost, s))) return 0;
    if (!CTX_CHECK(ctx, hipStreamSynchronize(s))) return 0;
    return 1;
}

void zgml_hip_dense_cache_invalidate(zgml_hip_ctx* ctx, const float* b) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (!b) {
        ctx->drop_b_cache();
        return;
    }
    auto it = ctx->b_cache.find(b);
    if (it == ctx->b_cache.end()) return;
    ctx->b_cache_bytes -= it->second.span * 4;
    hipFree(it->second.dev);
    ctx->b_cache.erase(it);
}

void zgml_hip_dense_cache_stats(zgml_hip_ctx* ctx, uint64_t* hits, uint64_t* misses, uint64_t* bytes) {
    if (!ctx) return;
    if (hits) *hits = ctx->b_cache_hits;
    if (misses) *misses = ctx->b_cache_misses;
    if (bytes) *bytes = ctx->b_cache_bytes;
}

zgml_hip_program* zgml_hip_compile_program(zgml_hip_ctx* ctx, const zgml_device_program* prog) {
    if (!ctx || !prog) return nullptr;
    if (!program_supported(prog)) {
        ctx->fail("compile_program: program not supported by Capabilities.hip");
        return nullptr;
    }
    hipSetDevice(ctx->device);
    zgml_hip_program* p = new zgml_hip_program();
    p->ctx = ctx;
    p->ksplit = ctx->opt_ksplit;
    own_ops(p, prog->ops, prog->n_ops);
    const size_t nb = prog->n_buffers;
    p->sizes.resize(nb);
    for (size_t i = 0; i < nb; i++) p->sizes[i] = prog->buffer_sizes[i] ? prog->buffer_sizes[i] : 1;

    // liveness: buffers some op references (F8: the f32 master copy of a quantized weight is not)
    std::vector<char> live(nb, ctx->opt_skip_dead ? 0 : 1);
    std::vector<uint16_t> ids;
    std::vector<char> qw_live(prog->n_qweights, 0);
    for (const auto& op : p->ops) {
        ids.clear();
        op_buffers(op, ids);
        for (uint16_t id : ids) live[id] = 1;
        if (op.kind == ZGML_DOP_QMATMUL) qw_live[op.u.qmatmul.weight_idx] = 1;
    }

    // f16 weight promotion (opt-in; src/backend/wgpu.zig:1071-1104): the B operand of `matmul` ops
    // that has an initial upload is kept only as an MFMA-packed f16 copy. Stricter than the
    // reference, which trusts the first user's geometry: every reader of the buffer must be a
    // matmul using it as B with the same geometry, and no op may write it.
    p->f16_weights.assign(nb, nullptr);
    struct Promo {
        uint64_t K, N, b_off, b_rs, b_cs;
    };
    std::vector<int> promo_state(nb, 0); // 0 unseen, 1 candidate, -1 rejected
    std::vector<Promo> promo(nb);
    if (ctx->opt_f16_dense) {
        std::vector<char> has_upload(nb, 0);
        for (uint64_t i = 0; i < prog->n_initial_uploads; i++)
            if (prog->initial_uploads[i].buf_idx < nb) has_upload[prog->initial_uploads[i].buf_idx] = 1;
        const Schedule acc = build_schedule(p->ops, p->sizes, {});
        for (size_t oi = 0; oi < p->ops.size(); oi++) {
            const auto& op = p->ops[oi];
            for (const Span& sp : acc.access[oi].writes) promo_state[sp.buf] = -1;
            if (op.kind == ZGML_DOP_MATMUL) {
                const auto& m = op.u.matmul;
                const Promo g{m.geom.K, m.geom.N, m.geom.b_offset, m.geom.b_row_stride, m.geom.b_col_stride};
                if (m.a == m.b || m.dst == m.b || !has_upload[m.b] || m.geom.a_col_stride != 1 || !f16_packable(g.K, g.N)) {
                    promo_state[m.b] = -1;
                } else if (promo_state[m.b] == 0) {
                    promo_state[m.b] = 1, promo[m.b] = g;
                } else if (promo_state[m.b] == 1) {
                    const Promo& q = promo[m.b];
                    if (q.K != g.K || q.N != g.N || q.b_off != g.b_off || q.b_rs != g.b_rs || q.b_cs != g.b_cs) promo_state[m.b] = -1;
                }
                promo_state[m.a] = -1;
            } else {
                for (const Span& sp : acc.access[oi].reads) promo_state[sp.buf] = -1;
            }
        }
        for (size_t i = 0; i < nb; i++)
            if (promo_state[i] == 1) live[i] = 0; // no f32 copy on the device
    }

    // one arena for all live buffers, 256-byte aligned slots, zero-initialised like
    // OwnedBufferTable.init (src/backend/reference.zig:81-97)
    uint64_t total = 0;
    std::vector<uint64_t> offs(nb, 0);
    for (size_t i = 0; i < nb; i++) {
        if (!live[i]) continue;
        offs[i] = total;
        total += (p->sizes[i] * sizeof(float) + 255) / 256 * 256;
    }
    bool ok = true;
    if (total) {
        ok = CTX_CHECK(ctx, hipMalloc(&p->arena, total)) && CTX_CHECK(ctx, hipMemsetAsync(p->arena, 0, total, ctx->stream));
    }
    if (ok) ok = CTX_CHECK(ctx, hipMalloc((void**)&p->zero_word, 256)) && CTX_CHECK(ctx, hipMemsetAsync(p->zero_word, 0, 256, ctx->stream));
    p->bufs.assign(nb, nullptr);
    if (ok)
        for (size_t i = 0; i < nb; i++)
            if (live[i]) p->bufs[i] = (float*)((char*)p->arena + offs[i]);

    // initial uploads (skipped for elided buffers)
    for (uint64_t i = 0; ok && i < prog->n_initial_uploads; i++) {
        const zgml_program_io& io = prog->initial_uploads[i];
        if (io.buf_idx >= nb) {
            ctx->fail("initial upload names a buffer that does not exist");
            ok = false;
            break;
        }
        if (!p->bufs[io.buf_idx]) continue;
        if ((uint64_t)io.offset + io.size > p->sizes[io.buf_idx] * sizeof(float)) {
            ctx->fail("initial upload out of range");
            ok = false;
            break;
        }
        ok = CTX_CHECK(ctx, hipMemcpyAsync((char*)p->bufs[io.buf_idx] + io.offset, io.host_ptr, io.size,
                                           hipMemcpyHostToDevice, ctx->stream));
    }

    // promoted weights: stage the f32 image, pack to f16 on the device, drop the f32 image
    uint64_t f16_total = 0;
    for (size_t b = 0; ok && b < nb; b++) {
        if (promo_state[b] != 1) continue;
        const Promo& g = promo[b];
        float* tmp = nullptr;
        void* packed = nullptr;
        const uint64_t bytes = p->sizes[b] * sizeof(float);
        ok = CTX_CHECK(ctx, hipMalloc((void**)&tmp, bytes)) && CTX_CHECK(ctx, hipMemsetAsync(tmp, 0, bytes, ctx->stream)) &&
             CTX_CHECK(ctx, hipMalloc(&packed, f16_packed_bytes(g.K, g.N)));
        if (ok && (g.b_off + (g.K - 1) * g.b_rs + (g.N - 1) * g.b_cs >= p->sizes[b])) {
            ctx->fail("matmul B geometry exceeds its buffer");
            ok = false;
        }
        for (uint64_t i = 0; ok && i < prog->n_initial_uploads; i++) {
            const zgml_program_io& io = prog->initial_uploads[i];
            if (io.buf_idx != b) continue;
            if ((uint64_t)io.offset + io.size > bytes) {
                ctx->fail("initial upload out of range");
                ok = false;
                break;
            }
            ok = CTX_CHECK(ctx, hipMemcpyAsync((char*)tmp + io.offset, io.host_ptr, io.size, hipMemcpyHostToDevice, ctx->stream));
        }
        if (ok) {
            launch_pack_f16(ctx->stream, tmp + g.b_off, (uint32_t)g.b_rs, (uint32_t)g.b_cs, (uint32_t)g.K, (uint32_t)g.N, packed);
            ok = CTX_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        }
        hipFree(tmp);
        if (ok) {
            p->f16_weights[b] = packed;
            p->owned.push_back(packed);
            f16_total += f16_packed_bytes(g.K, g.N);
        } else {
            hipFree(packed);
        }
    }

    p->f16_stream_nt = f16_total >= (192ull << 20);

    // quantized weights: upload raw, classify, then re-pack on the device into TWO arenas (quants, scales)
    // in weight-index order with nothing between consecutive weights: the weights a grouped mat-vec launch
    // reads (q/k/v, gate/up) are then contiguous and the kernel derives every part's pointers from part 0's
    // (preloaded) ones instead of waiting for an argument-block fetch (qmatvec.hip, QMV_HEAD_PARAMS).
    p->qweights.resize(prog->n_qweights);
    uint32_t* flags = nullptr;
    if (ok && prog->n_qweights) ok = CTX_CHECK(ctx, hipMalloc((void**)&flags, 2 * sizeof(uint32_t)));
    struct PendingPack {
        void* raw_a = nullptr; // int8 values, or the GGUF blocks
        float* raw_s = nullptr;
        bool gguf = false;
    };
    std::vector<PendingPack> pending(prog->n_qweights);
    uint64_t qs_total = 0, sc_total = 0;
    // Q4_0-valued weights with f16 scales that only ever feed M = 1 mat-vecs take the K-on-lanes layout (QW_Q4K, qmatvec.hip);
    // a weight an M > 1 matmul reads keeps the n-on-lanes layout the tile kernels are built for
    static const bool kon_on = !(getenv("ZGML_HIP_QMV_KON") && atoi(getenv("ZGML_HIP_QMV_KON")) == 0);
    std::vector<char> qw_m1(prog->n_qweights, kon_on ? 1 : 0);
    std::vector<char> qw_m1_all(prog->n_qweights, 1); // every use is a dense M = 1 row (the W8A8 arm's condition, reference.zig:512-516)
    for (const auto& op : p->ops)
        if (op.kind == ZGML_DOP_QMATMUL && op.u.qmatmul.weight_idx < prog->n_qweights) {
            const auto& q = op.u.qmatmul;
            if (q.M != 1) qw_m1[q.weight_idx] = 0;
            if (q.M != 1 || (q.input_row_stride != 0 && q.input_row_stride != q.K) || (q.dst_row_stride != 0 && q.dst_row_stride != q.N)) qw_m1_all[q.weight_idx] = 0;
        }
    for (uint64_t i = 0; ok && i < prog->n_qweights; i++) {
        if (!qw_live[i]) continue;
        const zgml_qweight_upload& qw = prog->qweights[i];
        QWeightDev& w = p->qweights[i];
        w.K = (uint32_t)qw.rows, w.N = (uint32_t)qw.cols, w.bs = (uint32_t)qw.block_size;
        const uint64_t n_elems = qw.rows * qw.cols;
        const uint64_t n_blocks = (n_elems + qw.block_size - 1) / qw.block_size;
        if (const int form = gguf_form(qw)) { // file blocks straight to the device, unpacked + re-packed there
            w.format = form == 1 ? QW_Q4 : QW_Q8;
            w.scale_f16 = 1;
            pending[i].gguf = true;
            ok = CTX_CHECK(ctx, hipMalloc(&pending[i].raw_a, qw.data_len)) &&
                 CTX_CHECK(ctx, hipMemcpyAsync(pending[i].raw_a, qw.data, qw.data_len, hipMemcpyHostToDevice, ctx->stream));
        } else {
            ok = CTX_CHECK(ctx, hipMalloc(&pending[i].raw_a, n_elems ? n_elems : 1)) &&
                 CTX_CHECK(ctx, hipMalloc((void**)&pending[i].raw_s, (n_blocks ? n_blocks : 1) * sizeof(float))) &&
                 CTX_CHECK(ctx, hipMemcpyAsync(pending[i].raw_a, qw.data, n_elems, hipMemcpyHostToDevice, ctx->stream)) &&
                 CTX_CHECK(ctx, hipMemcpyAsync(pending[i].raw_s, qw.scales, n_blocks * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
            if (!ok) break;
            if (ctx->opt_w8a8 && qw_m1_all[i] && w8a8_applies(qw.rows, qw.cols, qw.block_size)) { // the reference's W8A8 arm (w8a8.hip)
                w.format = QW_W8A8;
                w.KC = (uint32_t)(qw.rows / 32);
                w8a8_packed_bytes(qw.rows, qw.cols, &w.qs_bytes, &w.sc_bytes);
                qs_total += w.qs_bytes, sc_total += w.sc_bytes;
                continue;
            }
            if (!qweight_packable(qw.rows, qw.cols, qw.block_size)) { // odd shapes keep the raw form
                w.format = QW_RAW;
                w.qs = pending[i].raw_a, w.sc = pending[i].raw_s;
                w.qs_bytes = n_elems, w.sc_bytes = n_blocks * 4;
                p->owned.push_back(w.qs);
                p->owned.push_back(w.sc);
                pending[i] = PendingPack{};
                continue;
            }
            const uint32_t cls = classify_qweight(ctx->stream, (const int8_t*)pending[i].raw_a, n_elems, pending[i].raw_s, n_blocks, flags);
            w.format = (cls & 1) ? QW_Q4 : QW_Q8;
            w.scale_f16 = (cls & 2) ? 1 : 0;
        }
        if (!ok) break;
        // (short K stays n-on-lanes: those launches are one latency chain inside the decode stream, where the longer fold of
        // the K-on-lanes tail and the hand-over of the norm cost more than the cheaper inner loop saves: SmolLM-135M
        // 1770 tok/s either way without the hand-over, 1680 with it)
        static const uint32_t kon_min_k = getenv("ZGML_HIP_QMV_KON_MIN_K") ? (uint32_t)atoi(getenv("ZGML_HIP_QMV_KON_MIN_K")) : 2049u;
        if (w.format == QW_Q4 && w.scale_f16 && qw_m1[i] && w.K >= kon_min_k) w.format = QW_Q4K;
        w.KC = (uint32_t)((qw.rows + 31) / 32);
        packed_bytes(w.format, w.scale_f16, w.K, w.N, &w.qs_bytes, &w.sc_bytes);
        qs_total += w.qs_bytes, sc_total += w.sc_bytes;
    }
    char *qs_arena = nullptr, *sc_arena = nullptr;
    static const bool use_arena = !(getenv("ZGML_HIP_WEIGHT_ARENA") && atoi(getenv("ZGML_HIP_WEIGHT_ARENA")) == 0);
    if (ok && qs_total && use_arena) {
        ok = CTX_CHECK(ctx, hipMalloc((void**)&qs_arena, qs_total)) && CTX_CHECK(ctx, hipMalloc((void**)&sc_arena, sc_total ? sc_total : 16));
        if (qs_arena) p->owned.push_back(qs_arena);
        if (sc_arena) p->owned.push_back(sc_arena);
    }
    uint64_t qs_off = 0, sc_off = 0;
    for (uint64_t i = 0; i < prog->n_qweights; i++) {
        QWeightDev& w = p->qweights[i];
        if (ok && pending[i].raw_a) {
            if (use_arena) {
                w.qs = qs_arena + qs_off, w.sc = sc_arena + sc_off;
                qs_off += w.qs_bytes, sc_off += w.sc_bytes;
            } else { // experiment: one allocation per weight
                ok = CTX_CHECK(ctx, hipMalloc(&w.qs, w.qs_bytes)) && CTX_CHECK(ctx, hipMalloc(&w.sc, w.sc_bytes));
                if (!ok) break;
                p->owned.push_back(w.qs);
                p->owned.push_back(w.sc);
            }
            if (pending[i].gguf)
                launch_pack_gguf(ctx->stream, (const uint8_t*)pending[i].raw_a, w);
            else if (w.format == QW_W8A8)
                launch_pack_w8a8(ctx->stream, (const int8_t*)pending[i].raw_a, pending[i].raw_s, w);
            else
                launch_pack_qweight(ctx->stream, (const int8_t*)pending[i].raw_a, pending[i].raw_s, w);
        }
    }
    if (ok && prog->n_qweights) ok = CTX_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& pd : pending) { // raw images are only needed until the pack kernels have run
        if (pd.raw_a) hipFree(pd.raw_a);
        if (pd.raw_s) hipFree(pd.raw_s);
    }
    if (flags) hipFree(flags);
    { // weight sets beyond the 256 MB Infinity Cache are streamed with non-temporal loads (see qmatvec.hip: wload)
        static const uint64_t nt_min = getenv("ZGML_HIP_NT_MIN_BYTES") ? strtoull(getenv("ZGML_HIP_NT_MIN_BYTES"), nullptr, 0) : (192ull << 20);
        uint64_t total = 0;
        for (const QWeightDev& w : p->qweights) total += w.qs_bytes + w.sc_bytes;
        if (total >= nt_min)
            for (QWeightDev& w : p->qweights) w.stream_nt = 1;
    }

    // split-K scratch shared by all qmatmul launches (they are serialised on one stream)
    for (const auto& op : p->ops)
        if (op.kind == ZGML_DOP_QMATMUL) {
            uint64_t b = qmatmul_scratch_bytes(p->qweights[op.u.qmatmul.weight_idx], op.u.qmatmul.M);
            if (b > p->scratch_bytes) p->scratch_bytes = b;
        } else if (op.kind == ZGML_DOP_MATMUL && op.u.matmul.b < p->f16_weights.size() && p->f16_weights[op.u.matmul.b]) {
            uint64_t b = dense_f16_scratch_bytes((uint32_t)op.u.matmul.geom.M, (uint32_t)op.u.matmul.geom.K); // pre-rounded A operand
            if (b > p->scratch_bytes) p->scratch_bytes = b;
        }
    // the head in front of the block holds the K-split fan-in counters: zero once, every launch re-arms them (kernels.h)
    if (ok && p->scratch_bytes) {
        char* base = nullptr;
        ok = CTX_CHECK(ctx, hipMalloc((void**)&base, kQmmScratchHead + p->scratch_bytes)) && CTX_CHECK(ctx, hipMemsetAsync(base, 0, kQmmScratchHead, ctx->stream));
        if (base) p->scratch = (float*)(base + kQmmScratchHead);
    }

    const size_t n_dyn = p->ops.empty() ? 1 : p->ops.size();
    if (ok)
        ok = CTX_CHECK(ctx, hipMalloc((void**)&p->dyn_dev, n_dyn * sizeof(uint32_t))) &&
             CTX_CHECK(ctx, hipHostMalloc((void**)&p->dyn_host, n_dyn * sizeof(uint32_t), hipHostMallocDefault));
    if (ok) {
        memset(p->dyn_host, 0xFF, n_dyn * sizeof(uint32_t));
        set_dyn_from_ops(p);
        p->dyn_dirty = true;
        build_plan(p);
        ok = CTX_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (!ok) {
        zgml_hip_free_program(ctx, p);
        return nullptr;
    }
    return p;
}

// one op's dynamic field taken over from `src` (same kind, same static part): the program's copy, the device word, the attention's
// compile-time bound; false when the value leaves the span the level schedule assumed (dynamic_fields_in_bounds for this op)
static bool apply_dynamic(zgml_hip_program* p, uint32_t i, const zgml_device_op& src) {
    auto ext2 = [](uint64_t n0, uint64_t s0, uint64_t n1, uint64_t s1) -> uint64_t { return n0 && n1 ? (n0 - 1) * s0 + (n1 - 1) * s1 + 1 : 0; }; // (schedule.hip's extent of a strided 2-D view)
    zgml_device_op& op = p->ops[i];
    uint32_t v = 0;
    bool ok = true;
    const DynBound* const b = p->plan_batched && i < p->sched.bounds.size() ? &p->sched.bounds[i] : nullptr;
    switch (op.kind) {
        case ZGML_DOP_SLICE_ASSIGN: {
            auto& sa = op.u.slice_assign;
            v = sa.dst_offset = src.u.slice_assign.dst_offset;
            if (b && b->kind == 1) {
                const uint64_t lo = sa.dst_offset, hi = lo + ext2(sa.rows, sa.dst_row_stride, sa.cols, sa.dst_col_stride);
                ok = !(lo < b->lo || hi > b->hi);
            }
            break;
        }
        case ZGML_DOP_ATTENTION:
            v = op.u.attention.seq_kv = src.u.attention.seq_kv;
            if (b && b->kind == 2) ok = v <= b->max_seq_kv;
            if (i < p->seq_kv_bound.size()) p->seq_kv_bound[i] = std::max(p->seq_kv_bound[i], v);
            break;
        case ZGML_DOP_KVQ_STORE: v = op.u.kvq_store.col = src.u.kvq_store.col; break;
        case ZGML_DOP_ATTENTION_KVQ:
            v = op.u.attention_kvq.seq_kv = src.u.attention_kvq.seq_kv;
            if (b && b->kind == 3) ok = v <= b->max_seq_kv;
            if (i < p->seq_kv_bound.size()) p->seq_kv_bound[i] = std::max(p->seq_kv_bound[i], v);
            break;
        default: return true;
    }
    if (p->dyn_host[i] != v) p->dyn_host[i] = v, p->dyn_dirty = true;
    return ok;
}

// The per-token refresh reduced to what it is (src/backend/program.zig:7452-7490 StepDynamicParams, as the reference's wgpu backend
// applies it — src/backend/wgpu.zig:1162-1169): every KV store goes to column `slice_pos` (dst_offset = dst_base_offset +
// slice_pos * patch_stride; quantised caches: col = col_base + slice_pos * patch_stride) and every attention reads `seq_kv` keys.
// O(#dynamic ops); static fields are NOT looked at — a caller that may have changed them calls zgml_hip_refresh_program.
int zgml_hip_refresh_dynamic(zgml_hip_ctx* ctx, zgml_hip_program* p, uint32_t slice_pos, uint32_t seq_kv) {
    if (!ctx || !p) return -1;
    const uint64_t t_prof = ctx->host_prof ? now_ns() : 0;
    bool in_bounds = true;
    for (const uint32_t i : p->dyn_ops) {
        zgml_device_op src = p->ops[i];
        switch (src.kind) {
            case ZGML_DOP_SLICE_ASSIGN: src.u.slice_assign.dst_offset = src.u.slice_assign.dst_base_offset + slice_pos * src.u.slice_assign.patch_stride; break;
            case ZGML_DOP_ATTENTION: src.u.attention.seq_kv = seq_kv; break;
            case ZGML_DOP_KVQ_STORE: src.u.kvq_store.col = src.u.kvq_store.col_base + slice_pos * src.u.kvq_store.patch_stride; break;
            case ZGML_DOP_ATTENTION_KVQ: src.u.attention_kvq.seq_kv = seq_kv; break;
            default: break;
        }
        in_bounds = apply_dynamic(p, i, src) && in_bounds;
    }
    if (p->plan_batched && !in_bounds) p->batching_safe = false, p->plan_dirty = true;
    if (ctx->host_prof) ctx->prof_ns[0] += now_ns() - t_prof, ctx->prof_calls[0]++;
    return 0;
}

void zgml_hip_refresh_program(zgml_hip_ctx* ctx, zgml_hip_program* p, const zgml_device_op* ops, uint64_t n_ops) {
    if (!ctx || !p || !ops) return;
    const uint64_t t_prof = ctx->host_prof ? now_ns() : 0;
    bool static_same = n_ops == p->ops.size();
    for (uint64_t i = 0; static_same && i < n_ops; i++) static_same = same_static(p->ops[i], ops[i]);
    if (static_same) {
        // the common per-token case: only dynamic fields moved — of the ops that have one (p->dyn_ops)
        bool in_bounds = true;
        for (const uint32_t i : p->dyn_ops) in_bounds = apply_dynamic(p, i, ops[i]) && in_bounds;
        if (p->plan_batched && !in_bounds) {
            // a dynamic field left the span the level schedule assumed: the reordered plan is no
            // longer provably equivalent, fall back to program order for good
            p->batching_safe = false;
            p->plan_dirty = true;
        }
        if (ctx->host_prof) ctx->prof_ns[0] += now_ns() - t_prof, ctx->prof_calls[0]++;
        return;
    }
    if (n_ops != p->ops.size()) {
        ctx->fail("refresh_program: op list length changed");
        return;
    }
    // a static field changed: legal for the reference's CPU backend (it re-reads ops every
    // execute, src/backend/cpu.zig:128-131), so honour it by rebuilding the launch list
    for (uint64_t i = 0; i < n_ops; i++) {
        // ... except where compile_program specialised a weight's layout on the op list it was given: a Q4_0 weight that only
        // fed M = 1 mat-vecs was packed K-on-lanes (QW_Q4K), which the M > 1 tile kernels cannot read. Refuse the refresh (the
        // program keeps its previous ops) instead of skipping the launch later: a stale destination must never look like success.
        const zgml_device_op& op = ops[i];
        if (op.kind == ZGML_DOP_QMATMUL && op.u.qmatmul.M != 1 && op.u.qmatmul.weight_idx < p->qweights.size() &&
            (p->qweights[op.u.qmatmul.weight_idx].format == QW_Q4K || p->qweights[op.u.qmatmul.weight_idx].format == QW_W8A8)) {
            ctx->fail("refresh_program: op " + std::to_string(i) + " turns weight " + std::to_string(op.u.qmatmul.weight_idx) +
                      " into the operand of an M = " + std::to_string(op.u.qmatmul.M) +
                      " qmatmul, but the weight was packed for M = 1 mat-vecs at compile time (K-on-lanes layout): recompile the program");
            return;
        }
    }
    hipStreamSynchronize(ctx->stream);
    own_ops(p, ops, n_ops);
    set_dyn_from_ops(p);
    p->plan_dirty = true;
}

// inputs: pack -> one H2D -> scatter kernel
static bool upload_inputs(zgml_hip_ctx* ctx, zgml_hip_program* p, const zgml_program_io* inputs, uint64_t n_inputs) {
    hipStream_t s = ctx->stream;
    if (!prepare_io(p, p->in_plan, inputs, n_inputs)) return false;
    if (!n_inputs) return true;
    if (p->in_plan.word_aligned) {
        char* st = (char*)p->stage_host;
        uint64_t off = 0;
        for (uint64_t i = 0; i < n_inputs; i++) {
            memcpy(st + off, inputs[i].host_ptr, inputs[i].size);
            off += (inputs[i].size + 3) / 4 * 4;
        }
        uint32_t rows = (uint32_t)n_inputs;
        if (p->in_plan.dyn_row && p->dyn_dirty) { // the refreshed dynamic words ride along (prepare_io: the table's last row)
            memcpy(st + off, p->dyn_host, p->ops.size() * sizeof(uint32_t));
            off += p->ops.size() * sizeof(uint32_t);
            rows++;
            p->dyn_dirty = false;
        }
        hipMemcpyAsync(p->stage_dev, p->stage_host, off, hipMemcpyHostToDevice, s);
        scatter_words_kernel<<<dim3(rows, io_grid_y(p->in_plan.max_row_words)), 256, 0, s>>>(p->in_plan.table_dev, (const uint32_t*)p->stage_dev);
    } else {
        for (uint64_t i = 0; i < n_inputs; i++)
            hipMemcpyAsync((char*)p->bufs[inputs[i].buf_idx] + inputs[i].offset, inputs[i].host_ptr, inputs[i].size,
                           hipMemcpyHostToDevice, s);
    }
    return true;
}

// outputs: gather kernel -> one D2H -> sync -> unpack
static bool download_outputs(zgml_hip_ctx* ctx, zgml_hip_program* p, const zgml_program_io* outputs, uint64_t n_outputs) {
    hipStream_t s = ctx->stream;
    if (!prepare_io(p, p->out_plan, outputs, n_outputs)) return false;
    if (n_outputs) {
        if (p->out_plan.word_aligned && n_outputs == 1) { // (the decode step: the logits) straight from the buffer, no gather launch in front
            hipMemcpyAsync(p->stage_host, (char*)p->bufs[outputs[0].buf_idx] + outputs[0].offset, outputs[0].size, hipMemcpyDeviceToHost, s);
        } else if (p->out_plan.word_aligned) {
            gather_words_wide_kernel<<<dim3((uint32_t)n_outputs, io_grid_y(p->out_plan.max_row_words)), 256, 0, s>>>(p->out_plan.table_dev, (uint32_t*)p->stage_dev);
            hipMemcpyAsync(p->stage_host, p->stage_dev, (uint64_t)p->out_plan.total_words * 4, hipMemcpyDeviceToHost, s);
        } else {
            for (uint64_t i = 0; i < n_outputs; i++)
                hipMemcpyAsync(outputs[i].host_ptr, (char*)p->bufs[outputs[i].buf_idx] + outputs[i].offset,
                               outputs[i].size, hipMemcpyDeviceToHost, s);
        }
    }
    const uint64_t t0 = now_ns();
    const bool ok = CTX_CHECK(ctx, hipStreamSynchronize(s)) && ctx->handoff_ok("execute_program");
    p->profile.sync_time_ns += now_ns() - t0;
    p->profile.sync_count++;
    if (n_outputs && p->out_plan.word_aligned) {
        const char* st = (const char*)p->stage_host;
        uint64_t off = 0;
        for (uint64_t i = 0; i < n_outputs; i++) {
            memcpy(outputs[i].host_ptr, st + off, outputs[i].size);
            off += (outputs[i].size + 3) / 4 * 4;
        }
    }
    return ok;
}

// The decode step's fast path (round 5): inputs, dynamic words, the plan and the outputs as ONE graph launch (zgml_hip_program::
// io_graph). The GPU-side timeline of the copy-command form (profiles/r05_vtable_timeline.txt) shows what the commands around the
// plan's graph cost per token: H2D + wait for the scatter kernel + the kernel, then ~18 us from the last kernel to the D2H's start
// + the copy. Here the scatter kernel reads the pinned staging buffer through its device mapping, the gather kernel writes the
// outputs into a second pinned buffer, and both are nodes of the graph. false: not applicable (no graph, odd alignment, no
// inputs or outputs, profiling) — the caller takes the general path.
static bool execute_io_graph(zgml_hip_ctx* ctx, zgml_hip_program* p, const zgml_program_io* inputs, uint64_t n_inputs,
                             const zgml_program_io* outputs, uint64_t n_outputs, uint64_t t0) {
    static const bool enabled = !(getenv("ZGML_HIP_IO_GRAPH") && atoi(getenv("ZGML_HIP_IO_GRAPH")) == 0);
    if (!enabled || !ctx->opt_graph || ctx->opt_profile || !n_inputs || !n_outputs || !p->in_plan.word_aligned || !p->out_plan.word_aligned ||
        !p->in_plan.dyn_row || !p->in_plan.table_dev || !p->out_plan.table_dev)
        return false;
    if (p->plan_dirty || p->fuse_epoch != ctx->fuse_epoch) {
        free_graph(p);
        build_plan(p);
    }
    if (p->plan.empty()) return false;
    hipStream_t s = ctx->stream;
    { // pack: the inputs, then the dynamic words (always: the graph's scatter launch has a fixed number of rows)
        char* st = (char*)p->stage_host;
        uint64_t off = 0;
        for (uint64_t i = 0; i < n_inputs; i++) {
            memcpy(st + off, inputs[i].host_ptr, inputs[i].size);
            off += (inputs[i].size + 3) / 4 * 4;
        }
        memcpy(st + off, p->dyn_host, p->ops.size() * sizeof(uint32_t));
        p->dyn_dirty = true; // (until the graph that carries them has been launched: a fall-back to the general path must still send them)
    }
    const uint64_t t1 = ctx->host_prof ? now_ns() : 0;
    const uint32_t in_rows = (uint32_t)n_inputs + 1, out_rows = (uint32_t)n_outputs;
    // where the outputs land: the caller's own buffer once it has proved stable (zgml_hip_program::pin_host), else the staging buffer
    if (p->pin_allowed && !p->pin_off && n_outputs == 1) {
        const zgml_program_io& o = outputs[0];
        if (p->pin_host && (p->pin_host != o.host_ptr || p->pin_size != o.size)) { // moved: give the registration up for good
            hipStreamSynchronize(s);
            free_io_graph(p);
            (void)hipHostUnregister(p->pin_host);
            p->pin_host = p->pin_dev = nullptr, p->pin_off = true;
        } else if (!p->pin_host) {
            p->pin_seen = p->pin_cand == o.host_ptr ? p->pin_seen + 1 : 1;
            p->pin_cand = o.host_ptr;
            if (p->pin_seen >= 3 && o.size >= 4096) {
                void* dev = nullptr;
                if (hipHostRegister(o.host_ptr, o.size, hipHostRegisterMapped) != hipSuccess) { // (e.g. memory the caller has pinned itself: left alone)
                    (void)hipGetLastError();
                    p->pin_off = true;
                } else if (hipHostGetDevicePointer(&dev, o.host_ptr, 0) == hipSuccess && dev) {
                    p->pin_host = o.host_ptr, p->pin_dev = dev, p->pin_size = o.size;
                } else {
                    (void)hipGetLastError();
                    (void)hipHostUnregister(o.host_ptr); // (our own registration)
                    (void)hipGetLastError();
                    p->pin_off = true;
                }
            }
        }
    } else if (p->pin_host) {
        hipStreamSynchronize(s);
        free_io_graph(p);
        (void)hipHostUnregister(p->pin_host);
        p->pin_host = p->pin_dev = nullptr, p->pin_off = true;
    }
    void* out_dev = p->pin_host ? p->pin_dev : nullptr;
    if (!out_dev && hipHostGetDevicePointer(&out_dev, p->stage_out_host, 0) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (p->io_graph_exec && (p->io_in_rows != in_rows || p->io_out_rows != out_rows || p->io_out_dev != out_dev)) free_io_graph(p);
    if (!p->io_graph_exec) {
        void* in_dev = nullptr;
        if (hipHostGetDevicePointer(&in_dev, p->stage_host, 0) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        const uint32_t gy = io_grid_y(p->out_plan.max_row_words);
        hipGraph_t g = nullptr;
        if (!CTX_CHECK(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal))) return false;
        scatter_words_kernel<<<dim3(in_rows, io_grid_y(p->in_plan.max_row_words)), 256, 0, s>>>(p->in_plan.table_dev, (const uint32_t*)in_dev);
        run_plan(p, s, 0, p->plan.size());
        gather_words_wide_kernel<<<dim3(out_rows, gy), 256, 0, s>>>(p->out_plan.table_dev, (uint32_t*)out_dev);
        if (!CTX_CHECK(ctx, hipStreamEndCapture(s, &g)) || !g) return false;
        dump_graph(g, "program-io");
        hipGraphExec_t ge = nullptr;
        if (!CTX_CHECK(ctx, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0))) {
            hipGraphDestroy(g);
            return false;
        }
        p->io_graph = g, p->io_graph_exec = ge, p->io_in_rows = in_rows, p->io_out_rows = out_rows, p->io_out_dev = out_dev;
    }
    if (!CTX_CHECK(ctx, hipGraphLaunch(p->io_graph_exec, s))) return true; // (reported; nothing ran)
    p->dyn_dirty = false;
    const uint64_t t2 = ctx->host_prof ? now_ns() : 0;
    const uint64_t ts = now_ns();
    const bool ok = CTX_CHECK(ctx, hipStreamSynchronize(s)) && ctx->handoff_ok("execute_program");
    p->profile.sync_time_ns += now_ns() - ts;
    p->profile.sync_count++;
    if (ok && !p->pin_host) {
        const char* st = (const char*)p->stage_out_host;
        uint64_t off = 0;
        for (uint64_t i = 0; i < n_outputs; i++) {
            memcpy(outputs[i].host_ptr, st + off, outputs[i].size);
            off += (outputs[i].size + 3) / 4 * 4;
        }
    }
    if (ctx->host_prof) {
        const uint64_t t3 = now_ns();
        ctx->prof_ns[1] += t1 - t0, ctx->prof_ns[2] += t2 - t1, ctx->prof_ns[3] += t3 - t2;
        ctx->prof_calls[1]++, ctx->prof_calls[2]++, ctx->prof_calls[3]++;
    }
    p->profile.call_count++;
    p->profile.backend_op_count += p->ops.size();
    p->profile.backend_dispatch_count += p->plan.size();
    CTX_CHECK(ctx, hipGetLastError());
    return true;
}

int zgml_hip_program_pin_outputs(zgml_hip_ctx* ctx, zgml_hip_program* p, int on) {
    if (!ctx || !p) return -1;
    hipSetDevice(ctx->device);
    if (!on && p->pin_host) {
        hipStreamSynchronize(ctx->stream);
        free_io_graph(p);
        (void)hipHostUnregister(p->pin_host);
        (void)hipGetLastError();
        p->pin_host = p->pin_dev = nullptr;
    }
    p->pin_allowed = on != 0, p->pin_off = false, p->pin_seen = 0, p->pin_cand = nullptr;
    return 0;
}

void zgml_hip_execute_program(zgml_hip_ctx* ctx, zgml_hip_program* p, const zgml_program_io* inputs, uint64_t n_inputs,
                              const zgml_program_io* outputs, uint64_t n_outputs) {
    if (!ctx || !p) return;
    hipSetDevice(ctx->device);
    // both transfer tables are validated before anything is enqueued
    if (!prepare_io(p, p->in_plan, inputs, n_inputs) || !prepare_io(p, p->out_plan, outputs, n_outputs)) return;
    const uint64_t t0 = ctx->host_prof ? now_ns() : 0;
    if (execute_io_graph(ctx, p, inputs, n_inputs, outputs, n_outputs, t0)) return;
    if (!upload_inputs(ctx, p, inputs, n_inputs)) return;
    const uint64_t t1 = ctx->host_prof ? now_ns() : 0;
    enqueue(p);
    const uint64_t t2 = ctx->host_prof ? now_ns() : 0;
    download_outputs(ctx, p, outputs, n_outputs);
    if (ctx->host_prof) {
        const uint64_t t3 = now_ns();
        ctx->prof_ns[1] += t1 - t0, ctx->prof_ns[2] += t2 - t1, ctx->prof_ns[3] += t3 - t2;
        ctx->prof_calls[1]++, ctx->prof_calls[2]++, ctx->prof_calls[3]++;
    }
    p->profile.call_count++;
    p->profile.backend_op_count += p->ops.size();
    p->profile.backend_dispatch_count += p->plan.size();
    CTX_CHECK(ctx, hipGetLastError());
}

// ── capture-friendly split of a step (multi-GPU harness records ops + collectives into one graph) ──
// stage_inputs: host side only — validate the transfer table and copy the host leaves into the
// pinned staging buffer. enqueue_staged: device side only — one H2D of the staging buffer, the
// scatter kernel and an unconditional copy of the dynamic-parameter block; safe to record into a
// stream capture and replay, the replay picks up whatever stage_inputs / refresh_program wrote last.
int zgml_hip_stage_inputs(zgml_hip_ctx* ctx, zgml_hip_program* p, const zgml_program_io* inputs, uint64_t n_inputs) {
    if (!ctx || !p) return -1;
    hipSetDevice(ctx->device);
    if (!prepare_io(p, p->in_plan, inputs, n_inputs)) return -1;
    if (n_inputs && !p->in_plan.word_aligned) {
        ctx->fail("stage_inputs: inputs must be 4-byte aligned in offset and size");
        return -1;
    }
    char* st = (char*)p->stage_host;
    uint64_t off = 0;
    for (uint64_t i = 0; i < n_inputs; i++) {
        memcpy(st + off, inputs[i].host_ptr, inputs[i].size);
        off += (inputs[i].size + 3) / 4 * 4;
    }
    p->staged_bytes = off;
    p->staged_n = n_inputs;
    return 0;
}

void zgml_hip_enqueue_staged(zgml_hip_ctx* ctx, zgml_hip_program* p) {
    if (!ctx || !p) return;
    hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    if (p->staged_n) {
        hipMemcpyAsync(p->stage_dev, p->stage_host, p->staged_bytes, hipMemcpyHostToDevice, s);
        scatter_words_kernel<<<dim3((uint32_t)p->staged_n, io_grid_y(p->in_plan.max_row_words)), 256, 0, s>>>(p->in_plan.table_dev, (const uint32_t*)p->stage_dev);
    }
    if (!p->ops.empty()) hipMemcpyAsync(p->dyn_dev, p->dyn_host, p->ops.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s);
    p->dyn_dirty = false;
}

// argmax without the blocking read-back: result lands in pinned memory, read it with
// zgml_hip_argmax_result after synchronising the stream (or after a graph replay completed)
int zgml_hip_enqueue_argmax(zgml_hip_ctx* ctx, zgml_hip_program* p, uint16_t buf_idx, uint64_t offset, uint64_t n) {
    if (!ctx || !p || buf_idx >= p->bufs.size() || !p->bufs[buf_idx] || offset + n > p->sizes[buf_idx]) return -1;
    hipSetDevice(ctx->device);
    launch_argmax(ctx->stream, p->bufs[buf_idx] + offset, n, ctx->arg_val, ctx->arg_idx, ctx->arg_out);
    hipMemcpyAsync(ctx->arg_out_host, ctx->arg_out, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
    return 0;
}
int64_t zgml_hip_argmax_result(zgml_hip_ctx* ctx) { return ctx ? *ctx->arg_out_host : -1; }

void zgml_hip_upload_inputs(zgml_hip_ctx* ctx, zgml_hip_program* p, const zgml_program_io* inputs, uint64_t n_inputs) {
    if (!ctx || !p) return;
    hipSetDevice(ctx->device);
    upload_inputs(ctx, p, inputs, n_inputs);
}

void zgml_hip_download_outputs(zgml_hip_ctx* ctx, zgml_hip_program* p, const zgml_program_io* outputs, uint64_t n_outputs) {
    if (!ctx || !p) return;
    hipSetDevice(ctx->device);
    download_outputs(ctx, p, outputs, n_outputs);
}

void shard_peer_release(zgml_hip_program* p); // (defined with the row-shard path below)
void zgml_hip_free_program(zgml_hip_ctx* ctx, zgml_hip_program* p) {
    if (!p) return;
    if (ctx) {
        hipSetDevice(ctx->device);
        hipStreamSynchronize(ctx->stream);
    }
    if (ctx && !ctx->handoff_ok("free_program")) fprintf(stderr, "[zgml_hip] ERROR: %s\n", ctx->err.c_str());
    for (void* d : p->fuse_owned) hipFree(d);
    if (!p->attn_traces.empty()) { // stamps of the last execution, 100 MHz wall clock -> ns
        fprintf(stderr, "[zgml_hip] attention trace (ns since previous launch's end | start->params | ->dyn | ->rope | ->scores | ->max | ->pv | ->end)\n");
        unsigned long long prev_end = 0;
        for (size_t i = 0; i < p->attn_traces.size(); i++) {
            const unsigned long long* t = p->attn_traces[i];
            fprintf(stderr, "  L%02zu gap %6lld |", i, prev_end ? (long long)(t[0] - prev_end) * 10 : -1);
            for (int k = 1; k < 8; k++) fprintf(stderr, " %5lld", (long long)(t[k] - t[k - 1]) * 10);
            fprintf(stderr, "\n");
            prev_end = t[7];
        }
        for (auto* t : p->attn_traces) hipHostFree(t);
    }
    if (!p->ks_traces.empty()) {
        fprintf(stderr, "[zgml_hip] K-split launches, workgroup 0, ns between stamps (gap = since the previous launch's last stamp); ks-proj / ks-mlp: start | loads issued | "
                        "parts summed | vector whole | x staged | FMAs | barrier | folded (mlp: + SiLU) | (mlp: barrier | down partial); ks-attn-o: start | record | loads issued | attention | merged | partial O\n");
        unsigned long long prev_end = 0;
        for (const auto& k : p->ks_traces) {
            int last = 0;
            for (int j = 0; j < 16; j++)
                if (k.t[j]) last = j;
            fprintf(stderr, "  %-10s gap %6lld |", k.what, prev_end ? (long long)(k.t[0] - prev_end) * 10 : -1);
            for (int j = 1; j <= last; j++) fprintf(stderr, " %5lld", k.t[j] ? (long long)(k.t[j] - k.t[j - 1]) * 10 : -1);
            fprintf(stderr, " | span %5lld\n", (long long)(k.t[last] - k.t[0]) * 10);
            if (k.t[16]) { // lane 0 of the workgroup's last wave: its stamps relative to thread 0's start
                fprintf(stderr, "      last wave: start %+5lld |", (long long)(k.t[16] - k.t[0]) * 10);
                for (int j = 1; j <= last && j < 12; j++) fprintf(stderr, " %5lld", k.t[16 + j] ? (long long)(k.t[16 + j] - k.t[16 + j - 1]) * 10 : -1);
                fprintf(stderr, "\n");
            }
            prev_end = k.t[last];
        }
        for (auto& k : p->ks_traces) hipHostFree(k.t);
    }
    if (!p->qmv_traces.empty()) {
        fprintf(stderr, "[zgml_hip] mat-vec trace, workgroup 0 (ns: gap since previous mat-vec end | ->loads issued | ->x arrived(+sumsq) | ->x staged | ->streamed | ->reduced+epilogue)"
                        " || K-on-lanes launches also: LAST workgroup, start after workgroup 0's start | ->loads issued | ->streamed | ->end; launch span = first start -> last end\n");
        unsigned long long prev_end = 0;
        for (size_t i = 0; i < p->qmv_traces.size() && i < 48; i++) {
            const auto& q = p->qmv_traces[i];
            fprintf(stderr, "  #%02zu K=%5u N0=%5u parts=%u pro=%u gap %6lld |", i, q.K, q.N, q.parts, q.pro,
                    prev_end ? (long long)(q.t[0] - prev_end) * 10 : -1);
            for (int k = 1; k < 6; k++) fprintf(stderr, " %5lld", (long long)(q.t[k] - q.t[k - 1]) * 10);
            if (q.t[8]) {
                fprintf(stderr, " || %5lld | %5lld %5lld %5lld | span %5lld", (long long)(q.t[8] - q.t[0]) * 10, (long long)(q.t[9] - q.t[8]) * 10,
                        (long long)(q.t[12] - q.t[9]) * 10, (long long)(q.t[13] - q.t[12]) * 10, (long long)(std::max(q.t[13], q.t[5]) - q.t[0]) * 10);
                prev_end = std::max(q.t[13], q.t[5]);
            } else {
                prev_end = q.t[5];
            }
            fprintf(stderr, "\n");
        }
        for (auto& q : p->qmv_traces) hipHostFree(q.t);
    }
    free_graph(p);
    free_resident(p);
    free_param_blobs(p);
    shard_peer_release(p);
    if (p->arena) hipFree(p->arena);
    if (p->zero_word) hipFree(p->zero_word);
    for (void* d : p->owned) hipFree(d);
    if (p->scratch) hipFree((char*)p->scratch - kQmmScratchHead);
    if (p->dyn_dev) hipFree(p->dyn_dev);
    if (p->dyn_host) hipHostFree(p->dyn_host);
    if (p->pin_host) (void)hipHostUnregister(p->pin_host), (void)hipGetLastError();
    if (p->stage_host) hipHostFree(p->stage_host);
    if (p->stage_out_host) hipHostFree(p->stage_out_host);
    if (p->stage_dev) hipFree(p->stage_dev);
    if (p->in_plan.table_dev) hipFree(p->in_plan.table_dev);
    if (p->out_plan.table_dev) hipFree(p->out_plan.table_dev);
    delete p;
}

zgml_runtime_profile* zgml_hip_get_runtime_profile(zgml_hip_ctx*, zgml_hip_program* p) {
    return p ? &p->profile : nullptr;
}

// ── extensions ──────────────────────────────────────────────────────────────────────────────

// a caller about to touch a buffer some hoisted (run-once) repeat read or wrote: back to running every repeat in the plan
uint64_t zgml_hip_program_plan_text(zgml_hip_ctx* ctx, zgml_hip_program* p, char* out, uint64_t cap) {
    if (!ctx || !p) return 0;
    if (p->plan_dirty || p->fuse_epoch != ctx->fuse_epoch) {
        free_graph(p);
        build_plan(p);
    }
    static const char* const pro_names[4] = {"none", "mul", "rmsnorm", "prenorm"};
    std::string t;
    for (size_t i = 0; i < p->plan.size(); i++) {
        const Launch& L = p->plan[i];
        char line[256];
        snprintf(line, sizeof line, "%zu: kind %u ops %u [%u..%u]", i, L.kind, L.n_ops, L.op_lo, L.op_hi);
        t += line;
        if (L.qmv_desc) {
            const QmvLaunch& q = *L.qmv_desc;
            snprintf(line, sizeof line, " qmv parts %u K %u pro %s%s%s", q.n_parts, q.K, pro_names[q.pro.kind < 4 ? q.pro.kind : 0], q.next.xg_out ? " prepares-next-norm" : "",
                     q.pair_out ? " pair" : "");
            t += line;
        }
        if (L.adec_desc) t += L.qmv_desc ? " +decode-attention" : " decode-attention";
        if (L.tag) t += std::string(" ") + L.tag;
        if (L.hook && L.hook->ap && *L.hook->ap) t += " writes-A-operand";
        t += "\n";
    }
    if (out && cap) {
        const uint64_t n = std::min<uint64_t>(cap - 1, t.size());
        memcpy(out, t.data(), n);
        out[n] = 0;
    }
    return t.size();
}

void unhoist_if_guarded(zgml_hip_program* p, uint16_t buf_idx) {
    if (p->hoist_ok && buf_idx < p->hoist_guard.size() && p->hoist_guard[buf_idx]) p->hoist_ok = false, p->plan_dirty = true;
}

void* zgml_hip_program_buffer_ptr(zgml_hip_program* p, uint16_t buf_idx) {
    if (p) unhoist_if_guarded(p, buf_idx);
    return (p && buf_idx < p->bufs.size()) ? p->bufs[buf_idx] : nullptr;
}

void* zgml_hip_stream(zgml_hip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

void zgml_hip_enqueue_program(zgml_hip_ctx* ctx, zgml_hip_program* p) {
    if (!ctx || !p) return;
    hipSetDevice(ctx->device);
    enqueue(p);
}

int zgml_hip_copy_program_buffer(zgml_hip_ctx* ctx, zgml_hip_program* dst, uint16_t dst_buf, uint64_t dst_offset,
                                 zgml_hip_program* src, uint16_t src_buf, uint64_t src_offset, uint64_t n_elems) {
    if (!ctx || !dst || !src) return -1;
    if (dst_buf >= dst->bufs.size() || src_buf >= src->bufs.size() || !dst->bufs[dst_buf] || !src->bufs[src_buf]) {
        ctx->fail("copy_program_buffer: no such live buffer");
        return -1;
    }
    if (dst_offset + n_elems > dst->sizes[dst_buf] || src_offset + n_elems > src->sizes[src_buf]) {
        ctx->fail("copy_program_buffer: range exceeds a buffer");
        return -1;
    }
    unhoist_if_guarded(dst, dst_buf);
    hipSetDevice(ctx->device);
    return CTX_CHECK(ctx, hipMemcpyAsync(dst->bufs[dst_buf] + dst_offset, src->bufs[src_buf] + src_offset, n_elems * sizeof(float),
                                         hipMemcpyDeviceToDevice, ctx->stream))
               ? 0
               : -1;
}

void zgml_hip_enqueue_ops(zgml_hip_ctx* ctx, zgml_hip_program* p, uint64_t first, uint64_t count) {
    if (!ctx || !p) return;
    hipSetDevice(ctx->device);
    if (!p->ksplit_off && !(first == 0 && count >= p->ops.size())) // an op range must leave every buffer written: no deferred vectors in this program's plans
        p->ksplit_off = true, p->plan_dirty = p->plan_dirty || p->has_deferred;
    if (p->plan_dirty || p->fuse_epoch != p->ctx->fuse_epoch) {
        free_graph(p);
        build_plan(p);
    }
    flush_dyn(p);
    // a launch belongs to the range when every op it covers does; barriers (set_barriers) make
    // sure batching never straddles the harness's collective points
    for (auto& L : p->plan) {
        if (L.op_lo >= first && L.op_hi < first + count) {
            L.run(ctx->stream);
        } else if (!(L.op_hi < first || L.op_lo >= first + count)) {
            ctx->fail("enqueue_ops: range cuts through a batched launch; declare it with zgml_hip_program_set_barriers");
            return;
        }
    }
}

int zgml_hip_program_set_barriers(zgml_hip_ctx* ctx, zgml_hip_program* p, const uint64_t* op_indices, uint64_t n) {
    if (!ctx || !p) return -1;
    p->barriers.assign(op_indices, op_indices + n);
    std::sort(p->barriers.begin(), p->barriers.end());
    p->plan_dirty = true;
    return 0;
}

void zgml_hip_synchronize(zgml_hip_ctx* ctx) {
    if (!ctx) return;
    if (CTX_CHECK(ctx, hipStreamSynchronize(ctx->stream))) ctx->handoff_ok("synchronize");
}

int64_t zgml_hip_argmax(zgml_hip_ctx* ctx, zgml_hip_program* p, uint16_t buf_idx, uint64_t offset, uint64_t n) {
    if (!ctx || !p || buf_idx >= p->bufs.size() || !p->bufs[buf_idx] || offset + n > p->sizes[buf_idx]) return -1;
    hipSetDevice(ctx->device);
    launch_argmax(ctx->stream, p->bufs[buf_idx] + offset, n, ctx->arg_val, ctx->arg_idx, ctx->arg_out);
    hipMemcpyAsync(ctx->arg_out_host, ctx->arg_out, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (!CTX_CHECK(ctx, hipStreamSynchronize(ctx->stream)) || !ctx->handoff_ok("argmax")) return -1;
    return *ctx->arg_out_host;
}


int zgml_hip_resident_setup(zgml_hip_ctx* ctx, zgml_hip_program* p, const zgml_resident_llama* d) {
    if (!ctx || !p || !d) return -1;
    hipSetDevice(ctx->device);
    free_resident(p);
    auto live = [&](uint16_t b) { return b < p->bufs.size() && p->bufs[b]; };
    if (!live(d->buf_token_input) || !live(d->buf_attn_mask) || !live(d->buf_logits) || !d->d_model ||
        p->sizes[d->buf_token_input] < d->d_model || p->sizes[d->buf_attn_mask] < d->max_seq ||
        p->sizes[d->buf_logits] < d->vocab) {
        ctx->fail("resident_setup: bad buffer ids");
        return -1;
    }
    // token_len of the plan: token_input holds T embedding rows (T = 1: decode plan, T > 1: prefill chunk)
    const uint32_t T = (uint32_t)(p->sizes[d->buf_token_input] / d->d_model);
    if (T > 1 && (p->sizes[d->buf_attn_mask] < (uint64_t)T * d->max_seq || p->sizes[d->buf_logits] < (uint64_t)T * d->vocab)) {
        ctx->fail("resident_setup: attn_mask / logits smaller than token_len columns / rows");
        return -1;
    }
    Resident* r = new Resident();
    p->resident = r;
    r->token_len = T;
    r->vocab = d->vocab, r->d = d->d_model, r->max_seq = d->max_seq, r->dh = d->d_head, r->n_rope = d->n_rope;
    r->tok_in = p->bufs[d->buf_token_input], r->mask = p->bufs[d->buf_attn_mask], r->logits = p->bufs[d->buf_logits];
    const size_t n_ops = p->ops.size();
    std::vector<float*> ropes(d->n_rope);
    for (uint32_t l = 0; l < d->n_rope; l++) {
        if (!live(d->buf_rope[l]) || p->sizes[d->buf_rope[l]] < (uint64_t)T * 2 * d->d_head) {
            ctx->fail("resident_setup: bad rope buffer");
            return -1;
        }
        ropes[l] = p->bufs[d->buf_rope[l]];
    }
    std::vector<uint32_t> kind(n_ops, 0), base(n_ops, 0), stride(n_ops, 0);
    for (size_t i = 0; i < n_ops; i++) {
        if (p->ops[i].kind == ZGML_DOP_SLICE_ASSIGN && p->ops[i].u.slice_assign.patch_stride) {
            kind[i] = 1, base[i] = p->ops[i].u.slice_assign.dst_base_offset, stride[i] = p->ops[i].u.slice_assign.patch_stride;
        } else if (p->ops[i].kind == ZGML_DOP_ATTENTION || p->ops[i].kind == ZGML_DOP_ATTENTION_KVQ) {
            kind[i] = 2;
        } else if (p->ops[i].kind == ZGML_DOP_KVQ_STORE && p->ops[i].u.kvq_store.patch_stride) { // col = col_base + pos * stride
            kind[i] = 1, base[i] = p->ops[i].u.kvq_store.col_base, stride[i] = p->ops[i].u.kvq_store.patch_stride;
        }
    }
    const size_t tab = (size_t)d->max_seq * d->d_head * 4, emb = (size_t)d->vocab * d->d_model * 4;
    bool ok = CTX_CHECK(ctx, hipMalloc((void**)&r->embed, emb)) && CTX_CHECK(ctx, hipMalloc((void**)&r->cos, tab)) &&
              CTX_CHECK(ctx, hipMalloc((void**)&r->sin, tab)) &&
              CTX_CHECK(ctx, hipMalloc((void**)&r->rope_bufs, (ropes.size() + 1) * sizeof(float*))) &&
              CTX_CHECK(ctx, hipMalloc((void**)&r->dyn_kind, (n_ops + 1) * 4)) &&
              CTX_CHECK(ctx, hipMalloc((void**)&r->dyn_base, (n_ops + 1) * 4)) &&
              CTX_CHECK(ctx, hipMalloc((void**)&r->dyn_stride, (n_ops + 1) * 4)) &&
              CTX_CHECK(ctx, hipMalloc((void**)&r->state, 4 * 4)) && CTX_CHECK(ctx, hipMalloc((void**)&r->tok_dev, (size_t)T * 4)) &&
              CTX_CHECK(ctx, h2d_sync(ctx->stream, r->embed, d->token_embed, emb)) &&
              CTX_CHECK(ctx, h2d_sync(ctx->stream, r->cos, d->cos_table, tab)) &&
              CTX_CHECK(ctx, h2d_sync(ctx->stream, r->sin, d->sin_table, tab)) &&
              CTX_CHECK(ctx, h2d_sync(ctx->stream, r->rope_bufs, ropes.data(), ropes.size() * sizeof(float*))) &&
              CTX_CHECK(ctx, h2d_sync(ctx->stream, r->dyn_kind, kind.data(), n_ops * 4)) &&
              CTX_CHECK(ctx, h2d_sync(ctx->stream, r->dyn_base, base.data(), n_ops * 4)) &&
              CTX_CHECK(ctx, h2d_sync(ctx->stream, r->dyn_stride, stride.data(), n_ops * 4));
    if (!ok) {
        free_resident(p);
        return -1;
    }
    // The resident kernels write token_input, attn_mask and the rope buffers from the device, outside the op list: to the hoist
    // analysis those buffers look like constants (never written by an op), so a `repeat` that broadcasts one of them would have
    // been run once at plan build and its stale copy read ever after. Today's LLaMA stream only repeats gamma and the
    // scalar 1, but the guard must not depend on that: every buffer handed over here leaves the hoisted set.
    unhoist_if_guarded(p, d->buf_token_input);
    unhoist_if_guarded(p, d->buf_attn_mask);
    for (uint32_t l = 0; l < d->n_rope; l++) unhoist_if_guarded(p, d->buf_rope[l]);
    return 0;
}

int zgml_hip_resident_decode(zgml_hip_ctx* ctx, zgml_hip_program* p, uint32_t first_token, uint32_t start_pos,
                             uint32_t n_steps, int64_t* tokens_out) {
    if (!ctx || !p || !p->resident || !tokens_out) return -1;
    Resident* r = p->resident;
    if (r->token_len != 1) {
        ctx->fail("resident_decode: the program is a token_len > 1 plan (use zgml_hip_resident_prefill)");
        return -1;
    }
    if (first_token >= r->vocab || (uint64_t)start_pos + n_steps > r->max_seq) {
        ctx->fail("resident_decode: token or position out of range");
        return -1;
    }
    hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    if (p->plan_dirty || p->fuse_epoch != p->ctx->fuse_epoch) {
        free_graph(p);
        build_plan(p);
    }
    if (p->plan_batched && n_steps) {
        // the device patches slice_assign offsets / seq_kv itself (resident_prep_kernel): check the first and
        // last position of this call against the bounds the batched plan assumed, as refresh_program does
        bool ok = true;
        std::vector<zgml_device_op> probe = p->ops;
        for (uint32_t pos : {start_pos, start_pos + n_steps - 1}) {
            for (auto& o : probe) {
                if (o.kind == ZGML_DOP_SLICE_ASSIGN && o.u.slice_assign.patch_stride)
                    o.u.slice_assign.dst_offset = o.u.slice_assign.dst_base_offset + pos * o.u.slice_assign.patch_stride;
                if (o.kind == ZGML_DOP_ATTENTION) o.u.attention.seq_kv = pos + 1;
                if (o.kind == ZGML_DOP_ATTENTION_KVQ) o.u.attention_kvq.seq_kv = pos + 1;
            }
            ok = ok && dynamic_fields_in_bounds(p->sched, probe);
        }
        if (!ok) { // fall back to program order for good, like refresh_program
            p->batching_safe = false;
            free_graph(p);
            build_plan(p);
        }
    }
    if (r->tokens_cap < n_steps) {
        hipStreamSynchronize(s);
        hipFree(r->tokens);
        r->tokens = nullptr;
        if (!CTX_CHECK(ctx, hipMalloc((void**)&r->tokens, (size_t)n_steps * 8))) return -1;
        r->tokens_cap = n_steps;
        free_resident_graph(p); // the graph baked the old pointer/cap
    }
    ResidentPrepArgs a{r->embed, r->cos, r->sin, r->tok_in, r->mask, r->rope_bufs, r->dyn_kind, r->dyn_base, r->dyn_stride,
                       p->dyn_dev, r->state, r->state /* the token is state[0] */, r->d, r->max_seq, r->dh, r->n_rope, (uint32_t)p->ops.size(), 1};
    const uint32_t total = r->d + r->max_seq + r->n_rope * 2 * r->dh + (uint32_t)p->ops.size();
    // (off unless ZGML_HIP_TAIL_FUSED=1. First form — the last arriver walks all 8-16 K elements of the next token's patches alone —
    // measured SLOWER: SmolLM-135M 1756 against 1773 tok/s, Llama-2-7B 790 against 817. Second form — every workgroup writes the
    // position-only patches, the last arriver the embedding row — a wash: 1777-1782 against 1768-1780, 838 against 842;
    // profiles/r05_token_tail_ab.txt)
    static const bool tail_fused = getenv("ZGML_HIP_TAIL_FUSED") && atoi(getenv("ZGML_HIP_TAIL_FUSED")) != 0;
    auto one_token = [&](hipStream_t st) {
        if (!tail_fused) launch_resident_prep(st, a, total);
        // [prep] [plan] [argmax stage 1] [stage 2 + advance]; or, opt-in, [plan] [token tail]: the argmax of the logits, the advance
        // of the device state AND the next token's patches in ONE launch (launch_argmax_tail; the first token's patches then
        // come from a stand-alone prep launch in front of the loop) — two launches fewer per token.
        run_plan(p, st, 0, p->plan.size());
        if (tail_fused)
            launch_argmax_tail(st, r->logits, r->vocab, ctx->arg_val, ctx->arg_idx, ctx->arg_cnt, ctx->arg_out, ArgmaxAdvance{r->state, r->tokens, r->tokens_cap}, &a, total);
        else
            launch_argmax(st, r->logits, r->vocab, ctx->arg_val, ctx->arg_idx, ctx->arg_out, ArgmaxAdvance{r->state, r->tokens, r->tokens_cap});
    };
    // static dyn words (row stores with patch_stride 0) come from the host mirror; the prep kernel
    // only rewrites the position-dependent ones
    set_dyn_from_ops(p);
    p->dyn_dirty = true;
    flush_dyn(p);
    const uint32_t st0[4] = {first_token, start_pos, 0, 0};
    if (!CTX_CHECK(ctx, hipMemcpyAsync(r->state, st0, sizeof(st0), hipMemcpyHostToDevice, s))) return -1;
    if (tail_fused) launch_resident_prep(s, a, total); // the first token's patches (every later token's come from its predecessor's tail)
    if (ctx->opt_graph && !r->graph_exec) {
        hipGraph_t g = nullptr;
        if (CTX_CHECK(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal))) {
            one_token(s);
            if (CTX_CHECK(ctx, hipStreamEndCapture(s, &g)) && g) {
                dump_graph(g, "resident");
                if (CTX_CHECK(ctx, hipGraphInstantiate(&r->graph_exec, g, nullptr, nullptr, 0)))
                    r->graph = g;
                else
                    hipGraphDestroy(g);
            }
        }
    }
    // several tokens per graph launch: everything a token needs is produced on the device from the state words, so a graph may
    // simply hold the launches of G consecutive tokens (experiment: is there a per-graph gap on the device?)
    static const uint32_t per_graph = getenv("ZGML_HIP_RESIDENT_TOKENS_PER_GRAPH") ? (uint32_t)atoi(getenv("ZGML_HIP_RESIDENT_TOKENS_PER_GRAPH")) : 1u;
    if (ctx->opt_graph && r->graph_exec && per_graph > 1 && n_steps >= per_graph && !r->graph_multi_exec) {
        hipGraph_t g = nullptr;
        if (CTX_CHECK(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal))) {
            for (uint32_t t = 0; t < per_graph; t++) one_token(s);
            if (CTX_CHECK(ctx, hipStreamEndCapture(s, &g)) && g) {
                if (CTX_CHECK(ctx, hipGraphInstantiate(&r->graph_multi_exec, g, nullptr, nullptr, 0)))
                    r->graph_multi = g, r->multi_n = per_graph;
                else
                    hipGraphDestroy(g);
            }
        }
    }
    for (uint32_t i = 0; i < n_steps;) {
        if (r->graph_multi_exec && n_steps - i >= r->multi_n) {
            hipGraphLaunch(r->graph_multi_exec, s);
            i += r->multi_n;
        } else if (r->graph_exec) {
            hipGraphLaunch(r->graph_exec, s);
            i++;
        } else {
            one_token(s);
            i++;
        }
    }
    hipMemcpyAsync(tokens_out, r->tokens, (size_t)n_steps * 8, hipMemcpyDeviceToHost, s);
    bool ok = CTX_CHECK(ctx, hipStreamSynchronize(s));
    ok = ok && ctx->handoff_ok("resident_decode");
    // the device rewrote the dyn block behind the host mirror's back: force a re-upload next time
    memset(p->dyn_host, 0xFF, p->ops.size() * sizeof(uint32_t));
    set_dyn_from_ops(p);
    p->dyn_dirty = true;
    p->profile.call_count += n_steps;
    p->profile.backend_op_count += (uint64_t)n_steps * p->ops.size();
    p->profile.backend_dispatch_count += (uint64_t)n_steps * (p->plan.size() + (tail_fused ? 1 : 3));
    return ok ? 0 : -1;
}


// One execution of a token_len = T plan (a prefill chunk) with everything but the T token ids produced on the device.
int64_t zgml_hip_resident_prefill(zgml_hip_ctx* ctx, zgml_hip_program* p, const uint32_t* tokens, uint32_t n_tokens, uint32_t start_pos) {
    if (!ctx || !p || !p->resident || !tokens) return -1;
    Resident* r = p->resident;
    if (n_tokens != r->token_len || (uint64_t)start_pos + n_tokens > r->max_seq) {
        ctx->fail("resident_prefill: n_tokens must equal the plan's token_len and the chunk must fit max_seq");
        return -1;
    }
    for (uint32_t j = 0; j < n_tokens; j++)
        if (tokens[j] >= r->vocab) {
            ctx->fail("resident_prefill: token out of range");
            return -1;
        }
    hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    if (p->plan_dirty || p->fuse_epoch != p->ctx->fuse_epoch) {
        free_graph(p);
        build_plan(p);
    }
    if (p->plan_batched) { // the device patches offsets / seq_kv itself: same bound check as refresh_program
        std::vector<zgml_device_op> probe = p->ops;
        for (auto& o : probe) {
            if (o.kind == ZGML_DOP_SLICE_ASSIGN && o.u.slice_assign.patch_stride)
                o.u.slice_assign.dst_offset = o.u.slice_assign.dst_base_offset + start_pos * o.u.slice_assign.patch_stride;
            if (o.kind == ZGML_DOP_ATTENTION) o.u.attention.seq_kv = start_pos + n_tokens;
            if (o.kind == ZGML_DOP_ATTENTION_KVQ) o.u.attention_kvq.seq_kv = start_pos + n_tokens;
        }
        if (!dynamic_fields_in_bounds(p->sched, probe)) {
            p->batching_safe = false;
            free_graph(p);
            build_plan(p);
        }
    }
    set_dyn_from_ops(p); // static dyn words from the host mirror; the prep kernel rewrites the position-dependent ones
    p->dyn_dirty = true;
    flush_dyn(p);
    const uint32_t st0[4] = {0, start_pos, 0, 0};
    if (!CTX_CHECK(ctx, hipMemcpyAsync(r->state, st0, sizeof(st0), hipMemcpyHostToDevice, s)) ||
        !CTX_CHECK(ctx, hipMemcpyAsync(r->tok_dev, tokens, (size_t)n_tokens * 4, hipMemcpyHostToDevice, s)))
        return -1;
    ResidentPrepArgs a{r->embed, r->cos, r->sin, r->tok_in, r->mask, r->rope_bufs, r->dyn_kind, r->dyn_base, r->dyn_stride,
                       p->dyn_dev, r->state, r->tok_dev, r->d, r->max_seq, r->dh, r->n_rope, (uint32_t)p->ops.size(), n_tokens};
    const uint64_t total = (uint64_t)n_tokens * r->d + (uint64_t)n_tokens * r->max_seq + (uint64_t)r->n_rope * n_tokens * 2 * r->dh + p->ops.size();
    launch_resident_prep(s, a, (uint32_t)total);
    p->dyn_dirty = false;
    enqueue(p); // the plan (graph replay when enabled); flush_dyn is a no-op: the device words are current
    // the logits buffer holds one row per token: the greedy token comes from the LAST position's row
    launch_argmax(s, r->logits + (uint64_t)(n_tokens - 1) * r->vocab, r->vocab, ctx->arg_val, ctx->arg_idx, ctx->arg_out);
    hipMemcpyAsync(ctx->arg_out_host, ctx->arg_out, sizeof(int64_t), hipMemcpyDeviceToHost, s);
    const bool ok = CTX_CHECK(ctx, hipStreamSynchronize(s)) && ctx->handoff_ok("resident_prefill");
    // the device rewrote the dyn block behind the host mirror's back: force a re-upload next time
    memset(p->dyn_host, 0xFF, p->ops.size() * sizeof(uint32_t));
    set_dyn_from_ops(p);
    p->dyn_dirty = true;
    p->profile.call_count++;
    p->profile.backend_op_count += p->ops.size();
    p->profile.backend_dispatch_count += p->plan.size() + 3;
    return ok ? *ctx->arg_out_host : -1;
}

} // extern "C"

namespace zgml_rt { // what runtime_bench.hip / runtime_shard.hip call (runtime_internal.h)
void rt_build_plan(zgml_hip_program* p) { build_plan(p); }
void rt_free_graph(zgml_hip_program* p) { free_graph(p); }
void rt_run_plan(zgml_hip_program* p, hipStream_t s, size_t first, size_t count) { run_plan(p, s, first, count); }
bool rt_grow(zgml_hip_ctx* ctx, float** ptr, uint64_t* cap, uint64_t elems) { return grow(ctx, ptr, cap, elems); }
uint64_t rt_now_ns() { return now_ns(); }
void rt_dump_graph(hipGraph_t g, const char* tag) { dump_graph(g, tag); }
} // namespace zgml_rt
