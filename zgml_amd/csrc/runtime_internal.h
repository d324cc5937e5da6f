// runtime_internal.h — what the translation units of the runtime share (runtime.hip: the C ABI, planner and resident loops;
// runtime_bench.hip: the measurement entry points; runtime_shard.hip: the row-shard path): the context and program objects and the
// few internal functions the latter two call. Not part of the boundary (include/zgml_hip.h is).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "kernels.h"
#include "schedule.h"

using namespace zgml;

namespace zgml_rt {

extern std::string g_create_error; // (defined in runtime.hip)

// A launch that can also emit its result rows as the A pieces of an M > 1 quantized matmul (kernels.h: RowChainParams::ap):
// the planner arms it when the matmul that consumes exactly these rows follows with no other splitting launch in between.
struct SplitHook {
    const float* out = nullptr; // the rows the launch produces (dense: row stride == cols)
    uint32_t rows = 0, cols = 0; // rows == 0: a flat launch over `n` elements, any rows x cols = n of dense rows
    uint32_t n = 0;
    uint16_t** ap = nullptr;    // fields of the launch's (shared) parameter block
    uint32_t* ap_S = nullptr;
    uint32_t* ap_cols = nullptr; // flat launches: the row length is filled in too
};

struct AdecDesc {
    std::vector<AttnDecodeParams> host; // the records as uploaded
    const AttnDecodeParams* dev = nullptr;
    uint32_t nh = 0, dh = 0;
    AttnSplit sp;
    bool kvq = false;
};

struct Launch {
    uint32_t kind;           // DeviceOp tag the launch is accounted to
    uint32_t n_ops;          // DeviceOps covered (batching folds several)
    uint32_t op_lo, op_hi;   // smallest / largest op index covered
    std::function<void(hipStream_t)> run;
    std::shared_ptr<SplitHook> hook; // (after `run` so the aggregate initialisers elsewhere stay valid)
    // what a fused-launch pass needs to know about a grouped mat-vec launch / a decode-attention launch (fuse_qkv_attention)
    std::shared_ptr<QmvLaunch> qmv_desc;
    std::shared_ptr<struct AdecDesc> adec_desc;
    std::shared_ptr<RowChainParams> rc_desc; // a row-chain launch ([add ->] rmsnorm [-> mul]) and its row count (fuse_ksplit)
    uint32_t rc_rows = 0;
    const char* tag = nullptr; // plan text: what a fused launch is
    uint64_t prof_ns = 0;    // ZGML_HIP_OPT_PROFILE: accumulated event time of this launch
    uint32_t prof_calls = 0;
};

struct IoEntry {
    uint16_t buf_idx;
    uint32_t offset, size;
    bool operator==(const IoEntry& o) const { return buf_idx == o.buf_idx && offset == o.offset && size == o.size; }
};

struct IoTableDev { // one row per transfer, consumed by scatter/gather kernels
    float* dev;      // device address inside the program buffer
    uint32_t stage_off_words;
    uint32_t n_words;
};

struct IoPlan {
    std::vector<IoEntry> entries;
    IoTableDev* table_dev = nullptr;
    uint32_t total_words = 0, max_row_words = 0;
    bool word_aligned = true;
    bool dyn_row = false; // (input plan) the table's last row scatters the program's dynamic words (prepare_io)
};

} // namespace zgml_rt
using namespace zgml_rt;

struct ShardState;
struct zgml_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool opt_fusion = true, opt_graph = true, opt_profile = false, opt_skip_dead = true, opt_f16_dense = false;
    // ZGML_HIP_HOST_PROF=1: host time of the drop-in path's phases — [0] refresh, [1] upload (pack + H2D + scatter), [2] enqueue
    // (dynamic words + graph launch), [3] download (gather + D2H + wait + unpack) — printed by zgml_hip_destroy
    bool host_prof = false;
    uint64_t prof_ns[4] = {0, 0, 0, 0}, prof_calls[4] = {0, 0, 0, 0};
    bool opt_ksplit = false; // ZGML_HIP_OPT_KSPLIT (zgml_hip_create takes the default from ZGML_HIP_KSPLIT)
    bool opt_w8a8 = false;   // ZGML_HIP_OPT_W8A8: M = 1 qmatmuls through the reference's W8A8 arm (w8a8.hip); read at compile_program
    int64_t opt_attn_split_min_keys = -1; // -1: environment / default (attn_split_for)
    int64_t opt_fuse_resident_wgs = -1;   // -1: one 1024-thread workgroup per CU (fuse_qkv_attention)
    // host dense override scratch
    float *mm_a = nullptr, *mm_b = nullptr, *mm_c = nullptr;
    uint64_t mm_a_cap = 0, mm_b_cap = 0, mm_c_cap = 0;
    // host dense override: device copies of B operands keyed by host pointer; opt-in
    struct CachedB {
        float* dev;
        uint64_t span; // elements
    };
    std::map<const float*, CachedB> b_cache;
    uint64_t b_cache_cap = 0, b_cache_bytes = 0, b_cache_hits = 0, b_cache_misses = 0;
    void drop_b_cache() {
        for (auto& kv : b_cache) hipFree(kv.second.dev);
        b_cache.clear();
        b_cache_bytes = 0;
    }
    // argmax scratch
    float* arg_val = nullptr;
    int64_t* arg_idx = nullptr;
    uint32_t* arg_cnt = nullptr; // arrival counter of the fused token tail (launch_argmax_tail), re-armed by its last workgroup
    int64_t* arg_out = nullptr;
    int64_t* arg_out_host = nullptr; // pinned
    struct ShardState* shard = nullptr; // RCCL communicator of the row-shard path (zgml_hip_shard_*), else nullptr
    // Fused launches (q/k/v projection + decode attention): ONE host-visible word every bounded in-launch wait sets when it
    // gives up (pinned, device-mapped: the host reads it after any synchronisation without a copy). A set word means the
    // tokens of that run are wrong: every host sync point reports it (handoff_ok), clears it and switches the fusion off for
    // the context — plans are rebuilt in the two-launch form (fuse_epoch).
    uint32_t* handoff_flag = nullptr;     // host pointer
    uint32_t* handoff_flag_dev = nullptr; // the same word as the kernels see it
    bool fuse_qkv_off = false;
    uint64_t fuse_epoch = 0;
    int n_cu = 0; // compute units (residency guard of the fused launch)
    std::map<uint64_t, int> census; // (d_head, kvq, grid) -> qkv_attn_kon_census result: measured co-residency of the fused K-on-lanes launch
    bool handoff_ok(const char* where) {
        if (!handoff_flag || !*(volatile uint32_t*)handoff_flag) return true;
        const uint32_t cause = *(volatile uint32_t*)handoff_flag; // 1: a fused launch's in-launch hand-off; 2: a peer gather's wait (shard_peer.hip kHandoffPeer)
        *(volatile uint32_t*)handoff_flag = 0;
        if (cause == 2) { // a slow or failed PEER says nothing about the fusion: leave it on, no plan rebuild
            fail(std::string(where) + ": a peer gather's wait at a gather point timed out (or a peer had already failed the step) — the results of "
                                      "this step are wrong on every rank; all ranks must re-attach (zgml_hip_shard_attach + peer export / import) before stepping again");
            return false;
        }
        fuse_qkv_off = true, fuse_epoch++;
        fail(std::string(where) + ": an in-launch hand-off wait of a fused q/k/v + attention launch timed out — the results of this run are wrong; "
                                  "the fusion is now off for this context (later runs use two launches)");
        return false;
    }

    void fail(const std::string& what) {
        if (err.empty()) err = what;
    }
    bool check(hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        fail(std::string(what) + ": " + hipGetErrorString(e));
        return false;
    }
};

struct zgml_resident;
struct zgml_hip_program {
    zgml_hip_ctx* ctx = nullptr;
    std::vector<zgml_device_op> ops;
    std::vector<uint32_t> dyn_ops;                    // indices of the ops with a dynamic field (own_ops)
    std::vector<std::vector<zgml_fused_step>> steps; // owned copies, per op
    std::vector<uint64_t> sizes;                      // f32 elements
    std::vector<float*> bufs;                         // device pointers (nullptr = elided)
    void* arena = nullptr;
    std::vector<QWeightDev> qweights;
    std::vector<unsigned long long*> attn_traces; // diagnostics (ZGML_HIP_ATTN_TRACE)
    struct QmvTrace {
        unsigned long long* t;
        uint32_t parts, pro, K, N;
    };
    std::vector<QmvTrace> qmv_traces; // diagnostics (ZGML_HIP_QMV_TRACE)
    struct KsTrace {
        unsigned long long* t;
        const char* what;
    };
    std::vector<KsTrace> ks_traces;   // diagnostics (ZGML_HIP_KS_TRACE): stamps of the K-split launches' workgroup 0
    float* zero_word = nullptr;                   // a device 0.0f: mask operand of unmasked decode attention
    float* split_buf = nullptr;                   // long-context attention split: partials + arrival counters,
    uint32_t* split_cnt = nullptr;                // shared by the (stream-ordered) decode-attention launches
    uint64_t split_buf_floats = 0, split_cnt_words = 0;
    bool f16_stream_nt = false;     // promoted weights exceed the Infinity Cache: non-temporal loads
    std::vector<void*> f16_weights; // per buffer: MFMA-packed f16 copy of a promoted matmul B operand (else nullptr)
    std::vector<void*> owned; // other device allocations
    uint64_t fuse_epoch = 0;             // ctx->fuse_epoch the plan was built at (a time-out rebuilds it without the fusion)
    std::vector<void*> fuse_owned;       // counters / seen / idx blocks of the fused launches: freed with every plan rebuild
    // repeats of constant data (a weight broadcast to the activation shape: source never written by an op, destination written by
    // this op only) run ONCE when the plan is built instead of in every execution; a host input that ever targets one of the
    // buffers involved switches this off for the program (prepare_io)
    std::vector<char> hoist_op;    // per op: executed at plan-build time, not part of the plan
    std::vector<char> hoist_guard; // per buffer: read or written by a hoisted repeat
    bool hoist_ok = true;
    bool ksplit = false;                 // ZGML_HIP_OPT_KSPLIT as it stood at compile_program
    bool ksplit_off = false;             // fuse_ksplit: off for this program (a caller enqueues op ranges: a deferred vector must not outlive a range)
    bool has_deferred = false;           // the plan holds launches that leave a vector deferred to their successor
    float* prenorm_buf = nullptr;        // arm_prenorm: [x * gamma | partial sums of squares] handed from a residual epilogue to the next prologue
    size_t prenorm_bytes = 0;
    float* scratch = nullptr;
    uint64_t scratch_bytes = 0;
    // plan building: the last quantized matmul launch that split its input into the scratch (make_single)
    uint64_t split_pos = UINT64_MAX - 1;
    const float* split_input = nullptr;
    std::shared_ptr<std::vector<std::pair<QWeightDev, QMatmulParams>>> qmm_group; // parts of the launch at split_pos
    std::shared_ptr<std::vector<DenseF16Params>> f16_group;
    uint32_t split_M = 0, split_K = 0, split_in_rs = 0, split_kind = 0; // kind: 1 = bf16 pieces (quantized), 2 = f16 A (dense)
    // dynamic parameter block: one word per op
    uint32_t* dyn_dev = nullptr;
    uint32_t* dyn_host = nullptr; // pinned
    bool dyn_dirty = true;
    std::vector<Launch> plan;
    bool plan_dirty = true;
    Schedule sched;                  // valid when plan_batched
    bool plan_batched = false;       // plan was built from dependency levels
    bool batching_safe = true;       // cleared when a refresh leaves the assumed dynamic bounds
    std::vector<uint64_t> barriers;  // op indices nothing may be moved across (multi-GPU harness)
    // per attention op: the largest seq_kv the caller ever handed over (compile time or a refresh). The schedule
    // takes the attention's K/V read extent from here, never from the last refreshed value, so a plan rebuilt
    // after a small-seq_kv refresh cannot put a KV store and the attention that reads it on one level.
    std::vector<uint32_t> seq_kv_bound;
    std::vector<void*> param_blobs;  // device parameter arrays of batched launches
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    // ... in two pieces: hipGraphLaunch costs the host ~0.36 us per kernel node BEFORE the device sees the first one (45-50 us for a
    // 124-launch SmolLM token, all of it on the critical path of a blocking execute_program); the head piece is short, so the device
    // starts after ~10 us and runs it while the host submits the tail
    hipGraph_t graph_tail = nullptr;
    hipGraphExec_t graph_tail_exec = nullptr;
    // host I/O staging
    void* stage_host = nullptr; // pinned
    void* stage_dev = nullptr;
    uint64_t stage_cap = 0;
    // execute_program's own graph (round 5): [scatter the packed inputs + dynamic words straight from the pinned staging buffer]
    // [the plan] [gather the outputs into a second pinned buffer] — no copy commands, one hipGraphLaunch per token. The kernels
    // read / write host memory through its device mapping; captured for one (input table, output table) pair, dropped with the
    // plan's graphs, with either table and with the staging buffers.
    void* stage_out_host = nullptr; // pinned, outputs
    hipGraph_t io_graph = nullptr;
    hipGraphExec_t io_graph_exec = nullptr;
    uint32_t io_in_rows = 0, io_out_rows = 0;
    // zgml_hip_program_pin_outputs (opt-in: the caller promises its output buffer stays allocated): a single output handed over at
    // the same address call after call (DeviceInference's logits slice) is registered with the driver after three such calls and
    // the gather kernel then writes INTO it — no staging copy, no unpack (197 KB per SmolLM token). A different address later:
    // unregistered, staging from then on.
    bool pin_allowed = false;
    void* pin_host = nullptr;
    void* pin_dev = nullptr;
    uint64_t pin_size = 0;
    const void* pin_cand = nullptr;
    uint32_t pin_seen = 0;
    bool pin_off = false;
    void* io_out_dev = nullptr; // the address the captured gather kernel writes to
    IoPlan in_plan, out_plan;
    uint64_t staged_bytes = 0, staged_n = 0; // zgml_hip_stage_inputs / enqueue_staged
    zgml_runtime_profile profile{};
    zgml_resident* resident = nullptr;
    // row-shard path (zgml_hip_shard_attach): all-gather points in op order, the logits for the greedy token, and the
    // captured graph of one whole step (staged inputs, op ranges, all-gathers, argmax)
    std::vector<zgml_shard_point> shard_points;
    uint16_t shard_logits_buf = 0;
    uint64_t shard_vocab = 0;
    hipGraph_t shard_graph = nullptr;
    hipGraphExec_t shard_graph_exec = nullptr;
    bool shard_capture_failed = false;
    std::vector<double> shard_point_us; // per gather point: microseconds in the last zgml_hip_shard_profile_step
    // the greedy token as a (max, index) pair per rank instead of a gather of the logits (the last gather point covers the logits
    // buffer): device pairs [world] for the collective mode; and the peer-store gather's state (shard_peer.hip)
    bool shard_pair_argmax = false;
    unsigned long long* shard_pairs = nullptr;
    struct ShardPeer* shard_peer = nullptr;
};

#define CTX_CHECK(ctx, expr) (ctx)->check((expr), #expr)

// Host -> device copies and fills of set-up data, ordered on the CONTEXT's stream and waited for there — never on the legacy
// stream: a synchronous hipMemcpy / hipMemset synchronises with every blocking stream of the process, so with two contexts stepping
// from two host threads (the single-GPU rehearsal of world size 2) it invalidated the other thread's stream capture ("operation
// would make the legacy stream depend on a capturing blocking stream") or waited for the other rank's gather kernel, which in
// turn waits for this rank (a 5 s stall ended by the bounded wait).
namespace {
inline hipError_t h2d_sync(hipStream_t s, void* dst, const void* src, size_t bytes) {
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}
inline hipError_t memset_sync(hipStream_t s, void* dst, int value, size_t bytes) {
    const hipError_t e = hipMemsetAsync(dst, value, bytes, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}
} // namespace

namespace zgml_rt {
// defined in runtime.hip (thin exported forms of its file-local functions)
void rt_build_plan(zgml_hip_program* p);
void rt_free_graph(zgml_hip_program* p);
void rt_run_plan(zgml_hip_program* p, hipStream_t s, size_t first, size_t count);
bool rt_grow(zgml_hip_ctx* ctx, float** ptr, uint64_t* cap, uint64_t elems);
uint64_t rt_now_ns();
void rt_dump_graph(hipGraph_t g, const char* tag);
} // namespace zgml_rt
#ifndef ZGML_RUNTIME_MAIN // the other translation units call them by the names runtime.hip uses
namespace {
inline void build_plan(zgml_hip_program* p) { zgml_rt::rt_build_plan(p); }
inline void free_graph(zgml_hip_program* p) { zgml_rt::rt_free_graph(p); }
inline void run_plan(zgml_hip_program* p, hipStream_t s, size_t first, size_t count) { zgml_rt::rt_run_plan(p, s, first, count); }
inline bool grow(zgml_hip_ctx* ctx, float** ptr, uint64_t* cap, uint64_t elems) { return zgml_rt::rt_grow(ctx, ptr, cap, elems); }
inline uint64_t now_ns() { return zgml_rt::rt_now_ns(); }
inline void dump_graph(hipGraph_t g, const char* tag) { zgml_rt::rt_dump_graph(g, tag); }
} // namespace
#endif
