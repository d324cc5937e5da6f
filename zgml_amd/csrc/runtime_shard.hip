// runtime_shard.hip — the row-shard (N-split) path of include/zgml_hip.h: zgml_hip_shard_* (RCCL all-gathers or peer-store
// gathers between op ranges, one process per GPU). Split out of runtime.hip in round 4.
#include "runtime_internal.h"

// ── row-shard (N-split) path behind the C ABI: RCCL all-gathers between op ranges ─────────────────────
// One process per GPU (the caller's launcher decides ranks). librccl.so is opened at run time by zgml_hip_shard_*
// only, so single-GPU users of the library carry no dependency on it. The communicator lives in the context; a step's
// whole device side (H2D of the staged inputs, op ranges, in-place ncclAllGather of the replicated activations, argmax)
// is recorded once into a graph on the context stream and replayed per token — the messages are 2-44 KB, so the step is
// bound by launch and collective latency, not by xGMI bandwidth.
#include <dlfcn.h>
#include <unistd.h>
struct ShardState {
    void* lib = nullptr;
    void* comm = nullptr;
    int rank = 0, world = 1;
    bool peer = false; // gathers by peer stores (shard_peer.hip) instead of ncclAllGather: no RCCL library, no communicator
    int (*get_unique_id)(void*) = nullptr;
    int (*comm_init_rank)(void**, int, /* ncclUniqueId by value: 128 bytes */ struct Id128, int) = nullptr;
    int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*comm_destroy)(void*) = nullptr;
    const char* (*get_error_string)(int) = nullptr;
};
struct Id128 {
    char b[128];
};
namespace {
constexpr int kNcclFloat = 7; // ncclFloat32 (rccl.h ncclDataType_t)
bool shard_load(zgml_hip_ctx* ctx, ShardState* st) {
    if (st->lib) return true;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        st->lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (st->lib) break;
    }
    if (!st->lib) {
        if (ctx) ctx->fail(std::string("shard: cannot open librccl.so: ") + dlerror());
        return false;
    }
    st->get_unique_id = (int (*)(void*))dlsym(st->lib, "ncclGetUniqueId");
    st->comm_init_rank = (int (*)(void**, int, Id128, int))dlsym(st->lib, "ncclCommInitRank");
    st->all_gather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(st->lib, "ncclAllGather");
    st->comm_destroy = (int (*)(void*))dlsym(st->lib, "ncclCommDestroy");
    st->get_error_string = (const char* (*)(int))dlsym(st->lib, "ncclGetErrorString");
    if (!st->get_unique_id || !st->comm_init_rank || !st->all_gather || !st->comm_destroy) {
        if (ctx) ctx->fail("shard: librccl.so lacks an expected entry point");
        return false;
    }
    return true;
}
ShardState g_shard_loader; // for zgml_hip_shard_unique_id (no context yet)
} // namespace
// Peer-store gather state of one program (shard_peer.hip): ONE fine-grained block per rank —
//   [arrival counters: (n_points + 1) x 128 B][poison word: 128 B][staging of point 0: world x len_0 f32][...][pairs: world x 8 B]
// — that every peer maps (hipIpc, or the raw pointer inside one process) and writes; this rank's private `seen` words and the
// device table of the mapped blocks. The layout is a function of the gather points only, so it is the same on every rank.
struct ShardPeer {
    char* block = nullptr;
    uint64_t bytes = 0;
    std::vector<uint64_t> stage_off; // per point, bytes
    uint64_t pairs_off = 0;
    uint64_t poison_off = 0;         // one word every rank's give-up sets in every block (shard_peer.hip: peer_poison)
    uint32_t* seen = nullptr;        // [n_points + 1] device words
    std::vector<char*> mapped;       // per rank: its block as this process sees it (own = block)
    std::vector<char> ipc_opened;    // per rank: mapped through hipIpcOpenMemHandle (closed on free)
    char** table_dev = nullptr;      // device copy of `mapped`
    bool table_ready = false;
};
namespace {
void free_shard_peer(zgml_hip_program* p) {
    ShardPeer* sp = p->shard_peer;
    if (!sp) return;
    for (size_t r = 0; r < sp->mapped.size(); r++)
        if (sp->ipc_opened[r] && sp->mapped[r]) hipIpcCloseMemHandle(sp->mapped[r]);
    hipFree(sp->block);
    hipFree(sp->seen);
    hipFree(sp->table_dev);
    delete sp;
    p->shard_peer = nullptr;
}
} // namespace
extern "C" void shard_peer_release(zgml_hip_program* p) { free_shard_peer(p); }
namespace {
uint64_t peer_wait_ticks() { // 100 MHz ticks
    static const uint64_t ms = getenv("ZGML_SHARD_PEER_WAIT_MS") ? (uint64_t)std::max(1, atoi(getenv("ZGML_SHARD_PEER_WAIT_MS"))) : 5000u;
    return ms * 100000ull;
}

// the device side of one sharded step on stream order: op ranges separated by in-place all-gathers
bool shard_segments(zgml_hip_ctx* ctx, zgml_hip_program* p, std::vector<hipEvent_t>* ev = nullptr) { // ev: 2 events per gather point (profile_step)
    ShardState* st = ctx->shard;
    uint64_t prev = 0;
    size_t gi = 0;
    ShardPeer* const sp = p->shard_peer;
    if (st->peer && (!sp || !sp->table_ready)) {
        ctx->fail("shard: peer gather mode, but not every rank's block has been imported (zgml_hip_shard_peer_import)");
        return false;
    }
    const size_t n_pts = p->shard_points.size();
    for (const zgml_shard_point& gp : p->shard_points) {
        if (gp.op_end > prev) zgml_hip_enqueue_ops(ctx, p, prev, gp.op_end - prev);
        float* full = p->bufs[gp.buf_idx] + gp.offset;
        if (ev) hipEventRecord((*ev)[2 * gi], ctx->stream);
        int rc = 0;
        if (p->shard_pair_argmax && gi + 1 == n_pts) {
            // the logits: one (max, index) pair per rank instead of the vector (every op is already enqueued: this is the last point)
            if (p->ops.size() > gp.op_end) zgml_hip_enqueue_ops(ctx, p, gp.op_end, p->ops.size() - gp.op_end);
            const float* const slice = full + (uint64_t)st->rank * gp.len_per_rank;
            if (st->peer) {
                PeerArgmaxArgs a{sp->table_dev, slice, sp->seen + gi, ctx->handoff_flag_dev, ctx->arg_out, (uint64_t)gi * 128, sp->pairs_off, peer_wait_ticks(),
                                 gp.len_per_rank, (uint32_t)st->rank, (uint32_t)st->world, sp->poison_off};
                launch_peer_argmax(ctx->stream, a);
            } else {
                launch_local_argmax_pair(ctx->stream, slice, gp.len_per_rank, (uint32_t)st->rank, p->shard_pairs);
                rc = st->all_gather(p->shard_pairs + st->rank, p->shard_pairs, 2, kNcclFloat, st->comm, ctx->stream); // 8 bytes per rank
                launch_reduce_pairs(ctx->stream, p->shard_pairs, (uint32_t)st->world, ctx->arg_out);
            }
        } else if (st->peer) {
            PeerGatherArgs a{sp->table_dev, full, sp->seen + gi, ctx->handoff_flag_dev, (uint64_t)gi * 128, sp->stage_off[gi], peer_wait_ticks(), gp.len_per_rank,
                             (uint32_t)st->rank, (uint32_t)st->world, sp->poison_off};
            launch_peer_gather(ctx->stream, a);
        } else {
            rc = st->all_gather(full + (uint64_t)st->rank * gp.len_per_rank, full, gp.len_per_rank, kNcclFloat, st->comm, ctx->stream);
        }
        if (ev) hipEventRecord((*ev)[2 * gi + 1], ctx->stream);
        gi++;
        if (rc != 0) {
            ctx->fail(std::string("shard: ncclAllGather: ") + (st->get_error_string ? st->get_error_string(rc) : "error"));
            return false;
        }
        prev = gp.op_end;
    }
    if (p->ops.size() > prev && !p->shard_pair_argmax) zgml_hip_enqueue_ops(ctx, p, prev, p->ops.size() - prev);
    if (!p->shard_pair_argmax) launch_argmax(ctx->stream, p->bufs[p->shard_logits_buf], p->shard_vocab, ctx->arg_val, ctx->arg_idx, ctx->arg_out);
    return ctx->err.empty();
}
} // namespace
extern "C" {

int zgml_hip_shard_unique_id(unsigned char id_out[128]) {
    if (!id_out || !shard_load(nullptr, &g_shard_loader)) return -1;
    return g_shard_loader.get_unique_id(id_out) == 0 ? 0 : -1;
}

int zgml_hip_shard_init(zgml_hip_ctx* ctx, const unsigned char id[128], int rank, int world) {
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return -1;
    hipSetDevice(ctx->device);
    if (ctx->shard) return -1; // one communicator per context
    ShardState* st = new ShardState();
    if (!shard_load(ctx, st)) {
        delete st;
        return -1;
    }
    Id128 uid;
    memcpy(uid.b, id, 128);
    st->rank = rank, st->world = world;
    const int rc = st->comm_init_rank(&st->comm, world, uid, rank);
    if (rc != 0) {
        ctx->fail(std::string("shard: ncclCommInitRank: ") + (st->get_error_string ? st->get_error_string(rc) : "error"));
        delete st;
        return -1;
    }
    ctx->shard = st;
    return 0;
}

void zgml_hip_shard_destroy(zgml_hip_ctx* ctx) {
    if (!ctx || !ctx->shard) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->shard->comm && ctx->shard->comm_destroy) ctx->shard->comm_destroy(ctx->shard->comm);
    delete ctx->shard;
    ctx->shard = nullptr;
}

int zgml_hip_shard_attach(zgml_hip_ctx* ctx, zgml_hip_program* p, const zgml_shard_point* points, uint64_t n_points, uint16_t logits_buf,
                          uint64_t vocab) {
    if (!ctx || !p || (n_points && !points)) return -1;
    const uint32_t world = ctx->shard ? (uint32_t)ctx->shard->world : 1u;
    std::vector<uint64_t> bars;
    uint64_t prev = 0;
    for (uint64_t i = 0; i < n_points; i++) {
        const zgml_shard_point& gp = points[i];
        if (gp.op_end < prev || gp.op_end > p->ops.size() || gp.buf_idx >= p->bufs.size() || !p->bufs[gp.buf_idx] ||
            (uint64_t)gp.offset + (uint64_t)world * gp.len_per_rank > p->sizes[gp.buf_idx]) {
            ctx->fail("shard_attach: gather point " + std::to_string(i) + " out of range");
            return -1;
        }
        prev = gp.op_end;
        bars.push_back(gp.op_end);
    }
    if (logits_buf >= p->bufs.size() || !p->bufs[logits_buf] || vocab > p->sizes[logits_buf]) {
        ctx->fail("shard_attach: bad logits buffer");
        return -1;
    }
    p->shard_points.assign(points, points + n_points);
    p->shard_logits_buf = logits_buf, p->shard_vocab = vocab;
    p->shard_capture_failed = false;
    hipSetDevice(ctx->device);
    free_shard_peer(p);
    // the last gather point covers the logits: gather one (max, index) pair per rank instead
    p->shard_pair_argmax = false;
    if (n_points && ctx->shard) {
        const zgml_shard_point& last = points[n_points - 1];
        static const bool pairs_on = !(getenv("ZGML_SHARD_PAIR_ARGMAX") && atoi(getenv("ZGML_SHARD_PAIR_ARGMAX")) == 0);
        p->shard_pair_argmax = pairs_on && last.buf_idx == logits_buf && last.offset == 0 && (uint64_t)world * last.len_per_rank == vocab;
    }
    if (p->shard_pair_argmax && world > 64) { // (the pair block — device array or the peer block's slots — holds 64 entries, in BOTH gather modes)
        ctx->fail("shard_attach: the (max, index) pair gather of the greedy token holds at most 64 ranks");
        return -1;
    }
    if (p->shard_pair_argmax && !p->shard_pairs) {
        if (!CTX_CHECK(ctx, hipMalloc((void**)&p->shard_pairs, 64 * sizeof(unsigned long long)))) return -1;
        p->owned.push_back(p->shard_pairs);
    }
    if (ctx->shard && ctx->shard->peer) {
        if (n_points < 2 || world > 64) {
            ctx->fail("shard_attach: the peer gather needs at least two gather points per step (its flow control) and at most 64 ranks");
            return -1;
        }
        ShardPeer* sp = new ShardPeer();
        sp->poison_off = (n_points + 1) * 128;
        uint64_t off = (n_points + 2) * 128;
        for (uint64_t i = 0; i < n_points; i++) {
            const zgml_shard_point& gp = points[i];
            if (gp.len_per_rank % 2 || ((uintptr_t)(p->bufs[gp.buf_idx] + gp.offset) % 8) != 0) {
                delete sp;
                ctx->fail("shard_attach: peer gather: a slice that is not a whole number of aligned 8-byte units");
                return -1;
            }
            sp->stage_off.push_back(off);
            off += ((uint64_t)world * gp.len_per_rank * 4 + 255) / 256 * 256;
        }
        sp->pairs_off = off;
        off += 64 * 8;
        sp->bytes = off;
        // fine-grained device memory: peers' stores and this rank's system-scope loads meet without a cache between them
        if (!CTX_CHECK(ctx, hipExtMallocWithFlags((void**)&sp->block, sp->bytes, hipDeviceMallocFinegrained)) ||
            !CTX_CHECK(ctx, memset_sync(ctx->stream, sp->block, 0, sp->bytes)) || !CTX_CHECK(ctx, hipMalloc((void**)&sp->seen, (n_points + 1) * 4)) ||
            !CTX_CHECK(ctx, memset_sync(ctx->stream, sp->seen, 0, (n_points + 1) * 4)) || !CTX_CHECK(ctx, hipMalloc((void**)&sp->table_dev, (size_t)world * sizeof(char*)))) {
            hipFree(sp->block), hipFree(sp->seen), hipFree(sp->table_dev);
            delete sp;
            return -1;
        }
        sp->mapped.assign(world, nullptr);
        sp->ipc_opened.assign(world, 0);
        sp->mapped[ctx->shard->rank] = sp->block;
        p->shard_peer = sp;
    }
    return zgml_hip_program_set_barriers(ctx, p, bars.data(), bars.size()); // batched launches never straddle a collective
}

int zgml_hip_shard_init_peer(zgml_hip_ctx* ctx, int rank, int world) {
    if (!ctx || world < 1 || rank < 0 || rank >= world) return -1;
    if (ctx->shard) return -1; // one shard state per context
    if (!ctx->handoff_flag_dev) {
        ctx->fail("shard_init_peer: the context has no host-visible hand-off word");
        return -1;
    }
    ShardState* st = new ShardState();
    st->rank = rank, st->world = world, st->peer = true;
    ctx->shard = st;
    return 0;
}

int zgml_hip_shard_peer_export(zgml_hip_ctx* ctx, zgml_hip_program* p, zgml_shard_peer_handle* out) {
    if (!ctx || !p || !out || !p->shard_peer) return -1;
    hipSetDevice(ctx->device);
    memset(out, 0, sizeof(*out));
    hipIpcMemHandle_t h;
    static_assert(sizeof(h) <= sizeof(out->ipc), "hipIpcMemHandle_t fits the handle record");
    if (hipIpcGetMemHandle(&h, p->shard_peer->block) == hipSuccess)
        memcpy(out->ipc, &h, sizeof(h));
    else
        (void)hipGetLastError(); // (same-process importers only need `raw`; a cross-process import then fails loudly)
    out->pid = (uint64_t)getpid(), out->raw = (uint64_t)(uintptr_t)p->shard_peer->block, out->bytes = p->shard_peer->bytes;
    return 0;
}

int zgml_hip_shard_peer_import(zgml_hip_ctx* ctx, zgml_hip_program* p, int peer_rank, const zgml_shard_peer_handle* h) {
    if (!ctx || !p || !h || !p->shard_peer || !ctx->shard || peer_rank < 0 || peer_rank >= ctx->shard->world) return -1;
    hipSetDevice(ctx->device);
    ShardPeer* sp = p->shard_peer;
    if (h->bytes != sp->bytes) {
        ctx->fail("shard_peer_import: rank " + std::to_string(peer_rank) + "'s block has a different layout (gather points differ)");
        return -1;
    }
    if (peer_rank != ctx->shard->rank) {
        if (h->pid == (uint64_t)getpid()) {
            sp->mapped[peer_rank] = (char*)(uintptr_t)h->raw;
        } else {
            hipIpcMemHandle_t ih;
            memcpy(&ih, h->ipc, sizeof(ih));
            void* ptr = nullptr;
            if (!CTX_CHECK(ctx, hipIpcOpenMemHandle(&ptr, ih, hipIpcMemLazyEnablePeerAccess))) return -1;
            sp->mapped[peer_rank] = (char*)ptr, sp->ipc_opened[peer_rank] = 1;
        }
    }
    bool all = true;
    for (char* m : sp->mapped) all = all && m != nullptr;
    if (all) {
        if (!CTX_CHECK(ctx, h2d_sync(ctx->stream, sp->table_dev, sp->mapped.data(), sp->mapped.size() * sizeof(char*)))) return -1;
        sp->table_ready = true;
    }
    return 0;
}

int64_t zgml_hip_shard_step(zgml_hip_ctx* ctx, zgml_hip_program* p, const zgml_program_io* inputs, uint64_t n_inputs) {
    if (!ctx || !p || !ctx->shard) return -1;
    hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    if (p->plan_dirty || p->fuse_epoch != p->ctx->fuse_epoch) {
        free_graph(p);
        build_plan(p);
    }
    if (zgml_hip_stage_inputs(ctx, p, inputs, n_inputs) != 0) return -1;
    static const bool want_graph = !(getenv("ZGML_SHARD_GRAPH") && atoi(getenv("ZGML_SHARD_GRAPH")) == 0);
    if (want_graph && ctx->opt_graph && !p->shard_graph_exec && !p->shard_capture_failed) {
        // relaxed capture: RCCL may touch its own (already created) resources while it enqueues
        hipGraph_t g = nullptr;
        bool ok = hipStreamSynchronize(s) == hipSuccess && hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess;
        if (ok) {
            zgml_hip_enqueue_staged(ctx, p);
            const bool seg = shard_segments(ctx, p); // (incl. the greedy token into ctx->arg_out)
            hipMemcpyAsync(ctx->arg_out_host, ctx->arg_out, sizeof(int64_t), hipMemcpyDeviceToHost, s);
            ok = hipStreamEndCapture(s, &g) == hipSuccess && g && seg;
        }
        if (ok) ok = hipGraphInstantiate(&p->shard_graph_exec, g, nullptr, nullptr, 0) == hipSuccess;
        if (ok) {
            p->shard_graph = g;
        } else {
            if (g) hipGraphDestroy(g);
            p->shard_graph_exec = nullptr;
            p->shard_capture_failed = true; // eager from now on (all ranks decide alike: same program, same runtime)
            (void)hipGetLastError();
            if (!ctx->err.empty()) return -1;
        }
    }
    if (p->shard_graph_exec) {
        if (!CTX_CHECK(ctx, hipGraphLaunch(p->shard_graph_exec, s))) return -1;
    } else {
        zgml_hip_enqueue_staged(ctx, p);
        if (!shard_segments(ctx, p)) return -1;
        hipMemcpyAsync(ctx->arg_out_host, ctx->arg_out, sizeof(int64_t), hipMemcpyDeviceToHost, s);
    }
    if (!CTX_CHECK(ctx, hipStreamSynchronize(s)) || !ctx->handoff_ok("shard_step")) return -1;
    p->profile.call_count++;
    return *ctx->arg_out_host;
}

// One EAGER step with HIP events around every all-gather (diagnostics for bench.py's N > 1 line): the device time of the whole
// step and the part of it spent inside the collectives. Every rank must call it (the gathers are collective). Returns the
// greedy token, -1 on error.
int64_t zgml_hip_shard_profile_step(zgml_hip_ctx* ctx, zgml_hip_program* p, const zgml_program_io* inputs, uint64_t n_inputs, double* step_us,
                                    double* gather_us) {
    if (!ctx || !p || !ctx->shard) return -1;
    hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    if (p->plan_dirty || p->fuse_epoch != p->ctx->fuse_epoch) {
        free_graph(p);
        build_plan(p);
    }
    if (zgml_hip_stage_inputs(ctx, p, inputs, n_inputs) != 0) return -1;
    std::vector<hipEvent_t> ev(2 * p->shard_points.size() + 2);
    for (auto& e : ev) hipEventCreate(&e);
    hipEventRecord(ev[ev.size() - 2], s);
    zgml_hip_enqueue_staged(ctx, p);
    bool ok = shard_segments(ctx, p, &ev);
    hipMemcpyAsync(ctx->arg_out_host, ctx->arg_out, sizeof(int64_t), hipMemcpyDeviceToHost, s);
    hipEventRecord(ev[ev.size() - 1], s);
    ok = ok && CTX_CHECK(ctx, hipStreamSynchronize(s)) && ctx->handoff_ok("shard_profile_step");
    if (ok) {
        float ms = 0;
        double g = 0;
        p->shard_point_us.assign(p->shard_points.size(), 0.0);
        for (size_t i = 0; i < p->shard_points.size(); i++)
            if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) == hipSuccess) g += ms * 1e3, p->shard_point_us[i] = ms * 1e3;
        if (gather_us) *gather_us = g;
        if (step_us && hipEventElapsedTime(&ms, ev[ev.size() - 2], ev[ev.size() - 1]) == hipSuccess) *step_us = ms * 1e3;
    }
    for (auto& e : ev) hipEventDestroy(e);
    return ok ? *ctx->arg_out_host : -1;
}

uint64_t zgml_hip_shard_last_point_us(zgml_hip_program* p, double* out, uint64_t cap) {
    if (!p) return 0;
    for (uint64_t i = 0; out && i < cap && i < p->shard_point_us.size(); i++) out[i] = p->shard_point_us[i];
    return p->shard_point_us.size();
}
int zgml_hip_device_can_access_peer(int device, int peer) {
    int can = 0;
    if (device == peer) return 1;
    if (hipDeviceCanAccessPeer(&can, device, peer) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return can ? 1 : 0;
}
int zgml_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int zgml_hip_shard_step_mode(zgml_hip_program* p) { return !p ? -1 : (p->shard_graph_exec ? 1 : 0); } // 1 = one graph per token

} // extern "C"
