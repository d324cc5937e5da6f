/*
 * zgml_oracle.c — CPU restatement of zgml's forward-inference arithmetic (see zgml_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: parity oracle + host-CPU baseline. Not part of the product path.
 * Parity pinned by the reference's own known-answer tests (tests/golden/kat.json); the reference
 * itself (Zig 0.16) cannot be built in this image.
 *
 * Accumulation orders follow the reference: where it accumulates in 8-lane vectors and then
 * @reduce(.Add)s, we keep 8 partial sums and add them in lane order (LLVM lowers a strict-mode
 * float @reduce to an ordered reduction).
 */
#include "zgml_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ZO_V 8 /* reference.Context.V, src/backend/reference.zig:178 */

/* ───────────────────────────── thread pool ───────────────────────────── */

typedef void (*zo_range_fn)(void* ctx, uint64_t begin, uint64_t end);

#define ZO_MAX_THREADS 64

static struct {
    int n_threads; /* requested */
    int spawned;   /* worker threads alive (n-1 of them) */
    pthread_t threads[ZO_MAX_THREADS];
    pthread_mutex_t mutex;
    pthread_cond_t work_ready, work_done;
    uint64_t generation;
    int pending;
    int shutdown;
    zo_range_fn fn;
    void* ctx;
    uint64_t n_items, chunk;
    int n_active;
    int inited;
} g_pool = {.n_threads = 1};

/* a worker starts at the generation current when it was spawned: after zo_set_threads() re-created the pool
 * `generation` is no longer 0, and a worker starting from 0 would run the PREVIOUS dispatch's stale fn / ctx */
static struct {
    int id;
    uint64_t gen;
} g_worker_args[ZO_MAX_THREADS];

static void* zo_worker(void* arg) {
    int id = ((int*)arg)[0];
    uint64_t my_gen = g_worker_args[id].gen;
    for (;;) {
        pthread_mutex_lock(&g_pool.mutex);
        while (g_pool.generation == my_gen && !g_pool.shutdown)
            pthread_cond_wait(&g_pool.work_ready, &g_pool.mutex);
        if (g_pool.shutdown) {
            pthread_mutex_unlock(&g_pool.mutex);
            return NULL;
        }
        my_gen = g_pool.generation;
        zo_range_fn fn = g_pool.fn;
        void* ctx = g_pool.ctx;
        uint64_t n = g_pool.n_items, chunk = g_pool.chunk;
        int active = id < g_pool.n_active;
        pthread_mutex_unlock(&g_pool.mutex);
        if (active) {
            uint64_t b = (uint64_t)id * chunk, e = b + chunk;
            if (e > n) e = n;
            if (b < e) fn(ctx, b, e);
        }
        pthread_mutex_lock(&g_pool.mutex);
        if (--g_pool.pending == 0) pthread_cond_signal(&g_pool.work_done);
        pthread_mutex_unlock(&g_pool.mutex);
    }
}

static void zo_pool_shutdown(void) {
    if (!g_pool.inited) return;
    pthread_mutex_lock(&g_pool.mutex);
    g_pool.shutdown = 1;
    pthread_cond_broadcast(&g_pool.work_ready);
    pthread_mutex_unlock(&g_pool.mutex);
    for (int i = 1; i <= g_pool.spawned; i++) pthread_join(g_pool.threads[i], NULL);
    g_pool.spawned = 0;
    g_pool.shutdown = 0;
}

void zo_set_threads(int n) {
    if (n < 1) n = 1;
    if (n > ZO_MAX_THREADS) n = ZO_MAX_THREADS;
    if (n == g_pool.n_threads) return;
    zo_pool_shutdown();
    g_pool.n_threads = n;
}

int zo_get_threads(void) { return g_pool.n_threads; }

static void zo_pool_ensure(void) {
    if (!g_pool.inited) {
        pthread_mutex_init(&g_pool.mutex, NULL);
        pthread_cond_init(&g_pool.work_ready, NULL);
        pthread_cond_init(&g_pool.work_done, NULL);
        g_pool.inited = 1;
        atexit(zo_pool_shutdown);
    }
    while (g_pool.spawned < g_pool.n_threads - 1) {
        int id = g_pool.spawned + 1;
        g_worker_args[id].id = id;
        g_worker_args[id].gen = g_pool.generation; /* only this (dispatching) thread ever bumps it */
        if (pthread_create(&g_pool.threads[id], NULL, zo_worker, &g_worker_args[id]) != 0) break;
        g_pool.spawned = id;
    }
}

/* Split [0,n_items) into at most `max_active` contiguous chunks of a multiple of `align` items;
 * the calling thread takes chunk 0 (as GemvPool.dispatch does, src/quant.zig:187-190). */
static void zo_parallel_for(uint64_t n_items, uint64_t align, int max_active, zo_range_fn fn, void* ctx) {
    int n_active = g_pool.n_threads;
    if (max_active > 0 && n_active > max_active) n_active = max_active;
    if ((uint64_t)n_active > n_items) n_active = (int)(n_items ? n_items : 1);
    if (n_active <= 1) {
        fn(ctx, 0, n_items);
        return;
    }
    zo_pool_ensure();
    if (g_pool.spawned + 1 < n_active) n_active = g_pool.spawned + 1;
    if (n_active <= 1) {
        fn(ctx, 0, n_items);
        return;
    }
    uint64_t chunk = (n_items + (uint64_t)n_active - 1) / (uint64_t)n_active;
    if (align > 1) chunk = (chunk + align - 1) / align * align;
    pthread_mutex_lock(&g_pool.mutex);
    g_pool.fn = fn;
    g_pool.ctx = ctx;
    g_pool.n_items = n_items;
    g_pool.chunk = chunk;
    g_pool.n_active = n_active;
    g_pool.pending = g_pool.spawned;
    g_pool.generation++;
    pthread_cond_broadcast(&g_pool.work_ready);
    pthread_mutex_unlock(&g_pool.mutex);
    fn(ctx, 0, chunk < n_items ? chunk : n_items);
    pthread_mutex_lock(&g_pool.mutex);
    while (g_pool.pending > 0) pthread_cond_wait(&g_pool.work_done, &g_pool.mutex);
    pthread_mutex_unlock(&g_pool.mutex);
}

/* ───────────────────────────── small helpers ───────────────────────────── */

static inline float zo_maxf(float a, float b) { return fmaxf(a, b); } /* Zig @max: NaN-ignoring */

/* ordered 8-lane reduction, the lowering of a strict-mode @reduce(.Add, @Vector(8,f32)) */
static inline float zo_reduce8(const float* acc) {
    float s = acc[0];
    for (int i = 1; i < ZO_V; i++) s += acc[i];
    return s;
}

/* ───────────────────────────── elementwise ───────────────────────────── */

/* reference.Context.elementwise, src/backend/reference.zig:201-273.
 * sgn/step: the reference executor falls through to a copy (:271) although Capabilities admits
 * them (src/backend.zig:113-118); we implement the true ops of src/tensor/forward.zig:959-1007,
 * as the WGSL/MSL uber-kernels do (last bullet). */
static void zo_elementwise(const zo_buffer* b, const zgml_op_elementwise* e) {
    float* dst = b[e->dst].ptr + e->dst_offset;
    const float* s0 = b[e->src0].ptr + e->src0_offset;
    const float* s1 = b[e->src1].ptr + e->src1_offset;
    uint64_t n = e->n;
    switch (e->op) {
        case ZGML_OP_ADD:
            for (uint64_t i = 0; i < n; i++) dst[i] = s0[i] + s1[i];
            break;
        case ZGML_OP_MUL:
            for (uint64_t i = 0; i < n; i++) dst[i] = s0[i] * s1[i];
            break;
        case ZGML_OP_NEG:
            for (uint64_t i = 0; i < n; i++) dst[i] = -s0[i];
            break;
        case ZGML_OP_ABS:
            for (uint64_t i = 0; i < n; i++) dst[i] = fabsf(s0[i]);
            break;
        case ZGML_OP_SGN:
            for (uint64_t i = 0; i < n; i++) dst[i] = s0[i] > 0 ? 1.0f : (s0[i] < 0 ? -1.0f : 0.0f);
            break;
        case ZGML_OP_STEP:
            for (uint64_t i = 0; i < n; i++) dst[i] = s0[i] > 0 ? 1.0f : 0.0f;
            break;
        case ZGML_OP_RELU:
            for (uint64_t i = 0; i < n; i++) dst[i] = zo_maxf(s0[i], 0.0f);
            break;
        case ZGML_OP_SQRT:
            for (uint64_t i = 0; i < n; i++) dst[i] = sqrtf(s0[i]);
            break;
        case ZGML_OP_RECIP:
            for (uint64_t i = 0; i < n; i++) dst[i] = 1.0f / s0[i];
            break;
        case ZGML_OP_EXP:
            for (uint64_t i = 0; i < n; i++) dst[i] = expf(s0[i]);
            break;
        case ZGML_OP_LOG:
            for (uint64_t i = 0; i < n; i++) dst[i] = logf(s0[i]);
            break;
        case ZGML_OP_GELU: {
            /* vector body :258-264 uses (e^{2k}-1)/(e^{2k}+1); scalar tail :265-269 uses tanh */
            uint64_t i = 0;
            for (; i + ZO_V <= n; i += ZO_V) {
                for (int j = 0; j < ZO_V; j++) {
                    float a = s0[i + j];
                    float k = 0.7978845608f * (a + 0.044715f * a * a * a);
                    float e2k = expf(k + k);
                    dst[i + j] = 0.5f * a * (1.0f + (e2k - 1.0f) / (e2k + 1.0f));
                }
            }
            for (; i < n; i++) {
                float a = s0[i];
                float kk = 0.7978845608f * (a + 0.044715f * a * a * a);
                dst[i] = 0.5f * a * (1.0f + tanhf(kk));
            }
            break;
        }
        default:
            memmove(dst, s0, n * sizeof(float));
            break;
    }
}

/* reference.Context.fusedElementwise, src/backend/reference.zig:275-307 */
static void zo_fused_elementwise(const zo_buffer* b, const zgml_op_fused_elementwise* fe) {
    float* dst = b[fe->dst].ptr + fe->dst_offset;
    const float* src = b[fe->src].ptr + fe->src_offset;
    for (uint64_t i = 0; i < fe->n; i++) {
        float v = src[i];
        for (uint32_t s = 0; s < fe->n_steps; s++) {
            const zgml_fused_step* st = &fe->steps[s];
            switch (st->op) {
                case ZGML_OP_NEG: v = -v; break;
                case ZGML_OP_ABS: v = fabsf(v); break;
                case ZGML_OP_SGN: v = v > 0 ? 1.0f : (v < 0 ? -1.0f : 0.0f); break;
                case ZGML_OP_STEP: v = v > 0 ? 1.0f : 0.0f; break;
                case ZGML_OP_RELU: v = zo_maxf(v, 0.0f); break;
                case ZGML_OP_SQRT: v = sqrtf(v); break;
                case ZGML_OP_RECIP: v = 1.0f / v; break;
                case ZGML_OP_EXP: v = expf(v); break;
                case ZGML_OP_LOG: v = logf(v); break;
                case ZGML_OP_GELU: {
                    float kk = 0.7978845608f * (v + 0.044715f * v * v * v);
                    v = 0.5f * v * (1.0f + tanhf(kk));
                    break;
                }
                case ZGML_OP_ADD: {
                    const float* sp = b[st->secondary_buf].ptr + st->secondary_offset;
                    v = st->is_swapped ? sp[i] + v : v + sp[i];
                    break;
                }
                case ZGML_OP_MUL: {
                    const float* sp = b[st->secondary_buf].ptr + st->secondary_offset;
                    v = st->is_swapped ? sp[i] * v : v * sp[i];
                    break;
                }
                default: break;
            }
        }
        dst[i] = v;
    }
}

/* ───────────────────────────── row-wise ops ───────────────────────────── */

/* reference.Context.softmax, src/backend/reference.zig:309-327, with the finite-shift guard of
 * computeSoftmax, src/tensor/forward.zig:1306-1322 (an all -inf row gives zeros, the KAT at
 * forward.zig:2189-2203; the bare executor would give NaN*0 there). */
static void zo_softmax(const zo_buffer* b, const zgml_op_rowwise* s) {
    const float* src = b[s->src].ptr;
    float* dst = b[s->dst].ptr;
    uint64_t cols = s->cols;
    for (uint64_t row = 0; row < s->rows; row++) {
        uint64_t sb = (uint64_t)s->src_offset + row * cols, db = (uint64_t)s->dst_offset + row * cols;
        float m = -INFINITY;
        for (uint64_t j = 0; j < cols; j++) m = zo_maxf(m, src[sb + j]);
        float sum = 0;
        for (uint64_t j = 0; j < cols; j++) {
            float shifted = src[sb + j] - m;
            float v = isfinite(shifted) ? expf(shifted) : 0.0f;
            dst[db + j] = v;
            sum += v;
        }
        float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
        for (uint64_t j = 0; j < cols; j++) dst[db + j] *= inv;
    }
}

/* reference.Context.layernorm, src/backend/reference.zig:329-347 */
static void zo_layernorm(const zo_buffer* b, const zgml_op_rowwise* l) {
    const float* src = b[l->src].ptr;
    float* dst = b[l->dst].ptr;
    uint64_t cols = l->cols;
    for (uint64_t row = 0; row < l->rows; row++) {
        uint64_t base = (uint64_t)l->src_offset + row * cols, dbase = (uint64_t)l->dst_offset + row * cols;
        float mu = 0;
        for (uint64_t j = 0; j < cols; j++) mu += src[base + j];
        mu /= (float)cols;
        float v = 0;
        for (uint64_t j = 0; j < cols; j++) {
            float diff = src[base + j] - mu;
            v += diff * diff;
        }
        float inv_std = 1.0f / sqrtf(v / (float)cols + l->eps);
        for (uint64_t j = 0; j < cols; j++) dst[dbase + j] = (src[base + j] - mu) * inv_std;
    }
}

/* reference.Context.rmsnorm, src/backend/reference.zig:349-374 (8-lane sum of squares, ordered
 * reduce, scalar tail; inv = 1/sqrt(ss/cols + eps)) */
static void zo_rmsnorm(const zo_buffer* b, const zgml_op_rowwise* r) {
    const float* src = b[r->src].ptr;
    float* dst = b[r->dst].ptr;
    uint64_t cols = r->cols;
    for (uint64_t row = 0; row < r->rows; row++) {
        const float* s = src + r->src_offset + row * cols;
        float* d = dst + r->dst_offset + row * cols;
        float acc[ZO_V] = {0};
        uint64_t i = 0;
        for (; i + ZO_V <= cols; i += ZO_V)
            for (int j = 0; j < ZO_V; j++) acc[j] += s[i + j] * s[i + j];
        float ss = zo_reduce8(acc);
        for (; i < cols; i++) ss += s[i] * s[i];
        float inv = 1.0f / sqrtf(ss / (float)cols + r->eps);
        for (i = 0; i < cols; i++) d[i] = s[i] * inv;
    }
}

/* reference.Context.reduce, src/backend/reference.zig:376-389 */
static void zo_reduce(const zo_buffer* b, const zgml_op_reduce* rd) {
    const float* src = b[rd->src].ptr;
    float* dst = b[rd->dst].ptr;
    uint64_t rs = rd->reduce_size;
    for (uint64_t i = 0; i < rd->n_out; i++) {
        uint64_t sb = (uint64_t)rd->src_offset + i * rs;
        float val = rd->op == ZGML_OP_MAX ? -INFINITY : 0.0f;
        for (uint64_t k = 0; k < rs; k++) {
            float v = src[sb + k];
            val = rd->op == ZGML_OP_MAX ? zo_maxf(val, v) : val + v;
        }
        dst[(uint64_t)rd->dst_offset + i] = val;
    }
}

/* reference.Context.repeat, src/backend/reference.zig:391-433 */
static void zo_repeat(const zo_buffer* b, const zgml_op_repeat* rp) {
    const float* src = b[rp->src].ptr;
    float* dst = b[rp->dst].ptr;
    uint64_t n = rp->n;
    float* d = dst + rp->dst_offset;
    const float* s = src + rp->src_offset;
    uint64_t src_n = (uint64_t)rp->src_ne[0] * rp->src_ne[1] * rp->src_ne[2] * rp->src_ne[3];
    if (src_n == 1) {
        for (uint64_t i = 0; i < n; i++) d[i] = s[0];
        return;
    }
    if (src_n >= n) {
        memmove(d, s, n * sizeof(float));
        return;
    }
    if (n % src_n == 0 && rp->src_strides[0] == 1 &&
        (rp->src_ne[1] <= 1 || rp->src_strides[1] == rp->src_ne[0]) &&
        (rp->src_ne[2] <= 1 || rp->src_strides[2] == rp->src_ne[0] * rp->src_ne[1]) &&
        (rp->src_ne[3] <= 1 || rp->src_strides[3] == rp->src_ne[0] * rp->src_ne[1] * rp->src_ne[2])) {
        for (uint64_t off = 0; off + src_n <= n; off += src_n) memcpy(d + off, s, src_n * sizeof(float));
        return;
    }
    for (uint64_t gid = 0; gid < n; gid++) {
        uint64_t idx = gid, src_idx = rp->src_offset;
        for (int dim = 3; dim >= 0; dim--) {
            uint64_t coord = idx / rp->dst_strides[dim];
            idx = idx % rp->dst_strides[dim];
            src_idx += (coord % rp->src_ne[dim]) * rp->src_strides[dim];
        }
        dst[(uint64_t)rp->dst_offset + gid] = src[src_idx];
    }
}

/* reference.Context.sliceAssign, src/backend/reference.zig:435-455 */
static void zo_slice_assign(const zo_buffer* b, const zgml_op_slice_assign* sa) {
    const float* src = b[sa->src].ptr;
    float* dst = b[sa->dst].ptr;
    uint64_t rows = sa->rows, cols = sa->cols, doff = sa->dst_offset, soff = sa->src_offset;
    uint64_t drs = sa->dst_row_stride, dcs = sa->dst_col_stride, srs = sa->src_row_stride,
             scs = sa->src_col_stride;
    for (uint64_t col = 0; col < cols; col++)
        for (uint64_t row = 0; row < rows; row++)
            dst[doff + row * drs + col * dcs] = src[soff + row * srs + col * scs];
}

/* reference.Context.rope, src/backend/reference.zig:457-478. NOTE the DeviceOp convention: sin
 * is read at cs_off + pair + half_d (metal.zig:792-793 and compute.wgsl agree), whereas the CPU
 * tensor op computeRope (src/tensor/forward.zig:451-471) reads it at cs_off + d + pair. With the
 * LLaMA packing [cos(d) | sin(d)] (src/nn.zig:347-354) the device path therefore multiplies by
 * the duplicated cos half. Parity is with the DeviceOp executor. */
static void zo_rope(const zo_buffer* b, const zgml_op_rope* rr) {
    const float* src = b[rr->src].ptr;
    const float* cs = b[rr->cos_sin].ptr;
    float* dst = b[rr->dst].ptr;
    uint64_t hd = rr->half_d;
    for (uint64_t col = 0; col < rr->seq_len; col++) {
        for (uint64_t pair = 0; pair < hd; pair++) {
            float x_lo = src[rr->src_off + pair * rr->src_rs + col * rr->src_cs];
            float x_hi = src[rr->src_off + (pair + hd) * rr->src_rs + col * rr->src_cs];
            float cos_v = cs[rr->cs_off + pair + col * rr->cs_cs];
            float sin_v = cs[rr->cs_off + pair + hd + col * rr->cs_cs];
            dst[rr->dst_off + pair + col * 2 * hd] = x_lo * cos_v - x_hi * sin_v;
            dst[rr->dst_off + pair + hd + col * 2 * hd] = x_hi * cos_v + x_lo * sin_v;
        }
    }
}

/* ───────────────────────────── matmul ───────────────────────────── */

typedef struct {
    const float *a, *bm;
    float* dst;
    const zgml_matmul_geom* g;
} zo_mm_ctx;

/* reference.Context.matmul -> forward.blasSgemm index contract, src/tensor/forward.zig:686-753:
 * C[dst_offset + m*dst_row_stride + n] = sum_k A[a_offset + m*a_rs + k*a_cs] * B[b_offset + k*b_rs + n*b_cs].
 * k-sequential f32 accumulation (BLAS/TiledMatMul orders differ; tests use a relative bound). */
static void zo_mm_range(void* vctx, uint64_t n0, uint64_t n1) {
    zo_mm_ctx* c = (zo_mm_ctx*)vctx;
    const zgml_matmul_geom* g = c->g;
    for (uint64_t m = 0; m < g->M; m++) {
        for (uint64_t n = n0; n < n1; n++) {
            const float* ap = c->a + g->a_offset + m * g->a_row_stride;
            const float* bp = c->bm + g->b_offset + n * g->b_col_stride;
            float acc = 0;
            for (uint64_t k = 0; k < g->K; k++) acc += ap[k * g->a_col_stride] * bp[k * g->b_row_stride];
            c->dst[g->dst_offset + m * g->dst_row_stride + n] = acc;
        }
    }
}

/* f16 weight promotion (src/backend/wgpu.zig:1071-1104: B operands of `matmul` ops with an initial
 * upload are packed to f16 [K,N]; kernels src/backend/metal.zig:680-760):
 *   matvec_f16 (M == 1): sum += A[k] * float(B16[k*N + n])            — A stays f32
 *   matmul_f16 (M  > 1): A is staged as half(A[...]), f32 accumulate (simdgroup_float8x8)
 * `shadow` is that packed f16 image held as f32 values. k-sequential here; orders differ. */
float zo_f16_to_f32(uint16_t h);
uint16_t zo_f32_to_f16(float f);
typedef struct {
    const float *a, *shadow;
    float* dst;
    const zgml_matmul_geom* g;
} zo_mm16_ctx;

static void zo_mm16_range(void* vctx, uint64_t n0, uint64_t n1) {
    zo_mm16_ctx* c = (zo_mm16_ctx*)vctx;
    const zgml_matmul_geom* g = c->g;
    for (uint64_t m = 0; m < g->M; m++) {
        const float* ap = c->a + g->a_offset + m * g->a_row_stride;
        for (uint64_t n = n0; n < n1; n++) {
            float acc = 0;
            for (uint64_t k = 0; k < g->K; k++) {
                float av = ap[k * g->a_col_stride];
                if (g->M > 1) av = zo_f16_to_f32(zo_f32_to_f16(av));
                acc += av * c->shadow[k * g->N + n];
            }
            c->dst[g->dst_offset + m * g->dst_row_stride + n] = acc;
        }
    }
}

static void zo_matmul_f16(const zo_buffer* b, const zgml_op_matmul* m, const float* shadow) {
    zo_mm16_ctx c = {b[m->a].ptr, shadow, b[m->dst].ptr, &m->geom};
    if (m->geom.N * m->geom.K * m->geom.M >= (1u << 16))
        zo_parallel_for(m->geom.N, 1, 0, zo_mm16_range, &c);
    else
        zo_mm16_range(&c, 0, m->geom.N);
}

static void zo_matmul(const zo_buffer* b, const zgml_op_matmul* m) {
    zo_mm_ctx c = {b[m->a].ptr, b[m->b].ptr, b[m->dst].ptr, &m->geom};
    if (m->geom.N * m->geom.K * m->geom.M >= (1u << 16))
        zo_parallel_for(m->geom.N, 1, 0, zo_mm_range, &c);
    else
        zo_mm_range(&c, 0, m->geom.N);
}

typedef struct {
    const int8_t* data;
    const float* scales;
    uint64_t bs, N, K;
    const float* input_row;
    float* dst_row;
} zo_qmm_ctx;

/* one input row, output columns [n0,n1): k outer (sequential), n inner, scale*input hoisted per
 * block chunk — reference.zig:540-564. Each output's accumulation order (k ascending, product
 * (scale*x_k)*w then add, no contraction) is independent of the N split. */
static void zo_qmm_range(void* vctx, uint64_t n0, uint64_t n1) {
    zo_qmm_ctx* c = (zo_qmm_ctx*)vctx;
    for (uint64_t n = n0; n < n1; n++) c->dst_row[n] = 0.0f;
    for (uint64_t k = 0; k < c->K; k++) {
        float input_v = c->input_row[k];
        uint64_t w_base = k * c->N;
        uint64_t n = n0;
        while (n < n1) {
            uint64_t flat = w_base + n;
            float scale = c->scales[flat / c->bs] * input_v;
            uint64_t block_rem = c->bs - (flat % c->bs);
            uint64_t chunk = block_rem < n1 - n ? block_rem : n1 - n;
            const int8_t* w = c->data + flat;
            float* d = c->dst_row + n;
            for (uint64_t j = 0; j < chunk; j++) d[j] = d[j] + (float)w[j] * scale;
            n += chunk;
        }
    }
}

void zo_qmatmul_exact(const int8_t* data, const float* scales, uint64_t bs, const float* input, float* dst,
                      uint64_t M, uint64_t N, uint64_t K, uint64_t input_row_stride,
                      uint64_t dst_row_stride) {
    if (input_row_stride == 0) input_row_stride = K;
    if (dst_row_stride == 0) dst_row_stride = N;
    for (uint64_t row = 0; row < M; row++) {
        zo_qmm_ctx c = {data, scales, bs, N, K, input + row * input_row_stride, dst + row * dst_row_stride};
        if (N * K >= (1u << 16))
            zo_parallel_for(N, 32, 0, zo_qmm_range, &c);
        else
            zo_qmm_range(&c, 0, N);
    }
}

/* reference.Context.qmatmul, src/backend/reference.zig:499-566. On x86_64 the aarch64-only W8A8
 * arm (:512-528) is compiled out, so the generic exact-dequant loop is the behaviour. */
void zo_quantize_input(const float* input, uint64_t K, uint64_t bs, int8_t* inp_q, float* inp_scales);
void zo_gemv_pool_dispatch(const int8_t* t_d, const float* t_s, const int8_t* inp_q, const float* inp_scales,
                           float* dst, uint64_t N, uint64_t K, uint64_t bs, int n_workers);

static void zo_qmatmul(const zo_buffer* b, const zo_qweight* qws, const zgml_op_qmatmul* q) {
    const zo_qweight* w = &qws[q->weight_idx];
    /* the W8A8 arm (reference.zig:512-528, taken on aarch64 when the weight carries its transposed copy;
     * inference_utils.zig:157-176 runs the same arithmetic through GemvPool): only when zo_set_w8a8(1) made
     * zo_compile_program prepare t_data / t_scales */
    const uint64_t K = q->K, N = q->N, bs = w->block_size, bpr = bs ? (K + bs - 1) / bs : 0;
    const uint64_t in_rs = q->input_row_stride ? q->input_row_stride : K, dst_rs = q->dst_row_stride ? q->dst_row_stride : N;
    if (w->t_data && w->t_scales && q->M == 1 && in_rs == K && dst_rs == N && K <= 16384 && bpr <= 512) {
        int8_t inp_q[16384];
        float inp_scales[512];
        zo_quantize_input(b[q->input].ptr + q->input_offset, K, bs, inp_q, inp_scales);
        zo_gemv_pool_dispatch(w->t_data, w->t_scales, inp_q, inp_scales, b[q->dst].ptr + q->dst_offset, N, K, bs,
                              zo_get_threads());
        return;
    }
    zo_qmatmul_exact(w->data, w->scales, w->block_size, b[q->input].ptr + q->input_offset,
                     b[q->dst].ptr + q->dst_offset, q->M, q->N, q->K, q->input_row_stride, q->dst_row_stride);
}

/* ───────────────────────────── attention ───────────────────────────── */

/* reference.Context.attention, src/backend/reference.zig:568-672: streaming softmax per query
 * column, mask-first skip, non-finite score skip, l==0 -> zeros. */
static void zo_attention(const zo_buffer* b, const zgml_op_attention* att) {
    const float* q_ptr = b[att->q].ptr;
    const float* k_ptr = b[att->k].ptr;
    const float* v_ptr = b[att->v].ptr;
    const float* mask_ptr = b[att->mask].ptr;
    float* dst = b[att->dst].ptr;
    uint64_t dh = att->d_head, sq = att->seq_q, skv = att->seq_kv;
    uint64_t qrs = att->q_rs, qcs = att->q_cs, krs = att->k_rs, kcs = att->k_cs, vrs = att->v_rs,
             vcs = att->v_cs, mrs = att->mask_rs, mcs = att->mask_cs, drs = att->dst_rs, dcs = att->dst_cs;
    int unit_qk = (qrs == 1 && krs == 1);
    float acc[512];
    if (dh > 512) return; /* std.debug.assert(dh <= 512), :593 */
    for (uint64_t qi = 0; qi < sq; qi++) {
        uint64_t q_off = (uint64_t)att->q_off + qi * qcs, d_off = (uint64_t)att->dst_off + qi * dcs;
        uint64_t mask_q_off = (uint64_t)att->mask_off + qi * mcs;
        float m_val = -INFINITY, l = 0;
        for (uint64_t r = 0; r < dh; r++) acc[r] = 0;
        for (uint64_t s = 0; s < skv; s++) {
            float mask_add = att->has_mask ? mask_ptr[mask_q_off + s * mrs] : 0.0f;
            if (!isfinite(mask_add)) continue;
            float dot = 0;
            if (unit_qk) {
                float dv[ZO_V] = {0};
                uint64_t r = 0, kb = (uint64_t)att->k_off + s * kcs;
                for (; r + ZO_V <= dh; r += ZO_V)
                    for (int j = 0; j < ZO_V; j++) dv[j] += q_ptr[q_off + r + j] * k_ptr[kb + r + j];
                dot = zo_reduce8(dv);
                for (; r < dh; r++) dot += q_ptr[q_off + r] * k_ptr[kb + r];
            } else {
                for (uint64_t r = 0; r < dh; r++)
                    dot += q_ptr[q_off + r * qrs] * k_ptr[(uint64_t)att->k_off + r * krs + s * kcs];
            }
            float score = dot * att->scale + mask_add;
            if (!isfinite(score)) continue;
            float new_m = zo_maxf(m_val, score);
            float alpha = (m_val == -INFINITY) ? 0.0f : expf(m_val - new_m);
            float w = expf(score - new_m);
            l = l * alpha + w;
            m_val = new_m;
            for (uint64_t r = 0; r < dh; r++)
                acc[r] = acc[r] * alpha + w * v_ptr[(uint64_t)att->v_off + r * vrs + s * vcs];
        }
        float inv_l = l > 0 ? 1.0f / l : 0.0f;
        for (uint64_t r = 0; r < dh; r++) dst[d_off + r * drs] = acc[r] * inv_l;
    }
}

/* ───────────────────── quantised KV cache (extension ops) ───────────────────── */
/* Cache buffer layout (include/zgml_hip.h): int8 q_data[n_cols*d_head], then f32 scales[n_cols*bpc]
 * at element offset n_cols*d_head/4. */
void zo_quantize_input(const float* input, uint64_t K, uint64_t bs, int8_t* inp_q, float* inp_scales);

static inline int8_t* zo_kvq_data(float* base) { return (int8_t*)base; }
static inline float* zo_kvq_scales(float* base, uint64_t n_cols, uint64_t dh) { return base + n_cols * dh / 4; }

/* QuantizedKVCache.storeColumn, src/quant.zig:687-699 */
static void zo_kvq_store(const zo_buffer* b, const zgml_op_kvq_store* st) {
    const uint64_t dh = st->d_head, bpc = dh / st->block_size;
    int8_t* qd = zo_kvq_data(b[st->cache].ptr) + (uint64_t)st->col * dh;
    float* sc = zo_kvq_scales(b[st->cache].ptr, st->n_cols, dh) + (uint64_t)st->col * bpc;
    zo_quantize_input(b[st->src].ptr + st->src_offset, dh, st->block_size, qd, sc);
}

/* dotI8F32, src/quant.zig:801-828: per block 8 lanes of f*q, lane-order reduce, times the block scale */
static float zo_kvq_dot(const float* f, const int8_t* q, const float* sc, uint64_t bs, uint64_t nb) {
    float total = 0;
    for (uint64_t bl = 0; bl < nb; bl++) {
        const uint64_t base = bl * bs;
        float lanes[ZO_V] = {0};
        uint64_t i = 0;
        for (; i + ZO_V <= bs; i += ZO_V)
            for (int j = 0; j < ZO_V; j++) lanes[j] += f[base + i + j] * (float)q[base + i + j];
        float sub = zo_reduce8(lanes);
        for (; i < bs; i++) sub += f[base + i] * (float)q[base + i];
        total += sub * sc[bl];
    }
    return total;
}

/* attentionQuantized, src/quant.zig:925-1091, the non-SDOT (x86) path: flash tiles of Bs = 8 columns
 * (whole-tile mask skip, one rescale per tile, accumBatchI8F32 :861-908), then a one-column tail
 * (accumI8F32 :830-857). */
static void zo_attention_kvq(const zo_buffer* b, const zgml_op_attention_kvq* a) {
    enum { BS_TILE = 8 };
    const uint64_t dh = a->d_head, bs = a->block_size, nb = dh / bs, ncols = a->n_cols;
    const int8_t* kq = zo_kvq_data(b[a->k].ptr);
    const float* ks = zo_kvq_scales(b[a->k].ptr, ncols, dh);
    const int8_t* vq = zo_kvq_data(b[a->v].ptr);
    const float* vs = zo_kvq_scales(b[a->v].ptr, ncols, dh);
    const float* mask = b[a->mask].ptr;
    float acc[512];
    if (dh > 512) return;
    for (uint64_t qi = 0; qi < a->seq_q; qi++) {
        const float* q = b[a->q].ptr + a->q_off + qi * a->q_cs;
        const uint64_t mask_base = (uint64_t)a->mask_off + qi * a->mask_cs;
        float m_val = -INFINITY, l = 0;
        for (uint64_t r = 0; r < dh; r++) acc[r] = 0;
        uint64_t s = 0;
        for (; s + BS_TILE <= a->seq_kv; s += BS_TILE) {
            float scores[BS_TILE], ws[BS_TILE];
            if (a->has_mask) {
                int any = 0;
                for (int t = 0; t < BS_TILE; t++) any |= isfinite(mask[mask_base + (s + t) * a->mask_rs]) != 0;
                if (!any) continue;
            }
            float tile_max = -INFINITY;
            for (int t = 0; t < BS_TILE; t++) {
                const float mask_add = a->has_mask ? mask[mask_base + (s + t) * a->mask_rs] : 0.0f;
                if (isfinite(mask_add)) {
                    const uint64_t c = a->k_col_start + s + t;
                    const float score = zo_kvq_dot(q, kq + c * dh, ks + c * nb, bs, nb) * a->scale + mask_add;
                    scores[t] = score;
                    if (score > tile_max) tile_max = score;
                } else {
                    scores[t] = -INFINITY;
                }
            }
            if (tile_max == -INFINITY) continue;
            const float new_m = (m_val == -INFINITY) ? tile_max : zo_maxf(m_val, tile_max);
            const float alpha = (m_val == -INFINITY) ? 0.0f : expf(m_val - new_m);
            float tile_l = 0;
            for (int t = 0; t < BS_TILE; t++) {
                ws[t] = expf(scores[t] - new_m);
                tile_l += ws[t];
            }
            if (m_val != -INFINITY && alpha != 1.0f)
                for (uint64_t r = 0; r < dh; r++) acc[r] *= alpha;
            for (uint64_t bl = 0; bl < nb; bl++) { /* accumBatchI8F32 */
                float wsc[BS_TILE];
                for (int t = 0; t < BS_TILE; t++) wsc[t] = ws[t] * vs[(a->v_col_start + s + t) * nb + bl];
                for (uint64_t i = 0; i < bs; i++) {
                    float sum = acc[bl * bs + i];
                    for (int t = 0; t < BS_TILE; t++) sum = sum + wsc[t] * (float)vq[(a->v_col_start + s + t) * dh + bl * bs + i];
                    acc[bl * bs + i] = sum;
                }
            }
            l = l * alpha + tile_l;
            m_val = new_m;
        }
        for (; s < a->seq_kv; s++) { /* tail, one column at a time */
            const float mask_add = a->has_mask ? mask[mask_base + s * a->mask_rs] : 0.0f;
            if (!isfinite(mask_add)) continue;
            const uint64_t c = a->k_col_start + s;
            const float score = zo_kvq_dot(q, kq + c * dh, ks + c * nb, bs, nb) * a->scale + mask_add;
            if (!isfinite(score)) continue;
            const float new_m = zo_maxf(m_val, score);
            const float alpha = (m_val == -INFINITY) ? 0.0f : expf(m_val - new_m);
            const float w = expf(score - new_m);
            if (m_val != -INFINITY && alpha != 1.0f)
                for (uint64_t r = 0; r < dh; r++) acc[r] *= alpha;
            const uint64_t cv = a->v_col_start + s;
            for (uint64_t bl = 0; bl < nb; bl++) { /* accumI8F32 */
                const float wsc = w * vs[cv * nb + bl];
                for (uint64_t i = 0; i < bs; i++) acc[bl * bs + i] = acc[bl * bs + i] + wsc * (float)vq[cv * dh + bl * bs + i];
            }
            l = l * alpha + w;
            m_val = new_m;
        }
        const float inv_l = l > 0 ? 1.0f / l : 0.0f;
        float* dst = b[a->dst].ptr + a->dst_off + qi * a->dst_cs;
        for (uint64_t r = 0; r < dh; r++) dst[r] = acc[r] * inv_l;
    }
}

/* ───────────────────────────── dispatch ───────────────────────────── */

void zo_execute_op(const zo_buffer* buffers, const zo_qweight* qweights, const zgml_device_op* op) {
    switch (op->kind) {
        case ZGML_DOP_ELEMENTWISE: zo_elementwise(buffers, &op->u.elementwise); break;
        case ZGML_DOP_MATMUL: zo_matmul(buffers, &op->u.matmul); break;
        case ZGML_DOP_QMATMUL: zo_qmatmul(buffers, qweights, &op->u.qmatmul); break;
        case ZGML_DOP_SOFTMAX: zo_softmax(buffers, &op->u.softmax); break;
        case ZGML_DOP_LAYERNORM: zo_layernorm(buffers, &op->u.layernorm); break;
        case ZGML_DOP_RMSNORM: zo_rmsnorm(buffers, &op->u.rmsnorm); break;
        case ZGML_DOP_REDUCE: zo_reduce(buffers, &op->u.reduce); break;
        case ZGML_DOP_REPEAT: zo_repeat(buffers, &op->u.repeat); break;
        case ZGML_DOP_SLICE_ASSIGN: zo_slice_assign(buffers, &op->u.slice_assign); break;
        case ZGML_DOP_ROPE: zo_rope(buffers, &op->u.rope); break;
        case ZGML_DOP_ATTENTION: zo_attention(buffers, &op->u.attention); break;
        case ZGML_DOP_FUSED_ELEMENTWISE: zo_fused_elementwise(buffers, &op->u.fused_elementwise); break;
        case ZGML_DOP_KVQ_STORE: zo_kvq_store(buffers, &op->u.kvq_store); break;
        case ZGML_DOP_ATTENTION_KVQ: zo_attention_kvq(buffers, &op->u.attention_kvq); break;
        default: break;
    }
}

void zo_execute_ops(const zo_buffer* buffers, const zo_qweight* qweights, const zgml_device_op* ops,
                    uint64_t n_ops) {
    for (uint64_t i = 0; i < n_ops; i++) zo_execute_op(buffers, qweights, &ops[i]);
}

/* ───────────────────────────── compiled program (CpuBackend) ───────────────────────────── */

void zo_gguf_q4_0_to_int8(const uint8_t* raw, uint64_t n_elems, int8_t* data, float* scales);
void zo_gguf_q8_0_to_int8(const uint8_t* raw, uint64_t n_elems, int8_t* data, float* scales);
/* packed-GGUF pass-through form of a QuantizedWeightUpload (include/zgml_hip.h): 1 = Q4_0, 2 = Q8_0 */
static int zo_gguf_form(const zgml_qweight_upload* qw) {
    if (qw->scales || qw->scales_len || qw->block_size != 32 || !qw->data) return 0;
    const uint64_t n = qw->rows * qw->cols;
    if (!n || n % 32) return 0;
    if (qw->data_len == n / 32 * 18) return 1;
    if (qw->data_len == n / 32 * 34) return 2;
    return 0;
}

struct zo_program {
    uint64_t n_buffers;
    zo_buffer* buffers;
    uint64_t n_qweights;
    zo_qweight* qweights;
    int8_t** q_data;
    float** q_scales;
    uint64_t n_ops;
    zgml_device_op* ops;
    zgml_fused_step** steps; /* owned copies, one per op (NULL unless fused) */
    float** f16_shadow;      /* per buffer: packed f16 [K,N] image of a promoted matmul B (else NULL) */
    int8_t** t_data;         /* per qweight: prepareTransposed copy (zo_set_w8a8), else NULL */
    float** t_scales;
};

static int g_f16_dense = 0;
void zo_set_f16_dense(int on) { g_f16_dense = on; } /* consulted by zo_compile_program */
static int g_w8a8 = 0;
/* programs compiled afterwards carry prepareTransposed copies of their quantized weights, so M = 1 qmatmuls take
 * the W8A8 arm (quantizeInput + gemvRange over GemvPool) — CPU-baseline variant B3, BASELINE.md §3 */
void zo_set_w8a8(int on) { g_w8a8 = on; }
void zo_prepare_transposed(const int8_t* data, const float* scales, uint64_t K, uint64_t N, uint64_t bs, int8_t* t_data,
                           float* t_scales);

static void zo_program_run(zo_program* p, const zgml_device_op* ops, uint64_t n_ops) {
    for (uint64_t i = 0; i < n_ops; i++) {
        const zgml_device_op* op = &ops[i];
        if (op->kind == ZGML_DOP_MATMUL && p->f16_shadow[op->u.matmul.b])
            zo_matmul_f16(p->buffers, &op->u.matmul, p->f16_shadow[op->u.matmul.b]);
        else
            zo_execute_op(p->buffers, p->qweights, op);
    }
}

static void zo_copy_ops(zo_program* p, const zgml_device_op* ops, uint64_t n_ops) {
    for (uint64_t i = 0; i < p->n_ops; i++) free(p->steps[i]);
    free(p->steps);
    free(p->ops);
    p->n_ops = n_ops;
    p->ops = (zgml_device_op*)malloc(sizeof(zgml_device_op) * (n_ops ? n_ops : 1));
    p->steps = (zgml_fused_step**)calloc(n_ops ? n_ops : 1, sizeof(*p->steps));
    memcpy(p->ops, ops, sizeof(zgml_device_op) * n_ops);
    for (uint64_t i = 0; i < n_ops; i++) {
        if (ops[i].kind == ZGML_DOP_FUSED_ELEMENTWISE) {
            uint32_t ns = ops[i].u.fused_elementwise.n_steps;
            p->steps[i] = (zgml_fused_step*)malloc(sizeof(zgml_fused_step) * (ns ? ns : 1));
            memcpy(p->steps[i], ops[i].u.fused_elementwise.steps, sizeof(zgml_fused_step) * ns);
            p->ops[i].u.fused_elementwise.steps = p->steps[i];
        }
    }
}

static void zo_upload(zo_program* p, const zgml_program_io* io, uint64_t n) {
    for (uint64_t i = 0; i < n; i++)
        memcpy((char*)p->buffers[io[i].buf_idx].ptr + io[i].offset, io[i].host_ptr, io[i].size);
}

static void zo_download(zo_program* p, const zgml_program_io* io, uint64_t n) {
    for (uint64_t i = 0; i < n; i++)
        memcpy(io[i].host_ptr, (const char*)p->buffers[io[i].buf_idx].ptr + io[i].offset, io[i].size);
}

/* cpu.zig compileProgram :62-121: zeroed owned buffers (OwnedBufferTable.init,
 * reference.zig:81-97), initial uploads applied, qweights duplicated. */
zo_program* zo_compile_program(const zgml_device_program* prog) {
    zo_program* p = (zo_program*)calloc(1, sizeof(*p));
    p->n_buffers = prog->n_buffers;
    p->buffers = (zo_buffer*)calloc(p->n_buffers ? p->n_buffers : 1, sizeof(zo_buffer));
    for (uint64_t i = 0; i < p->n_buffers; i++) {
        uint64_t len = prog->buffer_sizes[i] ? prog->buffer_sizes[i] : 1;
        p->buffers[i].ptr = (float*)calloc(len, sizeof(float));
        p->buffers[i].len = len;
    }
    p->n_qweights = prog->n_qweights;
    p->qweights = (zo_qweight*)calloc(p->n_qweights ? p->n_qweights : 1, sizeof(zo_qweight));
    p->q_data = (int8_t**)calloc(p->n_qweights ? p->n_qweights : 1, sizeof(int8_t*));
    p->q_scales = (float**)calloc(p->n_qweights ? p->n_qweights : 1, sizeof(float*));
    for (uint64_t i = 0; i < p->n_qweights; i++) {
        const zgml_qweight_upload* qw = &prog->qweights[i];
        const int form = zo_gguf_form(qw);
        if (form) { /* packed pass-through: what the loader would have expanded (gguf_loader.zig:99-154) */
            const uint64_t n = qw->rows * qw->cols;
            p->q_data[i] = (int8_t*)malloc(n);
            p->q_scales[i] = (float*)malloc(sizeof(float) * (n / 32));
            if (form == 1)
                zo_gguf_q4_0_to_int8((const uint8_t*)qw->data, n, p->q_data[i], p->q_scales[i]);
            else
                zo_gguf_q8_0_to_int8((const uint8_t*)qw->data, n, p->q_data[i], p->q_scales[i]);
            p->qweights[i].data = p->q_data[i];
            p->qweights[i].scales = p->q_scales[i];
            p->qweights[i].block_size = 32;
            continue;
        }
        p->q_data[i] = (int8_t*)malloc(qw->data_len ? qw->data_len : 1);
        memcpy(p->q_data[i], qw->data, qw->data_len);
        p->q_scales[i] = (float*)malloc(sizeof(float) * (qw->scales_len ? qw->scales_len : 1));
        memcpy(p->q_scales[i], qw->scales, sizeof(float) * qw->scales_len);
        p->qweights[i].data = p->q_data[i];
        p->qweights[i].scales = p->q_scales[i];
        p->qweights[i].block_size = qw->block_size;
    }
    p->t_data = (int8_t**)calloc(p->n_qweights ? p->n_qweights : 1, sizeof(int8_t*));
    p->t_scales = (float**)calloc(p->n_qweights ? p->n_qweights : 1, sizeof(float*));
    for (uint64_t i = 0; g_w8a8 && i < p->n_qweights; i++) {
        const uint64_t K = prog->qweights[i].rows, N = prog->qweights[i].cols, bs = p->qweights[i].block_size;
        if (!K || !N || !bs) continue;
        p->t_data[i] = (int8_t*)malloc(N * K);
        p->t_scales[i] = (float*)malloc(sizeof(float) * N * ((K + bs - 1) / bs));
        zo_prepare_transposed(p->q_data[i], p->q_scales[i], K, N, bs, p->t_data[i], p->t_scales[i]);
        p->qweights[i].t_data = p->t_data[i];
        p->qweights[i].t_scales = p->t_scales[i];
    }
    zo_copy_ops(p, prog->ops, prog->n_ops);
    zo_upload(p, prog->initial_uploads, prog->n_initial_uploads);
    p->f16_shadow = (float**)calloc(p->n_buffers ? p->n_buffers : 1, sizeof(float*));
    if (g_f16_dense) { /* wgpu.zig:1071-1104: first matmul user's geometry packs the buffer */
        for (uint64_t i = 0; i < prog->n_ops; i++) {
            if (prog->ops[i].kind != ZGML_DOP_MATMUL) continue;
            const zgml_op_matmul* m = &prog->ops[i].u.matmul;
            int has_upload = 0;
            for (uint64_t u = 0; u < prog->n_initial_uploads; u++)
                if (prog->initial_uploads[u].buf_idx == m->b) has_upload = 1;
            if (!has_upload || p->f16_shadow[m->b]) continue;
            const zgml_matmul_geom* g = &m->geom;
            float* sh = (float*)malloc(sizeof(float) * (g->K * g->N ? g->K * g->N : 1));
            const float* src = p->buffers[m->b].ptr;
            for (uint64_t k = 0; k < g->K; k++)
                for (uint64_t n = 0; n < g->N; n++)
                    sh[k * g->N + n] = zo_f16_to_f32(zo_f32_to_f16(src[g->b_offset + k * g->b_row_stride + n * g->b_col_stride]));
            p->f16_shadow[m->b] = sh;
        }
    }
    return p;
}

void zo_refresh_program(zo_program* p, const zgml_device_op* ops, uint64_t n_ops) { zo_copy_ops(p, ops, n_ops); }

void zo_execute_program(zo_program* p, const zgml_program_io* inputs, uint64_t n_inputs,
                        const zgml_program_io* outputs, uint64_t n_outputs) {
    zo_upload(p, inputs, n_inputs);
    zo_program_run(p, p->ops, p->n_ops);
    zo_download(p, outputs, n_outputs);
}

void zo_program_execute_range(zo_program* p, uint64_t first, uint64_t count) {
    if (first > p->n_ops) return;
    if (first + count > p->n_ops) count = p->n_ops - first;
    zo_program_run(p, p->ops + first, count);
}

void zo_program_upload(zo_program* p, const zgml_program_io* io, uint64_t n) { zo_upload(p, io, n); }
void zo_program_download(zo_program* p, const zgml_program_io* io, uint64_t n) { zo_download(p, io, n); }

void zo_free_program(zo_program* p) {
    if (!p) return;
    for (uint64_t i = 0; i < p->n_buffers; i++) free(p->buffers[i].ptr);
    if (p->f16_shadow)
        for (uint64_t i = 0; i < p->n_buffers; i++) free(p->f16_shadow[i]);
    free(p->f16_shadow);
    free(p->buffers);
    for (uint64_t i = 0; i < p->n_qweights; i++) {
        free(p->q_data[i]);
        free(p->q_scales[i]);
        if (p->t_data) free(p->t_data[i]);
        if (p->t_scales) free(p->t_scales[i]);
    }
    free(p->t_data);
    free(p->t_scales);
    free(p->q_data);
    free(p->q_scales);
    free(p->qweights);
    for (uint64_t i = 0; i < p->n_ops; i++) free(p->steps[i]);
    free(p->steps);
    free(p->ops);
    free(p);
}

float* zo_program_buffer(zo_program* p, uint16_t idx, uint64_t* len_out) {
    if (idx >= p->n_buffers) return NULL;
    if (len_out) *len_out = p->buffers[idx].len;
    return p->buffers[idx].ptr;
}

/* DeviceProgram.isSupportedBy + opBuffersValid, src/backend.zig:277-325 */
static int zo_elementwise_op_ok(uint32_t op) { return op >= ZGML_OP_ADD && op <= ZGML_OP_GELU; }

int zo_program_supported(const zgml_device_program* pr, int fused_elementwise, int max_fused_steps,
                         int64_t attn_max_seq_kv, int64_t attn_max_d_head) {
    if ((uint64_t)pr->n_buffers != pr->n_buffer_sizes) return 0;
#define HAS(i) ((uint64_t)(i) < pr->n_buffer_sizes)
    for (uint64_t i = 0; i < pr->n_ops; i++) {
        const zgml_device_op* op = &pr->ops[i];
        switch (op->kind) {
            case ZGML_DOP_ELEMENTWISE:
                if (!zo_elementwise_op_ok(op->u.elementwise.op)) return 0;
                if (!HAS(op->u.elementwise.dst) || !HAS(op->u.elementwise.src0) || !HAS(op->u.elementwise.src1))
                    return 0;
                break;
            case ZGML_DOP_MATMUL:
                if (!HAS(op->u.matmul.dst) || !HAS(op->u.matmul.a) || !HAS(op->u.matmul.b)) return 0;
                break;
            case ZGML_DOP_QMATMUL: {
                const zgml_op_qmatmul* q = &op->u.qmatmul;
                if (!HAS(q->dst) || !HAS(q->input)) return 0;
                if ((uint64_t)q->weight_idx >= pr->n_qweights) return 0;
                const zgml_qweight_upload* qw = &pr->qweights[q->weight_idx];
                if (qw->block_size == 0) return 0;
                if (qw->rows != q->K || qw->cols != q->N) return 0;
                uint64_t n_elems = (uint64_t)q->K * q->N;
                uint64_t n_blocks = (n_elems + qw->block_size - 1) / qw->block_size;
                if (zo_gguf_form(qw)) break; /* packed-GGUF pass-through (include/zgml_hip.h) */
                if (qw->data_len < n_elems || qw->scales_len < n_blocks) return 0;
                break;
            }
            case ZGML_DOP_SOFTMAX:
            case ZGML_DOP_LAYERNORM:
            case ZGML_DOP_RMSNORM:
                if (!HAS(op->u.softmax.dst) || !HAS(op->u.softmax.src)) return 0;
                break;
            case ZGML_DOP_REDUCE:
                if (op->u.reduce.op != ZGML_OP_SUM && op->u.reduce.op != ZGML_OP_MAX) return 0;
                if (!HAS(op->u.reduce.dst) || !HAS(op->u.reduce.src)) return 0;
                break;
            case ZGML_DOP_REPEAT:
                if (!HAS(op->u.repeat.dst) || !HAS(op->u.repeat.src)) return 0;
                break;
            case ZGML_DOP_SLICE_ASSIGN:
                if (!HAS(op->u.slice_assign.dst) || !HAS(op->u.slice_assign.src)) return 0;
                break;
            case ZGML_DOP_ROPE:
                if (!HAS(op->u.rope.dst) || !HAS(op->u.rope.src) || !HAS(op->u.rope.cos_sin)) return 0;
                break;
            case ZGML_DOP_ATTENTION: {
                const zgml_op_attention* a = &op->u.attention;
                if (attn_max_seq_kv >= 0 && (int64_t)a->seq_kv > attn_max_seq_kv) return 0;
                if (attn_max_d_head >= 0 && (int64_t)a->d_head > attn_max_d_head) return 0;
                if (!HAS(a->dst) || !HAS(a->q) || !HAS(a->k) || !HAS(a->v) || !HAS(a->mask)) return 0;
                break;
            }
            case ZGML_DOP_KVQ_STORE: { /* extension: the reference's CpuBackend has no such op; shape checks only */
                const zgml_op_kvq_store* st = &op->u.kvq_store;
                if (!HAS(st->cache) || !HAS(st->src) || !st->block_size || st->d_head % st->block_size) return 0;
                if (((uint64_t)st->n_cols * st->d_head) % 4) return 0;
                break;
            }
            case ZGML_DOP_ATTENTION_KVQ: {
                const zgml_op_attention_kvq* a = &op->u.attention_kvq;
                if (!HAS(a->dst) || !HAS(a->q) || !HAS(a->k) || !HAS(a->v) || !HAS(a->mask)) return 0;
                if (!a->block_size || a->d_head % a->block_size || a->d_head > 512) return 0;
                break;
            }
            case ZGML_DOP_FUSED_ELEMENTWISE: {
                const zgml_op_fused_elementwise* fe = &op->u.fused_elementwise;
                if (!fused_elementwise) return 0;
                if (max_fused_steps >= 0 && fe->n_steps > (uint32_t)max_fused_steps) return 0;
                if (!HAS(fe->dst) || !HAS(fe->src)) return 0;
                for (uint32_t s = 0; s < fe->n_steps; s++) {
                    if (!zo_elementwise_op_ok(fe->steps[s].op)) return 0;
                    int binary = fe->steps[s].op == ZGML_OP_ADD || fe->steps[s].op == ZGML_OP_MUL;
                    if (binary && !HAS(fe->steps[s].secondary_buf)) return 0;
                }
                break;
            }
            default: return 0;
        }
    }
#undef HAS
    return 1;
}

/* ───────────────────────────── quantisation ───────────────────────────── */

static inline float zo_clamp127(float q) { return fmaxf(-127.0f, fminf(q, 127.0f)); }

void zo_quantize_from_slice(const float* weights, uint64_t rows, uint64_t cols, uint64_t bs, int8_t* data,
                            float* scales) {
    uint64_t n_elems = rows * cols, n_blocks = (n_elems + bs - 1) / bs;
    for (uint64_t b = 0; b < n_blocks; b++) {
        uint64_t start = b * bs, end = start + bs < n_elems ? start + bs : n_elems;
        float max_abs = 0;
        for (uint64_t j = start; j < end; j++) {
            float a = fabsf(weights[j]);
            if (a > max_abs) max_abs = a;
        }
        float scale = max_abs > 0 ? max_abs / 127.0f : 1.0f;
        float inv_scale = max_abs > 0 ? 127.0f / max_abs : 0.0f;
        scales[b] = scale;
        for (uint64_t j = start; j < end; j++)
            data[j] = (int8_t)zo_clamp127(weights[j] * inv_scale); /* @intFromFloat: truncation */
    }
}

void zo_prepare_transposed(const int8_t* data, const float* scales, uint64_t K, uint64_t N, uint64_t bs,
                           int8_t* t_data, float* t_scales) {
    uint64_t bpr = (K + bs - 1) / bs;
    for (uint64_t n = 0; n < N; n++) {
        for (uint64_t b = 0; b < bpr; b++) {
            uint64_t k0 = b * bs, k1 = k0 + bs < K ? k0 + bs : K;
            float max_abs = 0;
            for (uint64_t k = k0; k < k1; k++) {
                uint64_t flat = k * N + n;
                float val = (float)data[flat] * scales[flat / bs];
                float a = fabsf(val);
                if (a > max_abs) max_abs = a;
            }
            float scale = max_abs > 0 ? max_abs / 127.0f : 1.0f;
            float inv_scale = max_abs > 0 ? 127.0f / max_abs : 0.0f;
            t_scales[n * bpr + b] = scale;
            for (uint64_t k = k0; k < k1; k++) {
                uint64_t flat = k * N + n;
                float val = (float)data[flat] * scales[flat / bs];
                t_data[n * K + k] = (int8_t)zo_clamp127(val * inv_scale);
            }
        }
    }
}

void zo_quantize_input(const float* input, uint64_t K, uint64_t bs, int8_t* inp_q, float* inp_scales) {
    uint64_t bpr = (K + bs - 1) / bs;
    for (uint64_t b = 0; b < bpr; b++) {
        uint64_t k0 = b * bs, k1 = k0 + bs < K ? k0 + bs : K;
        float max_abs = 0;
        for (uint64_t k = k0; k < k1; k++) {
            float a = fabsf(input[k]);
            if (a > max_abs) max_abs = a;
        }
        float scale = max_abs > 0 ? max_abs / 127.0f : 1.0f;
        float inv_scale = max_abs > 0 ? 127.0f / max_abs : 0.0f;
        inp_scales[b] = scale;
        for (uint64_t k = k0; k < k1; k++) inp_q[k] = (int8_t)zo_clamp127(input[k] * inv_scale);
    }
}

/* The sdot lanes sum exactly in int32, so any summation order gives the same integer; the f32
 * combine order (blocks ascending, acc += f32(int) * (s_x*s_w)) is kept (quant.zig:382-409). */
static void zo_gemv_range_scalar(const int8_t* t_d, const float* t_s, const int8_t* inp_q, const float* inp_scales,
                                 float* dst, uint64_t n_start, uint64_t n_end, uint64_t K, uint64_t bs);

/* The reference's kernel is AArch64 `sdot` (16-lane int8 dot into int32, 4 rows unrolled: quant.zig:358-440). The x86 arm of
 * the same arithmetic: AVX-512 VNNI `vpdpbusd` on one 32-byte block of 8 rows at a time. vpdpbusd
 * multiplies UNSIGNED by signed bytes, so the sign of w moves to x first: |w| * (x * sgn w) = w * x. |w| is taken as an
 * UNSIGNED byte, so w = -128 (raw GGUF Q8_0 bytes may hold it; prepareTransposed's own output is clamped to +-127) is 128,
 * not a wrapped -128 (with the roles the other way round, _mm256_sign_epi8(w, x) kept w = -128 at -128 for
 * x < 0 and the product's sign flipped). x * sgn w only wraps for x = -128 with w < 0; quantizeInput clamps x to +-127, and
 * a block of an arbitrary caller that does hold -128 takes the scalar loop. The eight 8-lane partial vectors are reduced to one int32 per row by a horizontal-add tree
 * — the integers are exact, so this is the scalar loop's integer — and the f32 combine (convert, multiply by s_x * s_w, add,
 * blocks ascending, no FMA) is the scalar loop's, lane by lane: bit-identical results (tests/test_oracle_w8a8.py). */
#if defined(__x86_64__)
#include <immintrin.h>
static int zo_vnni = -1; /* -1: not probed yet; ZGML_ORACLE_VNNI=0 forces the scalar loop */
static int zo_have_vnni(void) {
    if (zo_vnni < 0) {
        const char* e = getenv("ZGML_ORACLE_VNNI");
        zo_vnni = (e && atoi(e) == 0) ? 0 : (__builtin_cpu_supports("avx512vnni") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx2"));
    }
    return zo_vnni;
}
__attribute__((target("avx2,avx512f,avx512vl,avx512vnni"))) static void zo_gemv_range_vnni(const int8_t* t_d, const float* t_s, const int8_t* inp_q,
                                                                                          const float* inp_scales, float* dst, uint64_t n_start,
                                                                                          uint64_t n_end, uint64_t K) {
    const uint64_t bpr = K / 32; /* caller: bs == 32, K % 32 == 0 */
    uint64_t n = n_start;
    for (; n + 8 <= n_end; n += 8) {
        __m256 acc = _mm256_setzero_ps();
        for (uint64_t b = 0; b < bpr; b++) {
            const __m256i x = _mm256_loadu_si256((const __m256i*)(inp_q + 32 * b));
            __m256i ia;
            if (_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, _mm256_set1_epi8((char)-128))) != 0) { /* x = -128 somewhere: the scalar integers */
                int32_t iv[8];
                for (int r = 0; r < 8; r++) {
                    const int8_t* w = t_d + (n + r) * K + 32 * b;
                    int32_t t = 0;
                    for (int k = 0; k < 32; k++) t += (int32_t)inp_q[32 * b + k] * (int32_t)w[k];
                    iv[r] = t;
                }
                ia = _mm256_loadu_si256((const __m256i*)iv);
            } else {
                __m256i v[8];
                for (int r = 0; r < 8; r++) {
                    const __m256i w = _mm256_loadu_si256((const __m256i*)(t_d + (n + r) * K + 32 * b));
                    v[r] = _mm256_dpbusd_epi32(_mm256_setzero_si256(), _mm256_abs_epi8(w), _mm256_sign_epi8(x, w)); /* |w| (u8: -128 -> 128) x (x sgn w) */
                }
                const __m256i h01 = _mm256_hadd_epi32(v[0], v[1]), h23 = _mm256_hadd_epi32(v[2], v[3]);
                const __m256i h45 = _mm256_hadd_epi32(v[4], v[5]), h67 = _mm256_hadd_epi32(v[6], v[7]);
                const __m256i q0 = _mm256_hadd_epi32(h01, h23), q1 = _mm256_hadd_epi32(h45, h67); /* rows 0-3 / 4-7, low | high half sums */
                ia = _mm256_add_epi32(_mm256_permute2x128_si256(q0, q1, 0x20), _mm256_permute2x128_si256(q0, q1, 0x31));
            }
            const float sx = inp_scales[b];
            const __m256 comb = _mm256_mul_ps(_mm256_set1_ps(sx), _mm256_set_ps(t_s[(n + 7) * bpr + b], t_s[(n + 6) * bpr + b], t_s[(n + 5) * bpr + b],
                                                                                t_s[(n + 4) * bpr + b], t_s[(n + 3) * bpr + b], t_s[(n + 2) * bpr + b],
                                                                                t_s[(n + 1) * bpr + b], t_s[(n + 0) * bpr + b]));
            acc = _mm256_add_ps(acc, _mm256_mul_ps(_mm256_cvtepi32_ps(ia), comb));
        }
        _mm256_storeu_ps(dst + n, acc);
    }
    if (n < n_end) zo_gemv_range_scalar(t_d, t_s, inp_q, inp_scales, dst, n, n_end, K, 32);
}
#endif

void zo_gemv_range(const int8_t* t_d, const float* t_s, const int8_t* inp_q, const float* inp_scales,
                   float* dst, uint64_t n_start, uint64_t n_end, uint64_t K, uint64_t bs) {
#if defined(__x86_64__)
    if (bs == 32 && K % 32 == 0 && zo_have_vnni()) {
        zo_gemv_range_vnni(t_d, t_s, inp_q, inp_scales, dst, n_start, n_end, K);
        return;
    }
#endif
    zo_gemv_range_scalar(t_d, t_s, inp_q, inp_scales, dst, n_start, n_end, K, bs);
}
/* 1: the AVX-512 VNNI arm is in use on this host */
int zo_gemv_uses_vnni(void) {
#if defined(__x86_64__)
    return zo_have_vnni();
#else
    return 0;
#endif
}
void zo_set_vnni(int on) { /* tests: 0 = scalar loop, 1 = VNNI if the host has it, -1 = probe again */
#if defined(__x86_64__)
    zo_vnni = on < 0 ? -1 : (on && __builtin_cpu_supports("avx512vnni") && __builtin_cpu_supports("avx512vl"));
#else
    (void)on;
#endif
}

static void zo_gemv_range_scalar(const int8_t* t_d, const float* t_s, const int8_t* inp_q, const float* inp_scales,
                                 float* dst, uint64_t n_start, uint64_t n_end, uint64_t K, uint64_t bs) {
    uint64_t bpr = (K + bs - 1) / bs;
    for (uint64_t n = n_start; n < n_end; n++) {
        float acc = 0;
        const int8_t* w = t_d + n * K;
        for (uint64_t b = 0; b < bpr; b++) {
            uint64_t k0 = b * bs, k1 = k0 + bs < K ? k0 + bs : K;
            float combined = inp_scales[b] * t_s[n * bpr + b];
            int32_t ia = 0;
            for (uint64_t k = k0; k < k1; k++) ia += (int32_t)inp_q[k] * (int32_t)w[k];
            acc += (float)ia * combined;
        }
        dst[n] = acc;
    }
}

typedef struct {
    const int8_t* t_d;
    const float* t_s;
    const int8_t* inp_q;
    const float* inp_scales;
    float* dst;
    uint64_t K, bs;
} zo_gemv_ctx;

static void zo_gemv_fn(void* vctx, uint64_t n0, uint64_t n1) {
    zo_gemv_ctx* c = (zo_gemv_ctx*)vctx;
    zo_gemv_range(c->t_d, c->t_s, c->inp_q, c->inp_scales, c->dst, n0, n1, c->K, c->bs);
}

void zo_gemv_pool_dispatch(const int8_t* t_d, const float* t_s, const int8_t* inp_q, const float* inp_scales,
                           float* dst, uint64_t N, uint64_t K, uint64_t bs, int n_workers) {
    const uint64_t min_work_per_thread = 1024 * 1024; /* quant.zig:150 */
    uint64_t useful = (N * K) / min_work_per_thread;
    if (useful < 1) useful = 1;
    if (n_workers > 16) n_workers = 16; /* GemvPool.max_workers, quant.zig:27 */
    int n_active = (int)(useful < (uint64_t)n_workers ? useful : (uint64_t)n_workers);
    zo_gemv_ctx c = {t_d, t_s, inp_q, inp_scales, dst, K, bs};
    if (n_active <= 1) {
        zo_gemv_range(t_d, t_s, inp_q, inp_scales, dst, 0, N, K, bs);
        return;
    }
    zo_parallel_for(N, 4, n_active, zo_gemv_fn, &c); /* chunk rounded up to 4, quant.zig:164 */
}

void zo_dequantize(const int8_t* data, const float* scales, uint64_t n_elems, uint64_t bs, float* out) {
    for (uint64_t i = 0; i < n_elems; i++) out[i] = (float)data[i] * scales[i / bs];
}

/* ───────────────────────────── GGUF blocks ───────────────────────────── */

float zo_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000) << 16, exp = (h >> 10) & 0x1F, man = h & 0x3FF, bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal */
            int e = -1;
            do {
                e++;
                man <<= 1;
            } while (!(man & 0x400));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FF) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

uint16_t zo_f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000;
    int32_t exp = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
    uint32_t man = x & 0x7FFFFF;
    if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00 | (man ? 0x200 : 0));
    if (exp >= 31) return (uint16_t)(sign | 0x7C00);
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;
        man |= 0x800000;
        uint32_t shift = (uint32_t)(14 - exp);
        uint32_t half = man >> shift, rem = man & ((1u << shift) - 1), mid = 1u << (shift - 1);
        if (rem > mid || (rem == mid && (half & 1))) half++;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)exp << 10) | (man >> 13), rem = man & 0x1FFF;
    if (rem > 0x1000 || (rem == 0x1000 && (half & 1))) half++;
    return (uint16_t)(sign | half);
}

static inline float zo_block_scale(const uint8_t* blk) { return zo_f16_to_f32((uint16_t)(blk[0] | (blk[1] << 8))); }

/* interleaved nibble order: element i -> byte i/2, even = low nibble, odd = high (F3) */
void zo_gguf_q4_0_to_int8(const uint8_t* raw, uint64_t n_elems, int8_t* data, float* scales) {
    uint64_t n_blocks = (n_elems + 31) / 32;
    for (uint64_t b = 0; b < n_blocks; b++) {
        const uint8_t* blk = raw + b * 18;
        scales[b] = zo_block_scale(blk);
        uint64_t elems = n_elems - b * 32 < 32 ? n_elems - b * 32 : 32;
        for (uint64_t i = 0; i < elems; i++) {
            uint8_t byte = blk[2 + i / 2];
            uint8_t nib = (i % 2 == 0) ? (byte & 0x0F) : (byte >> 4);
            data[b * 32 + i] = (int8_t)((int16_t)nib - 8);
        }
    }
}

void zo_gguf_q8_0_to_int8(const uint8_t* raw, uint64_t n_elems, int8_t* data, float* scales) {
    uint64_t n_blocks = (n_elems + 31) / 32;
    for (uint64_t b = 0; b < n_blocks; b++) {
        const uint8_t* blk = raw + b * 34;
        scales[b] = zo_block_scale(blk);
        uint64_t elems = n_elems - b * 32 < 32 ? n_elems - b * 32 : 32;
        for (uint64_t i = 0; i < elems; i++) data[b * 32 + i] = (int8_t)blk[2 + i];
    }
}

void zo_gguf_dequant_q4_0(float* dst, const uint8_t* raw, uint64_t n_elems) {
    uint64_t n_blocks = (n_elems + 31) / 32;
    for (uint64_t b = 0; b < n_blocks; b++) {
        const uint8_t* blk = raw + b * 18;
        float scale = zo_block_scale(blk);
        uint64_t elems = n_elems - b * 32 < 32 ? n_elems - b * 32 : 32;
        for (uint64_t i = 0; i < elems; i++) {
            uint8_t byte = blk[2 + i / 2];
            uint8_t nib = (i % 2 == 0) ? (byte & 0x0F) : (byte >> 4);
            dst[b * 32 + i] = (float)(int8_t)((int16_t)nib - 8) * scale;
        }
    }
}

void zo_gguf_dequant_q8_0(float* dst, const uint8_t* raw, uint64_t n_elems) {
    uint64_t n_blocks = (n_elems + 31) / 32;
    for (uint64_t b = 0; b < n_blocks; b++) {
        const uint8_t* blk = raw + b * 34;
        float scale = zo_block_scale(blk);
        uint64_t elems = n_elems - b * 32 < 32 ? n_elems - b * 32 : 32;
        for (uint64_t i = 0; i < elems; i++) dst[b * 32 + i] = (float)(int8_t)blk[2 + i] * scale;
    }
}

/* ───────────────────────────── nn helpers ───────────────────────────── */

void zo_rope_tables(uint64_t d, uint64_t max_seq, float base, float* cos_table, float* sin_table) {
    for (uint64_t pos = 0; pos < max_seq; pos++) {
        for (uint64_t i = 0; i < d / 2; i++) {
            float p = (float)pos, dim = (float)(2 * i), dm = (float)d;
            float freq = p / powf(base, dim / dm);
            float c = cosf(freq), s = sinf(freq);
            cos_table[pos * d + i] = c;
            cos_table[pos * d + i + d / 2] = c;
            sin_table[pos * d + i] = s;
            sin_table[pos * d + i + d / 2] = s;
        }
    }
}

int64_t zo_argmax(const float* v, uint64_t n) {
    if (n == 0) return -1;
    uint64_t best = 0;
    float best_val = v[0];
    for (uint64_t c = 1; c < n; c++) {
        if (v[c] > best_val) {
            best_val = v[c];
            best = c;
        }
    }
    return (int64_t)best;
}

/* ── ctx-taking wrappers with the exact signatures of the zgml_hip_* entry points, so the host
 *    session code (zgml_amd/host) can be driven by the oracle in parity tests ─────────────── */
void* zo_vt_compile_program(void* ctx, const zgml_device_program* program) {
    (void)ctx;
    return zo_compile_program(program);
}
void zo_vt_refresh_program(void* ctx, void* handle, const zgml_device_op* ops, uint64_t n_ops) {
    (void)ctx;
    zo_refresh_program((zo_program*)handle, ops, n_ops);
}
void zo_vt_execute_program(void* ctx, void* handle, const zgml_program_io* inputs, uint64_t n_inputs,
                           const zgml_program_io* outputs, uint64_t n_outputs) {
    (void)ctx;
    zo_execute_program((zo_program*)handle, inputs, n_inputs, outputs, n_outputs);
}
void zo_vt_free_program(void* ctx, void* handle) {
    (void)ctx;
    zo_free_program((zo_program*)handle);
}
