/*
 * zgml_hip.h — C ABI of the MI355X (gfx950) backend for zgml's forward-inference path.
 *
 * This header is the drop-in boundary. Every entry point replaces one slot of the
 * reference's backend plugin surface (all citations relative to the zgml tree):
 *
 *   zgml_hip_create / zgml_hip_destroy   <-> Backend.ctx lifetime           src/backend.zig:330-336
 *   zgml_hip_dense_matmul_f32            <-> VTable.dense_matmul_f32        src/backend.zig:341
 *   zgml_hip_compile_program             <-> VTable.compile_program         src/backend.zig:343
 *   zgml_hip_refresh_program             <-> VTable.refresh_program         src/backend.zig:345
 *   zgml_hip_execute_program             <-> VTable.execute_program         src/backend.zig:347
 *   zgml_hip_free_program                <-> VTable.free_program            src/backend.zig:349
 *   zgml_hip_get_runtime_profile         <-> VTable.get_runtime_profile     src/backend.zig:351
 *   zgml_hip_capabilities                <-> Backend.capabilities           src/backend.zig:14-141
 *   zgml_hip_program_supported           <-> DeviceProgram.isSupportedBy    src/backend.zig:277-325
 *
 * The structs below mirror the Zig types field for field (zgml_device_op <-> DeviceOp
 * src/backend.zig:179-249, zgml_program_io <-> ProgramIO :252-257, zgml_qweight_upload <->
 * QuantizedWeightUpload :260-266, zgml_device_program <-> DeviceProgram :270-275). Zig slices
 * become (pointer, length) pairs and the tagged union becomes `kind` + a C union, so a Zig
 * adapter can fill them with `extern struct`s (INTEGRATION.md shows it).
 *
 * Plain C: no C++ types, no torch types, no HIP types in any signature. Offsets and strides of
 * device ops are in f32 ELEMENTS; zgml_program_io offsets/sizes are in BYTES; buffer_sizes are
 * f32 element counts (same units as the reference).
 *
 * Extension entry points (zgml_hip_*_ext, zgml_hip_qmatvec_bench_*) have no reference
 * counterpart; they exist for measurement and for the row-sharded multi-GPU path and are marked
 * as such below.
 */
#ifndef ZGML_HIP_H
#define ZGML_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZGML_HIP_ABI_VERSION 1

/* ── Op enum ordinals: src/op.zig:11-62 (graph IR `Op`) ─────────────────────────────────── */
enum {
    ZGML_OP_NONE = 0,
    ZGML_OP_VIEW = 1,
    ZGML_OP_RESHAPE = 2,
    ZGML_OP_TRANSPOSE = 3,
    ZGML_OP_PERMUTE = 4,
    ZGML_OP_AS_STRIDED = 5,
    ZGML_OP_BROADCAST_TO = 6,
    ZGML_OP_ADD = 7,
    ZGML_OP_MUL = 8,
    ZGML_OP_NEG = 9,
    ZGML_OP_ABS = 10,
    ZGML_OP_SGN = 11,
    ZGML_OP_STEP = 12,
    ZGML_OP_RELU = 13,
    ZGML_OP_SQRT = 14,
    ZGML_OP_RECIP = 15,
    ZGML_OP_EXP = 16,
    ZGML_OP_LOG = 17,
    ZGML_OP_GELU = 18,
    ZGML_OP_SUM = 19,
    ZGML_OP_MAX = 20,
    ZGML_OP_REPEAT = 21
};

/* ── DeviceOp tags, in the declaration order of the Zig union (src/backend.zig:179-249) ──── */
enum {
    ZGML_DOP_ELEMENTWISE = 0,
    ZGML_DOP_MATMUL = 1,
    ZGML_DOP_QMATMUL = 2,
    ZGML_DOP_SOFTMAX = 3,
    ZGML_DOP_LAYERNORM = 4,
    ZGML_DOP_RMSNORM = 5,
    ZGML_DOP_REDUCE = 6,
    ZGML_DOP_REPEAT = 7,
    ZGML_DOP_SLICE_ASSIGN = 8,
    ZGML_DOP_ROPE = 9,
    ZGML_DOP_ATTENTION = 10,
    ZGML_DOP_FUSED_ELEMENTWISE = 11,
    ZGML_DOP_COUNT = 12,
    /* Extension kinds (no reference DeviceOp): the quantised KV cache of
     * src/quant.zig:645-1091 as device ops. Never counted in the reference-shaped profile arrays. */
    ZGML_DOP_KVQ_STORE = 13,     /* QuantizedKVCache.storeColumn */
    ZGML_DOP_ATTENTION_KVQ = 14  /* attentionQuantized */
};

/* MatMulGeometry, src/backend.zig:146-158 (usize -> uint64_t). */
typedef struct zgml_matmul_geom {
    uint64_t M, N, K;
    uint64_t a_row_stride, a_col_stride;
    uint64_t b_row_stride, b_col_stride;
    uint64_t a_offset, b_offset;
    uint64_t dst_offset, dst_row_stride;
} zgml_matmul_geom;

/* FusedEwStep, src/backend.zig:170-175. */
typedef struct zgml_fused_step {
    uint32_t op;               /* ZGML_OP_* */
    uint8_t is_swapped;        /* chain value sits in the src1 position */
    uint8_t _pad;
    uint16_t secondary_buf;    /* external operand of a binary step */
    uint32_t secondary_offset; /* elements */
} zgml_fused_step;

typedef struct zgml_op_elementwise {
    uint32_t op;
    uint16_t dst, src0, src1, _pad;
    uint32_t n, dst_offset, src0_offset, src1_offset;
} zgml_op_elementwise;

typedef struct zgml_op_matmul {
    uint16_t dst, a, b, _pad;
    zgml_matmul_geom geom;
} zgml_op_matmul;

typedef struct zgml_op_qmatmul {
    uint16_t dst, input, weight_idx, _pad;
    uint32_t M, N, K;
    uint32_t input_offset, input_row_stride; /* stride 0 => K */
    uint32_t dst_offset, dst_row_stride;     /* stride 0 => N */
} zgml_op_qmatmul;

typedef struct zgml_op_rowwise { /* softmax / layernorm / rmsnorm */
    uint16_t dst, src;
    uint32_t rows, cols;
    float eps; /* ignored by softmax */
    uint32_t src_offset, dst_offset;
} zgml_op_rowwise;

typedef struct zgml_op_reduce {
    uint32_t op; /* ZGML_OP_SUM or ZGML_OP_MAX */
    uint16_t dst, src;
    uint32_t n_out, reduce_size;
    uint32_t src_offset, dst_offset;
} zgml_op_reduce;

typedef struct zgml_op_repeat {
    uint16_t dst, src;
    uint32_t n;
    uint32_t src_ne[4], dst_ne[4], src_strides[4], dst_strides[4];
    uint32_t src_offset, dst_offset;
} zgml_op_repeat;

typedef struct zgml_op_slice_assign {
    uint16_t dst, src;
    uint32_t rows, cols;
    uint32_t dst_base_offset;
    uint32_t dst_offset; /* dynamic: dst_base_offset + pos * patch_stride */
    uint32_t dst_row_stride, dst_col_stride;
    uint32_t src_offset, src_row_stride, src_col_stride;
    uint32_t patch_stride; /* 0 => static */
} zgml_op_slice_assign;

typedef struct zgml_op_rope {
    uint16_t dst, src, cos_sin, _pad;
    uint32_t half_d, seq_len;
    uint32_t src_off, cs_off, dst_off;
    uint32_t src_rs, src_cs, cs_cs;
} zgml_op_rope;

typedef struct zgml_op_attention {
    uint16_t dst, q, k, v, mask;
    uint8_t has_mask, _pad;
    uint32_t d_head, seq_q;
    uint32_t seq_kv; /* dynamic */
    float scale;
    uint32_t q_off, k_off, v_off, mask_off, dst_off;
    uint32_t q_rs, q_cs, k_rs, k_cs, v_rs, v_cs, mask_rs, mask_cs, dst_rs, dst_cs;
} zgml_op_attention;

/* A quantised KV cache lives in ONE program buffer (f32-element sized like every buffer):
 *   int8 q_data[n_cols * d_head]            column c at byte offset c * d_head
 *   f32  scales[n_cols * d_head/block_size] at element offset n_cols * d_head / 4, column-major
 * so buffer_sizes[cache] >= n_cols*d_head/4 + n_cols*d_head/block_size (n_cols*d_head % 4 == 0). */

/* QuantizedKVCache.storeColumn (src/quant.zig:687-699 -> quantizeInput :320-341): quantise the d_head
 * f32 values at src[src_offset ..] (unit stride) into column `col`. `col` is the dynamic field
 * (refresh: col_base + pos * patch_stride, like slice_assign.dst_offset). */
typedef struct zgml_op_kvq_store {
    uint16_t cache, src;
    uint32_t d_head, block_size, n_cols;
    uint32_t src_offset;
    uint32_t col_base;
    uint32_t col; /* dynamic */
    uint32_t patch_stride;
} zgml_op_kvq_store;

/* attentionQuantized (src/quant.zig:925-1091): q / dst dense f32 columns (unit row stride), K and V
 * columns [k_col_start, k_col_start + seq_kv) / [v_col_start, ...) of two quantised caches, optional
 * additive mask, streaming softmax; a query with no valid key yields zeros. */
typedef struct zgml_op_attention_kvq {
    uint16_t dst, q, k, v, mask;
    uint8_t has_mask, _pad;
    uint32_t d_head, seq_q;
    uint32_t seq_kv; /* dynamic */
    float scale;
    uint32_t block_size, n_cols;
    uint32_t k_col_start, v_col_start;
    uint32_t q_off, q_cs, dst_off, dst_cs;
    uint32_t mask_off, mask_rs, mask_cs;
} zgml_op_attention_kvq;

typedef struct zgml_op_fused_elementwise {
    const zgml_fused_step* steps; /* borrowed: must outlive the compiled program's use of `ops` */
    uint32_t n_steps;
    uint32_t n;
    uint16_t dst, src;
    uint32_t dst_offset, src_offset;
} zgml_op_fused_elementwise;

/* DeviceOp, src/backend.zig:179-249. */
typedef struct zgml_device_op {
    uint32_t kind; /* ZGML_DOP_* */
    uint32_t _pad;
    union {
        zgml_op_elementwise elementwise;
        zgml_op_matmul matmul;
        zgml_op_qmatmul qmatmul;
        zgml_op_rowwise softmax;
        zgml_op_rowwise layernorm;
        zgml_op_rowwise rmsnorm;
        zgml_op_reduce reduce;
        zgml_op_repeat repeat;
        zgml_op_slice_assign slice_assign;
        zgml_op_rope rope;
        zgml_op_attention attention;
        zgml_op_fused_elementwise fused_elementwise;
        zgml_op_kvq_store kvq_store;         /* extension */
        zgml_op_attention_kvq attention_kvq; /* extension */
    } u;
} zgml_device_op;

/* ProgramIO, src/backend.zig:252-257. offset/size in bytes. */
typedef struct zgml_program_io {
    uint16_t buf_idx;
    uint16_t _pad;
    uint32_t offset;
    void* host_ptr;
    uint32_t size;
    uint32_t _pad2;
} zgml_program_io;

/* QuantizedWeightUpload, src/backend.zig:260-266: int8 data in flat [K,N] row-major order
 * (index k*N+n), one f32 scale per `block_size` consecutive FLAT elements.
 *
 * Extension (no reference counterpart yet): the packed-GGUF pass-through form. With
 * `scales == NULL && scales_len == 0 && block_size == 32`, `data` holds the tensor's GGUF blocks exactly
 * as they sit in the file — block b covers flat elements [32b, 32b+32) —
 *     data_len == rows*cols/32 * 18  ->  Q4_0 blocks {f16 scale, 16 bytes}; element j of the block is the
 *                                        low (j even) / high (j odd) nibble of byte j/2, value = nibble - 8
 *                                        — the reference loader's interleaved order (gguf_loader.zig:137-141)
 *     data_len == rows*cols/32 * 34  ->  Q8_0 blocks {f16 scale, 32 int8}
 * i.e. what quantizedWeightFromInfo (src/models/gguf_loader.zig:99-154) expands on the host; here the
 * expansion and the re-pack happen on the device and half (Q4_0) of the bytes cross PCIe.
 * Requires rows*cols % 32 == 0 and cols % 32 == 0. */
#define ZGML_QW_GGUF_Q4_0_BLOCK_BYTES 18
#define ZGML_QW_GGUF_Q8_0_BLOCK_BYTES 34
typedef struct zgml_qweight_upload {
    const int8_t* data;
    uint64_t data_len;
    const float* scales;
    uint64_t scales_len;
    uint64_t rows; /* K */
    uint64_t cols; /* N */
    uint64_t block_size;
} zgml_qweight_upload;

/* DeviceProgram, src/backend.zig:270-275. */
typedef struct zgml_device_program {
    const zgml_device_op* ops;
    uint64_t n_ops;
    uint16_t n_buffers;
    const uint64_t* buffer_sizes; /* f32 elements, n_buffers entries */
    uint64_t n_buffer_sizes;
    const zgml_program_io* initial_uploads;
    uint64_t n_initial_uploads;
    const zgml_qweight_upload* qweights;
    uint64_t n_qweights;
} zgml_device_program;

/* Capabilities, src/backend.zig:14-58. Optionals: *_has = 0 means "null" (no limit). */
typedef struct zgml_capabilities {
    uint8_t compiled_programs;
    uint8_t host_visible_program_memory;
    uint8_t dense_matmul_f32;
    uint8_t dense_matmul_f16;
    uint8_t qmatmul;
    uint8_t fused_elementwise;
    uint8_t f16_weight_promotion;
    uint8_t dynamic_program_refresh;
    uint8_t prefill_attention;
    uint8_t decode_attention;
    uint8_t quantized_kv;
    uint8_t command_buffer_execution;
    uint8_t max_fused_elementwise_steps_has;
    uint8_t attention_supported;
    uint8_t attention_max_seq_kv_has;
    uint8_t attention_max_d_head_has;
    uint32_t max_fused_elementwise_steps;
    uint32_t attention_max_seq_kv;
    uint32_t attention_max_d_head;
} zgml_capabilities;

/* Subset of profile.RuntimeProfile (src/profile.zig:820-843) a device backend can fill. */
typedef struct zgml_runtime_profile {
    uint64_t time_ns[ZGML_DOP_COUNT]; /* per DeviceOp tag; filled only in profiling mode */
    uint64_t backend_op_count;
    uint64_t fallback_op_count; /* always 0: there is no CPU fallback */
    uint64_t backend_dispatch_count;
    uint64_t sync_time_ns;
    uint64_t sync_count;
    uint32_t call_count;
    uint32_t _pad;
} zgml_runtime_profile;

typedef struct zgml_hip_ctx zgml_hip_ctx;         /* Backend.ctx */
typedef struct zgml_hip_program zgml_hip_program; /* Backend.CompiledHandle */

/* Create a backend context on HIP device `device_ordinal`. NULL on failure (no device, not
 * gfx950, allocation failure); zgml_hip_last_error(NULL) then describes why. */
zgml_hip_ctx* zgml_hip_create(int device_ordinal);
void zgml_hip_destroy(zgml_hip_ctx* ctx);

/* Sticky, human-readable description of the first error recorded on the context (or of the
 * last failed zgml_hip_create when ctx == NULL). Empty string when there is none. The vtable has
 * no error channel on execute (src/backend.zig:347 returns void), hence the side channel. */
const char* zgml_hip_last_error(const zgml_hip_ctx* ctx);
void zgml_hip_clear_error(zgml_hip_ctx* ctx);

/* Capabilities.hip — what DeviceInference consults when lowering (src/device_inference.zig:108). */
void zgml_hip_capabilities(zgml_capabilities* out);

/* DeviceProgram.isSupportedBy(Capabilities.hip): 1 if supported, else 0. Pure host logic. */
int zgml_hip_program_supported(const zgml_device_program* program);

/* VTable.dense_matmul_f32: host slices in, result in dst on return. Returns 1 if handled,
 * 0 to make the caller fall back (src/tensor/forward.zig:2022-2031). dst_len/a_len/b_len are
 * element counts of the host slices. */
int zgml_hip_dense_matmul_f32(zgml_hip_ctx* ctx, float* dst, uint64_t dst_len, const float* a,
                              uint64_t a_len, const float* b, uint64_t b_len,
                              const zgml_matmul_geom* geom);

/* VTable.compile_program. Every array is borrowed for the duration of the call only: the backend
 * copies the op list (including fused steps), so the caller may free everything afterwards (the
 * reference's cpu/metal backends keep `ops` borrowed, src/backend/cpu.zig:115; copying is a
 * superset of that contract). Returns NULL on failure or when the program is unsupported. */
zgml_hip_program* zgml_hip_compile_program(zgml_hip_ctx* ctx, const zgml_device_program* program);

/* VTable.refresh_program: same-length op list whose dynamic fields changed
 * (slice_assign.dst_offset, attention.seq_kv). Other fields must be unchanged. */
void zgml_hip_refresh_program(zgml_hip_ctx* ctx, zgml_hip_program* handle,
                              const zgml_device_op* ops, uint64_t n_ops);
/* (extension, round 5) Outputs written straight into the caller's buffer. By default execute_program lands outputs in the
 * library's pinned staging buffer and copies them to `host_ptr` (197 KB of logits per SmolLM-135M token). With on = 1 a program
 * whose execute_program calls name ONE output at the SAME host address three times in a row registers that buffer with the
 * driver (hipHostRegister) and the step's last kernel writes into it. THE CALLER PROMISES that such a buffer stays allocated
 * and mapped until it passes a different address, switches this off, or frees the program: pages are pinned at registration,
 * so a buffer that is freed and re-allocated at the same address would silently stop receiving data. DeviceInference's
 * session-owned logits slice (src/device_inference.zig:262) qualifies. Returns 0. */
int zgml_hip_program_pin_outputs(zgml_hip_ctx* ctx, zgml_hip_program* program, int on);

/* The per-token refresh reduced to its two numbers (src/backend/program.zig:7452-7490 StepDynamicParams, what the reference's wgpu
 * backend uploads per step, src/backend/wgpu.zig:1162-1169): every dynamic KV store goes to column `slice_pos` (dst_offset =
 * dst_base_offset + slice_pos * patch_stride), every attention reads `seq_kv` keys. O(#dynamic ops) instead of a compare of the
 * whole op list; static fields are not examined — an adapter derives (slice_pos, seq_kv) with stepDynamicStateFromOps and calls
 * this when `needsUpload()`, and calls zgml_hip_refresh_program (which detects static changes and rebuilds) otherwise.
 * Returns 0, -1 on a null argument. */
int zgml_hip_refresh_dynamic(zgml_hip_ctx* ctx, zgml_hip_program* handle, uint32_t slice_pos, uint32_t seq_kv);

/* VTable.execute_program: upload inputs, run all ops in order, download outputs; blocking. */
void zgml_hip_execute_program(zgml_hip_ctx* ctx, zgml_hip_program* handle,
                              const zgml_program_io* inputs, uint64_t n_inputs,
                              const zgml_program_io* outputs, uint64_t n_outputs);

void zgml_hip_free_program(zgml_hip_ctx* ctx, zgml_hip_program* handle);

/* VTable.get_runtime_profile: pointer stays valid until free_program. */
zgml_runtime_profile* zgml_hip_get_runtime_profile(zgml_hip_ctx* ctx, zgml_hip_program* handle);

/* ── Extensions (no reference counterpart) ──────────────────────────────────────────────── */

/* The launch plan of a program as text, one line per launch, for diagnostics and tests (which ops a launch covers, and for
 * a quantized mat-vec launch its parts, K, prologue form — none | mul | rmsnorm | prenorm —, whether it prepares the NEXT
 * launch's norm, carries the gate / up pair product or the decode attention of its heads). Builds the plan if it is not
 * built yet. Writes at most cap - 1 bytes + NUL; returns the bytes the whole text needs (without the NUL). */
uint64_t zgml_hip_program_plan_text(zgml_hip_ctx* ctx, zgml_hip_program* handle, char* out, uint64_t cap);

/* Program options, set before compile on the context. */
enum {
    ZGML_HIP_OPT_FUSION = 1,         /* 0/1: pattern-fuse the op stream (default 1) */
    ZGML_HIP_OPT_GRAPH = 2,          /* 0/1: replay a captured hipGraph per execute (default 1) */
    ZGML_HIP_OPT_PROFILE = 3,        /* 0/1: per-op hipEvent timing into time_ns (default 0) */
    ZGML_HIP_OPT_SKIP_DEAD_UPLOADS = 4, /* 0/1: do not allocate/upload buffers no op touches (default 1) */
    ZGML_HIP_OPT_F16_DENSE_WEIGHTS = 5, /* 0/1: f16 weight promotion for dense matmul B (default 0) */
    /* bytes (0 = off, default): zgml_hip_dense_matmul_f32 keeps device copies of its B operands, keyed by
     * host pointer, up to this many bytes (plain ComputeGraph.compute() users whose
     * weights never move). The caller promises B is not mutated between calls, or invalidates. */
    ZGML_HIP_OPT_DENSE_WEIGHT_CACHE = 6,
    /* keys per workgroup at which a fused decode attention starts to split one head's context over
     * several workgroups (flash-decoding split; default 128, minimum 32, 0 = never split). A head splits
     * once seq_kv >= 2 * value; below that the launch behaves exactly as without the option. */
    ZGML_HIP_OPT_ATTN_SPLIT_MIN_KEYS = 7,
    /* workgroups (1024 threads) the device is assumed to keep resident at once, for the launch that carries the q/k/v
     * projection AND the decode attention that waits for it in one grid (the waiting workgroups spin on workgroups of the
     * same grid, so the whole grid must be resident): the fusion is only built when projection + attention workgroups fit,
     * the attention's split count shrinks to fit, otherwise the two launches stay apart. -1 (default): one 1024-thread workgroup
     * per compute unit for the short-K form; the 256-thread K-on-lanes form (K > 2048: Llama-2-7B) takes what the occupancy
     * query admits (four per compute unit, no margin) and is only built with ALL the splits the stand-alone attention would
     * use. 0: never fuse. Takes effect for programs compiled afterwards and at the next plan rebuild of existing ones.
     * A hand-off wait that gives up is bounded and reported at the next host synchronisation (sticky error on the context): the
     * results of THAT execution are wrong, including the KV-cache column it wrote — re-run the step at the same position (the
     * context has switched the fusion off and rebuilt the plan, the re-run rewrites the column); nothing older is affected. */
    ZGML_HIP_OPT_FUSE_RESIDENT_WGS = 8,
    /* 0/1 (default 0; the environment variable ZGML_HIP_KSPLIT=1 turns it on for every context): the decoder layer of a short-K
     * model (K <= 2048, Q4_0 weights with f16 scales, d_head 64 / 128) as launches that end at a K-split instead of an all-to-all
     * seam — each head's attention workgroup adds its partial of the O projection, each 32-column gate / up workgroup its partial
     * of the down projection, the next launch sums the partials in its prologue (zgml_amd/csrc/ksplit.hip). Same results within the
     * mat-vec tolerance, two launches per layer instead of four — and MEASURED SLOWER on MI355X (DESIGN.md section 4, round 5:
     * reading 48 partial vectors costs a consumer workgroup more than the launch boundary it replaces), hence off by default.
     * Latched per program at compile_program. */
    ZGML_HIP_OPT_KSPLIT = 9,
    /* 0/1 (default 0; ZGML_HIP_W8A8 in the environment): M = 1 qmatmuls take the reference's W8A8 arm — what its CPU executor does
     * when a weight carries a transposed image (src/backend/reference.zig:512-528): quantizeInput on the input row (int8 + one f32
     * scale per 32 values, src/quant.zig:604-640), the weight re-quantised per (column, 32 k) as prepareTransposed does
     * (src/quant.zig:560-603; done on the device at compile_program from the int8 + f32-scale upload) and gemvRange's block-ordered
     * f32 combine of int32 dots (src/quant.zig:320-440). BIT-IDENTICAL to that arm (zgml_amd/csrc/w8a8.hip); it is NOT the exact
     * dequantise-then-dot arithmetic the default path (and the reference's x86 / GPU backends) computes: results differ by the
     * activations' int8 rounding. Applies to weights of block size 32 with K % 64 == 0, K <= 16384, N % 16 == 0 every use of which
     * is a dense M = 1 row; such ops run unfused. Read at compile_program. */
    ZGML_HIP_OPT_W8A8 = 10
};
int zgml_hip_set_option(zgml_hip_ctx* ctx, int option, int64_t value);
/* Drop the cached device copy of host operand `b` (NULL: all of them). */
void zgml_hip_dense_cache_invalidate(zgml_hip_ctx* ctx, const float* b);
void zgml_hip_dense_cache_stats(zgml_hip_ctx* ctx, uint64_t* hits, uint64_t* misses, uint64_t* bytes);

/* Raw device access for harnesses that keep data resident (bench, multi-GPU all-gather glue):
 * device pointer of program buffer `buf_idx` (NULL if elided). */
void* zgml_hip_program_buffer_ptr(zgml_hip_program* handle, uint16_t buf_idx);
/* Device-to-device copy of n_elems f32 between buffers of two compiled programs of this context
 * (stream-ordered): hands the KV caches of a prefill plan (token_len = N program) to the decode plan,
 * which the reference does through host-visible program memory. Returns 0 on success. */
int zgml_hip_copy_program_buffer(zgml_hip_ctx* ctx, zgml_hip_program* dst, uint16_t dst_buf, uint64_t dst_offset,
                                 zgml_hip_program* src, uint16_t src_buf, uint64_t src_offset, uint64_t n_elems);
/* The HIP stream (hipStream_t as void*) the context launches on. */
void* zgml_hip_stream(zgml_hip_ctx* ctx);
/* Execute without host I/O and without blocking: enqueue the program on the context stream. */
void zgml_hip_enqueue_program(zgml_hip_ctx* ctx, zgml_hip_program* handle);
/* Enqueue ops [first, first+count) only (multi-GPU harness interleaves collectives). */
void zgml_hip_enqueue_ops(zgml_hip_ctx* ctx, zgml_hip_program* handle, uint64_t first, uint64_t count);
/* Declare op indices that batched launches must not straddle (the positions of the harness's
 * collectives); enqueue_ops ranges must start/end on them. */
int zgml_hip_program_set_barriers(zgml_hip_ctx* ctx, zgml_hip_program* handle, const uint64_t* op_indices,
                                  uint64_t n);
void zgml_hip_synchronize(zgml_hip_ctx* ctx);
/* Capture-friendly split of a step, for harnesses that record ops + collectives into ONE graph per
 * token (multi-GPU): stage_inputs does the host side (validation + copy into pinned staging),
 * enqueue_staged the device side (one H2D, scatter kernel, dynamic-parameter block) and may be recorded
 * into a stream capture; a replay picks up what stage_inputs / refresh_program wrote last.
 * enqueue_argmax leaves the index in pinned memory: read it with argmax_result after the stream (or
 * the graph replay) has completed. */
int zgml_hip_stage_inputs(zgml_hip_ctx* ctx, zgml_hip_program* handle, const zgml_program_io* inputs, uint64_t n_inputs);
void zgml_hip_enqueue_staged(zgml_hip_ctx* ctx, zgml_hip_program* handle);
int zgml_hip_enqueue_argmax(zgml_hip_ctx* ctx, zgml_hip_program* handle, uint16_t buf_idx, uint64_t offset, uint64_t n);
int64_t zgml_hip_argmax_result(zgml_hip_ctx* ctx);
/* The two halves of execute_program on their own, for harnesses that interleave collectives with
 * op ranges: enqueue the host->device transfers / run the device->host transfers (blocking). */
void zgml_hip_upload_inputs(zgml_hip_ctx* ctx, zgml_hip_program* handle, const zgml_program_io* inputs,
                            uint64_t n_inputs);
void zgml_hip_download_outputs(zgml_hip_ctx* ctx, zgml_hip_program* handle, const zgml_program_io* outputs,
                               uint64_t n_outputs);
/* On-device greedy argmax over f32 elements [offset, offset+n) of a program buffer: first index
 * of the maximum (strict >), as scripts/generate_llama.zig:101-110 / src/nn.zig:122-138. */
int64_t zgml_hip_argmax(zgml_hip_ctx* ctx, zgml_hip_program* handle, uint16_t buf_idx,
                        uint64_t offset, uint64_t n);

/* ── Row-shard (N-split) execution across the GPUs of one node (the reference has no distributed code) ──
 * One process and one context per GPU. Every quantized weight of the per-rank program is split along N (whole 32-column
 * scale blocks), activations are replicated, KV caches are head-sharded; between op ranges the replicated activation is
 * restored by ONE in-place all-gather (RCCL over xGMI, enqueued on the context stream). librccl.so is opened at run time
 * by these entry points only. Typical use (the host side of zgml_amd/host builds the per-rank program and its gather
 * points): rank 0 calls shard_unique_id and hands the 128 bytes to the other ranks by any channel; every rank calls
 * shard_init, compile_program, shard_attach, then per token refresh_program + shard_step. */
typedef struct zgml_shard_point {
    uint64_t op_end;       /* ops [previous op_end, op_end) run before this gather */
    uint16_t buf_idx;      /* program buffer holding the replicated vector */
    uint16_t _pad;
    uint32_t offset;       /* f32 elements: the vector is [offset, offset + world * len_per_rank) */
    uint32_t len_per_rank; /* rank r owns [offset + r * len_per_rank, ...) before the gather */
} zgml_shard_point;
int zgml_hip_shard_unique_id(unsigned char id_out[128]);                 /* ncclGetUniqueId; 0 on success */
int zgml_hip_shard_init(zgml_hip_ctx* ctx, const unsigned char id[128], int rank, int world); /* ncclCommInitRank on the context's device */
void zgml_hip_shard_destroy(zgml_hip_ctx* ctx);                            /* also done by zgml_hip_destroy */
/* Declare the program's all-gather points (ascending op_end) and the logits the greedy token is taken from
 * (after the last gather every rank holds the full vocabulary). Also sets the plan barriers at those op indices. */
int zgml_hip_shard_attach(zgml_hip_ctx* ctx, zgml_hip_program* program, const zgml_shard_point* points, uint64_t n_points,
                          uint16_t logits_buf, uint64_t vocab);
/* One decode step on this rank: stage + upload the inputs, run the op ranges with the all-gathers between them, argmax
 * of the gathered logits; blocking. The device side is recorded into one graph on the first call and replayed
 * afterwards (ZGML_SHARD_GRAPH=0 or a failed capture: issued eagerly). Returns the greedy token, < 0 on error. Every
 * rank must call it for every step (the collectives are collective). */
int64_t zgml_hip_shard_step(zgml_hip_ctx* ctx, zgml_hip_program* program, const zgml_program_io* inputs, uint64_t n_inputs);
int zgml_hip_shard_step_mode(zgml_hip_program* program); /* 1: steps replay one graph per token, 0: eager */
/* The greedy token of the N-sharded LM head: when the LAST gather point covers the logits buffer, that gather is replaced by
 * one (max, index) pair per rank (8 bytes instead of 4 * vocab / world; first maximum wins, src/nn.zig:122-138) in BOTH gather
 * modes; the logits buffer then holds only the rank's own slice after a step.
 *
 * PEER gather mode (no RCCL; zgml_amd/csrc/shard_peer.hip): every gather point is one small kernel per rank that stores the
 * rank's slice into a staging area in every peer's fine-grained block (xGMI peer stores), counts an arrival on every peer,
 * waits (bounded) for its own arrivals and lands the peers' slices in its program buffer. Use: every rank calls
 * shard_init_peer instead of shard_unique_id / shard_init, then compile_program + shard_attach as before, then exports its
 * block's handle, hands it to every other rank by any channel (64 + 24 bytes), imports the others', and steps as before.
 * Ranks may be processes (hipIpc handles) or contexts of ONE process (raw pointers: how the single-GPU tests run two ranks).
 * A wait that gives up (ZGML_SHARD_PEER_WAIT_MS, default 5000) makes shard_step return -1 with an error on the context. */
typedef struct zgml_shard_peer_handle {
    unsigned char ipc[64]; /* hipIpcMemHandle_t of the block */
    uint64_t pid;          /* exporting process: an importer in the same process uses `raw` */
    uint64_t raw;          /* the block's device pointer in the exporting process */
    uint64_t bytes;
} zgml_shard_peer_handle;
int zgml_hip_shard_init_peer(zgml_hip_ctx* ctx, int rank, int world);
int zgml_hip_shard_peer_export(zgml_hip_ctx* ctx, zgml_hip_program* program, zgml_shard_peer_handle* out);
int zgml_hip_shard_peer_import(zgml_hip_ctx* ctx, zgml_hip_program* program, int peer_rank, const zgml_shard_peer_handle* handle);
/* Diagnostics: one EAGER step with HIP events around every all-gather — device microseconds of the whole step and of the
 * collectives inside it (launch-bound: eager steps are slower than graph replays; the ratio is what it is for). */
int64_t zgml_hip_shard_profile_step(zgml_hip_ctx* ctx, zgml_hip_program* program, const zgml_program_io* inputs, uint64_t n_inputs,
                                    double* step_us, double* gather_us);
/* ... and, gather point by gather point, the microseconds of the last profiled step (writes min(cap, points) values; returns the
 * number of points). With the device ordinal and the peer-access row below, the first multi-device run can be read from one log. */
uint64_t zgml_hip_shard_last_point_us(zgml_hip_program* program, double* out, uint64_t cap);
/* hipDeviceCanAccessPeer(device, peer) as 0 / 1 (-1: the query failed); touches no context. */
int zgml_hip_device_can_access_peer(int device, int peer);
int zgml_hip_device_count(void);

/* Device-resident greedy decode for LLaMA-shaped programs (measurement protocol: inputs already
 * in HBM when the timed region starts). The reference's per-token host work — embedding-row
 * copy, causal-mask column, RoPE row, KV position / seq_kv patching, logits download + argmax
 * (src/llama_inference.zig:405-466, benchmarks/llama_smollm_bench.zig:290-314,
 * scripts/generate_llama.zig:101-110) — is done by one small device kernel before and one after
 * the program graph, from tables uploaded once; a token costs one graph launch and no host<->
 * device traffic. Results are identical to stepping through execute_program (tests check it). */
typedef struct zgml_resident_llama {
    const float* token_embed; /* host [vocab][d_model], uploaded once */
    const float* cos_table;   /* host [max_seq][d_head] */
    const float* sin_table;   /* host [max_seq][d_head] */
    uint32_t vocab, d_model, max_seq, d_head;
    uint16_t buf_token_input, buf_attn_mask, buf_logits, _pad;
    const uint16_t* buf_rope; /* one packed cos|sin leaf per layer */
    uint32_t n_rope;
    uint32_t _pad2;
} zgml_resident_llama;
int zgml_hip_resident_setup(zgml_hip_ctx* ctx, zgml_hip_program* handle, const zgml_resident_llama* desc);
/* Feed `first_token` at `start_pos`, then the argmax of each step, for n_steps; blocking. Writes
 * the n_steps produced tokens. Returns 0 on success. */
int zgml_hip_resident_decode(zgml_hip_ctx* ctx, zgml_hip_program* handle, uint32_t first_token,
                             uint32_t start_pos, uint32_t n_steps, int64_t* tokens_out);

/* The same for a token_len = T plan (a prefill chunk: LlamaInferenceSession.prefill, src/llama_inference.zig:474,
 * src/llm/device_prefill.zig): zgml_hip_resident_setup on the T-token program (T is read off token_input's size), then per
 * chunk only the T token ids cross PCIe — the T embedding rows, the T causal-mask columns, the T RoPE rows per layer,
 * the KV store offsets / seq_kv = start_pos + T and the argmax of the last position's logits are produced on the
 * device. Blocking. Returns the greedy next token (< 0 on error); results equal execute_program on host-patched inputs. */
int64_t zgml_hip_resident_prefill(zgml_hip_ctx* ctx, zgml_hip_program* handle, const uint32_t* tokens, uint32_t n_tokens,
                                  uint32_t start_pos);

/* Mat-vec roofline micro-benchmark: builds `n_matrices` distinct K x N quantized
 * matrices on the device from the deterministic synthetic generator (q4: nibbles in [-8,7];
 * otherwise int8), runs `warmup` + `iters` launches round-robin over the ring and returns the
 * average kernel time in microseconds measured with HIP events on the launch stream (<0 on
 * error). `bytes_per_launch` receives the algorithmic bytes (K*N/32*{18|34} + 4K + 4N). */
double zgml_hip_qmatvec_bench(zgml_hip_ctx* ctx, uint32_t K, uint32_t N, int q4, uint32_t n_matrices,
                              uint32_t warmup, uint32_t iters, uint64_t* bytes_per_launch);
/* A CHAIN of square (K x K) mat-vecs with a true data dependency: launch i computes y_i = (x_i^T W_i) * c (an
 * epilogue multiply by a constant vector keeps magnitudes bounded) and launch i + 1 reads y_i as its x (ping-pong
 * vectors; n_matrices even). The roofline figure of bench.py: launches are ordered by DATA, not only by the stream.
 * Microseconds per launch (< 0 on error). */
double zgml_hip_qmatvec_chain_bench(zgml_hip_ctx* ctx, uint32_t K, int q4, uint32_t n_matrices, uint32_t warmup, uint32_t iters,
                                    uint64_t* bytes_per_launch);
/* Same ring with M input rows (M > 1 runs the f32-MFMA tile kernel used by prefill plans);
 * bytes = weights + 4*M*K + 4*M*N, flops = 2*M*K*N. */
double zgml_hip_qmatmul_bench(zgml_hip_ctx* ctx, uint32_t M, uint32_t K, uint32_t N, int q4, uint32_t n_matrices,
                              uint32_t warmup, uint32_t iters, uint64_t* bytes_per_launch);
/* The same ring of M = 1 mat-vecs as independent launches: the captured graph forks the ring over n_streams
 * branches (distinct outputs), so launches may overlap on the device. Not the decode path (every mat-vec there
 * waits for its predecessor); it separates what the kernel can stream from what a dependent launch of this size
 * costs. Returns microseconds per launch (< 0 on error). */
double zgml_hip_qmatvec_overlap_bench(zgml_hip_ctx* ctx, uint32_t K, uint32_t N, int q4, uint32_t n_matrices, uint32_t n_streams,
                                      uint32_t iters, uint64_t* bytes_per_launch);
/* ... and on EXPLICIT streams: stream t replays its own graph of the matrices i = t (mod n_streams), all n_streams replays
 * in flight at once on their own queues (n_matrices % n_streams == 0). Microseconds per launch over all streams. */
double zgml_hip_qmatvec_streams_bench(zgml_hip_ctx* ctx, uint32_t K, uint32_t N, int q4, uint32_t n_matrices, uint32_t n_streams,
                                      uint32_t iters, uint64_t* bytes_per_launch);
/* Ring benchmark of the f16-promoted dense matmul (M == 1: f32 x times f16 weights; M > 1: f16
 * MFMA); bytes = 2*K*N + 4*M*K + 4*M*N. N % 16 == 0. */
double zgml_hip_dense_f16_bench(zgml_hip_ctx* ctx, uint32_t M, uint32_t K, uint32_t N, uint32_t n_matrices,
                                uint32_t warmup, uint32_t iters, uint64_t* bytes_per_launch);
/* One mat-vec y = x^T W with synthetic matrix `matrix_id` of the same generator (parity tests
 * rebuild that matrix on the host in int8 + f32-scale form and check y against the oracle).
 * Returns 0 on success. */
int zgml_hip_qmatvec_synth(zgml_hip_ctx* ctx, uint32_t K, uint32_t N, int q4, uint32_t matrix_id,
                           const float* x_host, float* y_host);
/* Device-to-device float4 copy of `bytes` bytes, average microseconds per launch (the measured
 * "achievable HBM" yardstick printed next to the 8 TB/s nominal peak). */
double zgml_hip_copy_bench(zgml_hip_ctx* ctx, uint64_t bytes, uint32_t warmup, uint32_t iters);

#ifdef __cplusplus
}
#endif
#endif /* ZGML_HIP_H */
