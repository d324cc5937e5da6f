/*
 * zgml_oracle.h — CPU restatement of zgml's forward-inference arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY. This is the parity oracle and the host-CPU baseline. Nothing in the
 * product path (zgml_amd/, include/) may link, import or call it; only tests/, the smoke check in
 * __graft_entry__.py and bench.py's cpu_baseline leg do.
 *
 * Parity pinning: the reference is Zig 0.16 and cannot be built in this image (no zig
 * toolchain; its M=1 int8 GEMV is AArch64 inline asm), so the oracle is pinned by the
 * reference's own known-answer tests, restated in tests/golden/kat.json and
 * checked by tests/test_oracle_kat.py.
 *
 * Every function cites the reference code it restates (paths relative to the zgml tree).
 * Build: see oracle/Makefile (-O2 -ffp-contract=off, no fast-math: Zig's default float mode is
 * strict, i.e. no mul+add contraction and no reassociation).
 */
#ifndef ZGML_ORACLE_H
#define ZGML_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/zgml_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference.Buffer, src/backend/reference.zig:13-16 */
typedef struct zo_buffer {
    float* ptr;
    uint64_t len;
} zo_buffer;

/* reference.QWeight, src/backend/reference.zig:18-24 */
typedef struct zo_qweight {
    const int8_t* data;
    const float* scales;
    uint64_t block_size;
    const int8_t* t_data;  /* may be NULL */
    const float* t_scales; /* may be NULL */
} zo_qweight;

/* ── thread pool used by the CPU baseline (a restatement of the N-split idea of GemvPool,
 *    src/quant.zig:24-198, and ThreadPool.parallelFor, src/thread_pool.zig) ───────────────── */
void zo_set_threads(int n_threads); /* 1 = sequential (default) */
int zo_get_threads(void);

/* ── DeviceOp executor: reference.executeOp / executeProgram, src/backend/reference.zig:129-176 */
void zo_execute_op(const zo_buffer* buffers, const zo_qweight* qweights, const zgml_device_op* op);
void zo_execute_ops(const zo_buffer* buffers, const zo_qweight* qweights, const zgml_device_op* ops,
                    uint64_t n_ops);

/* ── compiled-program object: CpuBackend, src/backend/cpu.zig:38-145 ─────────────────────── */
typedef struct zo_program zo_program;
/* 1: programs compiled from now on promote matmul B operands that have an initial upload to f16
 * (src/backend/wgpu.zig:1071-1104, kernels src/backend/metal.zig:680-760). Default 0. */
void zo_set_f16_dense(int on);
/* 1: programs compiled afterwards keep prepareTransposed copies and run M = 1 qmatmuls through the W8A8 arm
 * (reference.zig:512-528 / inference_utils.zig:157-176): quantizeInput + gemvRange over GemvPool (<= 16 workers) */
void zo_set_w8a8(int on);
zo_program* zo_compile_program(const zgml_device_program* program);
void zo_refresh_program(zo_program* p, const zgml_device_op* ops, uint64_t n_ops);
void zo_execute_program(zo_program* p, const zgml_program_io* inputs, uint64_t n_inputs,
                        const zgml_program_io* outputs, uint64_t n_outputs);
void zo_free_program(zo_program* p);
float* zo_program_buffer(zo_program* p, uint16_t idx, uint64_t* len_out);
/* pieces of execute for harnesses that interleave collectives with op ranges (row-shard tests) */
void zo_program_execute_range(zo_program* p, uint64_t first, uint64_t count);
void zo_program_upload(zo_program* p, const zgml_program_io* io, uint64_t n);
void zo_program_download(zo_program* p, const zgml_program_io* io, uint64_t n);

/* the same four calls with a leading (ignored) ctx, i.e. the zgml_hip_* signatures */
void* zo_vt_compile_program(void* ctx, const zgml_device_program* program);
void zo_vt_refresh_program(void* ctx, void* handle, const zgml_device_op* ops, uint64_t n_ops);
void zo_vt_execute_program(void* ctx, void* handle, const zgml_program_io* inputs, uint64_t n_inputs,
                           const zgml_program_io* outputs, uint64_t n_outputs);
void zo_vt_free_program(void* ctx, void* handle);

/* DeviceProgram.isSupportedBy(Capabilities.reference_cpu-like), src/backend.zig:277-325.
 * max_fused_steps < 0 means "no limit". */
int zo_program_supported(const zgml_device_program* program, int fused_elementwise, int max_fused_steps,
                         int64_t attn_max_seq_kv, int64_t attn_max_d_head);

/* ── quantisation: src/quant.zig ─────────────────────────────────────────────────────────── */
/* QuantizedWeight.fromSlice, src/quant.zig:216-256 */
void zo_quantize_from_slice(const float* weights, uint64_t rows, uint64_t cols, uint64_t block_size,
                            int8_t* data, float* scales);
/* QuantizedWeight.prepareTransposed, src/quant.zig:274-317 == prepareTransposedQWeight,
 * src/backend/reference.zig:26-70 */
void zo_prepare_transposed(const int8_t* data, const float* scales, uint64_t K, uint64_t N, uint64_t bs,
                           int8_t* t_data, float* t_scales);
/* QuantizedWeight.quantizeInput, src/quant.zig:320-341 */
void zo_quantize_input(const float* input, uint64_t K, uint64_t bs, int8_t* inp_q, float* inp_scales);
/* QuantizedWeight.gemvRange, src/quant.zig:358-440 (sdot replaced by an exact int32 dot) */
void zo_gemv_range(const int8_t* t_d, const float* t_s, const int8_t* inp_q, const float* inp_scales,
                   float* dst, uint64_t n_start, uint64_t n_end, uint64_t K, uint64_t bs);
/* x86: the AVX-512 VNNI arm of gemvRange (bit-identical to the scalar loop); 1 when it is in use. zo_set_vnni(0) forces the
 * scalar loop, (1) enables VNNI where the host has it, (-1) probes again (honours ZGML_ORACLE_VNNI). */
int zo_gemv_uses_vnni(void);
void zo_set_vnni(int on);
/* GemvPool.dispatch, src/quant.zig:135-196: N-chunks rounded up to 4, 1 Mi-element threshold,
 * at most 16 workers. Uses the oracle pool; `n_workers` caps the active count. */
void zo_gemv_pool_dispatch(const int8_t* t_d, const float* t_s, const int8_t* inp_q,
                           const float* inp_scales, float* dst, uint64_t N, uint64_t K, uint64_t bs,
                           int n_workers);
/* QuantizedWeight.matmul, src/quant.zig:475-578 == reference.qmatmul generic path,
 * src/backend/reference.zig:530-565. Strides of 0 mean dense (K / N). */
void zo_qmatmul_exact(const int8_t* data, const float* scales, uint64_t bs, const float* input,
                      float* dst, uint64_t M, uint64_t N, uint64_t K, uint64_t input_row_stride,
                      uint64_t dst_row_stride);
/* QuantizedWeight.dequantizeTo / dequant, src/quant.zig:462-465 */
void zo_dequantize(const int8_t* data, const float* scales, uint64_t n_elems, uint64_t bs, float* out);

/* ── GGUF block decode: src/models/gguf_loader.zig ───────────────────────────────────────── */
float zo_f16_to_f32(uint16_t h);
uint16_t zo_f32_to_f16(float f); /* round-to-nearest-even; used only to build fixtures */
/* quantizedWeightFromInfo .q4_0 / .q8_0 arms, src/models/gguf_loader.zig:116-143 */
void zo_gguf_q4_0_to_int8(const uint8_t* raw, uint64_t n_elems, int8_t* data, float* scales);
void zo_gguf_q8_0_to_int8(const uint8_t* raw, uint64_t n_elems, int8_t* data, float* scales);
/* dequantQ4_0 / dequantQ8_0, src/models/gguf_loader.zig:33-72 */
void zo_gguf_dequant_q4_0(float* dst, const uint8_t* raw, uint64_t n_elems);
void zo_gguf_dequant_q8_0(float* dst, const uint8_t* raw, uint64_t n_elems);

/* ── nn helpers: src/nn.zig ──────────────────────────────────────────────────────────────── */
/* RoPE.init tables, src/nn.zig:286-311: cos/sin [max_seq][d], half-split duplicated. */
void zo_rope_tables(uint64_t d, uint64_t max_seq, float base, float* cos_table, float* sin_table);
/* argmax, src/nn.zig:122-138 / scripts/generate_llama.zig:101-110: first max wins. */
int64_t zo_argmax(const float* v, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif
