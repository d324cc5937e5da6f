// backend.hpp — C++ host mirror of zgml's backend plugin surface (src/backend.zig), layered on the
// C ABI of include/zgml_hip.h. Zig is not available in this image, so this is the compiled host
// side a zgml maintainer would otherwise write in Zig (INTEGRATION.md has that adapter): same
// names, same argument meaning, same error behaviour (optionals -> nullptr/false, no exceptions
// across the vtable).
//
//   Capabilities / Capabilities::Attention   src/backend.zig:14-141
//   DeviceOp (named constructors per arm)    src/backend.zig:179-249
//   ProgramIO, QuantizedWeightUpload         src/backend.zig:252-266
//   DeviceProgram::isSupportedBy             src/backend.zig:277-325
//   Backend + VTable                         src/backend.zig:330-382
//   tryDenseMatMul                           src/backend.zig:386-390
#pragma once

#include <cstdint>
#include <cstring>
#include <optional>
#include <string>
#include <vector>

#include "../../include/zgml_hip.h"

namespace zgml::backend {

enum class Device { cpu, metal, cuda, npu, wgpu, hip }; // `.hip` is the one-line upstream edit

using Op = uint32_t; // ZGML_OP_* ordinals of src/op.zig

inline bool isBinary(Op op) { return op == ZGML_OP_ADD || op == ZGML_OP_MUL; }

struct Capabilities {
    struct Attention {
        bool supported = false;
        std::optional<uint32_t> max_seq_kv, max_d_head;
        bool supports(uint32_t seq_kv, uint32_t d_head) const {
            if (!supported) return false;
            if (max_seq_kv && seq_kv > *max_seq_kv) return false;
            if (max_d_head && d_head > *max_d_head) return false;
            return true;
        }
    };
    bool compiled_programs = false, host_visible_program_memory = false, dense_matmul_f32 = false,
         dense_matmul_f16 = false, qmatmul = false, fused_elementwise = false;
    std::optional<uint32_t> max_fused_elementwise_steps;
    bool f16_weight_promotion = false, dynamic_program_refresh = false, prefill_attention = false,
         decode_attention = false, quantized_kv = false, command_buffer_execution = false;
    Attention attention;

    static bool supportsElementwiseOp(Op op) { return op >= ZGML_OP_ADD && op <= ZGML_OP_GELU; }

    bool supportsOp(const zgml_device_op& op) const {
        if (!compiled_programs) return false;
        switch (op.kind) {
            case ZGML_DOP_ELEMENTWISE: return supportsElementwiseOp(op.u.elementwise.op);
            case ZGML_DOP_MATMUL: return dense_matmul_f32;
            case ZGML_DOP_QMATMUL: return qmatmul;
            case ZGML_DOP_SOFTMAX:
            case ZGML_DOP_LAYERNORM:
            case ZGML_DOP_RMSNORM:
            case ZGML_DOP_REPEAT:
            case ZGML_DOP_SLICE_ASSIGN:
            case ZGML_DOP_ROPE: return true;
            case ZGML_DOP_REDUCE: return op.u.reduce.op == ZGML_OP_SUM || op.u.reduce.op == ZGML_OP_MAX;
            case ZGML_DOP_ATTENTION: return attention.supports(op.u.attention.seq_kv, op.u.attention.d_head);
            case ZGML_DOP_KVQ_STORE: // extension ops (quantised KV cache) behind the reference's capability flag
                return quantized_kv && op.u.kvq_store.block_size && op.u.kvq_store.d_head % op.u.kvq_store.block_size == 0;
            case ZGML_DOP_ATTENTION_KVQ:
                return quantized_kv && op.u.attention_kvq.block_size && op.u.attention_kvq.d_head % op.u.attention_kvq.block_size == 0 &&
                       attention.supports(op.u.attention_kvq.seq_kv, op.u.attention_kvq.d_head);
            case ZGML_DOP_FUSED_ELEMENTWISE: {
                const auto& fe = op.u.fused_elementwise;
                if (!fused_elementwise) return false;
                if (max_fused_elementwise_steps && fe.n_steps > *max_fused_elementwise_steps) return false;
                for (uint32_t s = 0; s < fe.n_steps; s++)
                    if (!supportsElementwiseOp(fe.steps[s].op)) return false;
                return true;
            }
            default: return false;
        }
    }

    static Capabilities fromC(const zgml_capabilities& c) {
        Capabilities k;
        k.compiled_programs = c.compiled_programs, k.host_visible_program_memory = c.host_visible_program_memory;
        k.dense_matmul_f32 = c.dense_matmul_f32, k.dense_matmul_f16 = c.dense_matmul_f16, k.qmatmul = c.qmatmul;
        k.fused_elementwise = c.fused_elementwise;
        if (c.max_fused_elementwise_steps_has) k.max_fused_elementwise_steps = c.max_fused_elementwise_steps;
        k.f16_weight_promotion = c.f16_weight_promotion, k.dynamic_program_refresh = c.dynamic_program_refresh;
        k.prefill_attention = c.prefill_attention, k.decode_attention = c.decode_attention;
        k.quantized_kv = c.quantized_kv, k.command_buffer_execution = c.command_buffer_execution;
        k.attention.supported = c.attention_supported;
        if (c.attention_max_seq_kv_has) k.attention.max_seq_kv = c.attention_max_seq_kv;
        if (c.attention_max_d_head_has) k.attention.max_d_head = c.attention_max_d_head;
        return k;
    }
};

using MatMulGeometry = zgml_matmul_geom;
using FusedEwStep = zgml_fused_step;
using ProgramIO = zgml_program_io;

// DeviceOp: the flat C record with the union arms of the Zig type as named constructors.
struct DeviceOp : zgml_device_op {
    DeviceOp() { std::memset(static_cast<zgml_device_op*>(this), 0, sizeof(zgml_device_op)); }

    static DeviceOp elementwise(Op op, uint16_t dst, uint16_t src0, uint16_t src1, uint32_t n, uint32_t dst_offset = 0,
                                uint32_t src0_offset = 0, uint32_t src1_offset = 0) {
        DeviceOp o;
        o.kind = ZGML_DOP_ELEMENTWISE;
        o.u.elementwise = {op, dst, src0, src1, 0, n, dst_offset, src0_offset, src1_offset};
        return o;
    }
    static DeviceOp matmul(uint16_t dst, uint16_t a, uint16_t b, const MatMulGeometry& geom) {
        DeviceOp o;
        o.kind = ZGML_DOP_MATMUL;
        o.u.matmul.dst = dst, o.u.matmul.a = a, o.u.matmul.b = b, o.u.matmul.geom = geom;
        return o;
    }
    static DeviceOp qmatmul(uint16_t dst, uint16_t input, uint16_t weight_idx, uint32_t M, uint32_t N, uint32_t K,
                            uint32_t input_offset = 0, uint32_t input_row_stride = 0, uint32_t dst_offset = 0,
                            uint32_t dst_row_stride = 0) {
        DeviceOp o;
        o.kind = ZGML_DOP_QMATMUL;
        o.u.qmatmul = {dst, input, weight_idx, 0, M, N, K, input_offset, input_row_stride, dst_offset, dst_row_stride};
        return o;
    }
    static DeviceOp rowwise(uint32_t kind, uint16_t dst, uint16_t src, uint32_t rows, uint32_t cols, float eps,
                            uint32_t src_offset, uint32_t dst_offset) {
        DeviceOp o;
        o.kind = kind;
        o.u.rmsnorm = {dst, src, rows, cols, eps, src_offset, dst_offset};
        return o;
    }
    static DeviceOp softmax(uint16_t dst, uint16_t src, uint32_t rows, uint32_t cols, uint32_t src_offset = 0,
                            uint32_t dst_offset = 0) {
        return rowwise(ZGML_DOP_SOFTMAX, dst, src, rows, cols, 0.f, src_offset, dst_offset);
    }
    static DeviceOp layernorm(uint16_t dst, uint16_t src, uint32_t rows, uint32_t cols, float eps = 1e-5f,
                              uint32_t src_offset = 0, uint32_t dst_offset = 0) {
        return rowwise(ZGML_DOP_LAYERNORM, dst, src, rows, cols, eps, src_offset, dst_offset);
    }
    static DeviceOp rmsnorm(uint16_t dst, uint16_t src, uint32_t rows, uint32_t cols, float eps = 1e-5f,
                            uint32_t src_offset = 0, uint32_t dst_offset = 0) {
        return rowwise(ZGML_DOP_RMSNORM, dst, src, rows, cols, eps, src_offset, dst_offset);
    }
    static DeviceOp reduce(Op op, uint16_t dst, uint16_t src, uint32_t n_out, uint32_t reduce_size,
                           uint32_t src_offset = 0, uint32_t dst_offset = 0) {
        DeviceOp o;
        o.kind = ZGML_DOP_REDUCE;
        o.u.reduce = {op, dst, src, n_out, reduce_size, src_offset, dst_offset};
        return o;
    }
    static DeviceOp repeat(uint16_t dst, uint16_t src, uint32_t n, const uint32_t (&src_ne)[4], const uint32_t (&dst_ne)[4],
                           const uint32_t (&src_strides)[4], const uint32_t (&dst_strides)[4], uint32_t src_offset = 0,
                           uint32_t dst_offset = 0) {
        DeviceOp o;
        o.kind = ZGML_DOP_REPEAT;
        auto& r = o.u.repeat;
        r.dst = dst, r.src = src, r.n = n, r.src_offset = src_offset, r.dst_offset = dst_offset;
        for (int i = 0; i < 4; i++) {
            r.src_ne[i] = src_ne[i], r.dst_ne[i] = dst_ne[i];
            r.src_strides[i] = src_strides[i], r.dst_strides[i] = dst_strides[i];
        }
        return o;
    }
    static DeviceOp slice_assign(uint16_t dst, uint16_t src, uint32_t rows, uint32_t cols, uint32_t dst_base_offset,
                                 uint32_t dst_offset, uint32_t dst_row_stride, uint32_t dst_col_stride,
                                 uint32_t src_offset, uint32_t src_row_stride, uint32_t src_col_stride,
                                 uint32_t patch_stride) {
        DeviceOp o;
        o.kind = ZGML_DOP_SLICE_ASSIGN;
        o.u.slice_assign = {dst,           src,        rows,           cols,           dst_base_offset, dst_offset, dst_row_stride,
                            dst_col_stride, src_offset, src_row_stride, src_col_stride, patch_stride};
        return o;
    }
    static DeviceOp rope(uint16_t dst, uint16_t src, uint16_t cos_sin, uint32_t half_d, uint32_t seq_len, uint32_t src_off,
                         uint32_t cs_off, uint32_t dst_off, uint32_t src_rs, uint32_t src_cs, uint32_t cs_cs) {
        DeviceOp o;
        o.kind = ZGML_DOP_ROPE;
        o.u.rope = {dst, src, cos_sin, 0, half_d, seq_len, src_off, cs_off, dst_off, src_rs, src_cs, cs_cs};
        return o;
    }
    static DeviceOp attention(const zgml_op_attention& a) {
        DeviceOp o;
        o.kind = ZGML_DOP_ATTENTION;
        o.u.attention = a;
        return o;
    }
    // extension ops: quantised KV cache (include/zgml_hip.h; src/quant.zig:645-1091)
    static DeviceOp kvq_store(const zgml_op_kvq_store& a) {
        DeviceOp o;
        o.kind = ZGML_DOP_KVQ_STORE;
        o.u.kvq_store = a;
        return o;
    }
    static DeviceOp attention_kvq(const zgml_op_attention_kvq& a) {
        DeviceOp o;
        o.kind = ZGML_DOP_ATTENTION_KVQ;
        o.u.attention_kvq = a;
        return o;
    }
    // `steps` is borrowed (as the Zig slice is): the owner must outlive uses of the op.
    static DeviceOp fused_elementwise(const FusedEwStep* steps, uint32_t n_steps, uint32_t n, uint16_t dst, uint16_t src,
                                      uint32_t dst_offset = 0, uint32_t src_offset = 0) {
        DeviceOp o;
        o.kind = ZGML_DOP_FUSED_ELEMENTWISE;
        o.u.fused_elementwise = {steps, n_steps, n, dst, src, dst_offset, src_offset};
        return o;
    }
};
static_assert(sizeof(DeviceOp) == sizeof(zgml_device_op), "DeviceOp must stay layout-identical to the C record");

struct QuantizedWeightUpload {
    const int8_t* data = nullptr;
    size_t data_len = 0;
    const float* scales = nullptr;
    size_t scales_len = 0;
    size_t rows = 0, cols = 0, block_size = 0;
    // Packed-GGUF pass-through (include/zgml_hip.h): the tensor's Q4_0 (18-byte) or
    // Q8_0 (34-byte) file blocks over the flat [K,N] order, no host expansion. 0 = reference form.
    static QuantizedWeightUpload ggufBlocks(const uint8_t* blocks, size_t n_bytes, size_t rows, size_t cols) {
        return {reinterpret_cast<const int8_t*>(blocks), n_bytes, nullptr, 0, rows, cols, 32};
    }
    int ggufForm() const {
        if (scales || scales_len || block_size != 32 || !data) return 0;
        const size_t n = rows * cols;
        if (!n || n % 32 || cols % 32) return 0;
        return data_len == n / 32 * ZGML_QW_GGUF_Q4_0_BLOCK_BYTES ? 1 : data_len == n / 32 * ZGML_QW_GGUF_Q8_0_BLOCK_BYTES ? 2 : 0;
    }
};

struct DeviceProgram {
    std::vector<DeviceOp> ops;
    uint16_t n_buffers = 0;
    std::vector<uint64_t> buffer_sizes; // f32 elements
    std::vector<ProgramIO> initial_uploads;
    std::vector<QuantizedWeightUpload> qweights;

    bool hasBuffer(uint16_t idx) const { return idx < buffer_sizes.size(); }

    bool opBuffersValid(const zgml_device_op& op) const {
        switch (op.kind) {
            case ZGML_DOP_ELEMENTWISE:
                return hasBuffer(op.u.elementwise.dst) && hasBuffer(op.u.elementwise.src0) && hasBuffer(op.u.elementwise.src1);
            case ZGML_DOP_MATMUL: return hasBuffer(op.u.matmul.dst) && hasBuffer(op.u.matmul.a) && hasBuffer(op.u.matmul.b);
            case ZGML_DOP_QMATMUL: return hasBuffer(op.u.qmatmul.dst) && hasBuffer(op.u.qmatmul.input);
            case ZGML_DOP_SOFTMAX:
            case ZGML_DOP_LAYERNORM:
            case ZGML_DOP_RMSNORM: return hasBuffer(op.u.rmsnorm.dst) && hasBuffer(op.u.rmsnorm.src);
            case ZGML_DOP_REDUCE: return hasBuffer(op.u.reduce.dst) && hasBuffer(op.u.reduce.src);
            case ZGML_DOP_REPEAT: return hasBuffer(op.u.repeat.dst) && hasBuffer(op.u.repeat.src);
            case ZGML_DOP_SLICE_ASSIGN: return hasBuffer(op.u.slice_assign.dst) && hasBuffer(op.u.slice_assign.src);
            case ZGML_DOP_ROPE: return hasBuffer(op.u.rope.dst) && hasBuffer(op.u.rope.src) && hasBuffer(op.u.rope.cos_sin);
            case ZGML_DOP_ATTENTION: {
                const auto& a = op.u.attention;
                return hasBuffer(a.dst) && hasBuffer(a.q) && hasBuffer(a.k) && hasBuffer(a.v) && hasBuffer(a.mask);
            }
            case ZGML_DOP_KVQ_STORE: return hasBuffer(op.u.kvq_store.cache) && hasBuffer(op.u.kvq_store.src);
            case ZGML_DOP_ATTENTION_KVQ: {
                const auto& a = op.u.attention_kvq;
                return hasBuffer(a.dst) && hasBuffer(a.q) && hasBuffer(a.k) && hasBuffer(a.v) && hasBuffer(a.mask);
            }
            case ZGML_DOP_FUSED_ELEMENTWISE: {
                const auto& fe = op.u.fused_elementwise;
                if (!hasBuffer(fe.dst) || !hasBuffer(fe.src)) return false;
                for (uint32_t s = 0; s < fe.n_steps; s++)
                    if (isBinary(fe.steps[s].op) && !hasBuffer(fe.steps[s].secondary_buf)) return false;
                return true;
            }
            default: return false;
        }
    }

    bool isSupportedBy(const Capabilities& caps) const {
        if (!caps.compiled_programs) return false;
        if ((size_t)n_buffers != buffer_sizes.size()) return false;
        for (const auto& op : ops) {
            if (!caps.supportsOp(op)) return false;
            if (!opBuffersValid(op)) return false;
            if (op.kind == ZGML_DOP_QMATMUL) {
                const auto& q = op.u.qmatmul;
                if ((size_t)q.weight_idx >= qweights.size()) return false;
                const auto& qw = qweights[q.weight_idx];
                if (qw.block_size == 0) return false;
                if (qw.rows != q.K || qw.cols != q.N) return false;
                const size_t n_elems = (size_t)q.K * q.N, n_blocks = (n_elems + qw.block_size - 1) / qw.block_size;
                if (qw.ggufForm()) continue;
                if (qw.data_len < n_elems || qw.scales_len < n_blocks) return false;
            }
        }
        return true;
    }

    // Flat view for the C ABI; `qw_storage` must outlive the returned struct's use.
    zgml_device_program view(std::vector<zgml_qweight_upload>& qw_storage) const {
        qw_storage.clear();
        for (const auto& q : qweights)
            qw_storage.push_back({q.data, q.data_len, q.scales, q.scales_len, q.rows, q.cols, q.block_size});
        zgml_device_program p{};
        p.ops = ops.data(), p.n_ops = ops.size();
        p.n_buffers = n_buffers;
        p.buffer_sizes = buffer_sizes.data(), p.n_buffer_sizes = buffer_sizes.size();
        p.initial_uploads = initial_uploads.data(), p.n_initial_uploads = initial_uploads.size();
        p.qweights = qw_storage.data(), p.n_qweights = qw_storage.size();
        return p;
    }
};

struct DenseMatMulSpecF32 {
    float* dst;
    size_t dst_len;
    const float* a;
    size_t a_len;
    const float* b;
    size_t b_len;
    MatMulGeometry geom;
};

struct Backend {
    using CompiledHandle = void*;
    struct VTable {
        bool (*dense_matmul_f32)(void* ctx, const DenseMatMulSpecF32& spec);
        CompiledHandle (*compile_program)(void* ctx, const DeviceProgram& program);
        void (*refresh_program)(void* ctx, CompiledHandle handle, const DeviceOp* ops, size_t n_ops);
        void (*execute_program)(void* ctx, CompiledHandle handle, const ProgramIO* inputs, size_t n_inputs,
                                const ProgramIO* outputs, size_t n_outputs);
        void (*free_program)(void* ctx, CompiledHandle handle);
        zgml_runtime_profile* (*get_runtime_profile)(void* ctx, CompiledHandle handle);
    };

    void* ctx = nullptr;
    const VTable* vtable = nullptr;
    std::string name_str;
    Device device_type = Device::cpu;
    Capabilities capabilities;

    bool supportsProgram(const DeviceProgram& p) const { return p.isSupportedBy(capabilities); }
    CompiledHandle compileProgram(const DeviceProgram& p) const {
        if (!supportsProgram(p)) return nullptr;
        return vtable->compile_program(ctx, p);
    }
    void refreshProgram(CompiledHandle h, const DeviceOp* ops, size_t n) const { vtable->refresh_program(ctx, h, ops, n); }
    void executeProgram(CompiledHandle h, const ProgramIO* in, size_t n_in, const ProgramIO* out, size_t n_out) const {
        vtable->execute_program(ctx, h, in, n_in, out, n_out);
    }
    void freeProgram(CompiledHandle h) const { vtable->free_program(ctx, h); }
    zgml_runtime_profile* getRuntimeProfile(CompiledHandle h) const { return vtable->get_runtime_profile(ctx, h); }
    bool supportsAttention(uint32_t seq_kv, uint32_t d_head) const { return capabilities.attention.supports(seq_kv, d_head); }
};

inline bool tryDenseMatMul(const Backend* backend_opt, const DenseMatMulSpecF32& spec) {
    if (!backend_opt) return false;
    return backend_opt->vtable->dense_matmul_f32(backend_opt->ctx, spec);
}

// The MI355X backend: dlopens libzgml_hip.so and fills the vtable with thin forwards.
// Fails (returns false, `error` set) when the library or a gfx950 device is missing — there is
// no CPU fallback.
struct HipBackend {
    void* lib = nullptr;
    zgml_hip_ctx* ctx = nullptr;
    std::string error;
    bool init(const char* lib_path, int device_ordinal);
    void deinit();
    Backend backend();
    // raw entry points (resolved by init)
    struct Api;
    Api* api = nullptr;
};

} // namespace zgml::backend
