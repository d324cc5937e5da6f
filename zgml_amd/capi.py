"""ctypes mirror of include/zgml_hip.h (the C ABI of the MI355X backend).

The structures here are the flat form of the reference's plugin types
(`DeviceOp`, `ProgramIO`, `QuantizedWeightUpload`, `DeviceProgram`,
`Capabilities` — src/backend.zig:14-275 of zgml). `program.py` builds them with
the reference's field names; this module only declares layouts and loads the
shared library. There is no CPU fallback: if the HIP library is missing,
`load_hip()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

ABI_VERSION = 1

# ── Op ordinals (src/op.zig:11-62) ───────────────────────────────────────────
OP = {
    "none": 0, "view": 1, "reshape": 2, "transpose": 3, "permute": 4, "as_strided": 5,
    "broadcast_to": 6, "add": 7, "mul": 8, "neg": 9, "abs": 10, "sgn": 11, "step": 12,
    "relu": 13, "sqrt": 14, "recip": 15, "exp": 16, "log": 17, "gelu": 18, "sum": 19,
    "max": 20, "repeat": 21,
}
OP_NAME = {v: k for k, v in OP.items()}

# ── DeviceOp tags (src/backend.zig:179-249, declaration order) ───────────────
DOP_KINDS = [
    "elementwise", "matmul", "qmatmul", "softmax", "layernorm", "rmsnorm", "reduce",
    "repeat", "slice_assign", "rope", "attention", "fused_elementwise",
]
DOP = {name: i for i, name in enumerate(DOP_KINDS)}
DOP.update({"kvq_store": 13, "attention_kvq": 14})  # extension kinds (quantised KV cache)


class MatMulGeom(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "M", "N", "K", "a_row_stride", "a_col_stride", "b_row_stride", "b_col_stride",
        "a_offset", "b_offset", "dst_offset", "dst_row_stride")]


class FusedStep(C.Structure):
    _fields_ = [("op", C.c_uint32), ("is_swapped", C.c_uint8), ("_pad", C.c_uint8),
                ("secondary_buf", C.c_uint16), ("secondary_offset", C.c_uint32)]


class OpElementwise(C.Structure):
    _fields_ = [("op", C.c_uint32), ("dst", C.c_uint16), ("src0", C.c_uint16), ("src1", C.c_uint16),
                ("_pad", C.c_uint16), ("n", C.c_uint32), ("dst_offset", C.c_uint32),
                ("src0_offset", C.c_uint32), ("src1_offset", C.c_uint32)]


class OpMatmul(C.Structure):
    _fields_ = [("dst", C.c_uint16), ("a", C.c_uint16), ("b", C.c_uint16), ("_pad", C.c_uint16),
                ("geom", MatMulGeom)]


class OpQMatmul(C.Structure):
    _fields_ = [("dst", C.c_uint16), ("input", C.c_uint16), ("weight_idx", C.c_uint16), ("_pad", C.c_uint16),
                ("M", C.c_uint32), ("N", C.c_uint32), ("K", C.c_uint32),
                ("input_offset", C.c_uint32), ("input_row_stride", C.c_uint32),
                ("dst_offset", C.c_uint32), ("dst_row_stride", C.c_uint32)]


class OpRowwise(C.Structure):
    _fields_ = [("dst", C.c_uint16), ("src", C.c_uint16), ("rows", C.c_uint32), ("cols", C.c_uint32),
                ("eps", C.c_float), ("src_offset", C.c_uint32), ("dst_offset", C.c_uint32)]


class OpReduce(C.Structure):
    _fields_ = [("op", C.c_uint32), ("dst", C.c_uint16), ("src", C.c_uint16), ("n_out", C.c_uint32),
                ("reduce_size", C.c_uint32), ("src_offset", C.c_uint32), ("dst_offset", C.c_uint32)]


class OpRepeat(C.Structure):
    _fields_ = [("dst", C.c_uint16), ("src", C.c_uint16), ("n", C.c_uint32),
                ("src_ne", C.c_uint32 * 4), ("dst_ne", C.c_uint32 * 4),
                ("src_strides", C.c_uint32 * 4), ("dst_strides", C.c_uint32 * 4),
                ("src_offset", C.c_uint32), ("dst_offset", C.c_uint32)]


class OpSliceAssign(C.Structure):
    _fields_ = [("dst", C.c_uint16), ("src", C.c_uint16), ("rows", C.c_uint32), ("cols", C.c_uint32),
                ("dst_base_offset", C.c_uint32), ("dst_offset", C.c_uint32),
                ("dst_row_stride", C.c_uint32), ("dst_col_stride", C.c_uint32),
                ("src_offset", C.c_uint32), ("src_row_stride", C.c_uint32), ("src_col_stride", C.c_uint32),
                ("patch_stride", C.c_uint32)]


class OpRope(C.Structure):
    _fields_ = [("dst", C.c_uint16), ("src", C.c_uint16), ("cos_sin", C.c_uint16), ("_pad", C.c_uint16),
                ("half_d", C.c_uint32), ("seq_len", C.c_uint32), ("src_off", C.c_uint32),
                ("cs_off", C.c_uint32), ("dst_off", C.c_uint32), ("src_rs", C.c_uint32),
                ("src_cs", C.c_uint32), ("cs_cs", C.c_uint32)]


class OpAttention(C.Structure):
    _fields_ = [("dst", C.c_uint16), ("q", C.c_uint16), ("k", C.c_uint16), ("v", C.c_uint16),
                ("mask", C.c_uint16), ("has_mask", C.c_uint8), ("_pad", C.c_uint8),
                ("d_head", C.c_uint32), ("seq_q", C.c_uint32), ("seq_kv", C.c_uint32), ("scale", C.c_float),
                ("q_off", C.c_uint32), ("k_off", C.c_uint32), ("v_off", C.c_uint32),
                ("mask_off", C.c_uint32), ("dst_off", C.c_uint32),
                ("q_rs", C.c_uint32), ("q_cs", C.c_uint32), ("k_rs", C.c_uint32), ("k_cs", C.c_uint32),
                ("v_rs", C.c_uint32), ("v_cs", C.c_uint32), ("mask_rs", C.c_uint32), ("mask_cs", C.c_uint32),
                ("dst_rs", C.c_uint32), ("dst_cs", C.c_uint32)]


class OpFusedElementwise(C.Structure):
    _fields_ = [("steps", C.POINTER(FusedStep)), ("n_steps", C.c_uint32), ("n", C.c_uint32),
                ("dst", C.c_uint16), ("src", C.c_uint16), ("dst_offset", C.c_uint32),
                ("src_offset", C.c_uint32)]


class OpKvqStore(C.Structure):
    _fields_ = [("cache", C.c_uint16), ("src", C.c_uint16), ("d_head", C.c_uint32), ("block_size", C.c_uint32),
                ("n_cols", C.c_uint32), ("src_offset", C.c_uint32), ("col_base", C.c_uint32), ("col", C.c_uint32),
                ("patch_stride", C.c_uint32)]


class OpAttentionKvq(C.Structure):
    _fields_ = [("dst", C.c_uint16), ("q", C.c_uint16), ("k", C.c_uint16), ("v", C.c_uint16),
                ("mask", C.c_uint16), ("has_mask", C.c_uint8), ("_pad", C.c_uint8),
                ("d_head", C.c_uint32), ("seq_q", C.c_uint32), ("seq_kv", C.c_uint32), ("scale", C.c_float),
                ("block_size", C.c_uint32), ("n_cols", C.c_uint32), ("k_col_start", C.c_uint32),
                ("v_col_start", C.c_uint32), ("q_off", C.c_uint32), ("q_cs", C.c_uint32), ("dst_off", C.c_uint32),
                ("dst_cs", C.c_uint32), ("mask_off", C.c_uint32), ("mask_rs", C.c_uint32), ("mask_cs", C.c_uint32)]


class _OpUnion(C.Union):
    _fields_ = [("elementwise", OpElementwise), ("matmul", OpMatmul), ("qmatmul", OpQMatmul),
                ("softmax", OpRowwise), ("layernorm", OpRowwise), ("rmsnorm", OpRowwise),
                ("reduce", OpReduce), ("repeat", OpRepeat), ("slice_assign", OpSliceAssign),
                ("rope", OpRope), ("attention", OpAttention), ("fused_elementwise", OpFusedElementwise),
                ("kvq_store", OpKvqStore), ("attention_kvq", OpAttentionKvq)]


class DeviceOpC(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("_pad", C.c_uint32), ("u", _OpUnion)]


class ProgramIOC(C.Structure):
    _fields_ = [("buf_idx", C.c_uint16), ("_pad", C.c_uint16), ("offset", C.c_uint32),
                ("host_ptr", C.c_void_p), ("size", C.c_uint32), ("_pad2", C.c_uint32)]


class QWeightUploadC(C.Structure):
    _fields_ = [("data", C.c_void_p), ("data_len", C.c_uint64), ("scales", C.c_void_p),
                ("scales_len", C.c_uint64), ("rows", C.c_uint64), ("cols", C.c_uint64),
                ("block_size", C.c_uint64)]


class DeviceProgramC(C.Structure):
    _fields_ = [("ops", C.POINTER(DeviceOpC)), ("n_ops", C.c_uint64), ("n_buffers", C.c_uint16),
                ("buffer_sizes", C.POINTER(C.c_uint64)), ("n_buffer_sizes", C.c_uint64),
                ("initial_uploads", C.POINTER(ProgramIOC)), ("n_initial_uploads", C.c_uint64),
                ("qweights", C.POINTER(QWeightUploadC)), ("n_qweights", C.c_uint64)]


class CapabilitiesC(C.Structure):
    _fields_ = [(n, C.c_uint8) for n in (
        "compiled_programs", "host_visible_program_memory", "dense_matmul_f32", "dense_matmul_f16",
        "qmatmul", "fused_elementwise", "f16_weight_promotion", "dynamic_program_refresh",
        "prefill_attention", "decode_attention", "quantized_kv", "command_buffer_execution",
        "max_fused_elementwise_steps_has", "attention_supported", "attention_max_seq_kv_has",
        "attention_max_d_head_has")] + [
        ("max_fused_elementwise_steps", C.c_uint32), ("attention_max_seq_kv", C.c_uint32),
        ("attention_max_d_head", C.c_uint32)]


class ResidentLlamaC(C.Structure):
    _fields_ = [("token_embed", C.c_void_p), ("cos_table", C.c_void_p), ("sin_table", C.c_void_p),
                ("vocab", C.c_uint32), ("d_model", C.c_uint32), ("max_seq", C.c_uint32), ("d_head", C.c_uint32),
                ("buf_token_input", C.c_uint16), ("buf_attn_mask", C.c_uint16), ("buf_logits", C.c_uint16),
                ("_pad", C.c_uint16), ("buf_rope", C.POINTER(C.c_uint16)), ("n_rope", C.c_uint32), ("_pad2", C.c_uint32)]


class RuntimeProfileC(C.Structure):
    _fields_ = [("time_ns", C.c_uint64 * 12), ("backend_op_count", C.c_uint64),
                ("fallback_op_count", C.c_uint64), ("backend_dispatch_count", C.c_uint64),
                ("sync_time_ns", C.c_uint64), ("sync_count", C.c_uint64), ("call_count", C.c_uint32),
                ("_pad", C.c_uint32)]


# Every symbol include/zgml_hip.h declares; tests/test_abi_symbols.py checks the built library
# exports each of them.
HIP_SYMBOLS = [
    "zgml_hip_create", "zgml_hip_destroy", "zgml_hip_last_error", "zgml_hip_clear_error",
    "zgml_hip_capabilities", "zgml_hip_program_supported", "zgml_hip_dense_matmul_f32",
    "zgml_hip_compile_program", "zgml_hip_refresh_program", "zgml_hip_refresh_dynamic", "zgml_hip_execute_program",
    "zgml_hip_free_program", "zgml_hip_get_runtime_profile", "zgml_hip_set_option",
    "zgml_hip_program_buffer_ptr", "zgml_hip_copy_program_buffer", "zgml_hip_stage_inputs", "zgml_hip_enqueue_staged",
    "zgml_hip_enqueue_argmax", "zgml_hip_argmax_result", "zgml_hip_stream", "zgml_hip_enqueue_program",
    "zgml_hip_enqueue_ops", "zgml_hip_program_set_barriers", "zgml_hip_synchronize", "zgml_hip_upload_inputs", "zgml_hip_download_outputs", "zgml_hip_argmax", "zgml_hip_qmatvec_bench",
    "zgml_hip_qmatmul_bench", "zgml_hip_qmatvec_overlap_bench", "zgml_hip_qmatvec_streams_bench", "zgml_hip_qmatvec_chain_bench", "zgml_hip_dense_f16_bench", "zgml_hip_dense_cache_invalidate", "zgml_hip_dense_cache_stats",
    "zgml_hip_qmatvec_synth", "zgml_hip_copy_bench", "zgml_hip_resident_setup", "zgml_hip_resident_decode",
    "zgml_hip_resident_prefill", "zgml_hip_shard_unique_id", "zgml_hip_shard_init", "zgml_hip_shard_destroy", "zgml_hip_shard_attach", "zgml_hip_shard_step", "zgml_hip_shard_step_mode",
    "zgml_hip_shard_profile_step", "zgml_hip_shard_last_point_us", "zgml_hip_device_can_access_peer", "zgml_hip_device_count", "zgml_hip_shard_init_peer", "zgml_hip_shard_peer_export", "zgml_hip_shard_peer_import",
    "zgml_hip_program_plan_text", "zgml_hip_program_pin_outputs",
]

class ShardPointC(C.Structure):
    """zgml_shard_point (include/zgml_hip.h)."""
    _fields_ = [("op_end", C.c_uint64), ("buf_idx", C.c_uint16), ("_pad", C.c_uint16), ("offset", C.c_uint32),
                ("len_per_rank", C.c_uint32)]


class ShardPeerHandleC(C.Structure):
    """zgml_shard_peer_handle (include/zgml_hip.h): what a rank hands to its peers in the peer gather mode."""
    _fields_ = [("ipc", C.c_ubyte * 64), ("pid", C.c_uint64), ("raw", C.c_uint64), ("bytes", C.c_uint64)]


OPT_FUSION, OPT_GRAPH, OPT_PROFILE, OPT_SKIP_DEAD_UPLOADS, OPT_F16_DENSE_WEIGHTS, OPT_DENSE_WEIGHT_CACHE = 1, 2, 3, 4, 5, 6
OPT_ATTN_SPLIT_MIN_KEYS = 7
OPT_FUSE_RESIDENT_WGS = 8
OPT_KSPLIT = 9
OPT_W8A8 = 10

_PKG_DIR = Path(__file__).resolve().parent
HIP_LIB_PATH = _PKG_DIR / "lib" / "libzgml_hip.so"
HOST_LIB_PATH = Path(os.environ["ZGML_HOST_LIB"]) if os.environ.get("ZGML_HOST_LIB") else _PKG_DIR / "lib" / "libzgml_host.so"  # (env: a sanitizer build)

_hip_lib = None


def _bind_hip(lib: C.CDLL) -> None:
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    lib.zgml_hip_create.restype, lib.zgml_hip_create.argtypes = vp, [i32]
    lib.zgml_hip_destroy.restype, lib.zgml_hip_destroy.argtypes = None, [vp]
    lib.zgml_hip_last_error.restype, lib.zgml_hip_last_error.argtypes = C.c_char_p, [vp]
    lib.zgml_hip_clear_error.restype, lib.zgml_hip_clear_error.argtypes = None, [vp]
    lib.zgml_hip_capabilities.restype, lib.zgml_hip_capabilities.argtypes = None, [C.POINTER(CapabilitiesC)]
    lib.zgml_hip_program_supported.restype = i32
    lib.zgml_hip_program_supported.argtypes = [C.POINTER(DeviceProgramC)]
    lib.zgml_hip_dense_matmul_f32.restype = i32
    lib.zgml_hip_dense_matmul_f32.argtypes = [vp, vp, u64, vp, u64, vp, u64, C.POINTER(MatMulGeom)]
    lib.zgml_hip_compile_program.restype = vp
    lib.zgml_hip_compile_program.argtypes = [vp, C.POINTER(DeviceProgramC)]
    lib.zgml_hip_refresh_program.restype = None
    lib.zgml_hip_refresh_dynamic.restype = C.c_int
    lib.zgml_hip_refresh_dynamic.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    lib.zgml_hip_program_pin_outputs.restype = C.c_int
    lib.zgml_hip_program_pin_outputs.argtypes = [vp, vp, C.c_int]
    lib.zgml_hip_refresh_program.argtypes = [vp, vp, C.POINTER(DeviceOpC), u64]
    lib.zgml_hip_execute_program.restype = None
    lib.zgml_hip_execute_program.argtypes = [vp, vp, C.POINTER(ProgramIOC), u64, C.POINTER(ProgramIOC), u64]
    lib.zgml_hip_free_program.restype, lib.zgml_hip_free_program.argtypes = None, [vp, vp]
    lib.zgml_hip_get_runtime_profile.restype = C.POINTER(RuntimeProfileC)
    lib.zgml_hip_get_runtime_profile.argtypes = [vp, vp]
    lib.zgml_hip_program_plan_text.restype, lib.zgml_hip_program_plan_text.argtypes = u64, [vp, vp, C.c_char_p, u64]
    lib.zgml_hip_set_option.restype, lib.zgml_hip_set_option.argtypes = i32, [vp, i32, C.c_int64]
    lib.zgml_hip_program_buffer_ptr.restype, lib.zgml_hip_program_buffer_ptr.argtypes = vp, [vp, C.c_uint16]
    lib.zgml_hip_stage_inputs.restype, lib.zgml_hip_stage_inputs.argtypes = i32, [vp, vp, C.POINTER(ProgramIOC), u64]
    lib.zgml_hip_enqueue_staged.restype, lib.zgml_hip_enqueue_staged.argtypes = None, [vp, vp]
    lib.zgml_hip_enqueue_argmax.restype, lib.zgml_hip_enqueue_argmax.argtypes = i32, [vp, vp, C.c_uint16, u64, u64]
    lib.zgml_hip_argmax_result.restype, lib.zgml_hip_argmax_result.argtypes = C.c_int64, [vp]
    lib.zgml_hip_copy_program_buffer.restype = i32
    lib.zgml_hip_copy_program_buffer.argtypes = [vp, vp, C.c_uint16, u64, vp, C.c_uint16, u64, u64]
    lib.zgml_hip_stream.restype, lib.zgml_hip_stream.argtypes = vp, [vp]
    lib.zgml_hip_enqueue_program.restype, lib.zgml_hip_enqueue_program.argtypes = None, [vp, vp]
    lib.zgml_hip_enqueue_ops.restype, lib.zgml_hip_enqueue_ops.argtypes = None, [vp, vp, u64, u64]
    lib.zgml_hip_synchronize.restype, lib.zgml_hip_synchronize.argtypes = None, [vp]
    lib.zgml_hip_program_set_barriers.restype = i32
    lib.zgml_hip_program_set_barriers.argtypes = [vp, vp, C.POINTER(u64), u64]
    lib.zgml_hip_upload_inputs.restype = None
    lib.zgml_hip_upload_inputs.argtypes = [vp, vp, C.POINTER(ProgramIOC), u64]
    lib.zgml_hip_download_outputs.restype = None
    lib.zgml_hip_download_outputs.argtypes = [vp, vp, C.POINTER(ProgramIOC), u64]
    lib.zgml_hip_argmax.restype = C.c_int64
    lib.zgml_hip_argmax.argtypes = [vp, vp, C.c_uint16, u64, u64]
    lib.zgml_hip_qmatvec_bench.restype = C.c_double
    lib.zgml_hip_qmatvec_bench.argtypes = [vp, u32, u32, i32, u32, u32, u32, C.POINTER(u64)]
    lib.zgml_hip_dense_cache_invalidate.restype, lib.zgml_hip_dense_cache_invalidate.argtypes = None, [vp, vp]
    lib.zgml_hip_dense_cache_stats.restype = None
    lib.zgml_hip_dense_cache_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    lib.zgml_hip_dense_f16_bench.restype = C.c_double
    lib.zgml_hip_dense_f16_bench.argtypes = [vp, u32, u32, u32, u32, u32, u32, C.POINTER(u64)]
    lib.zgml_hip_qmatmul_bench.restype = C.c_double
    lib.zgml_hip_qmatmul_bench.argtypes = [vp, u32, u32, u32, i32, u32, u32, u32, C.POINTER(u64)]
    lib.zgml_hip_qmatvec_overlap_bench.restype = C.c_double
    lib.zgml_hip_qmatvec_overlap_bench.argtypes = [vp, u32, u32, i32, u32, u32, u32, C.POINTER(u64)]
    if hasattr(lib, "zgml_hip_qmatvec_streams_bench"):
        lib.zgml_hip_qmatvec_streams_bench.restype = C.c_double
        lib.zgml_hip_qmatvec_streams_bench.argtypes = [vp, u32, u32, i32, u32, u32, u32, C.POINTER(u64)]
    if hasattr(lib, "zgml_hip_resident_prefill"):
        lib.zgml_hip_resident_prefill.restype = C.c_int64
        lib.zgml_hip_resident_prefill.argtypes = [vp, vp, C.POINTER(u32), u32, u32]
    if hasattr(lib, "zgml_hip_shard_init"):
        lib.zgml_hip_shard_unique_id.restype, lib.zgml_hip_shard_unique_id.argtypes = i32, [vp]
        lib.zgml_hip_shard_init.restype, lib.zgml_hip_shard_init.argtypes = i32, [vp, vp, i32, i32]
        lib.zgml_hip_shard_destroy.restype, lib.zgml_hip_shard_destroy.argtypes = None, [vp]
        lib.zgml_hip_shard_attach.restype = i32
        lib.zgml_hip_shard_attach.argtypes = [vp, vp, C.POINTER(ShardPointC), u64, C.c_uint16, u64]
        lib.zgml_hip_shard_step.restype, lib.zgml_hip_shard_step.argtypes = C.c_int64, [vp, vp, C.POINTER(ProgramIOC), u64]
        lib.zgml_hip_shard_step_mode.restype, lib.zgml_hip_shard_step_mode.argtypes = i32, [vp]
    if hasattr(lib, "zgml_hip_shard_init_peer"):
        lib.zgml_hip_shard_init_peer.restype, lib.zgml_hip_shard_init_peer.argtypes = i32, [vp, i32, i32]
        lib.zgml_hip_shard_peer_export.restype, lib.zgml_hip_shard_peer_export.argtypes = i32, [vp, vp, C.POINTER(ShardPeerHandleC)]
        lib.zgml_hip_shard_peer_import.restype, lib.zgml_hip_shard_peer_import.argtypes = i32, [vp, vp, i32, C.POINTER(ShardPeerHandleC)]
    if hasattr(lib, "zgml_hip_shard_last_point_us"):
        lib.zgml_hip_shard_last_point_us.restype = u64
        lib.zgml_hip_shard_last_point_us.argtypes = [vp, C.POINTER(C.c_double), u64]
        lib.zgml_hip_device_can_access_peer.restype = C.c_int
        lib.zgml_hip_device_can_access_peer.argtypes = [C.c_int, C.c_int]
        lib.zgml_hip_device_count.restype = C.c_int
        lib.zgml_hip_device_count.argtypes = []
    if hasattr(lib, "zgml_hip_shard_profile_step"):
        lib.zgml_hip_shard_profile_step.restype = C.c_int64
        lib.zgml_hip_shard_profile_step.argtypes = [vp, vp, C.POINTER(ProgramIOC), u64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if hasattr(lib, "zgml_hip_qmatvec_chain_bench"):  # absent from an older build loaded through ZGML_HIP_LIB (diagnostics)
        lib.zgml_hip_qmatvec_chain_bench.restype = C.c_double
        lib.zgml_hip_qmatvec_chain_bench.argtypes = [vp, u32, i32, u32, u32, u32, C.POINTER(u64)]
    lib.zgml_hip_qmatvec_synth.restype = i32
    lib.zgml_hip_qmatvec_synth.argtypes = [vp, u32, u32, i32, u32, vp, vp]
    lib.zgml_hip_resident_setup.restype = i32
    lib.zgml_hip_resident_setup.argtypes = [vp, vp, C.POINTER(ResidentLlamaC)]
    lib.zgml_hip_resident_decode.restype = i32
    lib.zgml_hip_resident_decode.argtypes = [vp, vp, u32, u32, u32, vp]
    lib.zgml_hip_copy_bench.restype = C.c_double
    lib.zgml_hip_copy_bench.argtypes = [vp, u64, u32, u32]


def load_hip() -> C.CDLL:
    """Load the in-tree HIP backend library. Fails loudly when it has not been built."""
    global _hip_lib
    if _hip_lib is None:
        path = Path(os.environ.get("ZGML_HIP_LIB", HIP_LIB_PATH))
        if not path.exists():
            raise RuntimeError(
                f"{path} not found: the HIP backend is not built (run `python -c 'import __graft_entry__ as g; "
                "g.build()'`). zgml_amd has no CPU fallback.")
        lib = C.CDLL(str(path))
        _bind_hip(lib)
        _hip_lib = lib
    return _hip_lib
