"""Host-side mirror of zgml's backend plugin types (src/backend.zig), in Python, for tests and
harnesses. Names, argument meaning and error behaviour follow the reference:

  DeviceOp.<tag>(...)          src/backend.zig:179-249 (same field names and defaults)
  FusedEwStep                  src/backend.zig:170-175
  ProgramIO                    src/backend.zig:252-257 (offset/size in bytes)
  QuantizedWeightUpload        src/backend.zig:260-266
  DeviceProgram.isSupportedBy  src/backend.zig:277-325
  Capabilities (+ consts)      src/backend.zig:14-141
  Backend (vtable wrappers)    src/backend.zig:330-382

Everything is lowered to the flat C structs of include/zgml_hip.h (capi.py); the C++ mirror in
zgml_amd/host/ is what a compiled host would use.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field, replace
from typing import Optional, Sequence

import numpy as np

from . import capi
from .capi import DOP, OP

_ELEMENTWISE_OPS = ("add", "mul", "neg", "abs", "sgn", "step", "relu", "sqrt", "recip", "exp", "log", "gelu")


def _op_id(op) -> int:
    return OP[op] if isinstance(op, str) else int(op)


@dataclass
class FusedEwStep:
    op: str
    is_swapped: bool = False
    secondary_buf: int = 0
    secondary_offset: int = 0


@dataclass
class MatMulGeometry:
    M: int
    N: int
    K: int
    a_row_stride: int
    a_col_stride: int
    b_row_stride: int
    b_col_stride: int
    a_offset: int = 0
    b_offset: int = 0
    dst_offset: int = 0
    dst_row_stride: int = 0

    def to_c(self) -> capi.MatMulGeom:
        g = capi.MatMulGeom()
        for name, _ in capi.MatMulGeom._fields_:
            setattr(g, name, int(getattr(self, name)))
        return g


class DeviceOp:
    """One op of a device program: `kind` is the union tag, `f` the payload fields."""

    __slots__ = ("kind", "f")

    def __init__(self, kind: str, **fields):
        self.kind = kind
        self.f = fields

    def __repr__(self):
        return f"DeviceOp.{self.kind}({self.f})"

    def __getattr__(self, name):
        try:
            return self.f[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def with_(self, **changes) -> "DeviceOp":
        return DeviceOp(self.kind, **{**self.f, **changes})

    # constructors, one per union arm, defaults as in the Zig struct literals
    @staticmethod
    def elementwise(op, dst, src0, src1, n, dst_offset=0, src0_offset=0, src1_offset=0):
        return DeviceOp("elementwise", op=op, dst=dst, src0=src0, src1=src1, n=n, dst_offset=dst_offset,
                        src0_offset=src0_offset, src1_offset=src1_offset)

    @staticmethod
    def matmul(dst, a, b, geom: MatMulGeometry):
        return DeviceOp("matmul", dst=dst, a=a, b=b, geom=geom)

    @staticmethod
    def qmatmul(dst, input, weight_idx, M, N, K, input_offset=0, input_row_stride=0, dst_offset=0,
                dst_row_stride=0):
        return DeviceOp("qmatmul", dst=dst, input=input, weight_idx=weight_idx, M=M, N=N, K=K,
                        input_offset=input_offset, input_row_stride=input_row_stride, dst_offset=dst_offset,
                        dst_row_stride=dst_row_stride)

    @staticmethod
    def softmax(dst, src, rows, cols, src_offset=0, dst_offset=0):
        return DeviceOp("softmax", dst=dst, src=src, rows=rows, cols=cols, eps=0.0, src_offset=src_offset,
                        dst_offset=dst_offset)

    @staticmethod
    def layernorm(dst, src, rows, cols, eps=1e-5, src_offset=0, dst_offset=0):
        return DeviceOp("layernorm", dst=dst, src=src, rows=rows, cols=cols, eps=eps, src_offset=src_offset,
                        dst_offset=dst_offset)

    @staticmethod
    def rmsnorm(dst, src, rows, cols, eps=1e-5, src_offset=0, dst_offset=0):
        return DeviceOp("rmsnorm", dst=dst, src=src, rows=rows, cols=cols, eps=eps, src_offset=src_offset,
                        dst_offset=dst_offset)

    @staticmethod
    def reduce(op, dst, src, n_out, reduce_size, src_offset=0, dst_offset=0):
        return DeviceOp("reduce", op=op, dst=dst, src=src, n_out=n_out, reduce_size=reduce_size,
                        src_offset=src_offset, dst_offset=dst_offset)

    @staticmethod
    def repeat(dst, src, n, src_ne, dst_ne, src_strides, dst_strides, src_offset=0, dst_offset=0):
        return DeviceOp("repeat", dst=dst, src=src, n=n, src_ne=tuple(src_ne), dst_ne=tuple(dst_ne),
                        src_strides=tuple(src_strides), dst_strides=tuple(dst_strides), src_offset=src_offset,
                        dst_offset=dst_offset)

    @staticmethod
    def slice_assign(dst, src, rows, cols, dst_base_offset, dst_offset, dst_row_stride, dst_col_stride,
                     src_offset, src_row_stride, src_col_stride, patch_stride):
        return DeviceOp("slice_assign", dst=dst, src=src, rows=rows, cols=cols, dst_base_offset=dst_base_offset,
                        dst_offset=dst_offset, dst_row_stride=dst_row_stride, dst_col_stride=dst_col_stride,
                        src_offset=src_offset, src_row_stride=src_row_stride, src_col_stride=src_col_stride,
                        patch_stride=patch_stride)

    @staticmethod
    def rope(dst, src, cos_sin, half_d, seq_len, src_off, cs_off, dst_off, src_rs, src_cs, cs_cs):
        return DeviceOp("rope", dst=dst, src=src, cos_sin=cos_sin, half_d=half_d, seq_len=seq_len,
                        src_off=src_off, cs_off=cs_off, dst_off=dst_off, src_rs=src_rs, src_cs=src_cs,
                        cs_cs=cs_cs)

    @staticmethod
    def attention(dst, q, k, v, mask, has_mask, d_head, seq_q, seq_kv, scale, q_off, k_off, v_off, mask_off,
                  dst_off, q_rs, q_cs, k_rs, k_cs, v_rs, v_cs, mask_rs, mask_cs, dst_rs, dst_cs):
        return DeviceOp("attention", dst=dst, q=q, k=k, v=v, mask=mask, has_mask=has_mask, d_head=d_head,
                        seq_q=seq_q, seq_kv=seq_kv, scale=scale, q_off=q_off, k_off=k_off, v_off=v_off,
                        mask_off=mask_off, dst_off=dst_off, q_rs=q_rs, q_cs=q_cs, k_rs=k_rs, k_cs=k_cs,
                        v_rs=v_rs, v_cs=v_cs, mask_rs=mask_rs, mask_cs=mask_cs, dst_rs=dst_rs, dst_cs=dst_cs)

    # extension ops: quantised KV cache (include/zgml_hip.h; src/quant.zig:645-1091)
    @staticmethod
    def kvq_store(cache, src, d_head, block_size, n_cols, src_offset, col_base, col, patch_stride):
        return DeviceOp("kvq_store", cache=cache, src=src, d_head=d_head, block_size=block_size, n_cols=n_cols,
                        src_offset=src_offset, col_base=col_base, col=col, patch_stride=patch_stride)

    @staticmethod
    def attention_kvq(dst, q, k, v, mask, has_mask, d_head, seq_q, seq_kv, scale, block_size, n_cols, k_col_start,
                      v_col_start, q_off, q_cs, dst_off, dst_cs, mask_off=0, mask_rs=0, mask_cs=0):
        return DeviceOp("attention_kvq", dst=dst, q=q, k=k, v=v, mask=mask, has_mask=has_mask, d_head=d_head,
                        seq_q=seq_q, seq_kv=seq_kv, scale=scale, block_size=block_size, n_cols=n_cols,
                        k_col_start=k_col_start, v_col_start=v_col_start, q_off=q_off, q_cs=q_cs, dst_off=dst_off,
                        dst_cs=dst_cs, mask_off=mask_off, mask_rs=mask_rs, mask_cs=mask_cs)

    @staticmethod
    def fused_elementwise(steps: Sequence[FusedEwStep], n, dst, src, dst_offset=0, src_offset=0):
        return DeviceOp("fused_elementwise", steps=list(steps), n=n, dst=dst, src=src, dst_offset=dst_offset,
                        src_offset=src_offset)

    def buffers(self) -> list[int]:
        """Buffer ids the op touches (opBuffersValid, src/backend.zig:303-325)."""
        f = self.f
        k = self.kind
        if k == "elementwise":
            return [f["dst"], f["src0"], f["src1"]]
        if k == "matmul":
            return [f["dst"], f["a"], f["b"]]
        if k == "qmatmul":
            return [f["dst"], f["input"]]
        if k in ("softmax", "layernorm", "rmsnorm", "reduce", "repeat", "slice_assign"):
            return [f["dst"], f["src"]]
        if k == "rope":
            return [f["dst"], f["src"], f["cos_sin"]]
        if k in ("attention", "attention_kvq"):
            return [f["dst"], f["q"], f["k"], f["v"], f["mask"]]
        if k == "kvq_store":
            return [f["cache"], f["src"]]
        if k == "fused_elementwise":
            out = [f["dst"], f["src"]]
            out += [s.secondary_buf for s in f["steps"] if s.op in ("add", "mul")]
            return out
        raise ValueError(k)

    def fill_c(self, c: capi.DeviceOpC, keep: list) -> None:
        c.kind = DOP[self.kind]
        u = getattr(c.u, self.kind)
        f = self.f
        if self.kind == "matmul":
            u.dst, u.a, u.b = f["dst"], f["a"], f["b"]
            u.geom = f["geom"].to_c()
        elif self.kind == "fused_elementwise":
            steps = (capi.FusedStep * max(1, len(f["steps"])))()
            for i, s in enumerate(f["steps"]):
                steps[i].op = _op_id(s.op)
                steps[i].is_swapped = 1 if s.is_swapped else 0
                steps[i].secondary_buf = s.secondary_buf
                steps[i].secondary_offset = s.secondary_offset
            keep.append(steps)
            u.steps = C.cast(steps, C.POINTER(capi.FusedStep))
            u.n_steps = len(f["steps"])
            u.n, u.dst, u.src = f["n"], f["dst"], f["src"]
            u.dst_offset, u.src_offset = f["dst_offset"], f["src_offset"]
        else:
            for name, val in f.items():
                if name == "op":
                    val = _op_id(val)
                elif name == "has_mask":
                    val = 1 if val else 0
                elif isinstance(val, tuple):
                    val = (C.c_uint32 * 4)(*val)
                setattr(u, name, val)


def ops_to_c(ops: Sequence[DeviceOp]):
    keep: list = []
    arr = (capi.DeviceOpC * max(1, len(ops)))()
    for i, op in enumerate(ops):
        op.fill_c(arr[i], keep)
    return arr, keep


@dataclass
class ProgramIO:
    buf_idx: int
    host: np.ndarray  # host_ptr + size come from this array
    offset: int = 0   # bytes
    size: Optional[int] = None  # bytes; default host.nbytes

    def fill_c(self, c: capi.ProgramIOC) -> None:
        assert self.host.flags["C_CONTIGUOUS"]
        c.buf_idx = self.buf_idx
        c.offset = self.offset
        c.host_ptr = self.host.ctypes.data
        c.size = self.host.nbytes if self.size is None else self.size


def ios_to_c(ios: Sequence[ProgramIO]):
    arr = (capi.ProgramIOC * max(1, len(ios)))()
    for i, io in enumerate(ios):
        io.fill_c(arr[i])
    return arr


@dataclass
class QuantizedWeightUpload:
    data: np.ndarray    # int8, flat [K, N] row-major (index k*N + n)
    scales: np.ndarray  # float32, one per block_size flat elements
    rows: int           # K
    cols: int           # N
    block_size: int = 32
    # packed-GGUF pass-through (include/zgml_hip.h): `data` holds the raw uint8 file
    # blocks (18 B Q4_0 / 34 B Q8_0 per 32 flat elements) and `scales` is empty
    gguf: Optional[str] = None

    @staticmethod
    def from_gguf_blocks(raw: np.ndarray, rows: int, cols: int, kind: str) -> "QuantizedWeightUpload":
        assert kind in ("q4_0", "q8_0") and (rows * cols) % 32 == 0
        raw = np.ascontiguousarray(raw, dtype=np.uint8).ravel()
        assert raw.size == rows * cols // 32 * (18 if kind == "q4_0" else 34)
        return QuantizedWeightUpload(raw, np.zeros(0, np.float32), rows, cols, 32, gguf=kind)


@dataclass
class Attention:
    supported: bool = False
    max_seq_kv: Optional[int] = None
    max_d_head: Optional[int] = None

    def supports(self, seq_kv: int, d_head: int) -> bool:
        if not self.supported:
            return False
        if self.max_seq_kv is not None and seq_kv > self.max_seq_kv:
            return False
        if self.max_d_head is not None and d_head > self.max_d_head:
            return False
        return True


@dataclass
class Capabilities:
    compiled_programs: bool = False
    host_visible_program_memory: bool = False
    dense_matmul_f32: bool = False
    dense_matmul_f16: bool = False
    qmatmul: bool = False
    fused_elementwise: bool = False
    max_fused_elementwise_steps: Optional[int] = None
    f16_weight_promotion: bool = False
    dynamic_program_refresh: bool = False
    prefill_attention: bool = False
    decode_attention: bool = False
    quantized_kv: bool = False
    command_buffer_execution: bool = False
    attention: Attention = field(default_factory=Attention)

    def supportsElementwiseOp(self, op) -> bool:
        name = op if isinstance(op, str) else capi.OP_NAME.get(int(op), "?")
        return name in _ELEMENTWISE_OPS

    def supportsOp(self, op: DeviceOp) -> bool:
        if not self.compiled_programs:
            return False
        k = op.kind
        if k == "elementwise":
            return self.supportsElementwiseOp(op.op)
        if k == "matmul":
            return self.dense_matmul_f32
        if k == "qmatmul":
            return self.qmatmul
        if k in ("softmax", "layernorm", "rmsnorm", "repeat", "slice_assign", "rope"):
            return True
        if k == "reduce":
            return op.op in ("sum", "max", OP["sum"], OP["max"])
        if k == "attention":
            return self.attention.supports(op.seq_kv, op.d_head)
        if k in ("kvq_store", "attention_kvq"):  # extension ops behind the reference's own capability flag
            if not self.quantized_kv or op.block_size == 0 or op.d_head % op.block_size:
                return False
            return k == "kvq_store" or self.attention.supports(op.seq_kv, op.d_head)
        if k == "fused_elementwise":
            if not self.fused_elementwise:
                return False
            if self.max_fused_elementwise_steps is not None and len(op.steps) > self.max_fused_elementwise_steps:
                return False
            return all(self.supportsElementwiseOp(s.op) for s in op.steps)
        return False

    @staticmethod
    def from_c(c: capi.CapabilitiesC) -> "Capabilities":
        return Capabilities(
            compiled_programs=bool(c.compiled_programs),
            host_visible_program_memory=bool(c.host_visible_program_memory),
            dense_matmul_f32=bool(c.dense_matmul_f32), dense_matmul_f16=bool(c.dense_matmul_f16),
            qmatmul=bool(c.qmatmul), fused_elementwise=bool(c.fused_elementwise),
            max_fused_elementwise_steps=(c.max_fused_elementwise_steps if c.max_fused_elementwise_steps_has else None),
            f16_weight_promotion=bool(c.f16_weight_promotion),
            dynamic_program_refresh=bool(c.dynamic_program_refresh),
            prefill_attention=bool(c.prefill_attention), decode_attention=bool(c.decode_attention),
            quantized_kv=bool(c.quantized_kv), command_buffer_execution=bool(c.command_buffer_execution),
            attention=Attention(bool(c.attention_supported),
                                c.attention_max_seq_kv if c.attention_max_seq_kv_has else None,
                                c.attention_max_d_head if c.attention_max_d_head_has else None))


# the reference's capability constants (src/backend.zig:60-111), used by the gating tests
Capabilities.reference_cpu = Capabilities(
    compiled_programs=True, host_visible_program_memory=True, dense_matmul_f32=True, qmatmul=True,
    fused_elementwise=True, dynamic_program_refresh=True, prefill_attention=True, decode_attention=True,
    attention=Attention(True, None, 512))
Capabilities.metal = replace(Capabilities.reference_cpu, dense_matmul_f16=True, max_fused_elementwise_steps=8,
                             command_buffer_execution=True)
Capabilities.wgpu = Capabilities(
    compiled_programs=True, dense_matmul_f32=True, qmatmul=True, f16_weight_promotion=True,
    dynamic_program_refresh=True, prefill_attention=True, decode_attention=True,
    command_buffer_execution=True, attention=Attention(True, 4096, 512))


@dataclass
class DeviceProgram:
    ops: list
    buffer_sizes: list          # f32 elements per buffer
    initial_uploads: list = field(default_factory=list)
    qweights: list = field(default_factory=list)
    n_buffers: Optional[int] = None

    def __post_init__(self):
        if self.n_buffers is None:
            self.n_buffers = len(self.buffer_sizes)

    def isSupportedBy(self, caps: Capabilities) -> bool:
        if not caps.compiled_programs:
            return False
        if self.n_buffers != len(self.buffer_sizes):
            return False
        for op in self.ops:
            if not caps.supportsOp(op):
                return False
            if any(b >= len(self.buffer_sizes) for b in op.buffers()):
                return False
            if op.kind == "qmatmul":
                if op.weight_idx >= len(self.qweights):
                    return False
                qw = self.qweights[op.weight_idx]
                if qw.block_size == 0:
                    return False
                if qw.rows != op.K or qw.cols != op.N:
                    return False
                n_elems = op.K * op.N
                n_blocks = (n_elems + qw.block_size - 1) // qw.block_size
                if qw.gguf is not None:
                    if n_elems % 32 or op.N % 32 or qw.block_size != 32:
                        return False
                    continue
                if qw.data.size < n_elems or qw.scales.size < n_blocks:
                    return False
        return True

    def to_c(self):
        keep: list = []
        ops_c, k2 = ops_to_c(self.ops)
        keep += [ops_c, k2]
        sizes = (C.c_uint64 * max(1, len(self.buffer_sizes)))(*[int(s) for s in self.buffer_sizes])
        ups = ios_to_c(self.initial_uploads)
        qws = (capi.QWeightUploadC * max(1, len(self.qweights)))()
        for i, qw in enumerate(self.qweights):
            if qw.gguf is not None:
                d = np.ascontiguousarray(qw.data, dtype=np.uint8)
                keep += [d]
                qws[i].data, qws[i].data_len = d.ctypes.data, d.size
                qws[i].scales, qws[i].scales_len = None, 0
            else:
                d = np.ascontiguousarray(qw.data, dtype=np.int8)
                s = np.ascontiguousarray(qw.scales, dtype=np.float32)
                keep += [d, s]
                qws[i].data, qws[i].data_len = d.ctypes.data, d.size
                qws[i].scales, qws[i].scales_len = s.ctypes.data, s.size
            qws[i].rows, qws[i].cols, qws[i].block_size = qw.rows, qw.cols, qw.block_size
        p = capi.DeviceProgramC()
        p.ops, p.n_ops = C.cast(ops_c, C.POINTER(capi.DeviceOpC)), len(self.ops)
        p.n_buffers = self.n_buffers
        p.buffer_sizes, p.n_buffer_sizes = C.cast(sizes, C.POINTER(C.c_uint64)), len(self.buffer_sizes)
        p.initial_uploads, p.n_initial_uploads = C.cast(ups, C.POINTER(capi.ProgramIOC)), len(self.initial_uploads)
        p.qweights, p.n_qweights = C.cast(qws, C.POINTER(capi.QWeightUploadC)), len(self.qweights)
        keep += [sizes, ups, qws, self.initial_uploads]
        return p, keep


@dataclass
class DenseMatMulSpecF32:
    dst: np.ndarray
    a: np.ndarray
    b: np.ndarray
    geom: MatMulGeometry


class Backend:
    """`Backend{ctx, vtable, name_str, device_type, capabilities}` over the HIP C ABI."""

    name_str = "hip"
    device_type = "hip"

    def __init__(self, device_ordinal: int = 0):
        self._lib = capi.load_hip()
        self.ctx = self._lib.zgml_hip_create(device_ordinal)
        if not self.ctx:
            raise RuntimeError("zgml_hip_create failed: " + self._lib.zgml_hip_last_error(None).decode())
        c = capi.CapabilitiesC()
        self._lib.zgml_hip_capabilities(C.byref(c))
        self.capabilities = Capabilities.from_c(c)
        self._keep: dict = {}

    def close(self):
        if self.ctx:
            self._lib.zgml_hip_destroy(self.ctx)
            self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def last_error(self) -> str:
        return self._lib.zgml_hip_last_error(self.ctx).decode()

    def set_option(self, option: int, value: int) -> None:
        if self._lib.zgml_hip_set_option(self.ctx, option, value) != 0:
            raise ValueError(f"unknown option {option}")

    # vtable
    def supportsProgram(self, program: DeviceProgram) -> bool:
        return program.isSupportedBy(self.capabilities)

    def compileProgram(self, program: DeviceProgram):
        if not self.supportsProgram(program):
            return None
        pc, keep = program.to_c()
        handle = self._lib.zgml_hip_compile_program(self.ctx, C.byref(pc))
        return handle or None

    def refreshProgram(self, handle, ops: Sequence[DeviceOp]) -> None:
        arr, keep = ops_to_c(ops)
        self._lib.zgml_hip_refresh_program(self.ctx, handle, arr, len(ops))

    def executeProgram(self, handle, inputs: Sequence[ProgramIO], outputs: Sequence[ProgramIO]) -> None:
        self._lib.zgml_hip_execute_program(self.ctx, handle, ios_to_c(inputs), len(inputs), ios_to_c(outputs),
                                           len(outputs))
        err = self.last_error()
        if err:
            raise RuntimeError("zgml_hip: " + err)

    def freeProgram(self, handle) -> None:
        self._lib.zgml_hip_free_program(self.ctx, handle)

    def getRuntimeProfile(self, handle):
        p = self._lib.zgml_hip_get_runtime_profile(self.ctx, handle)
        return p.contents if p else None

    def dense_matmul_f32(self, spec: DenseMatMulSpecF32) -> bool:
        g = spec.geom.to_c()
        return bool(self._lib.zgml_hip_dense_matmul_f32(
            self.ctx, spec.dst.ctypes.data, spec.dst.size, spec.a.ctypes.data, spec.a.size,
            spec.b.ctypes.data, spec.b.size, C.byref(g)))

    def supportsAttention(self, seq_kv: int, d_head: int) -> bool:
        return self.capabilities.attention.supports(seq_kv, d_head)

    # extensions
    def argmax(self, handle, buf_idx: int, offset: int, n: int) -> int:
        return int(self._lib.zgml_hip_argmax(self.ctx, handle, buf_idx, offset, n))

    def synchronize(self) -> None:
        self._lib.zgml_hip_synchronize(self.ctx)

    def planText(self, handle) -> str:
        """zgml_hip_program_plan_text: one line per launch of the program's plan (diagnostics, tests)."""
        n = int(self._lib.zgml_hip_program_plan_text(self.ctx, handle, None, 0))
        buf = C.create_string_buffer(n + 1)
        self._lib.zgml_hip_program_plan_text(self.ctx, handle, buf, n + 1)
        return buf.value.decode()


def tryDenseMatMul(backend: Optional[Backend], spec: DenseMatMulSpecF32) -> bool:
    """src/backend.zig:386-390."""
    if backend is None:
        return False
    return backend.dense_matmul_f32(spec)
