"""ctypes wrapper over oracle/_build/libzgml_oracle.so (TEST INFRASTRUCTURE ONLY).

`OracleBackend` has the same compile/refresh/execute surface as zgml_amd.program.Backend so a
parity test runs one DeviceProgram through both and compares.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import Sequence

import numpy as np

from zgml_amd import capi
from zgml_amd.program import DeviceOp, DeviceProgram, ProgramIO, ios_to_c, ops_to_c

_DIR = Path(__file__).resolve().parent
# ZGML_ORACLE_LIB: a sanitizer build (oracle/Makefile SAN=asan|tsan, tools/run_sanitizers.sh)
LIB_PATH = Path(os.environ["ZGML_ORACLE_LIB"]) if os.environ.get("ZGML_ORACLE_LIB") else _DIR / "_build" / "libzgml_oracle.so"
_lib = None


def build() -> None:
    subprocess.run(["make", "-C", str(_DIR)], check=True, capture_output=True)


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        build()
    lib = C.CDLL(str(LIB_PATH))
    vp, u64, i32, f32 = C.c_void_p, C.c_uint64, C.c_int, C.c_float
    lib.zo_set_threads.argtypes, lib.zo_set_threads.restype = [i32], None
    lib.zo_set_f16_dense.argtypes, lib.zo_set_f16_dense.restype = [i32], None
    lib.zo_get_threads.argtypes, lib.zo_get_threads.restype = [], i32
    lib.zo_set_w8a8.argtypes, lib.zo_set_w8a8.restype = [i32], None
    lib.zo_compile_program.argtypes, lib.zo_compile_program.restype = [C.POINTER(capi.DeviceProgramC)], vp
    lib.zo_refresh_program.argtypes, lib.zo_refresh_program.restype = [vp, C.POINTER(capi.DeviceOpC), u64], None
    lib.zo_execute_program.argtypes = [vp, C.POINTER(capi.ProgramIOC), u64, C.POINTER(capi.ProgramIOC), u64]
    lib.zo_execute_program.restype = None
    lib.zo_free_program.argtypes, lib.zo_free_program.restype = [vp], None
    lib.zo_program_buffer.argtypes, lib.zo_program_buffer.restype = [vp, C.c_uint16, C.POINTER(u64)], vp
    lib.zo_program_supported.argtypes = [C.POINTER(capi.DeviceProgramC), i32, i32, C.c_int64, C.c_int64]
    lib.zo_program_supported.restype = i32
    lib.zo_quantize_from_slice.argtypes, lib.zo_quantize_from_slice.restype = [vp, u64, u64, u64, vp, vp], None
    lib.zo_prepare_transposed.argtypes, lib.zo_prepare_transposed.restype = [vp, vp, u64, u64, u64, vp, vp], None
    lib.zo_quantize_input.argtypes, lib.zo_quantize_input.restype = [vp, u64, u64, vp, vp], None
    lib.zo_gemv_range.argtypes, lib.zo_gemv_range.restype = [vp, vp, vp, vp, vp, u64, u64, u64, u64], None
    lib.zo_gemv_pool_dispatch.argtypes = [vp, vp, vp, vp, vp, u64, u64, u64, i32]
    lib.zo_gemv_pool_dispatch.restype = None
    lib.zo_qmatmul_exact.argtypes = [vp, vp, u64, vp, vp, u64, u64, u64, u64, u64]
    lib.zo_qmatmul_exact.restype = None
    lib.zo_dequantize.argtypes, lib.zo_dequantize.restype = [vp, vp, u64, u64, vp], None
    lib.zo_f16_to_f32.argtypes, lib.zo_f16_to_f32.restype = [C.c_uint16], f32
    lib.zo_f32_to_f16.argtypes, lib.zo_f32_to_f16.restype = [f32], C.c_uint16
    for name in ("zo_gguf_q4_0_to_int8", "zo_gguf_q8_0_to_int8"):
        getattr(lib, name).argtypes, getattr(lib, name).restype = [vp, u64, vp, vp], None
    for name in ("zo_gguf_dequant_q4_0", "zo_gguf_dequant_q8_0"):
        getattr(lib, name).argtypes, getattr(lib, name).restype = [vp, vp, u64], None
    lib.zo_rope_tables.argtypes, lib.zo_rope_tables.restype = [u64, u64, f32, vp, vp], None
    lib.zo_argmax.argtypes, lib.zo_argmax.restype = [vp, u64], C.c_int64
    _lib = lib
    return lib


def set_threads(n: int) -> None:
    load().zo_set_threads(n)


def set_f16_dense(on: bool) -> None:
    """Programs compiled afterwards promote uploaded matmul B operands to f16 (wgpu.zig:1071-1104)."""
    load().zo_set_f16_dense(int(on))


def set_w8a8(on: bool) -> None:
    """Programs compiled afterwards take the W8A8 arm for M = 1 qmatmuls (CPU-baseline variant B3)."""
    load().zo_set_w8a8(int(on))


class OracleBackend:
    """CpuBackend restatement (src/backend/cpu.zig) over the flat program structs."""

    name_str = "oracle-cpu"

    def __init__(self):
        self._lib = load()

    def compileProgram(self, program: DeviceProgram):
        pc, keep = program.to_c()
        return self._lib.zo_compile_program(C.byref(pc))

    def refreshProgram(self, handle, ops: Sequence[DeviceOp]) -> None:
        arr, keep = ops_to_c(ops)
        self._lib.zo_refresh_program(handle, arr, len(ops))

    def executeProgram(self, handle, inputs: Sequence[ProgramIO], outputs: Sequence[ProgramIO]) -> None:
        self._lib.zo_execute_program(handle, ios_to_c(inputs), len(inputs), ios_to_c(outputs), len(outputs))

    def freeProgram(self, handle) -> None:
        self._lib.zo_free_program(handle)

    def buffer(self, handle, idx: int) -> np.ndarray:
        n = C.c_uint64()
        ptr = self._lib.zo_program_buffer(handle, idx, C.byref(n))
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n.value,))


def run_program(program: DeviceProgram, out_idx: int, out_len: int, backend=None) -> np.ndarray:
    """conformance.expectedOutput / backendOutput (src/backend/conformance.zig:13-43)."""
    be = backend or OracleBackend()
    h = be.compileProgram(program)
    if not h:
        raise RuntimeError("compile failed")
    out = np.zeros(out_len, dtype=np.float32)
    try:
        be.executeProgram(h, [], [ProgramIO(out_idx, out)])
    finally:
        be.freeProgram(h)
    return out


# ── quantisation helpers (numpy in / numpy out) ──────────────────────────────

def _p(a: np.ndarray) -> int:
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def quantize_from_slice(weights: np.ndarray, rows: int, cols: int, bs: int = 32):
    w = np.ascontiguousarray(weights, dtype=np.float32).ravel()
    data = np.zeros(rows * cols, dtype=np.int8)
    scales = np.zeros((rows * cols + bs - 1) // bs, dtype=np.float32)
    load().zo_quantize_from_slice(_p(w), rows, cols, bs, _p(data), _p(scales))
    return data, scales


def prepare_transposed(data, scales, K, N, bs=32):
    t_data = np.zeros(N * K, dtype=np.int8)
    t_scales = np.zeros(N * ((K + bs - 1) // bs), dtype=np.float32)
    load().zo_prepare_transposed(_p(data), _p(scales), K, N, bs, _p(t_data), _p(t_scales))
    return t_data, t_scales


def quantize_input(x, bs=32):
    x = np.ascontiguousarray(x, dtype=np.float32)
    K = x.size
    q = np.zeros(K, dtype=np.int8)
    s = np.zeros((K + bs - 1) // bs, dtype=np.float32)
    load().zo_quantize_input(_p(x), K, bs, _p(q), _p(s))
    return q, s


def gemv_uses_vnni() -> bool:
    return bool(load().zo_gemv_uses_vnni())


def set_vnni(on: int) -> None:
    """0: scalar gemvRange loop, 1: the AVX-512 VNNI arm where the host has it, -1: probe again."""
    load().zo_set_vnni(on)


def gemv(t_data, t_scales, x, N, K, bs=32, workers: int = 0):
    q, s = quantize_input(x, bs)
    dst = np.zeros(N, dtype=np.float32)
    if workers > 0:
        load().zo_gemv_pool_dispatch(_p(t_data), _p(t_scales), _p(q), _p(s), _p(dst), N, K, bs, workers)
    else:
        load().zo_gemv_range(_p(t_data), _p(t_scales), _p(q), _p(s), _p(dst), 0, N, K, bs)
    return dst


def qmatmul_exact(data, scales, x, M, N, K, bs=32, input_row_stride=0, dst_row_stride=0, dst=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    if dst is None:
        dst = np.zeros(M * (dst_row_stride or N), dtype=np.float32)
    load().zo_qmatmul_exact(_p(data), _p(scales), bs, _p(x), _p(dst), M, N, K, input_row_stride, dst_row_stride)
    return dst


def dequantize(data, scales, bs=32):
    out = np.zeros(data.size, dtype=np.float32)
    load().zo_dequantize(_p(data), _p(scales), data.size, bs, _p(out))
    return out


def gguf_to_int8(raw: np.ndarray, n_elems: int, kind: str):
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    data = np.zeros(n_elems, dtype=np.int8)
    scales = np.zeros((n_elems + 31) // 32, dtype=np.float32)
    fn = {"q4_0": load().zo_gguf_q4_0_to_int8, "q8_0": load().zo_gguf_q8_0_to_int8}[kind]
    fn(_p(raw), n_elems, _p(data), _p(scales))
    return data, scales


def gguf_dequant(raw: np.ndarray, n_elems: int, kind: str):
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    out = np.zeros(n_elems, dtype=np.float32)
    fn = {"q4_0": load().zo_gguf_dequant_q4_0, "q8_0": load().zo_gguf_dequant_q8_0}[kind]
    fn(_p(out), _p(raw), n_elems)
    return out


def rope_tables(d: int, max_seq: int, base: float = 10000.0):
    cos = np.zeros((max_seq, d), dtype=np.float32)
    sin = np.zeros((max_seq, d), dtype=np.float32)
    load().zo_rope_tables(d, max_seq, base, _p(cos), _p(sin))
    return cos, sin


def argmax(v: np.ndarray) -> int:
    v = np.ascontiguousarray(v, dtype=np.float32)
    return int(load().zo_argmax(_p(v), v.size))


def backend_fns():
    """zgml_amd.llama.BackendFns table backed by the oracle (tests / cpu_baseline only)."""
    from zgml_amd.llama import BackendFns
    lib = load()

    def addr(name):
        return C.cast(getattr(lib, name), C.c_void_p).value
    return BackendFns(None, addr("zo_vt_compile_program"), addr("zo_vt_refresh_program"),
                      addr("zo_vt_execute_program"), addr("zo_vt_free_program"))
