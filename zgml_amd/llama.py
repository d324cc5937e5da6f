"""Python binding of the C++ host side (zgml_amd/host): synthetic LLaMA models, the decode
DeviceProgram (DeviceInference lowering) and the per-token session loop. The compute backend is a
function table with the zgml_hip_* signatures — the HIP library in the product, the oracle's
`zo_vt_*` wrappers in tests."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import capi

Q4_0, Q8_0, F32_DENSE, Q4_0_GGUF = 0, 1, 2, 3


class Config(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("vocab_size", "d_model", "n_heads", "n_kv_heads", "d_ff", "n_layers",
                                           "max_seq_len")] + [
        ("rope_base", C.c_float), ("rms_norm_eps", C.c_float), ("tied_lm_head", C.c_uint32),
        ("shard_rank", C.c_uint32), ("shard_world", C.c_uint32), ("kv_quant_block", C.c_uint32)]

    @property
    def d_head(self):
        return self.d_model // self.n_heads


class BackendFns(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("compile_program", C.c_void_p), ("refresh_program", C.c_void_p),
                ("execute_program", C.c_void_p), ("free_program", C.c_void_p)]


class GatherPoint(C.Structure):
    _fields_ = [("op_end", C.c_uint32), ("buf", C.c_uint16), ("_pad", C.c_uint16), ("offset", C.c_uint32),
                ("len_per_rank", C.c_uint32)]


_lib = None


def load_host() -> C.CDLL:
    global _lib
    if _lib is None:
        path = capi.HOST_LIB_PATH
        if not Path(path).exists():
            raise RuntimeError(f"{path} not found: run __graft_entry__.build()")
        lib = C.CDLL(str(path))
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        lib.zh_preset.argtypes, lib.zh_preset.restype = [C.c_char_p, u32, C.POINTER(Config)], None
        lib.zh_model_create.argtypes, lib.zh_model_create.restype = [C.POINTER(Config), C.c_int, C.c_int, C.c_int, C.c_int], vp
        lib.zh_model_create_ex.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_int, C.c_int, u32]
        lib.zh_model_create_ex.restype = vp
        lib.zh_model_patch_tokens.argtypes, lib.zh_model_patch_tokens.restype = [vp, vp, u32], None
        lib.zh_session_prefill.argtypes, lib.zh_session_prefill.restype = [vp, vp, u32, vp], C.c_int64
        lib.zh_model_free.argtypes, lib.zh_model_free.restype = [vp], None
        lib.zh_model_program.argtypes, lib.zh_model_program.restype = [vp], C.POINTER(capi.DeviceProgramC)
        lib.zh_model_patch.argtypes, lib.zh_model_patch.restype = [vp, u32, u32], None
        lib.zh_model_step_inputs.argtypes, lib.zh_model_step_inputs.restype = [vp, C.POINTER(u64)], C.POINTER(capi.ProgramIOC)
        lib.zh_model_step_outputs.argtypes, lib.zh_model_step_outputs.restype = [vp, C.POINTER(u64)], C.POINTER(capi.ProgramIOC)
        lib.zh_model_logits.argtypes, lib.zh_model_logits.restype = [vp], C.POINTER(C.c_float)
        lib.zh_model_token_embed.argtypes, lib.zh_model_token_embed.restype = [vp], C.POINTER(C.c_float)
        lib.zh_model_rope_table.argtypes, lib.zh_model_rope_table.restype = [vp, C.c_int], C.POINTER(C.c_float)
        lib.zh_model_buf.argtypes, lib.zh_model_buf.restype = [vp, C.c_int, u32], C.c_uint16
        lib.zh_model_gather_points.argtypes, lib.zh_model_gather_points.restype = [vp, C.POINTER(GatherPoint), u64], u64
        lib.zh_model_kv_buffers.argtypes = [vp, C.POINTER(C.c_uint16), C.POINTER(u64), u64]
        lib.zh_model_kv_buffers.restype = u64
        lib.zh_model_quant_bytes.argtypes, lib.zh_model_quant_bytes.restype = [vp, C.POINTER(u64)], u64
        lib.zh_session_create.argtypes, lib.zh_session_create.restype = [vp, C.POINTER(BackendFns)], vp
        lib.zh_session_free.argtypes, lib.zh_session_free.restype = [vp], None
        lib.zh_session_handle.argtypes, lib.zh_session_handle.restype = [vp], vp
        lib.zh_session_step.argtypes, lib.zh_session_step.restype = [vp, u32, u32, vp], C.c_int64
        lib.zh_session_decode.argtypes, lib.zh_session_decode.restype = [vp, u32, u32, u32, vp], C.c_double
        lib.zh_session_set_refresh_dynamic.argtypes, lib.zh_session_set_refresh_dynamic.restype = [vp, vp], None
        lib.zh_argmax.argtypes, lib.zh_argmax.restype = [vp, u32], u32
        lib.zh_argmax_reference.argtypes, lib.zh_argmax_reference.restype = [vp, u32], u32
        _lib = lib
    return _lib


def preset(name: str, max_seq: int = 0) -> Config:
    c = Config()
    load_host().zh_preset(name.encode(), max_seq, C.byref(c))
    return c


def _fn_addr(lib, name) -> int:
    return C.cast(getattr(lib, name), C.c_void_p).value


def hip_backend_fns(backend) -> BackendFns:
    """Function table of the HIP library for context `backend.ctx` (zgml_amd.Backend)."""
    lib = capi.load_hip()
    return BackendFns(backend.ctx, _fn_addr(lib, "zgml_hip_compile_program"), _fn_addr(lib, "zgml_hip_refresh_program"),
                      _fn_addr(lib, "zgml_hip_execute_program"), _fn_addr(lib, "zgml_hip_free_program"))


class Model:
    """Synthetic LLaMA weights + the decode DeviceProgram (`DeviceInference.init`, token_len = 1)."""

    def __init__(self, cfg: Config, weight_kind: int = Q4_0, fused_elementwise: bool = True,
                 include_dead_f32: bool = False, threads: int = 8, token_len: int = 1):
        self.lib = load_host()
        self.cfg = cfg
        self.token_len = token_len
        self.ptr = self.lib.zh_model_create_ex(C.byref(cfg), weight_kind, int(fused_elementwise), int(include_dead_f32),
                                               threads, token_len)
        if not self.ptr:
            raise ValueError("invalid LlamaConfig / shard spec")

    def close(self):
        if self.ptr:
            self.lib.zh_model_free(self.ptr)
            self.ptr = None

    @property
    def program(self) -> capi.DeviceProgramC:
        return self.lib.zh_model_program(self.ptr).contents

    def patch(self, token: int, pos: int) -> None:
        self.lib.zh_model_patch(self.ptr, token, pos)

    def buf(self, which: str, layer: int = 0) -> int:
        idx = {"token_input": 0, "attn_mask": 1, "logits": 2, "rope": 3, "k_cache": 4, "v_cache": 5}[which]
        return int(self.lib.zh_model_buf(self.ptr, idx, layer))

    def token_embed(self) -> np.ndarray:
        p = self.lib.zh_model_token_embed(self.ptr)
        return np.ctypeslib.as_array(p, shape=(self.cfg.vocab_size, self.cfg.d_model))

    def rope_tables(self):
        shape = (self.cfg.max_seq_len, self.cfg.d_head)
        return (np.ctypeslib.as_array(self.lib.zh_model_rope_table(self.ptr, 0), shape=shape),
                np.ctypeslib.as_array(self.lib.zh_model_rope_table(self.ptr, 1), shape=shape))

    def gather_points(self):
        n = self.lib.zh_model_gather_points(self.ptr, None, 0)
        arr = (GatherPoint * max(1, n))()
        self.lib.zh_model_gather_points(self.ptr, arr, n)
        return list(arr[:n])

    def kv_buffers(self):
        """[(buffer id, f32 elements)] of every KV-cache buffer, in builder order (same order in a prefill plan and the
        decode plan of one config: the hand-off of llama_inference.prefill -> step is a pairwise copy)."""
        n = self.lib.zh_model_kv_buffers(self.ptr, None, None, 0)
        bufs, elems = (C.c_uint16 * max(1, n))(), (C.c_uint64 * max(1, n))()
        self.lib.zh_model_kv_buffers(self.ptr, bufs, elems, n)
        return [(int(bufs[i]), int(elems[i])) for i in range(n)]

    def quant_bytes(self):
        n = C.c_uint64()
        b = self.lib.zh_model_quant_bytes(self.ptr, C.byref(n))
        return int(b), int(n.value)


class Session:
    """`LlamaDeviceSession`: compile once, then doStep per token through the vtable."""

    def __init__(self, model: Model, fns: BackendFns):
        self.model, self.lib, self.fns = model, model.lib, fns
        self.ptr = self.lib.zh_session_create(model.ptr, C.byref(fns))
        if not self.ptr:
            raise RuntimeError("compile_program failed")

    @property
    def handle(self):
        return self.lib.zh_session_handle(self.ptr)

    def use_dynamic_refresh(self, on: bool = True) -> None:
        """The adapter's per-token refresh (zig/backend_hip.zig: refreshProgram): (slice_pos, seq_kv) through
        zgml_hip_refresh_dynamic instead of the whole op list through zgml_hip_refresh_program. HIP sessions only."""
        self.lib.zh_session_set_refresh_dynamic(self.ptr, _fn_addr(capi.load_hip(), "zgml_hip_refresh_dynamic") if on else None)

    def pin_outputs(self, backend, on: bool = True) -> None:
        """The adapter's promise that the session's logits buffer outlives the program (zgml_hip_program_pin_outputs): the step's
        last kernel then writes the logits straight into it. HIP sessions only."""
        capi.load_hip().zgml_hip_program_pin_outputs(backend.ctx, self.handle, 1 if on else 0)

    def step(self, token: int, pos: int, want_logits: bool = True):
        logits = np.zeros(self.model.cfg.vocab_size, np.float32) if want_logits else None
        nxt = self.lib.zh_session_step(self.ptr, token, pos, logits.ctypes.data if want_logits else None)
        return int(nxt), logits

    def prefill(self, tokens, pos: int, want_logits: bool = True):
        """One execution of a token_len = N plan: returns (greedy token, logits of the last position)."""
        toks = np.ascontiguousarray(tokens, dtype=np.uint32)
        assert toks.size == self.model.token_len
        logits = np.zeros(self.model.cfg.vocab_size, np.float32) if want_logits else None
        nxt = self.lib.zh_session_prefill(self.ptr, toks.ctypes.data, pos, logits.ctypes.data if want_logits else None)
        return int(nxt), logits

    def decode(self, first_token: int, start_pos: int, n_steps: int):
        toks = np.zeros(n_steps, np.int64)
        secs = self.lib.zh_session_decode(self.ptr, first_token, start_pos, n_steps, toks.ctypes.data)
        return toks, secs

    # ── device-resident loop (HIP backend only; include/zgml_hip.h zgml_hip_resident_*) ──
    def resident_setup(self, backend) -> None:
        m, cfg = self.model, self.model.cfg
        hip = capi.load_hip()
        cos, sin = m.rope_tables()
        ropes = (C.c_uint16 * cfg.n_layers)(*[m.buf("rope", l) for l in range(cfg.n_layers)])
        d = capi.ResidentLlamaC()
        d.token_embed = m.token_embed().ctypes.data
        d.cos_table, d.sin_table = cos.ctypes.data, sin.ctypes.data
        d.vocab, d.d_model, d.max_seq, d.d_head = cfg.vocab_size, cfg.d_model, cfg.max_seq_len, cfg.d_head
        d.buf_token_input, d.buf_attn_mask, d.buf_logits = m.buf("token_input"), m.buf("attn_mask"), m.buf("logits")
        d.buf_rope, d.n_rope = C.cast(ropes, C.POINTER(C.c_uint16)), cfg.n_layers
        self._resident_keep = (ropes, cos, sin)
        if hip.zgml_hip_resident_setup(backend.ctx, self.handle, C.byref(d)) != 0:
            raise RuntimeError("resident_setup: " + backend.last_error())
        self._backend = backend

    def resident_decode(self, first_token: int, start_pos: int, n_steps: int) -> np.ndarray:
        toks = np.zeros(n_steps, np.int64)
        rc = capi.load_hip().zgml_hip_resident_decode(self._backend.ctx, self.handle, first_token, start_pos, n_steps,
                                                      toks.ctypes.data)
        if rc != 0:
            raise RuntimeError("resident_decode: " + self._backend.last_error())
        return toks

    def resident_prefill(self, tokens, start_pos: int) -> int:
        """One chunk of a token_len = N plan with on-device embedding gather / mask / RoPE rows / argmax."""
        toks = np.ascontiguousarray(tokens, dtype=np.uint32)
        nxt = int(capi.load_hip().zgml_hip_resident_prefill(self._backend.ctx, self.handle, toks.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                            toks.size, start_pos))
        if nxt < 0:
            raise RuntimeError("resident_prefill: " + self._backend.last_error())
        return nxt

    def close(self):
        if self.ptr:
            self.lib.zh_session_free(self.ptr)
            self.ptr = None
