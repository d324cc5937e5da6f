"""Executor over the CPU oracle with the interface zgml_amd.sharded.ShardedDecoder expects
(tests only)."""
import contextlib
import ctypes as C

import numpy as np
import torch

from zgml_amd import capi
from zgml_amd.llama import Model
from oracle import oracle as O


class OracleExecutor:
    def __init__(self, model: Model):
        self.model, self.lib = model, O.load()
        self.lib.zo_program_execute_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        self.lib.zo_program_upload.argtypes = [C.c_void_p, C.POINTER(capi.ProgramIOC), C.c_uint64]
        self.lib.zo_program_download.argtypes = [C.c_void_p, C.POINTER(capi.ProgramIOC), C.c_uint64]
        prog = model.program
        self.handle = self.lib.zo_compile_program(C.byref(prog))
        self.n_ops = int(prog.n_ops)
        n = C.c_uint64()
        self._in = model.lib.zh_model_step_inputs(model.ptr, C.byref(n)), n.value
        n = C.c_uint64()
        self._out = model.lib.zh_model_step_outputs(model.ptr, C.byref(n)), n.value

    def refresh(self):
        prog = self.model.program
        self.lib.zo_refresh_program(self.handle, prog.ops, prog.n_ops)

    def upload(self):
        self.lib.zo_program_upload(self.handle, self._in[0], self._in[1])

    def run_ops(self, first, count):
        if count > 0:
            self.lib.zo_program_execute_range(self.handle, first, count)

    def tensor(self, buf, offset, n):
        ln = C.c_uint64()
        ptr = self.lib.zo_program_buffer(self.handle, buf, C.byref(ln))
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(ln.value,))
        return torch.from_numpy(arr)[offset:offset + n]

    def collective_stream(self):
        return contextlib.nullcontext()

    def download_logits(self):
        self.lib.zo_program_download(self.handle, self._out[0], self._out[1])
        p = self.model.lib.zh_model_logits(self.model.ptr)
        return np.ctypeslib.as_array(p, shape=(self.model.cfg.vocab_size,)).copy()

    def close(self):
        self.lib.zo_free_program(self.handle)
