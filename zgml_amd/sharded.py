"""Row-sharded (N-split) Llama decode across ranks: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI) all-gathers between op ranges of the per-rank DeviceProgram.

The reference has no distributed code; this is the multi-GPU extension of its decode
path: every quantized weight is split along N (output features, whole 32-column scale
blocks), activations are replicated, KV caches are head-sharded. Per layer there are 4 gathers
(attention output, O-proj, SwiGLU activation, down-proj) plus one for the logits; each moves
N/ws f32 per rank (2-44 KB): latency-bound, not bandwidth-bound.

`ShardedDecoder` is executor-agnostic: `HipExecutor` drives the HIP library; the CPU tests plug
the oracle in through the same interface with the gloo backend.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import capi
from .llama import Model


class NativeShardedDecoder:
    """The row-shard step behind the C ABI (zgml_hip_shard_*): the library owns the RCCL communicator and enqueues the
    all-gathers itself, one graph per token. Python only patches the per-token leaves (the host mirror of
    LlamaDeviceSession) and hands the 128-byte communicator id around — `exchange_id` is any callable that returns
    rank 0's bytes on every rank (bench.py broadcasts it over a gloo control plane; world size 1 needs none)."""

    def __init__(self, backend, model: Model, rank: int, world: int, exchange_id=None, gather: str = "rccl", exchange_handles=None):
        """gather = "rccl": in-place ncclAllGather per gather point (`exchange_id` hands rank 0's 128-byte id to every rank);
        gather = "peer": the peer-store gather of zgml_amd/csrc/shard_peer.hip, no RCCL — `exchange_handles` is a callable that
        takes this rank's handle bytes and returns every rank's, in rank order (bench.py: all_gather_object over gloo), or None
        when the caller wires the ranks itself (`export_handle` / `import_handle`: ranks as contexts of one process)."""
        self.be, self.model, self.lib = backend, model, capi.load_hip()
        self.rank, self.world, self.gather = rank, world, gather
        if gather == "peer":
            if self.lib.zgml_hip_shard_init_peer(backend.ctx, rank, world) != 0:
                raise RuntimeError("shard_init_peer: " + backend.last_error())
        else:
            uid = (C.c_ubyte * 128)()
            if rank == 0 and self.lib.zgml_hip_shard_unique_id(uid) != 0:
                raise RuntimeError("shard_unique_id failed (librccl.so?)")
            if exchange_id is not None:
                raw = exchange_id(bytes(uid))
                uid = (C.c_ubyte * 128)(*raw)
            if self.lib.zgml_hip_shard_init(backend.ctx, uid, rank, world) != 0:
                raise RuntimeError("shard_init: " + backend.last_error())
        prog = model.program
        self.handle = self.lib.zgml_hip_compile_program(backend.ctx, C.byref(prog))
        if not self.handle:
            raise RuntimeError("compile failed: " + backend.last_error())
        pts = model.gather_points()
        arr = (capi.ShardPointC * max(1, len(pts)))()
        for i, gp in enumerate(pts):
            arr[i] = capi.ShardPointC(gp.op_end, gp.buf, 0, gp.offset, gp.len_per_rank)
        if self.lib.zgml_hip_shard_attach(backend.ctx, self.handle, arr, len(pts), model.buf("logits"), model.cfg.vocab_size) != 0:
            raise RuntimeError("shard_attach: " + backend.last_error())
        self.n_points = len(pts)
        n = C.c_uint64()
        self._in = model.lib.zh_model_step_inputs(model.ptr, C.byref(n)), n.value
        if gather == "peer" and exchange_handles is not None:
            for r, raw in enumerate(exchange_handles(self.export_handle())):
                self.import_handle(r, raw)

    def export_handle(self) -> bytes:
        h = capi.ShardPeerHandleC()
        if self.lib.zgml_hip_shard_peer_export(self.be.ctx, self.handle, C.byref(h)) != 0:
            raise RuntimeError("shard_peer_export: " + self.be.last_error())
        return bytes(h)

    def import_handle(self, peer_rank: int, raw: bytes) -> None:
        h = capi.ShardPeerHandleC.from_buffer_copy(raw)
        if self.lib.zgml_hip_shard_peer_import(self.be.ctx, self.handle, peer_rank, C.byref(h)) != 0:
            raise RuntimeError("shard_peer_import: " + self.be.last_error())

    def step(self, token: int, pos: int) -> int:
        self.model.patch(token, pos)
        prog = self.model.program
        self.lib.zgml_hip_refresh_program(self.be.ctx, self.handle, prog.ops, prog.n_ops)
        tok = int(self.lib.zgml_hip_shard_step(self.be.ctx, self.handle, self._in[0], self._in[1]))
        if tok < 0:
            raise RuntimeError("shard_step: " + self.be.last_error())
        return tok

    def profile_step(self, token: int, pos: int):
        """One EAGER step with HIP events around every all-gather: (token, device us of the step, us inside the collectives).
        Collective: every rank calls it."""
        self.model.patch(token, pos)
        prog = self.model.program
        self.lib.zgml_hip_refresh_program(self.be.ctx, self.handle, prog.ops, prog.n_ops)
        step_us, gather_us = C.c_double(), C.c_double()
        tok = int(self.lib.zgml_hip_shard_profile_step(self.be.ctx, self.handle, self._in[0], self._in[1], C.byref(step_us), C.byref(gather_us)))
        if tok < 0:
            raise RuntimeError("shard_profile_step: " + self.be.last_error())
        return tok, step_us.value, gather_us.value

    def last_point_us(self):
        """Microseconds of every gather point in the last profile_step (diagnostics)."""
        n = int(self.lib.zgml_hip_shard_last_point_us(self.handle, None, 0))
        buf = (C.c_double * max(1, n))()
        self.lib.zgml_hip_shard_last_point_us(self.handle, buf, n)
        return [round(buf[i], 2) for i in range(n)]

    @property
    def mode(self) -> str:
        return "graph" if self.lib.zgml_hip_shard_step_mode(self.handle) == 1 else "eager"

    def close(self):
        if self.handle:
            self.lib.zgml_hip_free_program(self.be.ctx, self.handle)
            self.handle = None
        self.lib.zgml_hip_shard_destroy(self.be.ctx)


class _DevArray:
    """Minimal __cuda_array_interface__ holder so torch can wrap a raw device pointer."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class HipExecutor:
    def __init__(self, backend, model: Model, device: int):
        self.be, self.model, self.lib = backend, model, capi.load_hip()
        prog = model.program
        self.handle = self.lib.zgml_hip_compile_program(backend.ctx, C.byref(prog))
        if not self.handle:
            raise RuntimeError("compile failed: " + backend.last_error())
        self.n_ops = int(prog.n_ops)
        pts = [gp.op_end for gp in model.gather_points()]
        if pts:  # batched launches must not straddle the collectives
            arr = (C.c_uint64 * len(pts))(*pts)
            self.lib.zgml_hip_program_set_barriers(backend.ctx, self.handle, arr, len(pts))
        self.device = torch.device("cuda", device)
        self.stream = torch.cuda.ExternalStream(self.lib.zgml_hip_stream(backend.ctx), device=self.device)
        self._views = {}
        n = C.c_uint64()
        self._in = model.lib.zh_model_step_inputs(model.ptr, C.byref(n)), n.value
        n = C.c_uint64()
        self._out = model.lib.zh_model_step_outputs(model.ptr, C.byref(n)), n.value

    def refresh(self):
        prog = self.model.program
        self.lib.zgml_hip_refresh_program(self.be.ctx, self.handle, prog.ops, prog.n_ops)

    def upload(self):
        self.lib.zgml_hip_upload_inputs(self.be.ctx, self.handle, self._in[0], self._in[1])

    # capture-friendly split (one graph per token): host side / device side of the upload, argmax
    def stage(self):
        if self.lib.zgml_hip_stage_inputs(self.be.ctx, self.handle, self._in[0], self._in[1]) != 0:
            raise RuntimeError("stage_inputs: " + self.be.last_error())

    def enqueue_staged(self):
        self.lib.zgml_hip_enqueue_staged(self.be.ctx, self.handle)

    def enqueue_argmax(self, buf: int, n: int):
        if self.lib.zgml_hip_enqueue_argmax(self.be.ctx, self.handle, buf, 0, n) != 0:
            raise RuntimeError("enqueue_argmax failed")

    def argmax_result(self) -> int:
        return int(self.lib.zgml_hip_argmax_result(self.be.ctx))

    def run_ops(self, first: int, count: int):
        if count > 0:
            self.lib.zgml_hip_enqueue_ops(self.be.ctx, self.handle, first, count)

    def tensor(self, buf: int, offset: int, n: int) -> torch.Tensor:
        key = (buf, offset, n)
        if key not in self._views:
            base = self.lib.zgml_hip_program_buffer_ptr(self.handle, buf)
            self._views[key] = torch.as_tensor(_DevArray(base + 4 * offset, n), device=self.device)
        return self._views[key]

    def collective_stream(self):
        return torch.cuda.stream(self.stream)

    def download_logits(self) -> np.ndarray:
        self.lib.zgml_hip_download_outputs(self.be.ctx, self.handle, self._out[0], self._out[1])
        p = self.model.lib.zh_model_logits(self.model.ptr)
        return np.ctypeslib.as_array(p, shape=(self.model.cfg.vocab_size,))

    def synchronize(self):
        self.lib.zgml_hip_synchronize(self.be.ctx)

    def close(self):
        if self.handle:
            self.lib.zgml_hip_free_program(self.be.ctx, self.handle)
            self.handle = None


class ShardedDecoder:
    """One decode step = op ranges separated by in-place all-gathers of replicated activations."""

    def __init__(self, model: Model, executor, group=None, inplace: bool = True):
        self.model, self.ex, self.group = model, executor, group
        self.ws = model.cfg.shard_world
        self.rank = model.cfg.shard_rank
        self.points = model.gather_points()
        self.inplace = inplace
        self.graph = None
        # rehearsal switch: issue the collectives even at world size 1 (exercises RCCL under capture)
        import os
        self.force_gather = os.environ.get("ZGML_SHARD_FORCE_GATHER", "0") == "1"

    def _all_gather(self, full: torch.Tensor, n: int):
        mine = full[self.rank * n:(self.rank + 1) * n]
        if self.inplace:
            dist.all_gather_into_tensor(full, mine, group=self.group)
        else:  # backends without an in-place flat all-gather (gloo)
            chunks = [torch.empty_like(mine) for _ in range(self.ws)]
            dist.all_gather(chunks, mine.clone(), group=self.group)
            for r, c in enumerate(chunks):
                full[r * n:(r + 1) * n].copy_(c)

    def step(self, token: int, pos: int, download: bool = True):
        self.model.patch(token, pos)
        self.ex.refresh()
        self.ex.upload()
        with self.ex.collective_stream():
            self._segments()
        return self.ex.download_logits() if download else None

    def _segments(self):
        prev = 0
        for gp in self.points:
            self.ex.run_ops(prev, gp.op_end - prev)
            if self.ws > 1 or self.force_gather:
                self._all_gather(self.ex.tensor(gp.buf, gp.offset, self.ws * gp.len_per_rank), gp.len_per_rank)
            prev = gp.op_end
        self.ex.run_ops(prev, self.ex.n_ops - prev)

    # ── one graph per token ───────────────────────────────────────────────────────────────────
    # The eager step costs ~130 host-issued collectives plus ~650 kernel launches per token; the
    # whole device side of a step (H2D of the staged inputs, op ranges, RCCL all-gathers, argmax)
    # is recorded ONCE into a graph on the library's stream and replayed per token. Host work per
    # token: patch the leaves, refresh the dynamic words, fill the pinned staging buffer, replay.
    def capture(self, logits_buf: int) -> None:
        import torch
        ex = self.ex
        ex.stage()  # builds the transfer table outside the capture
        ex.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=ex.stream, capture_error_mode="relaxed"):
            ex.enqueue_staged()
            self._segments()
            ex.enqueue_argmax(logits_buf, self.model.cfg.vocab_size)
        self.graph = g

    def step_graph(self, token: int, pos: int) -> int:
        self.model.patch(token, pos)
        self.ex.refresh()
        self.ex.stage()
        with self.ex.collective_stream():  # replay on the library's stream (torch replays on its CURRENT stream)
            self.graph.replay()
        self.ex.synchronize()
        return self.ex.argmax_result()
