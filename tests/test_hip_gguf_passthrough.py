"""Packed-GGUF pass-through: a QuantizedWeightUpload that carries the tensor's raw
Q4_0 / Q8_0 file blocks is unpacked and re-packed on the device. The result must be BIT-IDENTICAL to
uploading the reference's expanded form (int8 + f32 scales) of the same blocks, because both end in
the same packed device layout; and equal to the oracle (which expands the blocks with its restatement
of quantizedWeightFromInfo, src/models/gguf_loader.zig:99-154) within the mat-vec tolerance."""
import numpy as np
import pytest

from zgml_amd import DeviceOp, DeviceProgram, ProgramIO, QuantizedWeightUpload
from tests.synth import q4_0_blocks_from_int8

pytestmark = pytest.mark.gpu
f32 = np.float32


def q8_0_blocks(data, scales):
    nb = scales.size
    out = np.zeros((nb, 34), np.uint8)
    out[:, :2] = scales.astype(np.float16).view(np.uint8).reshape(nb, 2)
    out[:, 2:] = data.reshape(nb, 32).view(np.uint8)
    return out.ravel()


def make(kind, K, N, rng):
    if kind == "q4_0":
        data = rng.integers(-8, 8, K * N).astype(np.int8)
        scales = (rng.random(K * N // 32).astype(np.float16) * 0.05 + 0.001).astype(f32)
        return data, scales, q4_0_blocks_from_int8(data, scales)
    data = rng.integers(-127, 128, K * N).astype(np.int8)
    scales = (rng.random(K * N // 32).astype(np.float16) * 0.01 + 0.001).astype(f32)
    return data, scales, q8_0_blocks(data, scales)


@pytest.mark.parametrize("kind", ["q4_0", "q8_0"])
@pytest.mark.parametrize("M,K,N", [(1, 576, 192), (1, 4096, 512), (1, 100, 64), (5, 576, 192), (32, 1536, 96)])
def test_packed_upload_equals_expanded_upload(hip_backend, oracle, kind, M, K, N):
    rng = np.random.default_rng(K * 7 + N + M)
    data, scales, raw = make(kind, K, N, rng)
    d2, s2 = oracle.gguf_to_int8(raw, data.size, kind)  # the loader restatement agrees with the generator
    assert np.array_equal(d2, data) and np.array_equal(s2, scales)
    x = rng.standard_normal(M * K).astype(f32)

    def program(qw):
        return DeviceProgram(ops=[DeviceOp.qmatmul(1, 0, 0, M, N, K)], buffer_sizes=[M * K, M * N],
                             initial_uploads=[ProgramIO(0, x)], qweights=[qw])
    expanded = oracle.run_program(program(QuantizedWeightUpload(data, scales, K, N, 32)), 1, M * N, backend=hip_backend)
    packed_prog = program(QuantizedWeightUpload.from_gguf_blocks(raw, K, N, kind))
    packed = oracle.run_program(packed_prog, 1, M * N, backend=hip_backend)
    assert not hip_backend.last_error(), hip_backend.last_error()
    assert np.array_equal(packed, expanded)
    want = oracle.run_program(packed_prog, 1, M * N)  # the oracle accepts the packed form too
    w = np.abs(data.astype(np.float64) * np.repeat(scales.astype(np.float64), 32)).reshape(K, N)
    bound = np.abs(x.reshape(M, K).astype(np.float64)) @ w
    assert np.all(np.abs(packed.reshape(M, N) - want.reshape(M, N)) <= 2e-5 * bound + 1e-30)


def test_malformed_packed_upload_is_rejected(hip_backend):
    raw = np.zeros(64 * 32 // 32 * 18 + 1, np.uint8)  # one byte too many: neither form
    qw = QuantizedWeightUpload(raw, np.zeros(0, f32), 64, 32, 32, gguf="q4_0")
    prog = DeviceProgram(ops=[DeviceOp.qmatmul(1, 0, 0, 1, 32, 64)], buffer_sizes=[64, 32], qweights=[qw])
    assert hip_backend._lib.zgml_hip_program_supported(prog.to_c()[0]) == 0
