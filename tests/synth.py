"""Deterministic synthetic data of SURVEY §8d (RNG-free), host side, in the reference's in-memory
form (int8 data + f32 scale per 32 flat elements). Mirrors the device generator in
zgml_amd/csrc/qmatvec.hip (synth_q4 / synth_q8 / synth_scale)."""
import numpy as np


def synth_weight(K: int, N: int, q4: bool, matrix_id: int = 0):
    flat = np.arange(K * N, dtype=np.uint64)
    if q4:
        data = (((flat * 7 + (flat >> 5) * 3 + matrix_id * 5) & 15).astype(np.int16) - 8).astype(np.int8)
    else:
        data = (((flat * 13 + matrix_id * 29) % 255).astype(np.int16) - 127).astype(np.int8)
    blocks = np.arange((K * N + 31) // 32, dtype=np.uint64)
    scales = (0.015625 * (1.0 + ((blocks + matrix_id) % 7).astype(np.float32) * 0.125)).astype(np.float32)
    return data, scales


def synth_x(K: int):
    i = np.arange(K)
    return (((i % 17) - 8) * 0.03125).astype(np.float32)


def q4_0_blocks_from_int8(data: np.ndarray, scales: np.ndarray) -> np.ndarray:
    """Pack int8 [-8,7] + f16-exact scales into the reference's GGUF Q4_0 byte layout
    (18-byte blocks, interleaved nibbles: element i -> byte i/2, even = low nibble;
    src/models/gguf_loader.zig:137-141)."""
    n_blocks = scales.size
    out = np.zeros((n_blocks, 18), np.uint8)
    out[:, :2] = scales.astype(np.float16).view(np.uint8).reshape(n_blocks, 2)
    nib = (data.astype(np.int16) + 8).astype(np.uint8).reshape(n_blocks, 32)
    out[:, 2:] = nib[:, 0::2] | (nib[:, 1::2] << 4)
    return out.ravel()
