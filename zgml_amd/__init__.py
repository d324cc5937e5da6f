"""zgml_amd — MI355X (gfx950) backend for zgml's forward-inference hot path.

The product is the C-ABI library `zgml_amd/lib/libzgml_hip.so` (include/zgml_hip.h); this
package is the thin Python host mirror used by tests and harnesses. No CPU fallback exists:
everything that computes goes through the HIP library.
"""
from . import capi  # noqa: F401
from .program import (  # noqa: F401
    Attention, Backend, Capabilities, DenseMatMulSpecF32, DeviceOp, DeviceProgram, FusedEwStep,
    MatMulGeometry, ProgramIO, QuantizedWeightUpload, tryDenseMatMul)
