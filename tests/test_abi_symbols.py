"""CPU-side checks of the drop-in boundary: the built library exports every symbol that
include/zgml_hip.h declares, the ctypes mirror matches the C struct sizes, and the pure host
logic (capabilities, program support) behaves like src/backend.zig. No compute calls."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

from zgml_amd import Capabilities, DeviceOp, DeviceProgram, FusedEwStep, QuantizedWeightUpload, capi

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    if not capi.HIP_LIB_PATH.exists():
        import __graft_entry__ as g
        g.build_hip()
    return capi.load_hip()


def test_header_symbols_exported(lib):
    header = (ROOT / "include" / "zgml_hip.h").read_text()
    declared = set(re.findall(r"\b(zgml_hip_[a-z0-9_]+)\s*\(", header))
    assert declared == set(capi.HIP_SYMBOLS), declared ^ set(capi.HIP_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_ctypes_layout_matches_c(tmp_path):
    src = tmp_path / "sz.c"
    names = ["zgml_matmul_geom", "zgml_fused_step", "zgml_device_op", "zgml_program_io", "zgml_qweight_upload",
             "zgml_device_program", "zgml_capabilities", "zgml_runtime_profile", "zgml_op_attention",
             "zgml_op_repeat", "zgml_op_slice_assign"]
    body = "".join(f'printf("%zu\\n", sizeof({n}));' for n in names)
    body += 'printf("%zu\\n", offsetof(zgml_device_op, u));'
    body += 'printf("%zu\\n", offsetof(zgml_op_attention, seq_kv));'
    body += 'printf("%zu\\n", offsetof(zgml_op_slice_assign, dst_offset));'
    src.write_text(f'#include <stdio.h>\n#include "zgml_hip.h"\nint main(){{{body}return 0;}}')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    py = [C.sizeof(t) for t in (capi.MatMulGeom, capi.FusedStep, capi.DeviceOpC, capi.ProgramIOC,
                                capi.QWeightUploadC, capi.DeviceProgramC, capi.CapabilitiesC, capi.RuntimeProfileC,
                                capi.OpAttention, capi.OpRepeat, capi.OpSliceAssign)]
    py += [capi.DeviceOpC.u.offset, capi.OpAttention.seq_kv.offset, capi.OpSliceAssign.dst_offset.offset]
    assert out == py


def test_zig_abi_asserts_are_current():
    """zig/abi_asserts.zig (the `comptime` size / offset asserts of the Zig adapter, zig/backend_hip.zig) is what
    tools/gen_zig_abi_asserts.py generates from include/zgml_hip.h TODAY: header, ctypes mirror and Zig adapter share one table."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_zig_abi_asserts", ROOT / "tools" / "gen_zig_abi_asserts.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert (ROOT / "zig" / "abi_asserts.zig").read_text() == mod.render()
    adapter = (ROOT / "zig" / "backend_hip.zig").read_text()
    for entry in ("zgml_hip_create", "zgml_hip_destroy", "zgml_hip_dense_matmul_f32", "zgml_hip_compile_program", "zgml_hip_refresh_program",
                  "zgml_hip_execute_program", "zgml_hip_free_program", "zgml_hip_get_runtime_profile", "zgml_hip_last_error"):
        assert "c." + entry + "(" in adapter, entry  # every vtable entry of src/backend.zig:338-352 forwards to its C entry point
    for tag in capi.DOP_KINDS:
        assert f".{tag} => |" in adapter, tag  # every DeviceOp arm is flattened


def test_capabilities_hip(lib):
    c = capi.CapabilitiesC()
    lib.zgml_hip_capabilities(C.byref(c))
    caps = Capabilities.from_c(c)
    assert caps.compiled_programs and caps.qmatmul and caps.dense_matmul_f32 and caps.fused_elementwise
    assert not caps.host_visible_program_memory
    assert caps.max_fused_elementwise_steps == 8
    assert caps.attention.supports(1 << 20, 512) and not caps.attention.supports(1, 513)


def test_program_supported_matches_python(lib):
    c = capi.CapabilitiesC()
    lib.zgml_hip_capabilities(C.byref(c))
    caps = Capabilities.from_c(c)
    qd, sc = np.array([1, 2, 3, 4], np.int8), np.array([1], np.float32)
    progs = [
        DeviceProgram(ops=[DeviceOp.fused_elementwise([FusedEwStep("relu")] * 8, 1, 1, 0)], buffer_sizes=[1, 1]),
        DeviceProgram(ops=[DeviceOp.fused_elementwise([FusedEwStep("relu")] * 9, 1, 1, 0)], buffer_sizes=[1, 1]),
        DeviceProgram(ops=[DeviceOp.qmatmul(1, 0, 0, 1, 2, 2)], buffer_sizes=[2, 2],
                      qweights=[QuantizedWeightUpload(qd, sc, 2, 2, 4)]),
        DeviceProgram(ops=[DeviceOp.qmatmul(1, 0, 0, 1, 2, 2)], buffer_sizes=[2, 2],
                      qweights=[QuantizedWeightUpload(qd, sc, 3, 2, 4)]),
        DeviceProgram(ops=[DeviceOp.elementwise("add", 2, 0, 1, 4)], buffer_sizes=[4, 4]),  # buffer 2 missing
        DeviceProgram(ops=[DeviceOp.reduce("mul", 1, 0, 1, 2)], buffer_sizes=[2, 1]),
    ]
    want = [True, False, True, False, False, False]
    for p, w in zip(progs, want):
        pc, keep = p.to_c()
        assert bool(lib.zgml_hip_program_supported(C.byref(pc))) is w
        assert p.isSupportedBy(caps) is w


def test_create_fails_loudly_without_gpu(lib):
    import os
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    assert not lib.zgml_hip_create(0)
    assert lib.zgml_hip_last_error(None)
