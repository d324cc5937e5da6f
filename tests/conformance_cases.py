"""The backend conformance cases of the reference (src/backend/conformance.zig:62-346), restated
as DeviceProgram builders: fixed inputs, expected output = the reference executor's semantics
(our oracle). Each entry: (name, program, out_idx, out_len)."""
import numpy as np

from zgml_amd import (DeviceOp, DeviceProgram, FusedEwStep, MatMulGeometry, ProgramIO,
                      QuantizedWeightUpload)

f32 = np.float32


def _a(*v):
    return np.array(v, dtype=f32)


def core_cases():
    cases = []
    # matmul 2x3 * 3x2 (:63-79)
    cases.append(("matmul", DeviceProgram(
        ops=[DeviceOp.matmul(2, 0, 1, MatMulGeometry(2, 2, 3, 3, 1, 2, 1, 0, 0, 0, 2))],
        buffer_sizes=[6, 6, 4],
        initial_uploads=[ProgramIO(0, _a(1, 2, 3, 4, 5, 6)), ProgramIO(1, _a(7, 8, 9, 10, 11, 12))]), 2, 4))
    # qmatmul with offsets/strides, sentinels untouched (:81-112)
    cases.append(("qmatmul_offsets", DeviceProgram(
        ops=[DeviceOp.qmatmul(1, 0, 0, 2, 3, 3, input_offset=1, input_row_stride=4, dst_offset=1, dst_row_stride=4)],
        buffer_sizes=[9, 9],
        initial_uploads=[ProgramIO(0, _a(99, 1, 2, 3, 99, -1, 0.5, 4, 99)), ProgramIO(1, np.full(9, -7, f32))],
        qweights=[QuantizedWeightUpload(np.array([2, -1, 3, 4, -2, 1, -3, 5, 2], np.int8), _a(0.5, 0.25, 1.0), 3, 3, 4)]),
        1, 9))
    # elementwise add (:114-131)
    cases.append(("add", DeviceProgram(
        ops=[DeviceOp.elementwise("add", 2, 0, 1, 4)], buffer_sizes=[4, 4, 4],
        initial_uploads=[ProgramIO(0, _a(1, 2, 3, 4)), ProgramIO(1, _a(10, 20, 30, 40))]), 2, 4))
    # reduce sum + max (:133-156)
    cases.append(("reduce", DeviceProgram(
        ops=[DeviceOp.reduce("sum", 1, 0, 2, 3), DeviceOp.reduce("max", 1, 0, 2, 3, dst_offset=2)],
        buffer_sizes=[6, 4], initial_uploads=[ProgramIO(0, _a(1, -2, 3, 4, 5, -6))]), 1, 4))
    # repeat (:158-173)
    cases.append(("repeat", DeviceProgram(
        ops=[DeviceOp.repeat(1, 0, 6, (2, 1, 1, 1), (2, 3, 1, 1), (1, 2, 2, 2), (1, 2, 6, 6))],
        buffer_sizes=[2, 6], initial_uploads=[ProgramIO(0, _a(7, 8))]), 1, 6))
    # slice_assign with strides (:175-199)
    cases.append(("slice_assign", DeviceProgram(
        ops=[DeviceOp.slice_assign(1, 0, 2, 2, dst_base_offset=0, dst_offset=2, dst_row_stride=1, dst_col_stride=2,
                                   src_offset=1, src_row_stride=1, src_col_stride=2, patch_stride=2)],
        buffer_sizes=[6, 8],
        initial_uploads=[ProgramIO(0, _a(99, 2, 3, 5, 6, 77)), ProgramIO(1, _a(10, 11, 12, 13, 14, 15, 16, 17))]),
        1, 8))
    # softmax (:201-213)
    cases.append(("softmax", DeviceProgram(
        ops=[DeviceOp.softmax(1, 0, 2, 3)], buffer_sizes=[6, 6],
        initial_uploads=[ProgramIO(0, _a(1, 2, 3, -1, 0, 1))]), 1, 6))
    # layernorm + rmsnorm (:215-238)
    cases.append(("norms", DeviceProgram(
        ops=[DeviceOp.layernorm(1, 0, 2, 4, eps=1e-5), DeviceOp.rmsnorm(1, 0, 2, 4, eps=1e-5, dst_offset=8)],
        buffer_sizes=[8, 16], initial_uploads=[ProgramIO(0, _a(1, 2, 3, 4, -1, 0, 1, 2))]), 1, 16))
    # rope (:240-263)
    cases.append(("rope", DeviceProgram(
        ops=[DeviceOp.rope(2, 0, 1, half_d=2, seq_len=2, src_off=0, cs_off=0, dst_off=0, src_rs=1, src_cs=4, cs_cs=4)],
        buffer_sizes=[8, 8, 8],
        initial_uploads=[ProgramIO(0, _a(1, 2, 3, 4, 5, 6, 7, 8)), ProgramIO(1, _a(1, 1, 0, 0, 0, 0, 1, 1))]), 2, 8))
    # masked attention seq_q=2 seq_kv=3 with one -inf (:265-320)
    q = _a(0.2, 0.1, -0.3, 0.4, -0.1, 0.5, 0.2, -0.4)
    k = _a(0.1, 0.2, 0.3, 0.4, -0.2, 0.3, 0.1, -0.1, 0.5, -0.4, 0.2, 0.1)
    v = _a(1, 2, 3, 4, -1, 0.5, 2, -0.5, 0.25, -0.75, 1.5, 2.5)
    mask = _a(0, 0, -np.inf, 0, -0.25, 0)
    cases.append(("attention", DeviceProgram(
        ops=[DeviceOp.attention(4, 0, 1, 2, 3, True, d_head=4, seq_q=2, seq_kv=3, scale=0.5, q_off=0, k_off=0,
                                v_off=0, mask_off=0, dst_off=0, q_rs=1, q_cs=4, k_rs=1, k_cs=4, v_rs=1, v_cs=4,
                                mask_rs=1, mask_cs=3, dst_rs=1, dst_cs=4)],
        buffer_sizes=[8, 12, 12, 6, 8],
        initial_uploads=[ProgramIO(0, q), ProgramIO(1, k), ProgramIO(2, v), ProgramIO(3, mask)]), 4, 8))
    # fused relu -> sqrt -> add (:322-345)
    cases.append(("fused_elementwise", DeviceProgram(
        ops=[DeviceOp.fused_elementwise([FusedEwStep("relu"), FusedEwStep("sqrt"), FusedEwStep("add", False, 1, 0)],
                                        n=4, dst=2, src=0)],
        buffer_sizes=[4, 4, 4],
        initial_uploads=[ProgramIO(0, _a(1, -2, 4, 9)), ProgramIO(1, _a(10, 20, 30, 40))]), 2, 4))
    return cases
