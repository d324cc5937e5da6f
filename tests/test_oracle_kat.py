"""Pins the CPU oracle to the reference's own known-answer tests.
The expected numbers live in tests/golden/kat.json with their reference citations."""
import json
from pathlib import Path

import numpy as np
import pytest

from zgml_amd import (Capabilities, DeviceOp, DeviceProgram, FusedEwStep, MatMulGeometry, ProgramIO,
                      QuantizedWeightUpload)
from tests.conformance_cases import core_cases

KAT = json.loads((Path(__file__).parent / "golden" / "kat.json").read_text())
f32 = np.float32


def _a(v):
    return np.array(v, dtype=f32)


def test_dense_matmul_kat(oracle):
    k = KAT["dense_matmul_2x3_3x2"]
    p = DeviceProgram(ops=[DeviceOp.matmul(2, 0, 1, MatMulGeometry(**k["geom"]))], buffer_sizes=[6, 6, 4],
                      initial_uploads=[ProgramIO(0, _a(k["a"])), ProgramIO(1, _a(k["b"]))])
    assert oracle.run_program(p, 2, 4).tolist() == k["expected"]


def test_qmatmul_kat(oracle):
    k = KAT["qmatmul_row_major"]
    qw = QuantizedWeightUpload(np.array(k["data"], np.int8), _a(k["scales"]), k["K"], k["N"], k["block_size"])
    p = DeviceProgram(ops=[DeviceOp.qmatmul(1, 0, 0, k["M"], k["N"], k["K"])], buffer_sizes=[6, 6],
                      initial_uploads=[ProgramIO(0, _a(k["input"]))], qweights=[qw])
    np.testing.assert_allclose(oracle.run_program(p, 1, 6), k["expected"], atol=k["tol"], rtol=0)
    k = KAT["qmatmul_offset_stride"]
    p = DeviceProgram(
        ops=[DeviceOp.qmatmul(1, 0, 0, k["M"], k["N"], k["K"], k["input_offset"], k["input_row_stride"],
                              k["dst_offset"], k["dst_row_stride"])],
        buffer_sizes=[9, 9], initial_uploads=[ProgramIO(0, _a(k["input"])), ProgramIO(1, _a(k["dst_init"]))],
        qweights=[qw])
    np.testing.assert_allclose(oracle.run_program(p, 1, 9), k["expected"], atol=k["tol"], rtol=0)


def test_elementwise_add_kat(oracle):
    k = KAT["elementwise_add"]
    p = DeviceProgram(ops=[DeviceOp.elementwise("add", 2, 0, 1, 4)], buffer_sizes=[4, 4, 4],
                      initial_uploads=[ProgramIO(0, _a(k["a"])), ProgramIO(1, _a(k["b"]))])
    assert oracle.run_program(p, 2, 4).tolist() == k["expected"]


def test_rope_kat(oracle):
    k = KAT["rope_half2_seq2"]
    p = DeviceProgram(
        ops=[DeviceOp.rope(2, 0, 1, k["half_d"], k["seq_len"], 0, 0, 0, k["src_rs"], k["src_cs"], k["cs_cs"])],
        buffer_sizes=[8, 8, 8], initial_uploads=[ProgramIO(0, _a(k["src"])), ProgramIO(1, _a(k["cos_sin"]))])
    assert oracle.run_program(p, 2, 8).tolist() == k["expected"]


def slice_assign_refresh_program(k):
    rows, dst_cols = k["rows"], k["dst_cols"]
    op = DeviceOp.slice_assign(1, 0, rows, k["cols"], dst_base_offset=0, dst_offset=0, dst_row_stride=1,
                               dst_col_stride=rows, src_offset=0, src_row_stride=1, src_col_stride=rows,
                               patch_stride=rows)
    return DeviceProgram(ops=[op], buffer_sizes=[rows, rows * dst_cols], initial_uploads=[ProgramIO(0, _a(k["src"]))])


def run_slice_assign_refresh(be, k):
    """DeviceInference.patchSliceAssignOffset + execute (src/device_inference.zig:242-263)."""
    p = slice_assign_refresh_program(k)
    h = be.compileProgram(p)
    outs = []
    try:
        for pos in k["positions"]:
            sa = p.ops[0]
            p.ops[0] = sa.with_(dst_offset=sa.dst_base_offset + pos * sa.patch_stride)
            be.refreshProgram(h, p.ops)
            out = np.zeros(k["rows"] * k["dst_cols"], f32)
            be.executeProgram(h, [], [ProgramIO(1, out)])
            outs.append(out.tolist())
    finally:
        be.freeProgram(h)
    return outs


def test_slice_assign_refresh_kat(oracle):
    k = KAT["slice_assign_position_refresh"]
    assert run_slice_assign_refresh(oracle.OracleBackend(), k) == k["expected_after"]


def test_slice_assign_strided_kat(oracle):
    k = KAT["slice_assign_strided"]
    name, p, idx, n = [c for c in core_cases() if c[0] == "slice_assign"][0]
    assert oracle.run_program(p, idx, n).tolist() == k["expected"]


def test_softmax_all_neg_inf_and_rmsnorm_zero(oracle):
    k = KAT["softmax_all_neg_inf_is_zero"]
    p = DeviceProgram(ops=[DeviceOp.softmax(1, 0, 1, k["cols"])], buffer_sizes=[4, 4],
                      initial_uploads=[ProgramIO(0, np.full(4, -np.inf, f32)), ProgramIO(1, np.full(4, 5, f32))])
    assert oracle.run_program(p, 1, 4).tolist() == k["expected"]
    k = KAT["rmsnorm_zero_input_is_zero"]
    p = DeviceProgram(ops=[DeviceOp.rmsnorm(1, 0, 1, k["cols"], eps=k["eps"])], buffer_sizes=[4, 4],
                      initial_uploads=[ProgramIO(1, np.full(4, 5, f32))])
    assert oracle.run_program(p, 1, 4).tolist() == k["expected"]


def test_gguf_block_decode_kats(oracle):
    k = KAT["gguf_q4_0_block"]
    raw = np.zeros(18, np.uint8)
    raw[0], raw[1] = k["scale_f16_bits"] & 0xFF, k["scale_f16_bits"] >> 8
    raw[2] = k["byte0"]
    deq = oracle.gguf_dequant(raw, k["n_elems"], "q4_0")
    assert deq[:3].tolist() == k["expected_dequant_first3"]
    assert np.all(deq[3:] == k["expected_rest"])
    raw[0], raw[1] = k["import_scale_f16_bits"] & 0xFF, k["import_scale_f16_bits"] >> 8
    data, scales = oracle.gguf_to_int8(raw, 32, "q4_0")
    assert data[:3].tolist() == k["expected_int8"] and scales[0] == k["expected_import_scale"]
    k = KAT["gguf_q8_0_block"]
    raw = np.zeros(34, np.uint8)
    raw[0], raw[1] = k["scale_f16_bits"] & 0xFF, k["scale_f16_bits"] >> 8
    raw[2:5] = np.array(k["q"], np.int8).view(np.uint8)
    assert oracle.gguf_dequant(raw, 32, "q8_0")[:3].tolist() == k["expected_first3"]
    data, scales = oracle.gguf_to_int8(raw, 32, "q8_0")
    assert data[:3].tolist() == k["q"] and scales[0] == 0.5


def test_f16_conversion_exhaustive(oracle):
    lib = oracle.load()
    bits = np.arange(65536, dtype=np.uint16)
    ref = bits.view(np.float16).astype(np.float32)
    got = np.array([lib.zo_f16_to_f32(int(b)) for b in bits[::37]], dtype=np.float32)
    exp = ref[::37]
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    assert np.array_equal(got[~np.isnan(got)], exp[~np.isnan(exp)])
    vals = np.random.default_rng(0xD3FA).standard_normal(2000).astype(np.float32) * 3
    back = np.array([lib.zo_f32_to_f16(float(v)) for v in vals], dtype=np.uint16)
    assert np.array_equal(back, vals.astype(np.float16).view(np.uint16))


def test_capability_gating_kat():
    k = KAT["capability_gating"]
    one = DeviceProgram(ops=[DeviceOp.fused_elementwise([FusedEwStep("relu")], 1, 1, 0)], buffer_sizes=[1, 1])
    nine = DeviceProgram(ops=[DeviceOp.fused_elementwise([FusedEwStep("relu")] * 9, 1, 1, 0)], buffer_sizes=[1, 1])
    for name, want in k["fused_1_step"].items():
        assert one.isSupportedBy(getattr(Capabilities, name)) is want
    for name, want in k["fused_9_steps"].items():
        assert nine.isSupportedBy(getattr(Capabilities, name)) is want
    qd, sc = np.array([1, 2, 3, 4], np.int8), np.array([1], f32)
    qops = [DeviceOp.qmatmul(1, 0, 0, 1, 2, 2)]
    good = DeviceProgram(ops=qops, buffer_sizes=[2, 2], qweights=[QuantizedWeightUpload(qd, sc, 2, 2, 4)])
    bad = DeviceProgram(ops=qops, buffer_sizes=[2, 2], qweights=[QuantizedWeightUpload(qd, sc, 3, 2, 4)])
    assert good.isSupportedBy(Capabilities.wgpu) and not bad.isSupportedBy(Capabilities.wgpu)
    for name, seq_kv, d_head, want in k["attention_limits"]:
        assert getattr(Capabilities, name).attention.supports(seq_kv, d_head) is want


def test_oracle_c_program_supported_matches_python(oracle):
    import ctypes as C
    lib = oracle.load()
    for name, p, _, _ in core_cases():
        pc, keep = p.to_c()
        assert lib.zo_program_supported(C.byref(pc), 1, -1, -1, 512) == 1, name
        assert p.isSupportedBy(Capabilities.reference_cpu)


# ── closed-form checks of the conformance cases whose expected values the reference computes at
#    run time (conformance.zig:133-345): we verify the oracle against explicit float64 formulas.

def test_conformance_cases_vs_closed_forms(oracle):
    out = {name: oracle.run_program(p, idx, n) for name, p, idx, n in core_cases()}
    np.testing.assert_array_equal(out["reduce"], [2, 3, 3, 5])
    np.testing.assert_array_equal(out["repeat"], [7, 8, 7, 8, 7, 8])
    x = np.array([[1, 2, 3], [-1, 0, 1]], np.float64)
    sm = np.exp(x - x.max(1, keepdims=True))
    sm /= sm.sum(1, keepdims=True)
    np.testing.assert_allclose(out["softmax"], sm.ravel(), atol=1e-6)
    x = np.array([[1, 2, 3, 4], [-1, 0, 1, 2]], np.float64)
    ln = (x - x.mean(1, keepdims=True)) / np.sqrt(x.var(1, keepdims=True) + 1e-5)
    rms = x / np.sqrt((x * x).mean(1, keepdims=True) + 1e-5)
    np.testing.assert_allclose(out["norms"], np.concatenate([ln.ravel(), rms.ravel()]), atol=1e-6)
    np.testing.assert_allclose(out["fused_elementwise"], [11, 20, 32, 43], atol=1e-6)
    # attention: dense softmax(QK^T*scale+mask)V in float64
    q = np.array([[0.2, 0.1, -0.3, 0.4], [-0.1, 0.5, 0.2, -0.4]])
    k = np.array([[0.1, 0.2, 0.3, 0.4], [-0.2, 0.3, 0.1, -0.1], [0.5, -0.4, 0.2, 0.1]])
    v = np.array([[1, 2, 3, 4], [-1, 0.5, 2, -0.5], [0.25, -0.75, 1.5, 2.5]])
    mask = np.array([[0, 0, -np.inf], [0, -0.25, 0]])
    s = q @ k.T * 0.5 + mask
    p = np.exp(s - s.max(1, keepdims=True))
    p /= p.sum(1, keepdims=True)
    np.testing.assert_allclose(out["attention"], (p @ v).ravel(), atol=1e-6)


def test_attention_fully_masked_row_is_zero(oracle):
    """seq_kv refresh incl. a fully masked query (src/backend/wgpu.zig:1457-1597)."""
    rng = np.random.default_rng(0xD3FA)
    dh, sq, skv = 8, 2, 4
    q, k, v = (rng.standard_normal(n).astype(f32) for n in (dh * sq, dh * skv, dh * skv))
    mask = np.zeros(skv * sq, f32)
    mask[skv:] = -np.inf  # query 1 fully masked
    op = DeviceOp.attention(4, 0, 1, 2, 3, True, dh, sq, skv, 0.3, 0, 0, 0, 0, 0, 1, dh, 1, dh, 1, dh, 1, skv, 1, dh)
    p = DeviceProgram(ops=[op], buffer_sizes=[dh * sq, dh * skv, dh * skv, skv * sq, dh * sq],
                      initial_uploads=[ProgramIO(0, q), ProgramIO(1, k), ProgramIO(2, v), ProgramIO(3, mask)])
    be = oracle.OracleBackend()
    h = be.compileProgram(p)
    out4, out2 = np.zeros(dh * sq, f32), np.zeros(dh * sq, f32)
    be.executeProgram(h, [], [ProgramIO(4, out4)])
    p.ops[0] = op.with_(seq_kv=2)
    be.refreshProgram(h, p.ops)
    be.executeProgram(h, [], [ProgramIO(4, out2)])
    be.freeProgram(h)
    assert np.all(out4[dh:] == 0) and np.all(out2[dh:] == 0)

    def dense(n):
        s = (q[:dh].astype(np.float64) @ k.reshape(skv, dh)[:n].T.astype(np.float64)) * 0.3
        w = np.exp(s - s.max())
        return (w / w.sum()) @ v.reshape(skv, dh)[:n].astype(np.float64)
    np.testing.assert_allclose(out4[:dh], dense(4), atol=1e-5)
    np.testing.assert_allclose(out2[:dh], dense(2), atol=1e-5)


def test_f16_promotion_known_answer_and_rounding():
    """src/backend/wgpu.zig:1276-1318 (auto-promoted f16 matmul, small integers exact) plus the two
    kernels' A handling (src/backend/metal.zig:680-760): M == 1 keeps A in f32, M > 1 rounds it."""
    from zgml_amd import DeviceOp, DeviceProgram, MatMulGeometry, ProgramIO
    from oracle import oracle as O
    f32 = np.float32
    O.set_f16_dense(True)
    try:
        g = MatMulGeometry(M=2, N=2, K=3, a_row_stride=3, a_col_stride=1, b_row_stride=2, b_col_stride=1,
                           a_offset=0, b_offset=0, dst_offset=0, dst_row_stride=2)
        prog = DeviceProgram(ops=[DeviceOp.matmul(2, 0, 1, g)], buffer_sizes=[6, 6, 4],
                             initial_uploads=[ProgramIO(0, np.array([1, 2, 3, 4, 5, 6], f32)),
                                              ProgramIO(1, np.array([7, 8, 9, 10, 11, 12], f32))])
        assert np.array_equal(O.run_program(prog, 2, 4), np.array([58, 64, 139, 154], f32))
        x = f32(1 + 2.0 ** -12)  # not representable in f16
        for M, want in ((1, x), (2, f32(1.0))):
            g = MatMulGeometry(M=M, N=1, K=1, a_row_stride=1, a_col_stride=1, b_row_stride=1, b_col_stride=1,
                               a_offset=0, b_offset=0, dst_offset=0, dst_row_stride=1)
            prog = DeviceProgram(ops=[DeviceOp.matmul(2, 0, 1, g)], buffer_sizes=[M, 1, M],
                                 initial_uploads=[ProgramIO(0, np.full(M, x, f32)), ProgramIO(1, np.array([1.0], f32))])
            assert O.run_program(prog, 2, M)[0] == want
        # B is rounded: 1 + 2^-12 -> 1.0
        g = MatMulGeometry(M=1, N=1, K=1, a_row_stride=1, a_col_stride=1, b_row_stride=1, b_col_stride=1,
                           a_offset=0, b_offset=0, dst_offset=0, dst_row_stride=1)
        prog = DeviceProgram(ops=[DeviceOp.matmul(2, 0, 1, g)], buffer_sizes=[1, 1, 1],
                             initial_uploads=[ProgramIO(0, np.array([3.0], f32)), ProgramIO(1, np.array([x], f32))])
        assert O.run_program(prog, 2, 1)[0] == f32(3.0)
    finally:
        O.set_f16_dense(False)


def test_oracle_accepts_packed_gguf_upload():
    """The packed-GGUF pass-through form (include/zgml_hip.h) runs through the oracle's CpuBackend
    restatement with the loader's expansion, and equals the expanded upload bit for bit."""
    from zgml_amd import DeviceOp, DeviceProgram, ProgramIO, QuantizedWeightUpload
    from oracle import oracle as O
    from tests.synth import q4_0_blocks_from_int8
    rng = np.random.default_rng(3)
    K, N = 64, 32
    data = rng.integers(-8, 8, K * N).astype(np.int8)
    scales = (rng.random(K * N // 32).astype(np.float16) * 0.1 + 0.01).astype(np.float32)
    raw = q4_0_blocks_from_int8(data, scales)
    x = rng.standard_normal(K).astype(np.float32)

    def program(qw):
        return DeviceProgram(ops=[DeviceOp.qmatmul(1, 0, 0, 1, N, K)], buffer_sizes=[K, N],
                             initial_uploads=[ProgramIO(0, x)], qweights=[qw])
    a = O.run_program(program(QuantizedWeightUpload(data, scales, K, N, 32)), 1, N)
    b = O.run_program(program(QuantizedWeightUpload.from_gguf_blocks(raw, K, N, "q4_0")), 1, N)
    assert np.array_equal(a, b) and np.abs(a).max() > 0


def test_quantised_kv_ops_meet_the_reference_tests_bounds():
    """The reference's own acceptance tests for QuantizedKVCache / attentionQuantized
    (src/quant.zig:1259-1606) restated on the oracle's extension ops, with inputs uniform in [-1, 1) as
    fillRandF32 produces: store -> dequant round trip within 0.02 (:1259-1276), decode attention within
    0.01 of the streaming softmax over the dequantised K/V (:1339-1411), a causal broadcast mask
    (:1413-1485) and a column offset selecting the right slab (:1487-1545), tile + tail path (:1547+)."""
    from zgml_amd import DeviceOp, DeviceProgram, ProgramIO
    from oracle import oracle as O
    f32 = np.float32
    rng = np.random.default_rng(12)
    dh, bs, n_cols, start = 64, 32, 32, 3
    ce = n_cols * dh // 4 + n_cols * (dh // bs)
    for seq_kv, causal_pos in ((8, None), (8, 4), (19, None)):  # 19 = two tiles of 8 + tail of 3
        k = ((rng.random((seq_kv, dh)) - 0.5) * 2).astype(f32)
        v = ((rng.random((seq_kv, dh)) - 0.5) * 2).astype(f32)
        q = ((rng.random(dh) - 0.5) * 2).astype(f32)
        mask = np.zeros(seq_kv, f32)
        if causal_pos is not None:
            mask[causal_pos + 1:] = -np.inf
        ops = []
        for c in range(seq_kv):
            ops.append(DeviceOp.kvq_store(0, 2, dh, bs, n_cols, c * dh, start, start + c, 1))
            ops.append(DeviceOp.kvq_store(1, 3, dh, bs, n_cols, c * dh, start, start + c, 1))
        scale = float(1 / np.sqrt(dh))
        ops.append(DeviceOp.attention_kvq(5, 4, 0, 1, 6, True, dh, 1, seq_kv, scale, bs, n_cols, start, start, 0, dh, 0, dh, 0, 1, 0))
        prog = DeviceProgram(ops=ops, buffer_sizes=[ce, ce, seq_kv * dh, seq_kv * dh, dh, dh, seq_kv],
                             initial_uploads=[ProgramIO(2, k.ravel()), ProgramIO(3, v.ravel()), ProgramIO(4, q), ProgramIO(6, mask)])
        be = O.OracleBackend()
        h = be.compileProgram(prog)
        out, kc, vc = np.zeros(dh, f32), np.zeros(ce, f32), np.zeros(ce, f32)
        be.executeProgram(h, [], [ProgramIO(5, out), ProgramIO(0, kc), ProgramIO(1, vc)])
        be.freeProgram(h)

        def dequant(c):
            qd = c.view(np.int8)[:n_cols * dh].reshape(n_cols, dh).astype(f32)
            return (qd * np.repeat(c[n_cols * dh // 4:].reshape(n_cols, dh // bs), bs, axis=1))[start:start + seq_kv]
        kd, vd = dequant(kc), dequant(vc)
        assert np.abs(kd - k).max() <= 0.02 and np.abs(vd - v).max() <= 0.02
        s = kd.astype(np.float64) @ q.astype(np.float64) * scale + mask
        w = np.exp(s - s.max())
        ref = (w[:, None] * vd).sum(0) / w.sum()
        assert np.abs(out - ref).max() <= 0.01
        assert np.abs(out - ref).max() <= 1e-5  # in fact the same dequantised operands: only the order differs
