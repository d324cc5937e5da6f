"""Backend conformance on the MI355X: the reference's core cases (src/backend/conformance.zig)
run through the HIP C ABI and compared with the oracle, tolerance 1e-5 as for cpu/metal there."""
import json
from pathlib import Path

import numpy as np
import pytest

from zgml_amd import DeviceOp, DeviceProgram, FusedEwStep, MatMulGeometry, ProgramIO, DenseMatMulSpecF32, tryDenseMatMul
from tests.conformance_cases import core_cases
from tests.test_oracle_kat import run_slice_assign_refresh

pytestmark = pytest.mark.gpu
KAT = json.loads((Path(__file__).parent / "golden" / "kat.json").read_text())
f32 = np.float32


@pytest.mark.parametrize("graph", [1, 0])
def test_core_cases_match_reference(hip_backend, oracle, graph):
    from zgml_amd import capi
    hip_backend.set_option(capi.OPT_GRAPH, graph)
    try:
        for name, prog, idx, n in core_cases():
            assert hip_backend.supportsProgram(prog), name
            want = oracle.run_program(prog, idx, n)
            got = oracle.run_program(prog, idx, n, backend=hip_backend)
            np.testing.assert_allclose(got, want, atol=1e-5, rtol=0, err_msg=name)
    finally:
        hip_backend.set_option(capi.OPT_GRAPH, 1)


def test_exact_kats(hip_backend, oracle):
    out = {name: oracle.run_program(p, idx, n, backend=hip_backend) for name, p, idx, n in core_cases()}
    assert out["matmul"].tolist() == KAT["dense_matmul_2x3_3x2"]["expected"]
    np.testing.assert_allclose(out["qmatmul_offsets"], KAT["qmatmul_offset_stride"]["expected"], atol=1e-6, rtol=0)
    assert out["add"].tolist() == KAT["elementwise_add"]["expected"]
    assert out["rope"].tolist() == KAT["rope_half2_seq2"]["expected"]
    assert out["slice_assign"].tolist() == KAT["slice_assign_strided"]["expected"]
    assert out["repeat"].tolist() == [7, 8, 7, 8, 7, 8]
    assert out["reduce"].tolist() == [2, 3, 3, 5]


def test_slice_assign_position_refresh(hip_backend):
    k = KAT["slice_assign_position_refresh"]
    assert run_slice_assign_refresh(hip_backend, k) == k["expected_after"]


def test_softmax_all_neg_inf_and_rmsnorm_zero(hip_backend, oracle):
    p = DeviceProgram(ops=[DeviceOp.softmax(1, 0, 1, 4)], buffer_sizes=[4, 4],
                      initial_uploads=[ProgramIO(0, np.full(4, -np.inf, f32)), ProgramIO(1, np.full(4, 5, f32))])
    assert oracle.run_program(p, 1, 4, backend=hip_backend).tolist() == [0, 0, 0, 0]
    p = DeviceProgram(ops=[DeviceOp.rmsnorm(1, 0, 1, 4)], buffer_sizes=[4, 4], initial_uploads=[ProgramIO(1, np.full(4, 5, f32))])
    assert oracle.run_program(p, 1, 4, backend=hip_backend).tolist() == [0, 0, 0, 0]


def test_attention_seq_kv_refresh_and_masked_row(hip_backend, oracle):
    rng = np.random.default_rng(0xD3FA)
    dh, sq, skv = 64, 2, 700  # > 2 key tiles of 256
    q, k, v = (rng.standard_normal(n).astype(f32) for n in (dh * sq, dh * skv, dh * skv))
    mask = np.zeros(skv * sq, f32)
    mask[5] = -np.inf
    mask[skv:] = -np.inf  # query 1 fully masked -> zeros
    op = DeviceOp.attention(4, 0, 1, 2, 3, True, dh, sq, skv, 0.125, 0, 0, 0, 0, 0, 1, dh, 1, dh, 1, dh, 1, skv, 1, dh)
    prog = DeviceProgram(ops=[op], buffer_sizes=[dh * sq, dh * skv, dh * skv, skv * sq, dh * sq],
                         initial_uploads=[ProgramIO(0, q), ProgramIO(1, k), ProgramIO(2, v), ProgramIO(3, mask)])
    outs = {}
    for name, be in (("ref", oracle.OracleBackend()), ("hip", hip_backend)):
        h = be.compileProgram(prog)
        res = []
        for n in (skv, 300, 2, 513):
            ops = [op.with_(seq_kv=n)]
            be.refreshProgram(h, ops)
            o = np.zeros(dh * sq, f32)
            be.executeProgram(h, [], [ProgramIO(4, o)])
            res.append(o)
        be.freeProgram(h)
        outs[name] = res
    for r, g in zip(outs["ref"], outs["hip"]):
        np.testing.assert_allclose(g, r, atol=2e-5, rtol=0)
        assert np.all(g[dh:] == 0)


def test_elementwise_all_ops_and_fused_chain(hip_backend, oracle):
    rng = np.random.default_rng(1)
    n = 1000
    x = (rng.standard_normal(n) * 2).astype(f32)
    y = rng.standard_normal(n).astype(f32)
    for op in ("add", "mul", "neg", "abs", "sgn", "step", "relu", "sqrt", "recip", "exp", "log", "gelu"):
        src = np.abs(x) + 0.1 if op in ("sqrt", "log") else x
        prog = DeviceProgram(ops=[DeviceOp.elementwise(op, 2, 0, 1, n - 7, dst_offset=3, src0_offset=5, src1_offset=2)],
                             buffer_sizes=[n, n, n], initial_uploads=[ProgramIO(0, src), ProgramIO(1, y)])
        want = oracle.run_program(prog, 2, n)
        got = oracle.run_program(prog, 2, n, backend=hip_backend)
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6, err_msg=op)
    # SiLU chain as the LLaMA lowering emits it: [neg,exp] ; [add(1), recip, mul(x, swapped)]
    one = np.ones(n, f32)
    prog = DeviceProgram(
        ops=[DeviceOp.fused_elementwise([FusedEwStep("neg"), FusedEwStep("exp")], n, 2, 0),
             DeviceOp.fused_elementwise([FusedEwStep("add", False, 1, 0), FusedEwStep("recip"),
                                         FusedEwStep("mul", True, 0, 0)], n, 3, 2)],
        buffer_sizes=[n, n, n, n], initial_uploads=[ProgramIO(0, x), ProgramIO(1, one)])
    want = oracle.run_program(prog, 3, n)
    got = oracle.run_program(prog, 3, n, backend=hip_backend)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(got, x / (1 + np.exp(-x.astype(np.float64))), rtol=1e-5, atol=1e-6)


def test_rowwise_large(hip_backend, oracle):
    rng = np.random.default_rng(2)
    rows, cols = 3, 4096 + 5
    x = rng.standard_normal(rows * cols).astype(f32)
    for mk in (lambda: DeviceOp.softmax(1, 0, rows, cols), lambda: DeviceOp.layernorm(1, 0, rows, cols, 1e-5),
               lambda: DeviceOp.rmsnorm(1, 0, rows, cols, 1e-6)):
        prog = DeviceProgram(ops=[mk()], buffer_sizes=[rows * cols, rows * cols], initial_uploads=[ProgramIO(0, x)])
        want = oracle.run_program(prog, 1, rows * cols)
        got = oracle.run_program(prog, 1, rows * cols, backend=hip_backend)
        np.testing.assert_allclose(got, want, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("inplace", ["add", "norm", "add+norm", "mul"])
def test_row_chain_in_place_forms_at_few_rows(hip_backend, oracle, inplace):
    """at few rows the [add ->] rmsnorm [-> mul] launch puts several workgroups on one row (each re-reads the whole
    row, stores its own column chunks), which is only legal when nothing it stores is something a sibling still reads. The
    reference API has in-place forms (x.addInplace(y) — src/tensor/api.zig:132 — then rmsNorm; a norm or a gain written over its
    input): the launcher must fall back to one workgroup per row for them. rows = 32, cols = 4096 (the shape that splits)."""
    rng = np.random.default_rng(11)
    rows, cols = 32, 4096
    n = rows * cols
    x, y, g = (rng.standard_normal(n).astype(f32) for _ in range(3))
    X, Y, G, H, NRM, OUT = 0, 1, 2, 3, 4, 5
    h = X if "add" in inplace else H               # x.addInplace(y): the sum lands in x
    nrm = h if "norm" in inplace else NRM          # the norm written over its input
    out = nrm if inplace == "mul" else OUT         # the gain written over the normalised rows
    ops = [DeviceOp.elementwise("add", h, X, Y, n), DeviceOp.rmsnorm(nrm, h, rows, cols, 1e-5), DeviceOp.elementwise("mul", out, nrm, G, n)]
    prog = DeviceProgram(ops=ops, buffer_sizes=[n] * 6, initial_uploads=[ProgramIO(X, x), ProgramIO(Y, y), ProgramIO(G, g)])
    for b in sorted({h, nrm, out}):
        want = oracle.run_program(prog, b, n)
        got = oracle.run_program(prog, b, n, backend=hip_backend)
        assert not hip_backend.last_error(), hip_backend.last_error()
        np.testing.assert_allclose(got, want, atol=1e-5, rtol=1e-5, err_msg=f"{inplace}: buffer {b}")


def test_repeat_generic_and_rope_strided_bit_exact(hip_backend, oracle):
    rng = np.random.default_rng(3)
    # generic repeat: src [3,2] (strided view, row stride 4) -> dst [6,4]
    src = rng.standard_normal(16).astype(f32)
    op = DeviceOp.repeat(1, 0, 24, (3, 2, 1, 1), (6, 4, 1, 1), (1, 4, 8, 8), (1, 6, 24, 24), src_offset=1, dst_offset=2)
    prog = DeviceProgram(ops=[op], buffer_sizes=[16, 30], initial_uploads=[ProgramIO(0, src)])
    assert np.array_equal(oracle.run_program(prog, 1, 30, backend=hip_backend), oracle.run_program(prog, 1, 30))
    # rope on a strided row-slice of a projection (src_rs=1, src_cs=kv_dim), as LlamaBlock emits
    hd, seq, kv_dim = 32, 3, 192
    x = rng.standard_normal(kv_dim * seq).astype(f32)
    cs = rng.standard_normal(4 * hd * seq).astype(f32)
    op = DeviceOp.rope(2, 0, 1, hd, seq, src_off=64, cs_off=0, dst_off=0, src_rs=1, src_cs=kv_dim, cs_cs=4 * hd)
    prog = DeviceProgram(ops=[op], buffer_sizes=[kv_dim * seq, 4 * hd * seq, 2 * hd * seq],
                         initial_uploads=[ProgramIO(0, x), ProgramIO(1, cs)])
    np.testing.assert_allclose(oracle.run_program(prog, 2, 2 * hd * seq, backend=hip_backend),
                               oracle.run_program(prog, 2, 2 * hd * seq), atol=0, rtol=0)


def test_dense_matmul_layouts(hip_backend, oracle):
    rng = np.random.default_rng(4)
    M, K, N = 2, 576, 1000
    a = rng.standard_normal(M * K).astype(f32)
    b = rng.standard_normal(K * N).astype(f32)
    bound = (np.abs(a.reshape(M, K)) @ np.abs(b.reshape(K, N))).ravel()
    # B N-contiguous ([K,N] row-major) and B K-contiguous (tied LM head: trans1)
    for geom in (MatMulGeometry(M, N, K, K, 1, N, 1, 0, 0, 0, N), MatMulGeometry(M, N, K, K, 1, 1, K, 0, 0, 0, N)):
        prog = DeviceProgram(ops=[DeviceOp.matmul(2, 0, 1, geom)], buffer_sizes=[M * K, K * N, M * N],
                             initial_uploads=[ProgramIO(0, a), ProgramIO(1, b)])
        want = oracle.run_program(prog, 2, M * N)
        got = oracle.run_program(prog, 2, M * N, backend=hip_backend)
        if geom.b_col_stride != 1:
            bound = (np.abs(a.reshape(M, K)) @ np.abs(b.reshape(N, K).T)).ravel()
        assert np.all(np.abs(got - want) <= 2e-6 * bound + 1e-6)


def test_host_dense_matmul_override(hip_backend):
    k = KAT["dense_matmul_2x3_3x2"]
    dst = np.zeros(4, f32)
    ok = tryDenseMatMul(hip_backend, DenseMatMulSpecF32(dst, np.array(k["a"], f32), np.array(k["b"], f32),
                                                        MatMulGeometry(**k["geom"])))
    assert ok and dst.tolist() == k["expected"]
    assert tryDenseMatMul(None, DenseMatMulSpecF32(dst, dst, dst, MatMulGeometry(**k["geom"]))) is False
    # strided dst keeps the gaps
    dst = np.full(7, -1, f32)
    g = MatMulGeometry(2, 2, 3, 3, 1, 2, 1, 0, 0, 1, 3)
    assert tryDenseMatMul(hip_backend, DenseMatMulSpecF32(dst, np.array(k["a"], f32), np.array(k["b"], f32), g))
    assert dst.tolist() == [-1, 58, 64, -1, 139, 154, -1]


def test_host_dense_override_weight_cache(hip_backend):
    """opt-in device cache of the override's B operands keyed by host pointer; the
    second call with the same B skips its upload, invalidation makes a mutated B visible."""
    import ctypes as C
    from zgml_amd import capi
    rng = np.random.default_rng(11)
    M, K, N = 3, 64, 48
    a, b = rng.standard_normal(M * K).astype(f32), rng.standard_normal(K * N).astype(f32)
    g = MatMulGeometry(M, N, K, K, 1, N, 1, 0, 0, 0, N)

    def stats():
        h, m, by = C.c_uint64(), C.c_uint64(), C.c_uint64()
        hip_backend._lib.zgml_hip_dense_cache_stats(hip_backend.ctx, C.byref(h), C.byref(m), C.byref(by))
        return h.value, m.value, by.value

    def run():
        dst = np.zeros(M * N, f32)
        assert tryDenseMatMul(hip_backend, DenseMatMulSpecF32(dst, a, b, g))
        return dst
    want = (a.reshape(M, K).astype(np.float64) @ b.reshape(K, N).astype(np.float64)).ravel()
    hip_backend.set_option(capi.OPT_DENSE_WEIGHT_CACHE, 1 << 20)
    try:
        h0, m0, _ = stats()
        y1, y2 = run(), run()
        h1, m1, by = stats()
        assert (h1 - h0, m1 - m0) == (1, 1) and by == b.nbytes
        assert np.array_equal(y1, y2) and np.allclose(y1, want, rtol=1e-4, atol=1e-4)
        b *= 2.0  # mutated in place: the cache still holds the old copy ...
        assert np.array_equal(run(), y1)
        hip_backend._lib.zgml_hip_dense_cache_invalidate(hip_backend.ctx, b.ctypes.data)  # ... until invalidated
        assert np.allclose(run(), 2 * want, rtol=1e-4, atol=1e-4)
    finally:
        hip_backend.set_option(capi.OPT_DENSE_WEIGHT_CACHE, 0)
    assert stats()[2] == 0
    assert np.allclose(run(), 2 * want, rtol=1e-4, atol=1e-4)  # cache off: plain path


def test_profile_counters_and_no_fallback(hip_backend, oracle):
    name, prog, idx, n = core_cases()[3]
    h = hip_backend.compileProgram(prog)
    out = np.zeros(n, f32)
    hip_backend.executeProgram(h, [], [ProgramIO(idx, out)])
    hip_backend.executeProgram(h, [], [ProgramIO(idx, out)])
    prof = hip_backend.getRuntimeProfile(h)
    assert prof.call_count == 2 and prof.fallback_op_count == 0 and prof.backend_op_count == 2 * len(prog.ops)
    assert prof.sync_count == 2
    hip_backend.freeProgram(h)


def test_dead_buffer_elision_and_loud_io_error(hip_backend):
    # buffer 2 is a dead "f32 master copy": no op references it
    dead = np.arange(8, dtype=f32)
    prog = DeviceProgram(ops=[DeviceOp.elementwise("neg", 1, 0, 0, 4)], buffer_sizes=[4, 4, 8],
                         initial_uploads=[ProgramIO(0, np.arange(4, dtype=f32)), ProgramIO(2, dead)])
    h = hip_backend.compileProgram(prog)
    out = np.zeros(4, f32)
    hip_backend.executeProgram(h, [], [ProgramIO(1, out)])
    assert out.tolist() == [0, -1, -2, -3]
    with pytest.raises(RuntimeError, match="elided"):
        hip_backend.executeProgram(h, [], [ProgramIO(2, np.zeros(8, f32))])
    hip_backend._lib.zgml_hip_clear_error(hip_backend.ctx)
    hip_backend.freeProgram(h)


def test_constant_repeat_runs_once_until_the_host_writes_its_source(hip_backend):
    """A repeat whose source no op writes is executed when the plan is built, not per execution (the γ broadcasts of the
    LLaMA programs); an execute input on that source puts it back in the plan, with the new values visible."""
    import os
    from zgml_amd import capi
    if os.environ.get("ZGML_HIP_HOIST_REPEAT") == "0":
        pytest.skip("the diagnostic switch under test turns the hoisting off")
    prog = DeviceProgram(
        ops=[DeviceOp.repeat(1, 0, 6, (2, 1, 1, 1), (2, 3, 1, 1), (1, 2, 2, 2), (1, 2, 6, 6)), DeviceOp.elementwise("neg", 2, 1, 1, 6)],
        buffer_sizes=[2, 6, 6], initial_uploads=[ProgramIO(0, np.array([7, 8], f32))])
    h = hip_backend.compileProgram(prog)
    prof = lambda: capi.load_hip().zgml_hip_get_runtime_profile(hip_backend.ctx, h).contents.backend_dispatch_count
    out = np.zeros(6, f32)
    for _ in range(2):
        hip_backend.executeProgram(h, [], [ProgramIO(2, out)])
        assert out.tolist() == [-7, -8, -7, -8, -7, -8]
    assert prof() == 2  # one launch per execution: the neg
    hip_backend.executeProgram(h, [ProgramIO(0, np.array([1, 2], f32))], [ProgramIO(2, out)])
    assert out.tolist() == [-1, -2, -1, -2, -1, -2]
    hip_backend.executeProgram(h, [ProgramIO(0, np.array([3, 4], f32))], [ProgramIO(2, out)])
    assert out.tolist() == [-3, -4, -3, -4, -3, -4]
    assert prof() == 6  # repeat + neg from then on
    hip_backend.freeProgram(h)


def test_argmax_first_max_wins(hip_backend, oracle):
    rng = np.random.default_rng(5)
    v = rng.standard_normal(49152).astype(f32)
    v[[100, 40000]] = 9.0  # tie: first index wins
    prog = DeviceProgram(ops=[DeviceOp.elementwise("abs", 1, 0, 0, 1)], buffer_sizes=[v.size, 1],
                         initial_uploads=[ProgramIO(0, v)])
    h = hip_backend.compileProgram(prog)
    assert hip_backend.argmax(h, 0, 0, v.size) == oracle.argmax(v) == 100
    assert hip_backend.argmax(h, 0, 101, v.size - 101) == 40000 - 101
    hip_backend.freeProgram(h)


def test_refresh_beyond_assumed_bounds_falls_back_correctly(hip_backend, oracle):
    """The level-batched plan assumes seq_kv never exceeds its compile-time value and that a
    dynamic store stays inside its slab; a refresh that breaks the assumption must still give the
    reference's result (the runtime drops to program order)."""
    rng = np.random.default_rng(11)
    dh, skv = 32, 300
    q, k, v = (rng.standard_normal(n).astype(f32) for n in (dh, dh * skv, dh * skv))
    src = rng.standard_normal(dh).astype(f32)
    # store a new key column at a dynamic position, then attend over seq_kv keys
    sa = DeviceOp.slice_assign(1, 5, dh, 1, dst_base_offset=0, dst_offset=0, dst_row_stride=1, dst_col_stride=dh,
                               src_offset=0, src_row_stride=1, src_col_stride=dh, patch_stride=dh)
    att = DeviceOp.attention(4, 0, 1, 2, 3, False, dh, 1, 2, 0.2, 0, 0, 0, 0, 0, 1, dh, 1, dh, 1, dh, 0, 0, 1, dh)
    prog = DeviceProgram(ops=[sa, att], buffer_sizes=[dh, dh * skv, dh * skv, 1, dh, dh],
                         initial_uploads=[ProgramIO(0, q), ProgramIO(1, k), ProgramIO(2, v), ProgramIO(5, src)])
    res = {}
    for name, be in (("ref", oracle.OracleBackend()), ("hip", hip_backend)):
        h = be.compileProgram(prog)
        outs = []
        for pos, n in ((1, 2), (250, 300), (7, 100)):  # seq_kv 300 > compile-time 2
            be.refreshProgram(h, [sa.with_(dst_offset=pos * dh), att.with_(seq_kv=n)])
            o = np.zeros(dh, f32)
            be.executeProgram(h, [], [ProgramIO(4, o)])
            outs.append(o)
        be.freeProgram(h)
        res[name] = outs
    for r, g in zip(res["ref"], res["hip"]):
        np.testing.assert_allclose(g, r, atol=2e-5, rtol=0)


@pytest.mark.parametrize("dh,sq,n_heads,skv,has_mask", [(128, 64, 8, 300, False), (64, 33, 3, 90, True), (32, 40, 8, 1030, True),
                                                         (128, 20, 2, 37, True), (64, 128, 4, 128, True), (128, 16, 1, 5, True),
                                                         (128, 128, 32, 256, True), (128, 32, 2, 2040, True), (64, 17, 2, 17, False)])
def test_attention_many_queries_batched_heads(hip_backend, oracle, dh, sq, n_heads, skv, has_mask):
    """seq_q > 1 over several heads in one dependency level (the prefill shape). d_head 64 / 128 with >= 16 queries:
    attention_tiles_kernel (16-query tiles on the f32 matrix cores; ragged query and key tiles, causal tiles skipped,
    4- and 8-wave workgroups, key tiles split over waves and merged); otherwise the streaming kernel. Causal-style masks,
    multi-step contexts."""
    rng = np.random.default_rng(dh * 1000 + sq + skv)
    bufs, ups, ops = [], [], []
    mask = np.zeros(skv * sq, f32)
    for qi in range(sq):  # query qi sees keys [0, skv - sq + qi]
        mask[qi * skv + max(0, skv - sq + qi + 1):(qi + 1) * skv] = -np.inf
    for h in range(n_heads):
        q, k, v = (rng.standard_normal(n).astype(f32) for n in (dh * sq, dh * skv, dh * skv))
        base = len(bufs)
        bufs += [dh * sq, dh * skv, dh * skv, skv * sq, dh * sq]
        ups += [ProgramIO(base + 0, q), ProgramIO(base + 1, k), ProgramIO(base + 2, v), ProgramIO(base + 3, mask)]
        ops.append(DeviceOp.attention(base + 4, base + 0, base + 1, base + 2, base + 3, has_mask, dh, sq, skv, dh ** -0.5, 0, 0, 0, 0, 0,
                                      1, dh, 1, dh, 1, dh, 1, skv, 1, dh))
    prog = DeviceProgram(ops=ops, buffer_sizes=bufs, initial_uploads=ups)
    for h in (0, n_heads - 1):
        want = oracle.run_program(prog, 5 * h + 4, dh * sq)
        got = oracle.run_program(prog, 5 * h + 4, dh * sq, backend=hip_backend)
        assert not hip_backend.last_error(), hip_backend.last_error()
        np.testing.assert_allclose(got, want, atol=2e-5, rtol=0)
