//! MI355X (gfx950) backend for zgml: a thin extern-C forward to libzgml_hip.so (include/zgml_hip.h).
//!
//! Drop this file into the zgml tree as `src/backend/hip.zig` (with `abi_asserts.zig` beside it) and apply the four one-line
//! edits + `linkHip` of zig/PATCH.md. It follows the plumbing of the reference's own CPU backend
//! (src/backend/cpu.zig:62-147: a struct holding the state, `backend()` returning {ctx, vtable, name, device, capabilities},
//! static vtable of six functions) and the FFI style of the Metal shim (src/backend/metal_shim.h:1-63, build.zig:44-55).
//!
//! UNVERIFIED SOURCE: no Zig toolchain exists in the image this library is built in (Zig >= 0.16, build.zig.zon:5). The file is
//! deliberately declarative — struct flattening and six forwards — and every record it fills is pinned by the generated
//! `comptime` asserts of abi_asserts.zig (tools/gen_zig_abi_asserts.py measures the header with the C compiler; the same
//! numbers are checked against the Python mirror by tests/test_abi_symbols.py). The compiled twin of this adapter, which the
//! test-suite runs, is zgml_amd/host/hip_backend.cpp.
//!
//! Interface implemented: `Backend.VTable`, src/backend.zig:330-382.
const std = @import("std");
const backend_mod = @import("../backend.zig");
const profile = @import("../profile.zig");
const backend_program = @import("program.zig"); // src/backend/program.zig: stepDynamicStateFromOps
const c = @cImport(@cInclude("zgml_hip.h"));

comptime {
    @import("abi_asserts.zig").check(c);
    // the DeviceOp tags are sent as ordinals: the union's declaration order must be the header's enum (src/backend.zig:179-249)
    const Tag = std.meta.Tag(backend_mod.DeviceOp);
    std.debug.assert(@intFromEnum(Tag.elementwise) == c.ZGML_DOP_ELEMENTWISE);
    std.debug.assert(@intFromEnum(Tag.matmul) == c.ZGML_DOP_MATMUL);
    std.debug.assert(@intFromEnum(Tag.qmatmul) == c.ZGML_DOP_QMATMUL);
    std.debug.assert(@intFromEnum(Tag.softmax) == c.ZGML_DOP_SOFTMAX);
    std.debug.assert(@intFromEnum(Tag.layernorm) == c.ZGML_DOP_LAYERNORM);
    std.debug.assert(@intFromEnum(Tag.rmsnorm) == c.ZGML_DOP_RMSNORM);
    std.debug.assert(@intFromEnum(Tag.reduce) == c.ZGML_DOP_REDUCE);
    std.debug.assert(@intFromEnum(Tag.repeat) == c.ZGML_DOP_REPEAT);
    std.debug.assert(@intFromEnum(Tag.slice_assign) == c.ZGML_DOP_SLICE_ASSIGN);
    std.debug.assert(@intFromEnum(Tag.rope) == c.ZGML_DOP_ROPE);
    std.debug.assert(@intFromEnum(Tag.attention) == c.ZGML_DOP_ATTENTION);
    std.debug.assert(@intFromEnum(Tag.fused_elementwise) == c.ZGML_DOP_FUSED_ELEMENTWISE);
    // ... and the graph `Op` ordinals the kernels switch on (src/op.zig:11-62)
    const Op = @import("../op.zig").Op;
    std.debug.assert(@intFromEnum(Op.add) == c.ZGML_OP_ADD and @intFromEnum(Op.mul) == c.ZGML_OP_MUL);
    std.debug.assert(@intFromEnum(Op.neg) == c.ZGML_OP_NEG and @intFromEnum(Op.gelu) == c.ZGML_OP_GELU);
    std.debug.assert(@intFromEnum(Op.sum) == c.ZGML_OP_SUM and @intFromEnum(Op.max) == c.ZGML_OP_MAX);
}

pub const HipBackend = struct {
    ctx: *c.zgml_hip_ctx,
    profile_storage: profile.RuntimeProfile = .{},
    /// refresh_program re-sends the whole op list (the cpu backend's contract: a changed static field is honoured) instead of the
    /// two dynamic numbers (the wgpu backend's, the default here: see refreshProgram)
    strict_refresh: bool = false,
    /// compiled programs may register the caller's output slice with the driver and write results straight into it
    /// (zgml_hip_program_pin_outputs: the slice must outlive the program, as DeviceInference's does)
    pin_outputs: bool = true,

    /// null device / not a gfx950 / library missing: error.HipNotAvailable; `lastError(null)` says why.
    pub fn init(device_ordinal: c_int) !HipBackend {
        const ctx = c.zgml_hip_create(device_ordinal) orelse return error.HipNotAvailable;
        return .{ .ctx = ctx };
    }

    pub fn deinit(self: *HipBackend) void {
        c.zgml_hip_destroy(self.ctx);
    }

    /// The vtable has no error channel on execute (src/backend.zig:347 returns void): the library keeps a sticky string.
    pub fn lastError(self: ?*const HipBackend) []const u8 {
        const p = c.zgml_hip_last_error(if (self) |s| s.ctx else null);
        return std.mem.span(p);
    }

    pub fn backend(self: *HipBackend) backend_mod.Backend {
        return .{
            .ctx = @ptrCast(self),
            .vtable = &vtable,
            .name_str = "hip",
            .device_type = .hip, // zig/PATCH.md edit 1
            .capabilities = backend_mod.Capabilities.hip, // zig/PATCH.md edit 2
        };
    }
};

// ── flattening: Zig slices / tagged unions -> the flat C records ─────────────────────────────────────────────────────

fn flattenGeom(g: backend_mod.MatMulGeometry) c.zgml_matmul_geom {
    var o: c.zgml_matmul_geom = undefined;
    inline for (std.meta.fields(backend_mod.MatMulGeometry)) |f| @field(o, f.name) = @intCast(@field(g, f.name));
    return o;
}

fn flattenSteps(arena: std.mem.Allocator, steps: []const backend_mod.FusedEwStep) ![*]const c.zgml_fused_step {
    const out = try arena.alloc(c.zgml_fused_step, @max(steps.len, 1));
    for (steps, 0..) |s, i| out[i] = .{
        .op = @intFromEnum(s.op),
        .is_swapped = @intFromBool(s.is_swapped),
        ._pad = 0,
        .secondary_buf = s.secondary_buf,
        .secondary_offset = s.secondary_offset,
    };
    return out.ptr;
}

fn flattenOp(arena: std.mem.Allocator, op: backend_mod.DeviceOp) !c.zgml_device_op {
    var o = std.mem.zeroes(c.zgml_device_op);
    switch (op) {
        .elementwise => |e| {
            o.kind = c.ZGML_DOP_ELEMENTWISE;
            o.u.elementwise = .{ .op = @intFromEnum(e.op), .dst = e.dst, .src0 = e.src0, .src1 = e.src1, ._pad = 0, .n = e.n, .dst_offset = e.dst_offset, .src0_offset = e.src0_offset, .src1_offset = e.src1_offset };
        },
        .matmul => |m| {
            o.kind = c.ZGML_DOP_MATMUL;
            o.u.matmul = .{ .dst = m.dst, .a = m.a, .b = m.b, ._pad = 0, .geom = flattenGeom(m.geom) };
        },
        .qmatmul => |q| {
            o.kind = c.ZGML_DOP_QMATMUL;
            o.u.qmatmul = .{ .dst = q.dst, .input = q.input, .weight_idx = q.weight_idx, ._pad = 0, .M = q.M, .N = q.N, .K = q.K, .input_offset = q.input_offset, .input_row_stride = q.input_row_stride, .dst_offset = q.dst_offset, .dst_row_stride = q.dst_row_stride };
        },
        .softmax => |s| {
            o.kind = c.ZGML_DOP_SOFTMAX;
            o.u.softmax = .{ .dst = s.dst, .src = s.src, .rows = s.rows, .cols = s.cols, .eps = 0, .src_offset = s.src_offset, .dst_offset = s.dst_offset };
        },
        .layernorm => |l| {
            o.kind = c.ZGML_DOP_LAYERNORM;
            o.u.layernorm = .{ .dst = l.dst, .src = l.src, .rows = l.rows, .cols = l.cols, .eps = l.eps, .src_offset = l.src_offset, .dst_offset = l.dst_offset };
        },
        .rmsnorm => |r| {
            o.kind = c.ZGML_DOP_RMSNORM;
            o.u.rmsnorm = .{ .dst = r.dst, .src = r.src, .rows = r.rows, .cols = r.cols, .eps = r.eps, .src_offset = r.src_offset, .dst_offset = r.dst_offset };
        },
        .reduce => |r| {
            o.kind = c.ZGML_DOP_REDUCE;
            o.u.reduce = .{ .op = @intFromEnum(r.op), .dst = r.dst, .src = r.src, .n_out = r.n_out, .reduce_size = r.reduce_size, .src_offset = r.src_offset, .dst_offset = r.dst_offset };
        },
        .repeat => |r| {
            o.kind = c.ZGML_DOP_REPEAT;
            o.u.repeat = .{ .dst = r.dst, .src = r.src, .n = r.n, .src_ne = r.src_ne, .dst_ne = r.dst_ne, .src_strides = r.src_strides, .dst_strides = r.dst_strides, .src_offset = r.src_offset, .dst_offset = r.dst_offset };
        },
        .slice_assign => |s| {
            o.kind = c.ZGML_DOP_SLICE_ASSIGN;
            o.u.slice_assign = .{ .dst = s.dst, .src = s.src, .rows = s.rows, .cols = s.cols, .dst_base_offset = s.dst_base_offset, .dst_offset = s.dst_offset, .dst_row_stride = s.dst_row_stride, .dst_col_stride = s.dst_col_stride, .src_offset = s.src_offset, .src_row_stride = s.src_row_stride, .src_col_stride = s.src_col_stride, .patch_stride = s.patch_stride };
        },
        .rope => |r| {
            o.kind = c.ZGML_DOP_ROPE;
            o.u.rope = .{ .dst = r.dst, .src = r.src, .cos_sin = r.cos_sin, ._pad = 0, .half_d = r.half_d, .seq_len = r.seq_len, .src_off = r.src_off, .cs_off = r.cs_off, .dst_off = r.dst_off, .src_rs = r.src_rs, .src_cs = r.src_cs, .cs_cs = r.cs_cs };
        },
        .attention => |a| {
            o.kind = c.ZGML_DOP_ATTENTION;
            o.u.attention = .{ .dst = a.dst, .q = a.q, .k = a.k, .v = a.v, .mask = a.mask, .has_mask = @intFromBool(a.has_mask), ._pad = 0, .d_head = a.d_head, .seq_q = a.seq_q, .seq_kv = a.seq_kv, .scale = a.scale, .q_off = a.q_off, .k_off = a.k_off, .v_off = a.v_off, .mask_off = a.mask_off, .dst_off = a.dst_off, .q_rs = a.q_rs, .q_cs = a.q_cs, .k_rs = a.k_rs, .k_cs = a.k_cs, .v_rs = a.v_rs, .v_cs = a.v_cs, .mask_rs = a.mask_rs, .mask_cs = a.mask_cs, .dst_rs = a.dst_rs, .dst_cs = a.dst_cs };
        },
        .fused_elementwise => |fe| {
            o.kind = c.ZGML_DOP_FUSED_ELEMENTWISE;
            // (the library copies the steps at compile / refresh: the arena only has to outlive the call)
            o.u.fused_elementwise = .{ .steps = try flattenSteps(arena, fe.steps), .n_steps = @intCast(fe.steps.len), .n = fe.n, .dst = fe.dst, .src = fe.src, .dst_offset = fe.dst_offset, .src_offset = fe.src_offset };
        },
    }
    return o;
}

fn flattenOps(arena: std.mem.Allocator, ops: []const backend_mod.DeviceOp) ![]c.zgml_device_op {
    const out = try arena.alloc(c.zgml_device_op, ops.len);
    for (ops, 0..) |op, i| out[i] = try flattenOp(arena, op);
    return out;
}

fn flattenIO(io: backend_mod.ProgramIO) c.zgml_program_io {
    return .{ .buf_idx = io.buf_idx, ._pad = 0, .offset = io.offset, .host_ptr = @ptrCast(io.host_ptr), .size = io.size, ._pad2 = 0 };
}

fn flattenIOs(arena: std.mem.Allocator, ios: []const backend_mod.ProgramIO) ![]c.zgml_program_io {
    const out = try arena.alloc(c.zgml_program_io, ios.len);
    for (ios, 0..) |io, i| out[i] = flattenIO(io);
    return out;
}

fn flattenProgram(arena: std.mem.Allocator, program: backend_mod.DeviceProgram) !c.zgml_device_program {
    const ops = try flattenOps(arena, program.ops);
    const sizes = try arena.alloc(u64, program.buffer_sizes.len); // usize -> u64
    for (program.buffer_sizes, 0..) |s, i| sizes[i] = @intCast(s);
    const ups = try flattenIOs(arena, program.initial_uploads);
    const qws = try arena.alloc(c.zgml_qweight_upload, program.qweights.len);
    for (program.qweights, 0..) |q, i| qws[i] = .{
        // the reference's in-memory form: int8 [K,N] + one f32 scale per `block_size` flat elements (src/backend.zig:260-266).
        // The packed-GGUF pass-through (scales = null, data = the file's 18- / 34-byte blocks; include/zgml_hip.h) needs the
        // loader to keep the raw blocks: INTEGRATION.md section 4.
        .data = q.data.ptr,
        .data_len = q.data.len,
        .scales = q.scales.ptr,
        .scales_len = q.scales.len,
        .rows = q.rows,
        .cols = q.cols,
        .block_size = q.block_size,
    };
    return .{
        .ops = ops.ptr,
        .n_ops = ops.len,
        .n_buffers = program.n_buffers,
        .buffer_sizes = sizes.ptr,
        .n_buffer_sizes = sizes.len,
        .initial_uploads = ups.ptr,
        .n_initial_uploads = ups.len,
        .qweights = qws.ptr,
        .n_qweights = qws.len,
    };
}

// ── the six vtable entries (src/backend.zig:338-352) ─────────────────────────────────────────────────────────────────

fn selfOf(ctx: *anyopaque) *HipBackend {
    return @ptrCast(@alignCast(ctx));
}

fn denseMatMulF32(ctx: *anyopaque, spec: backend_mod.DenseMatMulSpecF32) bool {
    var g = flattenGeom(spec.geom);
    // 0 = not handled: the caller falls back to the CPU kernel (src/tensor/forward.zig:2022-2031)
    return c.zgml_hip_dense_matmul_f32(selfOf(ctx).ctx, spec.dst.ptr, spec.dst.len, spec.a.ptr, spec.a.len, spec.b.ptr, spec.b.len, &g) != 0;
}

fn compileProgram(ctx: *anyopaque, program: backend_mod.DeviceProgram) ?backend_mod.Backend.CompiledHandle {
    var arena = std.heap.ArenaAllocator.init(std.heap.page_allocator);
    defer arena.deinit();
    const flat = flattenProgram(arena.allocator(), program) catch return null;
    const self = selfOf(ctx);
    const h = c.zgml_hip_compile_program(self.ctx, &flat) orelse return null;
    // DeviceInference passes the same caller-owned output slice to every executeProgram (src/device_inference.zig:170-262) and
    // requires it to outlive the session: let the step's last kernel write the logits straight into it (no staging copy of
    // 197 KB per SmolLM token). `pin_outputs = false` on the backend struct keeps the copying path.
    if (self.pin_outputs) _ = c.zgml_hip_program_pin_outputs(self.ctx, h, 1);
    return @ptrCast(h);
}

fn refreshProgram(ctx: *anyopaque, handle: backend_mod.Backend.CompiledHandle, ops: []const backend_mod.DeviceOp) void {
    const self = selfOf(ctx);
    // The per-token case, as the wgpu backend handles it (src/backend/wgpu.zig:1162-1169): the refresh IS two numbers — the KV
    // stores' column and the attentions' seq_kv (src/backend/program.zig:7466-7490) — so derive them from the first dynamic ops and
    // hand those over: O(#dynamic ops) in the library, no flattening of 1654 ops here. `strict_refresh` (off by default) keeps
    // the cpu backend's semantics instead: every field of every op is re-read, a changed static field rebuilds the launch list.
    if (!self.strict_refresh) {
        const st = backend_program.stepDynamicStateFromOps(ops);
        if (st.needsUpload() and c.zgml_hip_refresh_dynamic(self.ctx, @ptrCast(handle), st.params.slice_pos, st.params.seq_kv) == 0) return;
    }
    var arena = std.heap.ArenaAllocator.init(std.heap.page_allocator);
    defer arena.deinit();
    const flat = flattenOps(arena.allocator(), ops) catch return;
    c.zgml_hip_refresh_program(self.ctx, @ptrCast(handle), flat.ptr, flat.len);
}

fn executeProgram(ctx: *anyopaque, handle: backend_mod.Backend.CompiledHandle, inputs: []const backend_mod.ProgramIO, outputs: []const backend_mod.ProgramIO) void {
    // the decode plan sends 2 + n_layers inputs and one output per step (llama_inference.zig:405-466): no allocation up to 64 / 8
    var in_buf: [64]c.zgml_program_io = undefined;
    var out_buf: [8]c.zgml_program_io = undefined;
    var arena = std.heap.ArenaAllocator.init(std.heap.page_allocator);
    defer arena.deinit();
    const in: []c.zgml_program_io = if (inputs.len <= in_buf.len) in_buf[0..inputs.len] else (arena.allocator().alloc(c.zgml_program_io, inputs.len) catch return);
    const out: []c.zgml_program_io = if (outputs.len <= out_buf.len) out_buf[0..outputs.len] else (arena.allocator().alloc(c.zgml_program_io, outputs.len) catch return);
    for (inputs, 0..) |io, i| in[i] = flattenIO(io);
    for (outputs, 0..) |io, i| out[i] = flattenIO(io);
    c.zgml_hip_execute_program(selfOf(ctx).ctx, @ptrCast(handle), in.ptr, in.len, out.ptr, out.len);
}

fn freeProgram(ctx: *anyopaque, handle: backend_mod.Backend.CompiledHandle) void {
    c.zgml_hip_free_program(selfOf(ctx).ctx, @ptrCast(handle));
}

fn getRuntimeProfile(ctx: *anyopaque, handle: backend_mod.Backend.CompiledHandle) ?*profile.RuntimeProfile {
    const self = selfOf(ctx);
    const p = c.zgml_hip_get_runtime_profile(self.ctx, @ptrCast(handle)) orelse return null;
    // the subset a device backend fills (src/profile.zig:820-843); there is no CPU fallback, so fallback_op_count stays 0
    self.profile_storage.backend_op_count = p.*.backend_op_count;
    self.profile_storage.fallback_op_count = 0;
    self.profile_storage.backend_dispatch_count = p.*.backend_dispatch_count;
    self.profile_storage.sync_time_ns = p.*.sync_time_ns;
    self.profile_storage.sync_count = p.*.sync_count;
    self.profile_storage.call_count = p.*.call_count;
    for (0..c.ZGML_DOP_COUNT) |i| self.profile_storage.time_ns[i] = p.*.time_ns[i]; // DeviceOp tag order is identical (asserted above)
    return &self.profile_storage;
}

const vtable = backend_mod.Backend.VTable{
    .dense_matmul_f32 = denseMatMulF32,
    .compile_program = compileProgram,
    .refresh_program = refreshProgram,
    .execute_program = executeProgram,
    .free_program = freeProgram,
    .get_runtime_profile = getRuntimeProfile,
};

// ── tests (mirror src/backend/conformance.zig:353-372 and the wgpu / metal backend tests) ────────────────────────────

test "hip backend conforms to the reference executor on the core cases" {
    var hip = HipBackend.init(0) catch return; // no device: skip silently, as the metal / wgpu tests do
    defer hip.deinit();
    try @import("conformance.zig").runCoreCases(hip.backend(), 1e-5);
}

test "hip dense matmul known answer" { // src/backend/reference.zig:690-708: {58, 64, 139, 154}
    var hip = HipBackend.init(0) catch return;
    defer hip.deinit();
    const a = [_]f32{ 1, 2, 3, 4, 5, 6 };
    const b = [_]f32{ 7, 8, 9, 10, 11, 12 };
    var dst = [_]f32{ 0, 0, 0, 0 };
    const ok = hip.backend().vtable.dense_matmul_f32(hip.backend().ctx, .{
        .dst = &dst,
        .a = &a,
        .b = &b,
        .geom = .{ .M = 2, .N = 2, .K = 3, .a_row_stride = 3, .a_col_stride = 1, .b_row_stride = 2, .b_col_stride = 1, .a_offset = 0, .b_offset = 0, .dst_offset = 0, .dst_row_stride = 2 },
    });
    try std.testing.expect(ok);
    try std.testing.expectEqualSlices(f32, &.{ 58, 64, 139, 154 }, &dst);
}
