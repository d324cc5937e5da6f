// conformance.cpp — the reference's backend conformance suite (src/backend/conformance.zig:13-372)
// against the MI355X backend, written with the C++ host mirror of the plugin types so it reads like
// the original: build a DeviceProgram, run it on the backend under test and on the reference
// executor (here: the CPU oracle, loaded with dlopen — test infrastructure), compare with `tol`.
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

#include "../../zgml_amd/host/backend.hpp"

using namespace zgml::backend;

// ── the oracle as a Backend (reference.executeProgram stand-in) ─────────────────────────────
struct OracleFns {
    void* (*compile)(void*, const zgml_device_program*);
    void (*refresh)(void*, void*, const zgml_device_op*, uint64_t);
    void (*execute)(void*, void*, const zgml_program_io*, uint64_t, const zgml_program_io*, uint64_t);
    void (*free_)(void*, void*);
} g_oracle;

static Backend::CompiledHandle o_compile(void*, const DeviceProgram& p) {
    std::vector<zgml_qweight_upload> qw;
    zgml_device_program flat = p.view(qw);
    return g_oracle.compile(nullptr, &flat);
}
static void o_refresh(void*, Backend::CompiledHandle h, const DeviceOp* ops, size_t n) { g_oracle.refresh(nullptr, h, ops, n); }
static void o_execute(void*, Backend::CompiledHandle h, const ProgramIO* in, size_t ni, const ProgramIO* out, size_t no) {
    g_oracle.execute(nullptr, h, in, ni, out, no);
}
static void o_free(void*, Backend::CompiledHandle h) { g_oracle.free_(nullptr, h); }
static bool o_dense(void*, const DenseMatMulSpecF32&) { return false; }
static zgml_runtime_profile* o_prof(void*, Backend::CompiledHandle) { return nullptr; }
static const Backend::VTable kOracleVT = {o_dense, o_compile, o_refresh, o_execute, o_free, o_prof};

static Backend referenceBackend() {
    Backend b;
    b.vtable = &kOracleVT;
    b.name_str = "reference";
    b.device_type = Device::cpu;
    b.capabilities.compiled_programs = b.capabilities.dense_matmul_f32 = b.capabilities.qmatmul = true;
    b.capabilities.fused_elementwise = true;
    b.capabilities.attention.supported = true;
    b.capabilities.attention.max_d_head = 512;
    return b;
}

// ── helpers mirroring conformance.zig:13-60 ──────────────────────────────────────────────────
static ProgramIO io(uint16_t buf, const float* data, size_t n) {
    ProgramIO p{};
    p.buf_idx = buf, p.offset = 0, p.host_ptr = (void*)data, p.size = (uint32_t)(n * sizeof(float));
    return p;
}

static std::vector<float> backendOutput(const Backend& be, const DeviceProgram& program, uint16_t out_idx, size_t out_len) {
    auto handle = be.compileProgram(program);
    if (!handle) {
        std::fprintf(stderr, "CompileFailed on %s\n", be.name_str.c_str());
        std::exit(2);
    }
    std::vector<float> out(out_len);
    ProgramIO o = io(out_idx, out.data(), out_len);
    be.executeProgram(handle, nullptr, 0, &o, 1);
    be.freeProgram(handle);
    return out;
}

static int g_failures = 0;
static void assertBackendMatchesReference(const char* name, const Backend& be, const Backend& ref, const DeviceProgram& program,
                                          uint16_t out_idx, size_t out_len, float tol) {
    if (!be.supportsProgram(program)) {
        std::printf("SKIP %s (unsupported)\n", name);
        return;
    }
    const auto expected = backendOutput(ref, program, out_idx, out_len);
    const auto got = backendOutput(be, program, out_idx, out_len);
    bool ok = true;
    for (size_t i = 0; i < out_len; i++)
        if (!(std::fabs(expected[i] - got[i]) <= tol)) ok = false;
    std::printf("%s %s\n", ok ? "PASS" : "FAIL", name);
    if (!ok) g_failures++;
}

static void runCoreCases(const Backend& be, const Backend& ref, float tol) {
    { // matmul 2x3 * 3x2
        float a[] = {1, 2, 3, 4, 5, 6}, b[] = {7, 8, 9, 10, 11, 12};
        DeviceProgram p;
        MatMulGeometry g{2, 2, 3, 3, 1, 2, 1, 0, 0, 0, 2};
        p.ops = {DeviceOp::matmul(2, 0, 1, g)};
        p.n_buffers = 3, p.buffer_sizes = {6, 6, 4};
        p.initial_uploads = {io(0, a, 6), io(1, b, 6)};
        assertBackendMatchesReference("matmul", be, ref, p, 2, 4, tol);
    }
    { // qmatmul with offsets/strides, sentinels untouched
        float input[] = {99, 1, 2, 3, 99, -1, 0.5f, 4, 99}, dst_init[] = {-7, -7, -7, -7, -7, -7, -7, -7, -7};
        const int8_t qdata[] = {2, -1, 3, 4, -2, 1, -3, 5, 2};
        const float scales[] = {0.5f, 0.25f, 1.0f};
        DeviceProgram p;
        p.ops = {DeviceOp::qmatmul(1, 0, 0, 2, 3, 3, 1, 4, 1, 4)};
        p.n_buffers = 2, p.buffer_sizes = {9, 9};
        p.initial_uploads = {io(0, input, 9), io(1, dst_init, 9)};
        p.qweights = {{qdata, 9, scales, 3, 3, 3, 4}};
        assertBackendMatchesReference("qmatmul_offsets", be, ref, p, 1, 9, tol);
    }
    { // elementwise add
        float a[] = {1, 2, 3, 4}, b[] = {10, 20, 30, 40};
        DeviceProgram p;
        p.ops = {DeviceOp::elementwise(ZGML_OP_ADD, 2, 0, 1, 4)};
        p.n_buffers = 3, p.buffer_sizes = {4, 4, 4};
        p.initial_uploads = {io(0, a, 4), io(1, b, 4)};
        assertBackendMatchesReference("add", be, ref, p, 2, 4, tol);
    }
    { // reduce sum + max
        float src[] = {1, -2, 3, 4, 5, -6};
        DeviceProgram p;
        p.ops = {DeviceOp::reduce(ZGML_OP_SUM, 1, 0, 2, 3), DeviceOp::reduce(ZGML_OP_MAX, 1, 0, 2, 3, 0, 2)};
        p.n_buffers = 2, p.buffer_sizes = {6, 4};
        p.initial_uploads = {io(0, src, 6)};
        assertBackendMatchesReference("reduce", be, ref, p, 1, 4, tol);
    }
    { // repeat
        float src[] = {7, 8};
        DeviceProgram p;
        p.ops = {DeviceOp::repeat(1, 0, 6, {2, 1, 1, 1}, {2, 3, 1, 1}, {1, 2, 2, 2}, {1, 2, 6, 6})};
        p.n_buffers = 2, p.buffer_sizes = {2, 6};
        p.initial_uploads = {io(0, src, 2)};
        assertBackendMatchesReference("repeat", be, ref, p, 1, 6, tol);
    }
    { // slice_assign with strides
        float src[] = {99, 2, 3, 5, 6, 77}, dst[] = {10, 11, 12, 13, 14, 15, 16, 17};
        DeviceProgram p;
        p.ops = {DeviceOp::slice_assign(1, 0, 2, 2, 0, 2, 1, 2, 1, 1, 2, 2)};
        p.n_buffers = 2, p.buffer_sizes = {6, 8};
        p.initial_uploads = {io(0, src, 6), io(1, dst, 8)};
        assertBackendMatchesReference("slice_assign", be, ref, p, 1, 8, tol);
    }
    { // softmax
        float src[] = {1, 2, 3, -1, 0, 1};
        DeviceProgram p;
        p.ops = {DeviceOp::softmax(1, 0, 2, 3)};
        p.n_buffers = 2, p.buffer_sizes = {6, 6};
        p.initial_uploads = {io(0, src, 6)};
        assertBackendMatchesReference("softmax", be, ref, p, 1, 6, tol);
    }
    { // layernorm + rmsnorm
        float src[] = {1, 2, 3, 4, -1, 0, 1, 2};
        DeviceProgram p;
        p.ops = {DeviceOp::layernorm(1, 0, 2, 4, 1e-5f), DeviceOp::rmsnorm(1, 0, 2, 4, 1e-5f, 0, 8)};
        p.n_buffers = 2, p.buffer_sizes = {8, 16};
        p.initial_uploads = {io(0, src, 8)};
        assertBackendMatchesReference("norms", be, ref, p, 1, 16, tol);
    }
    { // rope
        float src[] = {1, 2, 3, 4, 5, 6, 7, 8}, cs[] = {1, 1, 0, 0, 0, 0, 1, 1};
        DeviceProgram p;
        p.ops = {DeviceOp::rope(2, 0, 1, 2, 2, 0, 0, 0, 1, 4, 4)};
        p.n_buffers = 3, p.buffer_sizes = {8, 8, 8};
        p.initial_uploads = {io(0, src, 8), io(1, cs, 8)};
        assertBackendMatchesReference("rope", be, ref, p, 2, 8, tol);
    }
    { // masked attention, seq_q = 2, seq_kv = 3, one -inf
        float q[] = {0.2f, 0.1f, -0.3f, 0.4f, -0.1f, 0.5f, 0.2f, -0.4f};
        float k[] = {0.1f, 0.2f, 0.3f, 0.4f, -0.2f, 0.3f, 0.1f, -0.1f, 0.5f, -0.4f, 0.2f, 0.1f};
        float v[] = {1, 2, 3, 4, -1, 0.5f, 2, -0.5f, 0.25f, -0.75f, 1.5f, 2.5f};
        const float ninf = -std::numeric_limits<float>::infinity();
        float mask[] = {0, 0, ninf, 0, -0.25f, 0};
        zgml_op_attention a{};
        a.dst = 4, a.q = 0, a.k = 1, a.v = 2, a.mask = 3, a.has_mask = 1, a.d_head = 4, a.seq_q = 2, a.seq_kv = 3, a.scale = 0.5f;
        a.q_rs = 1, a.q_cs = 4, a.k_rs = 1, a.k_cs = 4, a.v_rs = 1, a.v_cs = 4, a.mask_rs = 1, a.mask_cs = 3, a.dst_rs = 1, a.dst_cs = 4;
        DeviceProgram p;
        p.ops = {DeviceOp::attention(a)};
        p.n_buffers = 5, p.buffer_sizes = {8, 12, 12, 6, 8};
        p.initial_uploads = {io(0, q, 8), io(1, k, 12), io(2, v, 12), io(3, mask, 6)};
        assertBackendMatchesReference("attention", be, ref, p, 4, 8, tol);
    }
    { // fused relu -> sqrt -> add
        float src[] = {1, -2, 4, 9}, addend[] = {10, 20, 30, 40};
        const FusedEwStep steps[] = {{ZGML_OP_RELU, 0, 0, 0, 0}, {ZGML_OP_SQRT, 0, 0, 0, 0}, {ZGML_OP_ADD, 0, 0, 1, 0}};
        DeviceProgram p;
        p.ops = {DeviceOp::fused_elementwise(steps, 3, 4, 2, 0)};
        p.n_buffers = 3, p.buffer_sizes = {4, 4, 4};
        p.initial_uploads = {io(0, src, 4), io(1, addend, 4)};
        assertBackendMatchesReference("fused_elementwise", be, ref, p, 2, 4, tol);
    }
}

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: conformance <libzgml_hip.so> <libzgml_oracle.so>\n");
        return 2;
    }
    void* ol = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
    if (!ol) {
        std::fprintf(stderr, "oracle: %s\n", dlerror());
        return 2;
    }
    g_oracle.compile = (decltype(g_oracle.compile))dlsym(ol, "zo_vt_compile_program");
    g_oracle.refresh = (decltype(g_oracle.refresh))dlsym(ol, "zo_vt_refresh_program");
    g_oracle.execute = (decltype(g_oracle.execute))dlsym(ol, "zo_vt_execute_program");
    g_oracle.free_ = (decltype(g_oracle.free_))dlsym(ol, "zo_vt_free_program");
    HipBackend hip;
    if (!hip.init(argv[1], 0)) { // the reference silently returns without a device (conformance.zig:353-372)
        std::printf("NO_DEVICE %s\n", hip.error.c_str());
        return 3;
    }
    Backend be = hip.backend(), ref = referenceBackend();
    // test "dispatch helper" / host dense override (src/backend/cpu.zig:149-163)
    {
        float dst[4] = {0}, a[] = {1, 2, 3, 4, 5, 6}, b[] = {7, 8, 9, 10, 11, 12};
        DenseMatMulSpecF32 spec{dst, 4, a, 6, b, 6, {2, 2, 3, 3, 1, 2, 1, 0, 0, 0, 2}};
        const bool ok = tryDenseMatMul(&be, spec) && dst[0] == 58 && dst[1] == 64 && dst[2] == 139 && dst[3] == 154 &&
                        !tryDenseMatMul(nullptr, spec);
        std::printf("%s host_dense_matmul\n", ok ? "PASS" : "FAIL");
        if (!ok) g_failures++;
    }
    runCoreCases(be, ref, 1e-5f);
    hip.deinit();
    std::printf("%s\n", g_failures ? "CONFORMANCE_FAILED" : "CONFORMANCE_OK");
    return g_failures ? 1 : 0;
}
