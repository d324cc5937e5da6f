// hip_backend.cpp — fills Backend::VTable with forwards into libzgml_hip.so (loaded with dlopen so
// the host library itself has no HIP link dependency). This is the C++ twin of the Zig adapter in
// flatten DeviceProgram -> zgml_device_program, forward the six vtable calls.
#include <dlfcn.h>

#include "backend.hpp"

namespace zgml::backend {

struct HipBackend::Api {
    zgml_hip_ctx* (*create)(int);
    void (*destroy)(zgml_hip_ctx*);
    const char* (*last_error)(const zgml_hip_ctx*);
    void (*capabilities)(zgml_capabilities*);
    int (*dense_matmul_f32)(zgml_hip_ctx*, float*, uint64_t, const float*, uint64_t, const float*, uint64_t,
                            const zgml_matmul_geom*);
    zgml_hip_program* (*compile_program)(zgml_hip_ctx*, const zgml_device_program*);
    void (*refresh_program)(zgml_hip_ctx*, zgml_hip_program*, const zgml_device_op*, uint64_t);
    int (*refresh_dynamic)(zgml_hip_ctx*, zgml_hip_program*, uint32_t, uint32_t);
    int (*pin_outputs)(zgml_hip_ctx*, zgml_hip_program*, int);
    void (*execute_program)(zgml_hip_ctx*, zgml_hip_program*, const zgml_program_io*, uint64_t, const zgml_program_io*,
                            uint64_t);
    void (*free_program)(zgml_hip_ctx*, zgml_hip_program*);
    zgml_runtime_profile* (*get_runtime_profile)(zgml_hip_ctx*, zgml_hip_program*);
};

namespace {

struct HipCtx {
    HipBackend::Api* api;
    zgml_hip_ctx* ctx;
};

bool vt_dense(void* c, const DenseMatMulSpecF32& s) {
    auto* h = (HipCtx*)c;
    return h->api->dense_matmul_f32(h->ctx, s.dst, s.dst_len, s.a, s.a_len, s.b, s.b_len, &s.geom) != 0;
}
Backend::CompiledHandle vt_compile(void* c, const DeviceProgram& p) {
    auto* h = (HipCtx*)c;
    std::vector<zgml_qweight_upload> qw;
    zgml_device_program flat = p.view(qw);
    zgml_hip_program* prog = h->api->compile_program(h->ctx, &flat);
    // DeviceInference hands execute_program the same session-owned output slice every step (src/device_inference.zig:170-262):
    // let the step's last kernel write into it (zig/backend_hip.zig does the same)
    if (prog && h->api->pin_outputs) h->api->pin_outputs(h->ctx, prog, 1);
    return prog;
}
void vt_refresh(void* c, Backend::CompiledHandle hd, const DeviceOp* ops, size_t n) {
    auto* h = (HipCtx*)c;
    // the per-token case as two numbers (zig/backend_hip.zig: refreshProgram; stepDynamicStateFromOps, src/backend/program.zig:7466-7490)
    uint32_t slice_pos = 0, seq_kv = 0;
    bool has_sa = false, has_att = false;
    for (size_t i = 0; i < n && !(has_sa && has_att); i++) {
        const DeviceOp& op = ops[i];
        if (op.kind == ZGML_DOP_SLICE_ASSIGN && !has_sa && op.u.slice_assign.patch_stride != 0 && op.u.slice_assign.dst_offset >= op.u.slice_assign.dst_base_offset)
            slice_pos = (op.u.slice_assign.dst_offset - op.u.slice_assign.dst_base_offset) / op.u.slice_assign.patch_stride, has_sa = true;
        else if (op.kind == ZGML_DOP_ATTENTION && !has_att)
            seq_kv = op.u.attention.seq_kv, has_att = true;
    }
    if ((has_sa || has_att) && h->api->refresh_dynamic && h->api->refresh_dynamic(h->ctx, (zgml_hip_program*)hd, slice_pos, seq_kv) == 0) return;
    h->api->refresh_program(h->ctx, (zgml_hip_program*)hd, ops, n);
}
void vt_execute(void* c, Backend::CompiledHandle hd, const ProgramIO* in, size_t n_in, const ProgramIO* out, size_t n_out) {
    auto* h = (HipCtx*)c;
    h->api->execute_program(h->ctx, (zgml_hip_program*)hd, in, n_in, out, n_out);
}
void vt_free(void* c, Backend::CompiledHandle hd) {
    auto* h = (HipCtx*)c;
    h->api->free_program(h->ctx, (zgml_hip_program*)hd);
}
zgml_runtime_profile* vt_profile(void* c, Backend::CompiledHandle hd) {
    auto* h = (HipCtx*)c;
    return h->api->get_runtime_profile(h->ctx, (zgml_hip_program*)hd);
}
const Backend::VTable kVTable = {vt_dense, vt_compile, vt_refresh, vt_execute, vt_free, vt_profile};

} // namespace

bool HipBackend::init(const char* lib_path, int device_ordinal) {
    lib = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        error = std::string("dlopen failed: ") + dlerror();
        return false;
    }
    api = new Api();
#define SYM(field, name)                                              \
    api->field = (decltype(api->field))dlsym(lib, name);              \
    if (!api->field) {                                                \
        error = std::string("missing symbol ") + name;                \
        return false;                                                 \
    }
    SYM(create, "zgml_hip_create")
    SYM(destroy, "zgml_hip_destroy")
    SYM(last_error, "zgml_hip_last_error")
    SYM(capabilities, "zgml_hip_capabilities")
    SYM(dense_matmul_f32, "zgml_hip_dense_matmul_f32")
    SYM(compile_program, "zgml_hip_compile_program")
    SYM(refresh_program, "zgml_hip_refresh_program")
    SYM(refresh_dynamic, "zgml_hip_refresh_dynamic")
    SYM(pin_outputs, "zgml_hip_program_pin_outputs")
    SYM(execute_program, "zgml_hip_execute_program")
    SYM(free_program, "zgml_hip_free_program")
    SYM(get_runtime_profile, "zgml_hip_get_runtime_profile")
#undef SYM
    ctx = api->create(device_ordinal);
    if (!ctx) {
        error = std::string("zgml_hip_create failed: ") + api->last_error(nullptr);
        return false;
    }
    return true;
}

void HipBackend::deinit() {
    if (ctx && api) api->destroy(ctx);
    ctx = nullptr;
    delete api;
    api = nullptr;
    if (lib) dlclose(lib);
    lib = nullptr;
}

Backend HipBackend::backend() {
    static thread_local HipCtx hc; // Backend is a value type; the ctx it points at lives here
    hc = {api, ctx};
    zgml_capabilities c;
    api->capabilities(&c);
    Backend b;
    b.ctx = &hc;
    b.vtable = &kVTable;
    b.name_str = "hip";
    b.device_type = Device::hip;
    b.capabilities = Capabilities::fromC(c);
    return b;
}

} // namespace zgml::backend
