// What does a dependent kernel boundary cost when the AQL packets are written by hand? (inside a
// hipGraph it is 1.67 us.) N dispatches of one kernel are written into a user-mode queue behind ONE doorbell; variants:
// barrier bit on / off, acquire / release fence scopes none / agent / system. bump_kernel checks that the chain is
// really ordered (every word must equal N at the end).
//   hipcc --offload-arch=gfx950 --offload-device-only --no-gpu-bundle-output -O3 tools/exp/aql_kernels.hip -o tools/exp/aql_kernels.hsaco
//   g++ -O2 -std=c++17 tools/exp/aql.cpp -I/opt/rocm/include -L/opt/rocm/lib -lhsa-runtime64 -o tools/exp/aql
//   tools/exp/aql tools/exp/aql_kernels.hsaco
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#define CK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(s_, &m_); fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, m_ ? m_ : "?"); exit(1); } } while (0)

static hsa_agent_t g_gpu{}, g_cpu{};
static hsa_amd_memory_pool_t g_kernarg{}, g_coarse{};
static hsa_status_t on_agent(hsa_agent_t a, void*) {
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_gpu.handle) g_gpu = a;
    if (t == HSA_DEVICE_TYPE_CPU && !g_cpu.handle) g_cpu = a;
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_cpu_pool(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if (seg == HSA_AMD_SEGMENT_GLOBAL && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_kernarg.handle) g_kernarg = p;
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_gpu_pool(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if (seg == HSA_AMD_SEGMENT_GLOBAL && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_coarse.handle) g_coarse = p;
    return HSA_STATUS_SUCCESS;
}

struct Kernel {
    uint64_t object = 0;
    uint32_t kernarg = 0, group = 0, priv = 0;
};
static Kernel symbol(hsa_executable_t ex, const char* name) {
    hsa_executable_symbol_t sym;
    CK(hsa_executable_get_symbol_by_name(ex, name, &g_gpu, &sym));
    Kernel k;
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
    return k;
}

int main(int argc, char** argv) {
    if (argc < 2) return fprintf(stderr, "usage: aql <code object>\n"), 2;
    CK(hsa_init());
    CK(hsa_iterate_agents(on_agent, nullptr));
    CK(hsa_amd_agent_iterate_memory_pools(g_cpu, on_cpu_pool, nullptr));
    CK(hsa_amd_agent_iterate_memory_pools(g_gpu, on_gpu_pool, nullptr));
    if (!g_gpu.handle || !g_kernarg.handle || !g_coarse.handle) return fprintf(stderr, "no gpu / pools\n"), 1;

    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (blob.empty()) return fprintf(stderr, "cannot read %s\n", argv[1]), 1;
    hsa_code_object_reader_t rd;
    CK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &rd));
    hsa_executable_t ex;
    CK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
    CK(hsa_executable_load_agent_code_object(ex, g_gpu, rd, nullptr, nullptr));
    CK(hsa_executable_freeze(ex, nullptr));
    const Kernel k_empty = symbol(ex, "empty_kernel.kd"), k_bump = symbol(ex, "bump_kernel.kd");

    const uint32_t QSIZE = 8192;
    hsa_queue_t* q = nullptr;
    CK(hsa_queue_create(g_gpu, QSIZE, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    hsa_signal_t done;
    CK(hsa_signal_create(1, 0, nullptr, &done));

    const uint32_t WGS = argc > 2 ? (uint32_t)atoi(argv[2]) : 256, WG = 256; // default: one workgroup per CU (the buffers below hold 256 x 256 words: WGS <= 256 for the bump / rotate kernels)
    unsigned* words = nullptr;
    CK(hsa_amd_memory_pool_allocate(g_coarse, (size_t)WGS * WG * 4, 0, (void**)&words));
    CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, words));
    void** kernarg = nullptr;
    CK(hsa_amd_memory_pool_allocate(g_kernarg, 4096, 0, (void**)&kernarg));
    CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kernarg));
    memset(kernarg, 0, 4096);
    kernarg[0] = words;
    unsigned* words2 = nullptr;
    CK(hsa_amd_memory_pool_allocate(g_coarse, (size_t)WGS * WG * 4, 0, (void**)&words2));
    CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, words2));
    // ping-pong argument blocks of the rotate kernels: even dispatches words -> words2, odd ones back
    kernarg[64] = words, kernarg[65] = words2;
    kernarg[128] = words2, kernarg[129] = words;
    // the argument blocks live in DEVICE memory (as HIP places them): from host memory every wave's s_load crosses PCIe
    // and a one-load kernel costs ~9 us per dispatch
    void** kernarg_host = kernarg;
    CK(hsa_amd_memory_pool_allocate(g_coarse, 4096, 0, (void**)&kernarg));
    CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kernarg));
    {
        hsa_signal_t cs;
        CK(hsa_signal_create(1, 0, nullptr, &cs));
        CK(hsa_amd_memory_async_copy(kernarg, g_gpu, kernarg_host, g_cpu, 4096, 0, nullptr, cs));
        hsa_signal_wait_scacquire(cs, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
        hsa_signal_destroy(cs);
    }
    unsigned* host = nullptr;
    CK(hsa_amd_memory_pool_allocate(g_kernarg, (size_t)WGS * WG * 4, 0, (void**)&host));
    CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, host));

    auto run = [&](const char* what, const Kernel& k, uint32_t n, bool barrier, int acq, int rel, bool check, bool rotate = false) {
        if (check) {
            memset(host, 0, (size_t)WGS * WG * 4);
            hsa_signal_t cs;
            CK(hsa_signal_create(1, 0, nullptr, &cs));
            CK(hsa_amd_memory_async_copy(words, g_gpu, host, g_cpu, (size_t)WGS * WG * 4, 0, nullptr, cs));
            hsa_signal_wait_scacquire(cs, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
            hsa_signal_destroy(cs);
        }
        double best = 1e30;
        for (int rep = 0; rep < 5; rep++) {
            hsa_signal_store_relaxed(done, 1);
            const uint64_t base = hsa_queue_add_write_index_relaxed(q, n);
            for (uint32_t i = 0; i < n; i++) {
                hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + ((base + i) & (QSIZE - 1));
                const bool lastp = i + 1 == n;
                p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
                p->workgroup_size_x = WG, p->workgroup_size_y = 1, p->workgroup_size_z = 1;
                p->grid_size_x = WGS * WG, p->grid_size_y = 1, p->grid_size_z = 1;
                p->private_segment_size = k.priv, p->group_segment_size = k.group;
                p->kernel_object = k.object;
                p->kernarg_address = rotate ? (void*)(kernarg + ((base + i) & 1 ? 128 : 64)) : (void*)kernarg;
                p->completion_signal = lastp ? done : hsa_signal_t{0};
                // first packet acquires at system scope (sees the host's writes), last releases to the system
                const int a = i == 0 ? HSA_FENCE_SCOPE_SYSTEM : acq, r = lastp ? HSA_FENCE_SCOPE_SYSTEM : rel;
                const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                        (a << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (r << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
                __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
            }
            const auto t0 = std::chrono::steady_clock::now();
            hsa_signal_store_screlease(q->doorbell_signal, base + n - 1);
            hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
        }
        long bad = -1;
        if (check) {
            hsa_signal_t cs;
            CK(hsa_signal_create(1, 0, nullptr, &cs));
            CK(hsa_amd_memory_async_copy(host, g_cpu, words, g_gpu, (size_t)WGS * WG * 4, 0, nullptr, cs));
            hsa_signal_wait_scacquire(cs, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
            hsa_signal_destroy(cs);
            bad = 0;
            for (uint32_t i = 0; i < WGS * WG; i++) bad += host[i] != 5u * n; // (rotate: n even, so the result is back in `words`; every word counts the dispatches)
        }
        printf("%-28s barrier=%d acq=%d rel=%d  n=%u  %8.3f us per dispatch%s", what, barrier, acq, rel, n, best / n,
               check ? (bad ? "  WRONG" : "  ordered ok") : "");
        if (check && bad) printf(" (%ld words differ; word 0 = %u, word 65535 = %u, want %u)", bad, host[0], host[WGS * WG - 1], 5u * n);
        printf("\n");
        fflush(stdout);
    };
    const uint32_t N = 2000;
    for (int barrier = 1; barrier >= 0; barrier--)
        for (int scope = 0; scope <= 2; scope++) run("empty kernel", k_empty, N, barrier, scope, scope, false);
    if (WGS != 256) { // grid-size sweep: empty kernels only
        hsa_queue_destroy(q);
        hsa_shut_down();
        return 0;
    }
    for (int scope = 0; scope <= 2; scope++) run("bump kernel (dependent)", k_bump, N, true, scope, scope, true);
    run("bump kernel, no barrier", k_bump, N, false, 1, 1, true);
    const Kernel k_bump1 = symbol(ex, "bump_sc1_kernel.kd"), k_rot = symbol(ex, "rotate_kernel.kd"), k_rot1 = symbol(ex, "rotate_sc1_kernel.kd");
    for (int scope = 0; scope <= 1; scope++) run("bump, sc1 accesses", k_bump1, N, true, scope, scope, true);
    for (int scope = 0; scope <= 2; scope++) run("rotate (cross-XCD reads)", k_rot, N, true, scope, scope, true, true);
    for (int scope = 0; scope <= 1; scope++) run("rotate, sc1 accesses", k_rot1, N, true, scope, scope, true, true);
    run("bump, acquire agent only", k_bump, N, true, 1, 0, true);
    run("bump, release agent only", k_bump, N, true, 0, 1, true);
    hsa_queue_destroy(q);
    hsa_shut_down();
    return 0;
}
