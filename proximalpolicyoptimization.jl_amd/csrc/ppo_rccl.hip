// ppo_rccl.hip -- optional native gradient all-reduce: RCCL called from the library itself, on the engine's stream,
// instead of a host-language callback per optimiser step (one sum all-reduce of the flat [np + 2]
// buffer per step).  RCCL is resolved at run time (dlopen), so the library still loads where RCCL is absent and a
// host that already carries an RCCL (torch) shares its copy.  The communicator is created from a unique id that the
// HOST distributes between its ranks (torch.distributed / MPI.jl / a file): the engine owns no rendezvous.
#include "ppo_internal.h"
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <mutex>

namespace {
typedef struct { char internal[128]; } rcclUniqueId;           // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rcclComm;
typedef int (*fn_get_unique_id)(rcclUniqueId*);
typedef int (*fn_comm_init_rank)(rcclComm*, int, rcclUniqueId, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, rcclComm, hipStream_t);
typedef int (*fn_comm_destroy)(rcclComm);
typedef const char* (*fn_error_string)(int);
typedef int (*fn_comm_count)(rcclComm, int*);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;                   // ncclDataType_t / ncclRedOp_t values (rccl.h)

void* g_lib = nullptr;
fn_get_unique_id p_get_unique_id = nullptr;
fn_comm_init_rank p_comm_init_rank = nullptr;
fn_all_reduce p_all_reduce = nullptr;
fn_comm_destroy p_comm_destroy = nullptr;
fn_error_string p_error_string = nullptr;
fn_comm_count p_comm_count = nullptr, p_comm_user_rank = nullptr;
thread_local rcclComm g_comm = nullptr;      // one communicator per engine = per host thread (ppo_api.hip)

std::mutex g_load_mu;
int32_t load_rccl() {
    std::lock_guard<std::mutex> lk(g_load_mu);
    if (g_lib) return PPO_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        g_lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_lib) break;
    }
    if (!g_lib) { ppo_set_error("native all-reduce: librccl.so not found"); return PPO_ERR_UNSUPPORTED; }
    p_get_unique_id = (fn_get_unique_id)dlsym(g_lib, "ncclGetUniqueId");
    p_comm_init_rank = (fn_comm_init_rank)dlsym(g_lib, "ncclCommInitRank");
    p_all_reduce = (fn_all_reduce)dlsym(g_lib, "ncclAllReduce");
    p_comm_destroy = (fn_comm_destroy)dlsym(g_lib, "ncclCommDestroy");
    p_error_string = (fn_error_string)dlsym(g_lib, "ncclGetErrorString");
    p_comm_count = (fn_comm_count)dlsym(g_lib, "ncclCommCount");
    p_comm_user_rank = (fn_comm_count)dlsym(g_lib, "ncclCommUserRank");
    if (!p_get_unique_id || !p_comm_init_rank || !p_all_reduce || !p_comm_destroy) {
        ppo_set_error("native all-reduce: librccl.so lacks the nccl* entry points");
        dlclose(g_lib); g_lib = nullptr;
        return PPO_ERR_UNSUPPORTED;
    }
    return PPO_OK;
}
int32_t rccl_fail(const char* what, int rc) {
    char msg[256];
    snprintf(msg, sizeof msg, "%s failed: %s (%d)", what, p_error_string ? p_error_string(rc) : "rccl error", rc);
    ppo_set_error(msg);
    return PPO_ERR_HIP;
}
}  // namespace

extern "C" {

// can this process resolve RCCL at all?  (local, not collective: the host agrees on the answer BEFORE any rank enters the
// collective ncclCommInitRank, where a rank that cannot load the library would leave the others waiting)
int32_t ppo_rccl_probe(void) { return load_rccl(); }

int32_t ppo_rccl_unique_id(uint8_t* out128) {
    ARG_CHECK(out128, "ppo_rccl_unique_id: null buffer");
    PPO_TRY(load_rccl());
    rcclUniqueId id;
    const int rc = p_get_unique_id(&id);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(out128, id.internal, 128);
    return PPO_OK;
}

int32_t ppo_rccl_init(int32_t rank, int32_t world, const uint8_t* id128) {
    ARG_CHECK(id128 && world >= 1 && rank >= 0 && rank < world, "ppo_rccl_init: bad rank / world / id");
    ARG_CHECK(!g_comm, "ppo_rccl_init: communicator already initialised");
    PPO_TRY(load_rccl());
    rcclUniqueId id;
    memcpy(id.internal, id128, 128);
    const int rc = p_comm_init_rank(&g_comm, world, id, rank);          // on the current HIP device (ppo_device_init)
    if (rc != 0) { g_comm = nullptr; return rccl_fail("ncclCommInitRank", rc); }
    return PPO_OK;
}

// ppo_allreduce_fn-compatible: in-place fp32 sum over all ranks, enqueued on the engine's stream
int32_t ppo_rccl_allreduce(void* /*ctx*/, void* grad_dev, int64_t n_floats) {
    if (!g_comm) { ppo_set_error("ppo_rccl_allreduce: ppo_rccl_init has not run"); return PPO_ERR_ARG; }
    const int rc = p_all_reduce(grad_dev, grad_dev, (size_t)n_floats, kNcclFloat32, kNcclSum, g_comm, ppo_stream());
    if (rc != 0) return rccl_fail("ncclAllReduce", rc);
    return PPO_OK;
}

// rank / size as the communicator itself reports them (bench.py prints these, not the launcher's environment)
int32_t ppo_rccl_comm_info(int32_t* rank, int32_t* world) {
    if (!g_comm) { ppo_set_error("ppo_rccl_comm_info: ppo_rccl_init has not run"); return PPO_ERR_ARG; }
    if (!p_comm_count || !p_comm_user_rank) { ppo_set_error("native all-reduce: librccl.so lacks ncclCommCount / ncclCommUserRank"); return PPO_ERR_UNSUPPORTED; }
    int n = 0, r = 0;
    int rc = p_comm_count(g_comm, &n);
    if (rc != 0) return rccl_fail("ncclCommCount", rc);
    rc = p_comm_user_rank(g_comm, &r);
    if (rc != 0) return rccl_fail("ncclCommUserRank", rc);
    if (rank) *rank = r;
    if (world) *world = n;
    return PPO_OK;
}

// one all-reduce of a known vector on the engine's stream, checked on the host: *ok = 1 when every element came back as
// the sum over all ranks.  Collective (every rank of the communicator calls it); run once after ppo_rccl_init before
// the communicator is trusted with gradients.
int32_t ppo_rccl_self_test(int32_t* ok) {
    ARG_CHECK(ok, "ppo_rccl_self_test: null out");
    *ok = 0;
    int32_t rank = 0, world = 0;
    PPO_TRY(ppo_rccl_comm_info(&rank, &world));
    constexpr int n = 1024;
    DevBuf<float> buf;
    PPO_TRY(buf.alloc(n));
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)(rank + 1) + (float)(i & 7);          // sum over ranks: world(world+1)/2 + world*(i&7)
    HIP_TRY(hipMemcpyAsync(buf.p, h.data(), n * sizeof(float), hipMemcpyHostToDevice, ppo_stream()));
    PPO_TRY(ppo_rccl_allreduce(nullptr, buf.p, n));
    HIP_TRY(hipMemcpyAsync(h.data(), buf.p, n * sizeof(float), hipMemcpyDeviceToHost, ppo_stream()));
    HIP_TRY(hipStreamSynchronize(ppo_stream()));
    bool good = true;
    for (int i = 0; i < n; ++i) good = good && h[i] == (float)(world * (world + 1) / 2) + (float)(world * (i & 7));
    *ok = good ? 1 : 0;
    return PPO_OK;
}

int32_t ppo_rccl_finalize(void) {
    if (g_comm) {
        (void)hipStreamSynchronize(ppo_stream());
        const int rc = p_comm_destroy(g_comm);
        g_comm = nullptr;
        if (rc != 0) return rccl_fail("ncclCommDestroy", rc);
    }
    return PPO_OK;
}

}  // extern "C"
