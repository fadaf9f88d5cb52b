// ppo_api.hip -- extern "C" surface of libppo_hip.so (include/ppo_hip.h): handle management and
// the host-side orchestration of collect_rollouts! / ppo_train! (launch order only; all math is
// in the kernels).  There is NO CPU fallback: every entry point runs on the GPU or fails.
#include "ppo_internal.h"
#include "ppo_device.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>


// ---------------------------------------------------------------- engine state: ONE ENGINE PER HOST THREAD
// Everything that makes up an engine instance -- the HIP device (hipSetDevice is per thread), the stream every kernel of
// that engine is enqueued on, the kernel-timing tables, the RCCL communicator (ppo_rccl.hip), the last error text -- is
// thread_local: a process may run several engines, one per host thread (e.g. one per GPU), each with its own stream and
// communicator, and handles belong to the thread that created them ("one stream per engine handle").  What
// stays process-wide are the tuning knobs (ppo_set_*) and the pinned-record pool of the disk sink (mutex-protected).
static thread_local std::string g_err;
static thread_local hipStream_t g_stream = nullptr;
static thread_local bool g_own_stream = false;
static thread_local bool g_init = false;
static thread_local bool g_prof = false;
struct ProfRec { hipEvent_t e0, e1; };
static thread_local std::map<std::string, std::vector<ProfRec>> g_prof_pending;
static thread_local std::map<std::string, std::pair<double, int64_t>> g_prof_done;

void ppo_set_error(const std::string& msg) { g_err = msg; }
hipStream_t ppo_stream() { return g_stream; }
int ppo_hip_fail(hipError_t e, const char* what, const char* file, int line) {
    g_err = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what + " at " + file + ":" + std::to_string(line);
    return PPO_ERR_HIP;
}

ProfScope::ProfScope(const char* n) : name(n), e0(nullptr), e1(nullptr), on(g_prof) {
    if (on) {
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, g_stream);
    }
}
ProfScope::~ProfScope() {
    if (on) {
        (void)hipEventRecord(e1, g_stream);
        g_prof_pending[name].push_back({e0, e1});
    }
}

static int32_t ensure_init() {
    if (g_init) return PPO_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        ppo_set_error("no HIP device available: libppo_hip has no CPU fallback");
        return PPO_ERR_HIP;
    }
    HIP_TRY(hipStreamCreate(&g_stream));
    g_own_stream = true;
    g_init = true;
    return PPO_OK;
}

template <typename T>
static int32_t h2d(T* dst, const T* src, size_t n) {
    if (n == 0) return PPO_OK;
    HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return PPO_OK;
}
template <typename T>
static int32_t d2h(T* dst, const T* src, size_t n) {
    if (n == 0) return PPO_OK;
    HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return PPO_OK;
}

extern "C" {

int32_t ppo_version(void) { return 200; }

int32_t ppo_last_error(char* buf, int64_t cap) {
    if (!buf || cap <= 0) return PPO_ERR_ARG;
    std::strncpy(buf, g_err.c_str(), (size_t)cap - 1);
    buf[cap - 1] = 0;
    return PPO_OK;
}

int32_t ppo_device_count(int32_t* out) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    if (out) *out = n;
    return PPO_OK;
}

int32_t ppo_device_init(int32_t device_ordinal) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { ppo_set_error("no HIP device available: libppo_hip has no CPU fallback"); return PPO_ERR_HIP; }
    ARG_CHECK(device_ordinal >= 0 && device_ordinal < n, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device_ordinal));
    if (!g_init) {
        HIP_TRY(hipStreamCreate(&g_stream));
        g_own_stream = true;
        g_init = true;
    }
    return PPO_OK;
}

int32_t ppo_set_stream(void* hip_stream) {
    PPO_TRY(ensure_init());
    if (g_own_stream && g_stream) { (void)hipStreamSynchronize(g_stream); (void)hipStreamDestroy(g_stream); }
    g_stream = (hipStream_t)hip_stream;
    g_own_stream = false;
    return PPO_OK;
}

int32_t ppo_device_synchronize(void) {
    PPO_TRY(ensure_init());
    HIP_TRY(hipStreamSynchronize(g_stream));
    return PPO_OK;
}

int32_t ppo_profile_enable(int32_t on) {
    g_prof = on != 0;
    if (on) { g_prof_pending.clear(); g_prof_done.clear(); }
    return PPO_OK;
}

int32_t ppo_profile_get(const char* kernel_name, double* total_ms, int64_t* launches) {
    PPO_TRY(ensure_init());
    HIP_TRY(hipStreamSynchronize(g_stream));
    for (auto& kv : g_prof_pending) {
        auto& acc = g_prof_done[kv.first];
        for (auto& r : kv.second) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { acc.first += ms; acc.second += 1; }
            (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
        }
        kv.second.clear();
    }
    auto it = g_prof_done.find(kernel_name ? kernel_name : "");
    if (total_ms) *total_ms = it == g_prof_done.end() ? 0.0 : it->second.first;
    if (launches) *launches = it == g_prof_done.end() ? 0 : it->second.second;
    return PPO_OK;
}

int32_t ppo_profile_returns(int64_t T, int64_t N, double discount, int32_t iters, double* avg_ms) {
    PPO_TRY(ensure_init());
    ARG_CHECK(T >= 1 && N >= 1 && iters >= 1 && avg_ms, "profile_returns: bad argument");
    const size_t n = (size_t)T * N;
    DevBuf<float> r, o; DevBuf<uint8_t> d;
    PPO_TRY(r.alloc(n)); PPO_TRY(o.alloc(n)); PPO_TRY(d.alloc(n));
    HIP_TRY(hipMemsetAsync(r.p, 0x3c, n * 4, g_stream));          // ~0.0115 everywhere
    HIP_TRY(hipMemsetAsync(d.p, 0, n, g_stream));
    PPO_TRY(launch_returns_tn(r.p, d.p, o.p, T, N, discount, 0)); // warm-up
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, g_stream));
    for (int i = 0; i < iters; ++i) PPO_TRY(launch_returns_tn(r.p, d.p, o.p, T, N, discount, 0));
    HIP_TRY(hipEventRecord(e1, g_stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_ms = (double)ms / iters;
    return PPO_OK;
}

int32_t ppo_profile_gae(int64_t T, int64_t N, double gamma, double lambda, int32_t iters, double* avg_ms) {
    PPO_TRY(ensure_init());
    ARG_CHECK(T >= 1 && N >= 1 && iters >= 1 && avg_ms, "profile_gae: bad argument");
    const size_t n = (size_t)T * N;
    DevBuf<float> r, v, a, o; DevBuf<uint8_t> d;
    PPO_TRY(r.alloc(n)); PPO_TRY(v.alloc(n + N)); PPO_TRY(a.alloc(n)); PPO_TRY(o.alloc(n)); PPO_TRY(d.alloc(n));
    HIP_TRY(hipMemsetAsync(r.p, 0x3c, n * 4, g_stream));          // ~0.0115 everywhere
    HIP_TRY(hipMemsetAsync(v.p, 0x3c, (n + N) * 4, g_stream));
    HIP_TRY(hipMemsetAsync(d.p, 0, n, g_stream));
    PPO_TRY(launch_gae_tn(r.p, d.p, v.p, a.p, o.p, T, N, gamma, lambda));   // warm-up
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, g_stream));
    for (int i = 0; i < iters; ++i) PPO_TRY(launch_gae_tn(r.p, d.p, v.p, a.p, o.p, T, N, gamma, lambda));
    HIP_TRY(hipEventRecord(e1, g_stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_ms = (double)ms / iters;
    return PPO_OK;
}

// ================================================================ standalone ops
int32_t ppo_compute_returns(const float* rewards, const uint8_t* terminal, int64_t n, double discount,
                            int32_t discount_is_f32, float* out) {
    PPO_TRY(ensure_init());
    ARG_CHECK(n >= 0 && (n == 0 || (rewards && terminal && out)), "compute_returns: null buffer");
    if (n == 0) return PPO_OK;
    DevBuf<float> r, o; DevBuf<uint8_t> t;
    PPO_TRY(r.alloc(n)); PPO_TRY(o.alloc(n)); PPO_TRY(t.alloc(n));
    PPO_TRY(h2d(r.p, rewards, n)); PPO_TRY(h2d(t.p, terminal, n));
    PPO_TRY(launch_returns_flat(r.p, t.p, o.p, n, discount, discount_is_f32));
    return d2h(out, o.p, n);
}

int32_t ppo_compute_returns_tn(const float* rewards, const uint8_t* done, int64_t T, int64_t N, double discount,
                               int32_t discount_is_f32, float* out) {
    PPO_TRY(ensure_init());
    ARG_CHECK(T >= 0 && N >= 0, "compute_returns_tn: negative size");
    const size_t n = (size_t)T * N;
    if (n == 0) return PPO_OK;
    ARG_CHECK(rewards && done && out, "compute_returns_tn: null buffer");
    DevBuf<float> r, o; DevBuf<uint8_t> t;
    PPO_TRY(r.alloc(n)); PPO_TRY(o.alloc(n)); PPO_TRY(t.alloc(n));
    PPO_TRY(h2d(r.p, rewards, n)); PPO_TRY(h2d(t.p, done, n));
    PPO_TRY(launch_returns_tn(r.p, t.p, o.p, T, N, discount, discount_is_f32));
    return d2h(out, o.p, n);
}

int32_t ppo_gae_tn(const float* rewards, const uint8_t* done, const float* values, int64_t T, int64_t N, double gamma,
                   double lambda, float* adv_out, float* ret_out) {
    PPO_TRY(ensure_init());
    const size_t n = (size_t)T * N;
    if (n == 0) return PPO_OK;
    ARG_CHECK(rewards && done && values && adv_out && ret_out, "gae_tn: null buffer");
    DevBuf<float> r, v, a, o; DevBuf<uint8_t> t;
    PPO_TRY(r.alloc(n)); PPO_TRY(v.alloc(n + N)); PPO_TRY(a.alloc(n)); PPO_TRY(o.alloc(n)); PPO_TRY(t.alloc(n));
    PPO_TRY(h2d(r.p, rewards, n)); PPO_TRY(h2d(t.p, done, n)); PPO_TRY(h2d(v.p, values, n + N));
    PPO_TRY(launch_gae_tn(r.p, t.p, v.p, a.p, o.p, T, N, gamma, lambda));
    PPO_TRY(d2h(adv_out, a.p, n));
    return d2h(ret_out, o.p, n);
}

int32_t ppo_categorical_sample(const float* probs, const float* u, int64_t B, int64_t A, int32_t* actions,
                               float* p_sel, int32_t* err) {
    PPO_TRY(ensure_init());
    ARG_CHECK(B >= 0 && A >= 1, "categorical_sample: bad shape");
    if (B == 0) return PPO_OK;
    DevBuf<float> p, uu, ps; DevBuf<int32_t> ac, er;
    PPO_TRY(p.alloc((size_t)B * A)); PPO_TRY(uu.alloc(B)); PPO_TRY(ps.alloc(B)); PPO_TRY(ac.alloc(B)); PPO_TRY(er.alloc(B));
    PPO_TRY(h2d(p.p, probs, (size_t)B * A)); PPO_TRY(h2d(uu.p, u, B));
    PPO_TRY(launch_categorical(p.p, uu.p, B, A, ac.p, ps.p, er.p));
    PPO_TRY(d2h(actions, ac.p, B)); PPO_TRY(d2h(p_sel, ps.p, B));
    return d2h(err, er.p, B);
}

int32_t ppo_linear_action_index(const int64_t* actions1, int64_t B, int64_t A, int64_t* out) {
    // src/train.jl:48-52: selected_actions + range(0, step=A, length=B).  Pure index arithmetic on
    // the host side of the ABI (it is fused into the loss kernel on the device path).
    ARG_CHECK(B >= 0 && A >= 1, "linear_action_index: bad shape");
    for (int64_t b = 0; b < B; ++b) out[b] = actions1[b] + b * A;
    return PPO_OK;
}

// ================================================================ env
int32_t ppo_env_create(int32_t kind, int64_t num_envs, int64_t global_env_offset, int32_t Q, int32_t max_actions,
                       float no_action_reward, uint64_t seed, ppo_env_t* out) {
    PPO_TRY(ensure_init());
    ARG_CHECK(out, "env_create: null out");
    ARG_CHECK(kind == 0, "env_create: only kind 0 (synthetic rand-poly-shaped env) is built in");
    ARG_CHECK(num_envs >= 1 && Q >= 2 && Q <= 32 && max_actions >= 1, "env_create: bad sizes");
    ppo_env_s* e = new ppo_env_s();
    e->kind = kind; e->Q = Q; e->H = 4 * Q; e->A = 16 * Q; e->V = 4 * Q; e->F = 2 * PPO_TPL;
    e->max_actions = max_actions; e->no_action_reward = no_action_reward; e->N = num_envs;
    e->global_offset = global_env_offset; e->seed = seed;
    const size_t N = (size_t)num_envs;
    int32_t s = PPO_OK;
    if ((s = e->score.alloc(N * e->V)) || (s = e->degree.alloc(N * e->V)) || (s = e->active.alloc(N)) ||
        (s = e->steps.alloc(N)) || (s = e->reward.alloc(N)) || (s = e->done.alloc(N)) || (s = e->episode.alloc(N)) ||
        (s = e->tick.alloc(N)) || (s = e->err.alloc(1)) || (s = e->actions_tmp.alloc(N)) ||
        (s = e->episodes_left.alloc(N)) || (s = e->tmpl.alloc((size_t)e->H * PPO_TPL))) { delete e; return s; }
    {   // the observation template depends on (half-edge, template row) only: tabulated once (1-4.5 KB, cache resident)
        std::vector<int8_t> t((size_t)e->H * PPO_TPL);
        for (int h = 0; h < e->H; ++h)
            for (int k = 0; k < PPO_TPL; ++k) t[(size_t)h * PPO_TPL + k] = (int8_t)env_template(Q, h, k);
        s = h2d(e->tmpl.p, t.data(), t.size());
        if (s) { delete e; return s; }
    }
    (void)hipMemsetAsync(e->episode.p, 0, N * 4, g_stream);
    (void)hipMemsetAsync(e->tick.p, 0, N * 4, g_stream);
    (void)hipMemsetAsync(e->err.p, 0, 4, g_stream);
    (void)hipMemsetAsync(e->done.p, 0, N, g_stream);
    (void)hipMemsetAsync(e->episodes_left.p, 0, N * 4, g_stream);
    s = launch_env_reset(e, 0);
    if (s) { delete e; return s; }
    *out = e;
    return PPO_OK;
}

int32_t ppo_env_destroy(ppo_env_t env) { if (env) { (void)hipStreamSynchronize(g_stream); delete env; } return PPO_OK; }

int32_t ppo_env_dims(ppo_env_t env, int64_t* N, int32_t* H, int32_t* F, int32_t* A) {
    ARG_CHECK(env, "env_dims: null env");
    if (N) *N = env->N; if (H) *H = env->H; if (F) *F = env->F; if (A) *A = env->A;
    return PPO_OK;
}

// -1 = auto (persistent where it is the faster form: Q = 8, wavefront-parallel env update), 0 = off, 1 = on wherever covered
static int g_rollout_persistent = [] { const char* v = std::getenv("PPO_ROLLOUT_PERSISTENT"); return (v && (v[0] == '0' || v[0] == '1')) ? v[0] - '0' : -1; }();
int32_t ppo_set_rollout_persistent(int32_t mode) { g_rollout_persistent = mode < 0 ? -1 : (mode != 0); return PPO_OK; }

// state storage of engine-collected rollouts: -1 = automatic (compact env snapshots when the expanded observations of
// the requested rollout would exceed PPO_COMPACT_AUTO_BYTES -- default 32 GiB: re-deriving the rows costs the train
// forward 3 % in fp32 and 17 % in bf16 mode, so below that the 288 GB of HBM are spent on speed -- or when a disk sink is
// attached: the stream then carries 64 + 4 instead of 2304 + 4 state bytes per env-step for Q = 8), 0 = always
// expanded, 1 = always compact
static int g_rollout_compact = [] { const char* v = std::getenv("PPO_ROLLOUT_COMPACT"); return (v && (v[0] == '0' || v[0] == '1')) ? v[0] - '0' : -1; }();
int32_t ppo_set_rollout_compact(int32_t mode) { g_rollout_compact = mode < 0 ? -1 : (mode != 0); return PPO_OK; }
static bool want_compact(const ppo_rollouts_s* ro, int64_t T) {
    if (ro->V == 0) return false;                             // created by shape: no env snapshot form
    if (g_rollout_compact >= 0) return g_rollout_compact == 1;
    if (ro->sink) return true;
    static const double limit = [] { const char* v = std::getenv("PPO_COMPACT_AUTO_BYTES"); return v ? atof(v) : 32.0 * 1024 * 1024 * 1024; }();
    return (double)T * (double)ro->N * ro->H * ro->F > limit;
}

int32_t ppo_env_set_strict_sampling(ppo_env_t env, int32_t strict) { ARG_CHECK(env, "null env"); env->strict_sampling = strict ? 1 : 0; return PPO_OK; }

int32_t ppo_env_reset(ppo_env_t env) { ARG_CHECK(env, "reset!: null env"); return launch_env_reset(env, 0); }

int32_t ppo_env_check_errors(ppo_env_t env, int32_t* flags_or_null) {
    ARG_CHECK(env, "null env");
    int32_t f = 0;
    PPO_TRY(d2h(&f, env->err.p, 1));
    if (flags_or_null) *flags_or_null = f;
    if (f & ~(env->strict_sampling ? 0 : 32)) {     // bit 32 is informational unless strict: a CDF rounding residue went to the last unmasked action
        std::string m = "AssertionError (device flag):";
        if (f & 1) m += " action on inactive quad;";
        if (f & 2) m += " action index out of range;";
        if (f & 4) m += " step! on a terminated env;";
        if (f & 8) m += " sampled action has probability 0 (ap[a] > 0.0 failed, src/collect_rollouts.jl:7);";
        if (f & 32) m += " the CDF walk ended on a masked action (ap[a] > 0.0 would fail in the reference; strict sampling);";
        ppo_set_error(m);
        return PPO_ERR_DEVICE_FLAG;
    }
    return PPO_OK;
}

int32_t ppo_env_step(ppo_env_t env, const int32_t* actions0) {
    ARG_CHECK(env && actions0, "step!: null argument");
    PPO_TRY(h2d(env->actions_tmp.p, actions0, (size_t)env->N));
    PPO_TRY(launch_env_step(env, env->actions_tmp.p, nullptr, nullptr, nullptr, 0, 0));
    return ppo_env_check_errors(env, nullptr);
}

int32_t ppo_env_get_state(ppo_env_t env, int8_t* obs, uint32_t* active) {
    ARG_CHECK(env && obs, "state: null argument");
    const size_t n = (size_t)env->N * env->H * env->F;
    PPO_TRY(env->obs_tmp.alloc(n));
    PPO_TRY(launch_env_observe(env, env->obs_tmp.p, nullptr));
    PPO_TRY(d2h(obs, env->obs_tmp.p, n));
    if (active) PPO_TRY(d2h(active, env->active.p, (size_t)env->N));
    return PPO_OK;
}

int32_t ppo_env_get_reward(ppo_env_t env, float* out) { ARG_CHECK(env && out, "reward: null"); return d2h(out, env->reward.p, (size_t)env->N); }
int32_t ppo_env_get_terminal(ppo_env_t env, uint8_t* out) { ARG_CHECK(env && out, "is_terminal: null"); return d2h(out, env->done.p, (size_t)env->N); }

int32_t ppo_env_get_internal(ppo_env_t env, int8_t* score, int8_t* degree, int32_t* steps, uint32_t* episode,
                             uint32_t* tick) {
    ARG_CHECK(env, "null env");
    const size_t N = (size_t)env->N;
    if (score) PPO_TRY(d2h(score, env->score.p, N * env->V));
    if (degree) PPO_TRY(d2h(degree, env->degree.p, N * env->V));
    if (steps) PPO_TRY(d2h(steps, env->steps.p, N));
    if (episode) PPO_TRY(d2h(episode, env->episode.p, N));
    if (tick) PPO_TRY(d2h(tick, env->tick.p, N));
    return PPO_OK;
}

// ================================================================ policy
// Hidden widths other than 128 / 256 run on the next wider kernel with ZERO-PADDED hidden units.  That is exact, not an
// approximation: a padded unit has zero input weights and zero bias, so its activation is leakyrelu(0) = 0; its outgoing
// weights are zero, so it adds fmaf(0, 0, acc) = acc to every sum it enters; and every gradient that touches it carries
// one of those zeros as a factor (dZ2[pad] = (W3^T dY)[pad] = 0, H1[pad] = 0, dH1[pad] = (W2^T dZ2)[pad] = 0), so Adam
// (m = v = 0 -> update 0 / (sqrt(0) + eps) = 0) never moves it.  The caller sees its own Policy(F, hidden, 2, 4): the flat
// Flux-order vectors that cross the ABI (parameters, gradient, Adam moments) have the caller's layout.
static __host__ __device__ inline int64_t np_of(int64_t F, int64_t hid, int64_t L = 2) {
    return hid * F + hid + (L - 1) * (hid * hid + hid) + (int64_t)PPO_OUT * hid + PPO_OUT;
}

// index of the caller's flat element i (width hu, L hidden layers) in the flat vector at width hp
__device__ __forceinline__ int64_t pad_index(int64_t i, int F, int hu, int hp, int L) {
    const int64_t uW2 = (int64_t)hu * F + hu, uper = (int64_t)hu * hu + hu, uW3 = uW2 + (L - 1) * uper;
    const int64_t pW2 = (int64_t)hp * F + hp, pper = (int64_t)hp * hp + hp, pW3 = pW2 + (L - 1) * pper;
    if (i < (int64_t)hu * F) return (i % hu) + (int64_t)hp * (i / hu);                  // W1[o][k]
    if (i < uW2) return (int64_t)hp * F + (i - (int64_t)hu * F);                        // b1
    if (i < uW3) {                                                                     // hidden->hidden layer: W[o][k], then b
        const int64_t lay = (i - uW2) / uper, e = (i - uW2) - lay * uper;
        if (e < (int64_t)hu * hu) return pW2 + lay * pper + (e % hu) + (int64_t)hp * (e / hu);
        return pW2 + lay * pper + (int64_t)hp * hp + (e - (int64_t)hu * hu);
    }
    return pW3 + (i - uW3);                                                            // W3[4][k] (k < hu), b3: b3 follows W3
}
__global__ void k_pad_copy(float* __restrict__ user, float* __restrict__ padded, int64_t n_user, int F, int hu, int hp, int L, int to_user) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_user) return;
    // b3 sits behind W3 in both layouts, but W3 is [4][hid]: the last 4 elements are b3
    const int64_t pi = (i >= n_user - PPO_OUT) ? (np_of(F, hp, L) - (n_user - i)) : pad_index(i, F, hu, hp, L);
    if (to_user) user[i] = padded[pi]; else padded[pi] = user[i];
}
// host flat vector (caller's layout) <-> device flat vector at the kernels' width
static int32_t flat_to_device(ppo_policy_s* p, const float* host, float* dev_padded) {
    if (p->hid_user == p->HID) return h2d(dev_padded, host, (size_t)p->np);
    DevBuf<float> tmp;
    PPO_TRY(tmp.alloc((size_t)p->np_user));
    PPO_TRY(h2d(tmp.p, host, (size_t)p->np_user));
    HIP_TRY(hipMemsetAsync(dev_padded, 0, (size_t)p->np * sizeof(float), g_stream));
    hipLaunchKernelGGL(k_pad_copy, dim3((unsigned)((p->np_user + 255) / 256)), dim3(256), 0, g_stream, tmp.p, dev_padded,
                       p->np_user, p->F, p->hid_user, p->HID, p->L, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g_stream));
    return PPO_OK;
}
static int32_t flat_to_host(ppo_policy_s* p, float* host, const float* dev_padded) {
    if (p->hid_user == p->HID) return d2h(host, dev_padded, (size_t)p->np);
    DevBuf<float> tmp;
    PPO_TRY(tmp.alloc((size_t)p->np_user));
    hipLaunchKernelGGL(k_pad_copy, dim3((unsigned)((p->np_user + 255) / 256)), dim3(256), 0, g_stream, tmp.p,
                       const_cast<float*>(dev_padded), p->np_user, p->F, p->hid_user, p->HID, p->L, 1);
    HIP_TRY(hipGetLastError());
    return d2h(host, tmp.p, (size_t)p->np_user);
}

int32_t ppo_policy_create(int32_t F, int32_t hidden_user, int32_t num_hidden_layers, int32_t out_per_edge,
                          ppo_policy_t* out) {
    PPO_TRY(ensure_init());
    ARG_CHECK(out, "policy_create: null out");
    const int32_t hidden = hidden_user <= 128 ? 128 : 256;                  // width of the kernels that run it
    if (num_hidden_layers < 1 || num_hidden_layers > 4 || out_per_edge != PPO_OUT || hidden_user < 1 || hidden_user > 256 ||
        !(F == 72 || F == 216)) {
        ppo_set_error("policy_create: the gfx950 kernels cover Policy(F, hidden, num_hidden_layers, 4) with F in {72, 216}, "
                      "hidden in 1..256 and num_hidden_layers in 1..4 (test/policy.jl:9-19, test/test_square_mesh.jl:29, "
                      "test/output/*.bson, BASELINE config 2); hidden widths other than 128 / 256 run zero-padded on the "
                      "next wider kernel; out must be the quad game's 4 actions per edge (test/quad_game_utilities.jl:95)");
        return PPO_ERR_UNSUPPORTED;
    }
    const int32_t NL2 = num_hidden_layers - 1;
    ppo_policy_s* p = new ppo_policy_s();
    p->F = F; p->HID = hidden; p->L = num_hidden_layers; p->OUT = out_per_edge;
    p->hid_user = hidden_user; p->np_user = np_of(F, hidden_user, num_hidden_layers);
    p->np = np_of(F, hidden, num_hidden_layers);
    int32_t s = PPO_OK;
    if ((s = p->params.alloc(p->np)) || (s = p->w1p.alloc((size_t)hidden * F + PPO_PACK_PAD)) ||
        (s = p->w2p.alloc((size_t)NL2 * hidden * hidden + PPO_PACK_PAD)) ||
        (s = p->w2tp.alloc((size_t)NL2 * hidden * hidden + PPO_PACK_PAD)) || (s = p->b1p.alloc(hidden)) ||
        (s = p->b2p.alloc((size_t)(NL2 > 0 ? NL2 : 1) * hidden)) ||
        (s = p->w3p.alloc((size_t)hidden * PPO_OUT)) || (s = p->b3.alloc(PPO_OUT)) || (s = p->grad.alloc(p->np + 2)) ||
        (s = p->err.alloc(1))) { delete p; return s; }
    if (num_hidden_layers == 2 && F == 72) {             // + 16 KiB: the operand ring of the last k-steps reads ahead (up to 12 KiB)
        if ((s = p->w2x.alloc((size_t)3 * hidden * hidden + 8192))) { delete p; return s; }
        (void)hipMemsetAsync(p->w2x.p, 0, p->w2x.n * 2, g_stream);
        if ((s = p->w2fx.alloc((size_t)3 * hidden * hidden + 8192)) || (s = p->w1x.alloc((size_t)(hidden / 32) * 5 * 3 * 512 + 4096))) { delete p; return s; }
        (void)hipMemsetAsync(p->w2fx.p, 0, p->w2fx.n * 2, g_stream);
        (void)hipMemsetAsync(p->w1x.p, 0, p->w1x.n * 2, g_stream);       // inputs 72 .. 79 of the last k-step stay zero
    }
    (void)hipMemsetAsync(p->params.p, 0, p->np * 4, g_stream);
    (void)hipMemsetAsync(p->w1p.p, 0, p->w1p.n * 4, g_stream);
    (void)hipMemsetAsync(p->w2p.p, 0, p->w2p.n * 4, g_stream);
    (void)hipMemsetAsync(p->w2tp.p, 0, p->w2tp.n * 4, g_stream);
    (void)hipMemsetAsync(p->grad.p, 0, (p->np + 2) * 4, g_stream);
    (void)hipMemsetAsync(p->err.p, 0, 4, g_stream);
    s = launch_pack_params(p);
    if (s) { delete p; return s; }
    *out = p;
    return PPO_OK;
}

int32_t ppo_policy_set_dtype(ppo_policy_t pol, int32_t dtype) {
    ARG_CHECK(pol, "policy_set_dtype: null policy");
    ARG_CHECK(dtype == PPO_DTYPE_F32 || dtype == PPO_DTYPE_BF16, "policy_set_dtype: dtype must be PPO_DTYPE_F32 or PPO_DTYPE_BF16");
    if (dtype == PPO_DTYPE_BF16 && (pol->L != 2 || pol->F != 72)) {
        ppo_set_error("policy_set_dtype: the bf16 kernels cover Policy(72, hidden, 2, 4) only");
        return PPO_ERR_UNSUPPORTED;
    }
    if (dtype == PPO_DTYPE_BF16 && !pol->w1b.p) {
        const size_t HID = pol->HID, KS1 = bf16_ks1(pol->F);
        PPO_TRY(pol->w1b.alloc(HID * KS1 * 16)); PPO_TRY(pol->w2b.alloc(HID * HID)); PPO_TRY(pol->w2tb.alloc(HID * HID));
        PPO_TRY(pol->w3c.alloc(HID * PPO_OUT)); PPO_TRY(pol->w3tb.alloc(HID * PPO_OUT));
        HIP_TRY(hipMemsetAsync(pol->w1b.p, 0, pol->w1b.n * 2, g_stream));      // k-slots >= F stay zero
    }
    pol->dtype = dtype;
    return launch_pack_params(pol);
}
int32_t ppo_policy_get_dtype(ppo_policy_t pol, int32_t* dtype) { ARG_CHECK(pol && dtype, "null"); *dtype = pol->dtype; return PPO_OK; }

int32_t ppo_policy_destroy(ppo_policy_t pol) { if (pol) { (void)hipStreamSynchronize(g_stream); delete pol; } return PPO_OK; }
int32_t ppo_policy_num_params(ppo_policy_t pol, int64_t* n) { ARG_CHECK(pol && n, "null"); *n = pol->np_user; return PPO_OK; }

int32_t ppo_policy_set_params(ppo_policy_t pol, const float* flat) {
    ARG_CHECK(pol && flat, "set_params: null");
    PPO_TRY(flat_to_device(pol, flat, pol->params.p));
    return launch_pack_params(pol);
}
int32_t ppo_policy_get_params(ppo_policy_t pol, float* flat) { ARG_CHECK(pol && flat, "get_params: null"); return flat_to_host(pol, flat, pol->params.p); }
int32_t ppo_policy_get_grad(ppo_policy_t pol, float* flat) { ARG_CHECK(pol && flat, "get_grad: null"); return flat_to_host(pol, flat, pol->grad.p); }
int32_t ppo_policy_grad_buffer_dev(ppo_policy_t pol, void** dev_ptr, int64_t* n_floats) {
    ARG_CHECK(pol && dev_ptr && n_floats, "grad_buffer_dev: null");
    *dev_ptr = pol->grad.p; *n_floats = pol->np + 2;
    return PPO_OK;
}

int32_t ppo_policy_forward(ppo_policy_t pol, const int8_t* states, const uint32_t* active, int64_t B, int32_t H,
                           float* probs) {
    ARG_CHECK(pol && states && active && probs, "batch_action_probabilities: null argument");
    ARG_CHECK(B >= 1, "batch_action_probabilities: empty batch");
    if (H != 32 && H != 128) { ppo_set_error("policy_forward: H must be 32 (Q=8) or 128 (Q=32) half-edges in this build"); return PPO_ERR_UNSUPPORTED; }
    DevBuf<int8_t> s; DevBuf<uint32_t> a; DevBuf<float> p;
    const size_t ns = (size_t)B * H * pol->F;
    PPO_TRY(s.alloc(ns)); PPO_TRY(a.alloc(B)); PPO_TRY(p.alloc((size_t)B * H * 4));
    PPO_TRY(h2d(s.p, states, ns)); PPO_TRY(h2d(a.p, active, (size_t)B));
    PPO_TRY(launch_policy_probs(pol, s.p, a.p, B, H, p.p));
    return d2h(probs, p.p, (size_t)B * H * 4);
}

// ================================================================ optimiser
int32_t ppo_adam_create(ppo_policy_t pol, double eta, double beta1, double beta2, double eps, ppo_adam_t* out) {
    ARG_CHECK(pol && out, "adam_create: null");
    ppo_adam_s* o = new ppo_adam_s();
    o->pol = pol; o->eta = eta; o->beta1 = beta1; o->beta2 = beta2; o->eps = eps;
    o->beta_pow[0] = beta1; o->beta_pow[1] = beta2;
    int32_t s;
    if ((s = o->m.alloc(pol->np)) || (s = o->v.alloc(pol->np))) { delete o; return s; }
    (void)hipMemsetAsync(o->m.p, 0, pol->np * 4, g_stream);
    (void)hipMemsetAsync(o->v.p, 0, pol->np * 4, g_stream);
    *out = o;
    return PPO_OK;
}
int32_t ppo_adam_destroy(ppo_adam_t opt) { if (opt) { (void)hipStreamSynchronize(g_stream); delete opt; } return PPO_OK; }
int32_t ppo_adam_get_lr(ppo_adam_t opt, double* eta) { ARG_CHECK(opt && eta, "null"); *eta = opt->eta; return PPO_OK; }
int32_t ppo_adam_set_lr(ppo_adam_t opt, double eta) { ARG_CHECK(opt, "null"); opt->eta = eta; return PPO_OK; }
int32_t ppo_adam_get_state(ppo_adam_t opt, float* m, float* v, double* bp) {
    ARG_CHECK(opt, "null");
    if (m) PPO_TRY(flat_to_host(opt->pol, m, opt->m.p));
    if (v) PPO_TRY(flat_to_host(opt->pol, v, opt->v.p));
    if (bp) { bp[0] = opt->beta_pow[0]; bp[1] = opt->beta_pow[1]; }
    return PPO_OK;
}
int32_t ppo_adam_set_state(ppo_adam_t opt, const float* m, const float* v, const double* bp) {
    ARG_CHECK(opt, "null");
    if (m) PPO_TRY(flat_to_device(opt->pol, m, opt->m.p));
    if (v) PPO_TRY(flat_to_device(opt->pol, v, opt->v.p));
    if (bp) { opt->beta_pow[0] = bp[0]; opt->beta_pow[1] = bp[1]; }
    return PPO_OK;
}

int32_t ppo_adam_get_epoch_count(ppo_adam_t opt, int64_t* epochs) { ARG_CHECK(opt && epochs, "null"); *epochs = opt->epochs_done; return PPO_OK; }
int32_t ppo_adam_set_epoch_count(ppo_adam_t opt, int64_t epochs) { ARG_CHECK(opt && epochs >= 0, "bad epoch count"); opt->epochs_done = epochs; return PPO_OK; }

// ================================================================ rollouts
// capacity for T steps in the requested state-storage form (the other form's buffer is released: the two differ by
// 36x in size and a buffer switches form only when the caller changes ppo_set_rollout_compact between collections)
int32_t rollouts_reserve(ppo_rollouts_s* r, int64_t T, bool compact) {
    const int64_t cap = std::max(T, r->capT);
    const size_t n = (size_t)cap * r->N;
    if (compact) { PPO_TRY(r->cstate.alloc(n * 2 * r->V)); r->states.release(); }
    else { PPO_TRY(r->states.alloc(n * r->H * r->F)); r->cstate.release(); }
    r->compact = compact;
    if (T <= r->capT) return PPO_OK;
    PPO_TRY(r->active.alloc(n)); PPO_TRY(r->actions.alloc(n));
    PPO_TRY(r->p_sel.alloc(n)); PPO_TRY(r->rewards.alloc(n)); PPO_TRY(r->returns.alloc(n)); PPO_TRY(r->done.alloc(n));
    PPO_TRY(r->valid.alloc(n)); PPO_TRY(r->index.alloc(n));
    r->capT = T;
    return PPO_OK;
}

int32_t ppo_rollouts_create(ppo_env_t env, int64_t capacity_T, ppo_rollouts_t* out) {
    PPO_TRY(ensure_init());
    ARG_CHECK(env && out && capacity_T >= 0, "BufferRollouts: bad argument");
    ppo_rollouts_s* r = new ppo_rollouts_s();
    r->N = env->N; r->H = env->H; r->F = env->F; r->A = env->A; r->V = env->V; r->capT = 0; r->T = 0; r->len = 0;
    int32_t s = r->tmpl.alloc((size_t)env->H * PPO_TPL);
    if (!s && hipMemcpyAsync(r->tmpl.p, env->tmpl.p, (size_t)env->H * PPO_TPL, hipMemcpyDeviceToDevice, g_stream) != hipSuccess) s = PPO_ERR_HIP;
    if (!s) s = rollouts_reserve(r, capacity_T, want_compact(r, capacity_T));
    if (s) { delete r; return s; }
    *out = r;
    return PPO_OK;
}
// BufferRollouts for states that do not come from the built-in env: any feature count the policy kernels take (F = 72 or
// 216 int8 features per half-edge row), H = 32 or 128 rows.  Host-supplied columns only (ppo_rollouts_set): the
// expanded storage form, no env template.
int32_t ppo_rollouts_create_shape(int64_t num_envs, int32_t H, int32_t F, int64_t capacity_T, ppo_rollouts_t* out) {
    PPO_TRY(ensure_init());
    ARG_CHECK(out && num_envs >= 1 && capacity_T >= 0, "BufferRollouts: bad argument");
    ARG_CHECK((H == 32 || H == 128) && F >= 8 && F % 8 == 0, "BufferRollouts: H must be 32 or 128 rows, F a multiple of 8");
    ppo_rollouts_s* r = new ppo_rollouts_s();
    r->N = num_envs; r->H = H; r->F = F; r->A = 4 * H; r->V = 0; r->capT = 0; r->T = 0; r->len = 0;
    const int32_t s = rollouts_reserve(r, capacity_T, false);
    if (s) { delete r; return s; }
    *out = r;
    return PPO_OK;
}
int32_t ppo_rollouts_destroy(ppo_rollouts_t ro) { if (ro) { (void)hipStreamSynchronize(g_stream); delete ro; } return PPO_OK; }
int32_t ppo_rollouts_len(ppo_rollouts_t ro, int64_t* n) { ARG_CHECK(ro && n, "length: null"); *n = ro->len; return PPO_OK; }
int32_t ppo_rollouts_dims(ppo_rollouts_t ro, int64_t* T, int64_t* N) {
    ARG_CHECK(ro, "null"); if (T) *T = ro->T; if (N) *N = ro->N; return PPO_OK;
}

__global__ void k_iota(int32_t* p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (int32_t)i;
}

int32_t set_index_all(ppo_rollouts_s* r) {
    const int64_t n = r->T * r->N;
    r->len = n; r->all_valid = true;
    if (n == 0) return PPO_OK;
    hipLaunchKernelGGL(k_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_stream, r->index.p, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(r->valid.p, 1, (size_t)n, g_stream));
    return PPO_OK;
}

static int32_t check_shapes(ppo_rollouts_s* ro, ppo_env_s* env, ppo_policy_s* pol) {
    ARG_CHECK(ro && env && pol, "collect_rollouts!: null argument");
    ARG_CHECK(ro->N == env->N && ro->H == env->H && ro->F == env->F, "collect_rollouts!: rollouts were created for another env shape");
    ARG_CHECK(pol->F == env->F, "collect_rollouts!: policy input width != env feature count");
    if (env->H != 32 && env->H != 128) { ppo_set_error("collect_rollouts!: H must be 32 (Q=8) or 128 (Q=32) half-edges in this build"); return PPO_ERR_UNSUPPORTED; }
    return PPO_OK;
}

int32_t ppo_collect_rollouts(ppo_rollouts_t ro, ppo_env_t env, ppo_policy_t pol, int64_t T, double discount,
                             int32_t discount_is_f32, int32_t record_probs) {
    PPO_TRY(check_shapes(ro, env, pol));
    ARG_CHECK(T >= 1, "collect_rollouts!: T must be >= 1");
    const bool compact = want_compact(ro, T);
    PPO_TRY(rollouts_reserve(ro, T, compact));
    const int64_t N = env->N;
    const size_t srow = (size_t)N * env->H * env->F, crow = (size_t)N * 2 * env->V;
    if (compact) PPO_TRY(env->obs_tmp.alloc(srow));           // per-step launches: the observation of one step only
    if (record_probs) PPO_TRY(ro->full_probs.alloc((size_t)T * N * env->A));
    // an env left terminal by a previous call starts a fresh episode (reset! before each episode)
    PPO_TRY(launch_env_reset(env, 1));
    // One launch for the whole rollout (every wave walks its envs through all T steps: k_policy_fwd MODE 3).  Default: on
    // for Q = 8, where the env update is wavefront-parallel (312.7 vs 314.6 ms per bench iteration, same box); off for
    // Q = 32, whose update still runs on one lane.  ppo_set_rollout_persistent / PPO_ROLLOUT_PERSISTENT=0|1 override.
    // With a disk sink attached the rollout is a CHAIN of such launches, a few steps each: the finished steps of launch k
    // are copied device -> pinned host on the copy stream while launch k + 1 runs.
    const bool persistent_ok = g_rollout_persistent == 1 || (g_rollout_persistent < 0 && env->Q == 8);
    int32_t ps = PPO_ERR_UNSUPPORTED;
    PPO_TRY(disk_sink_begin(ro, T));
    if (persistent_ok && !ro->sink) ps = launch_policy_rollout_persistent(pol, env, ro, T, record_probs);
    else if (persistent_ok) {
        const int64_t chunk = disk_sink_chunk(ro);                         // half the ring in flight (at most 8 steps), the rest draining
        for (int64_t t0 = 0; t0 < T; t0 += chunk) {
            const int64_t tc = std::min(chunk, T - t0);
            ps = launch_policy_rollout_persistent(pol, env, ro, tc, record_probs, t0);
            if (ps == PPO_ERR_UNSUPPORTED && t0 > 0) {
                // the per-step fallback restarts at t = 0, but the envs have advanced t0 steps and their records are
                // already enqueued to the sink: a shape that stops being covered mid-chain is a hard error
                ppo_set_error("collect_rollouts!: the one-launch rollout became unavailable in the middle of a streamed collection");
                return PPO_ERR_UNSUPPORTED;
            }
            if (ps != PPO_OK) break;                                       // t0 == 0: shape not covered -> per-step launches
            for (int64_t t = t0; t < t0 + tc; ++t) PPO_TRY(disk_sink_step(ro, t));
        }
    }
    if (ps != PPO_OK && ps != PPO_ERR_UNSUPPORTED) return ps;
    if (ps == PPO_ERR_UNSUPPORTED) {
    for (int64_t t = 0; t < T; ++t) {
        int8_t* st = compact ? env->obs_tmp.p : ro->states.p + (size_t)t * srow;
        uint32_t* am = ro->active.p + (size_t)t * N;
        PPO_TRY(launch_env_observe(env, st, am));                                               // state(env)
        if (compact) PPO_TRY(launch_env_snapshot(env, ro->cstate.p + (size_t)t * crow));       // what is kept of it
        PPO_TRY(launch_policy_rollout(pol, env, st, am, ro->actions.p + t * N, ro->p_sel.p + t * N,
                                      record_probs ? ro->full_probs.p + (size_t)t * N * env->A : nullptr));
        PPO_TRY(launch_env_step(env, ro->actions.p + t * N, ro->rewards.p + t * N, ro->done.p + t * N, nullptr, 1, 0));
        PPO_TRY(disk_sink_step(ro, t));                 // out-of-core store: async D2H of step t on the copy stream
    }
    }
    ro->T = T; ro->adv_T = -1;
    PPO_TRY(set_index_all(ro));
    // compute_state_value!: returns overwrite the rewards column (src/rollout_buffer.jl:55-64)
    PPO_TRY(launch_returns_tn(ro->rewards.p, ro->done.p, ro->returns.p, T, N, discount, discount_is_f32));
    PPO_TRY(disk_sink_finish(ro));
    return ppo_env_check_errors(env, nullptr);
}

// episode e of the call is played by env e mod N: env n plays episodes n, n + N, n + 2N, ... below num_episodes
__global__ void k_episode_quota(int32_t* left, int64_t N, int64_t num_episodes) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) left[n] = n < num_episodes ? (int32_t)((num_episodes - n + N - 1) / N) : 0;
}

int32_t ppo_collect_rollouts_episodes(ppo_rollouts_t ro, ppo_env_t env, ppo_policy_t pol, int64_t num_episodes,
                                      double discount, int32_t discount_is_f32) {
    PPO_TRY(check_shapes(ro, env, pol));
    ARG_CHECK(num_episodes >= 1, "collect_rollouts!: num_episodes must be >= 1");
    const int64_t N = env->N;
    const int64_t episodes_per_env = (num_episodes + N - 1) / N;       // the busiest env (env 0)
    const int64_t Tmax = episodes_per_env * env->max_actions;
    const bool compact = want_compact(ro, Tmax);
    PPO_TRY(rollouts_reserve(ro, Tmax, compact));
    const size_t srow = (size_t)N * env->H * env->F, crow = (size_t)N * 2 * env->V;
    if (compact) PPO_TRY(env->obs_tmp.alloc(srow));
    hipLaunchKernelGGL(k_episode_quota, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, g_stream, env->episodes_left.p, N,
                       num_episodes);
    PPO_TRY(launch_env_reset(env, 0));                                  // reset!(env) before the first episode
    std::vector<int32_t> left((size_t)N);
    int64_t T = 0;
    for (int64_t t = 0; t < Tmax; ++t) {
        int8_t* st = compact ? env->obs_tmp.p : ro->states.p + (size_t)t * srow;
        uint32_t* am = ro->active.p + (size_t)t * N;
        PPO_TRY(launch_env_observe(env, st, am));
        if (compact) PPO_TRY(launch_env_snapshot(env, ro->cstate.p + (size_t)t * crow));
        PPO_TRY(launch_policy_rollout(pol, env, st, am, ro->actions.p + t * N, ro->p_sel.p + t * N, nullptr));
        PPO_TRY(launch_env_step(env, ro->actions.p + t * N, ro->rewards.p + t * N, ro->done.p + t * N,
                                ro->valid.p + t * N, 0, 1));
        T = t + 1;
        if ((t & 7) == 7 || t + 1 == Tmax) {            // poll completion every 8 steps
            PPO_TRY(d2h(left.data(), env->episodes_left.p, (size_t)N));
            bool all = true;
            for (int64_t n = 0; n < N; ++n) if (left[n] > 0) { all = false; break; }
            if (all) break;
        }
    }
    ro->T = T; ro->adv_T = -1;
    // dataset order = env-major concatenation of whole episodes (the reference's flat buffer)
    std::vector<uint8_t> valid((size_t)T * N);
    PPO_TRY(d2h(valid.data(), ro->valid.p, (size_t)T * N));
    std::vector<int32_t> index;
    index.reserve((size_t)T * N);
    for (int64_t n = 0; n < N; ++n)
        for (int64_t t = 0; t < T; ++t)
            if (valid[(size_t)t * N + n]) index.push_back((int32_t)(t * N + n));
    ro->len = (int64_t)index.size();
    ro->all_valid = false;
    PPO_TRY(h2d(ro->index.p, index.data(), index.size()));
    PPO_TRY(launch_returns_tn(ro->rewards.p, ro->done.p, ro->returns.p, T, N, discount, discount_is_f32));
    return ppo_env_check_errors(env, nullptr);
}

// average_returns (src/evaluate.jl:18-25): undiscounted return of each whole episode, then Flux.mean / Flux.std
int32_t ppo_average_returns(ppo_policy_t pol, ppo_env_t env, ppo_rollouts_t scratch, int64_t num_trajectories,
                            double* mean, double* std) {
    ARG_CHECK(pol && env && scratch && mean && std && num_trajectories >= 1, "average_returns: bad argument");
    const int64_t N = env->N;
    const int64_t per_env = (num_trajectories + N - 1) / N;
    PPO_TRY(ppo_collect_rollouts_episodes(scratch, env, pol, num_trajectories, 1.0, 0));   // exactly num_trajectories (:19-22)
    const int64_t T = scratch->T;
    std::vector<float> r((size_t)T * N);
    std::vector<uint8_t> dn((size_t)T * N), va((size_t)T * N);
    PPO_TRY(d2h(r.data(), scratch->rewards.p, r.size()));
    PPO_TRY(d2h(dn.data(), scratch->done.p, dn.size()));
    PPO_TRY(d2h(va.data(), scratch->valid.p, va.size()));
    std::vector<double> rets;
    rets.reserve((size_t)per_env * N);
    for (int64_t n = 0; n < N; ++n) {
        double acc = 0.0;
        for (int64_t t = 0; t < T; ++t) {
            const size_t i = (size_t)t * N + n;
            if (!va[i]) continue;
            acc += (double)r[i];                         // ret += reward(env)   src/evaluate.jl:13
            if (dn[i]) { rets.push_back(acc); acc = 0.0; }
        }
    }
    ARG_CHECK(!rets.empty(), "average_returns: no complete episode");
    double m = 0.0;
    for (double x : rets) m += x;
    m /= (double)rets.size();
    double v = 0.0;
    for (double x : rets) v += (x - m) * (x - m);
    *mean = m;
    *std = rets.size() > 1 ? std::sqrt(v / (double)(rets.size() - 1)) : NAN;   // Flux.std: corrected (n-1)
    return PPO_OK;
}

// Evaluator variants on the device (test/quad_game_utilities.jl:280-307,369-387; kind 1 = src/evaluate.jl:1-16): exactly
// num_trajectories whole episodes, trajectory e on resident env e mod N like ppo_collect_rollouts_episodes; the
// per-episode value is tracked inside the env step (k_env_step, EvalView), nothing but one value per trajectory
// comes back.  Row 0 of `scratch` is the only rollout storage touched.  values: [num_trajectories], env-major.
static int32_t evaluate_impl(ppo_policy_s* pol, ppo_env_s* env, ppo_rollouts_s* scratch, int64_t num_traj, int32_t kind,
                             std::vector<double>& values) {
    PPO_TRY(check_shapes(scratch, env, pol));
    ARG_CHECK(num_traj >= 1, "evaluator: num_trajectories must be >= 1");
    ARG_CHECK(kind >= 1 && kind <= 3, "evaluator: kind must be 1 (return), 2 (best return) or 3 (normalised best return)");
    const int64_t N = env->N;
    const int64_t per_env = (num_traj + N - 1) / N;
    const int64_t Tmax = per_env * ((int64_t)env->max_actions + 1);      // + 1: a skipped (maxreturn == 0) episode costs one step
    PPO_TRY(rollouts_reserve(scratch, 1, false));
    DevBuf<double> ep_ret, out; DevBuf<int32_t> ep_i;
    PPO_TRY(ep_ret.alloc((size_t)N)); PPO_TRY(out.alloc((size_t)num_traj)); PPO_TRY(ep_i.alloc((size_t)4 * N));
    HIP_TRY(hipMemsetAsync(ep_ret.p, 0, (size_t)N * 8, g_stream));
    HIP_TRY(hipMemsetAsync(ep_i.p, 0, (size_t)4 * N * 4, g_stream));
    HIP_TRY(hipMemsetAsync(out.p, 0, (size_t)num_traj * 8, g_stream));
    EvalView ev;
    ev.kind = kind; ev.ep_ret = ep_ret.p; ev.ep_init = ep_i.p; ev.ep_min = ep_i.p + N; ev.ep_maxret = ep_i.p + 2 * N;
    ev.ep_count = ep_i.p + 3 * N; ev.out = out.p; ev.num_traj = num_traj;
    hipLaunchKernelGGL(k_episode_quota, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, g_stream, env->episodes_left.p, N,
                       num_traj);
    PPO_TRY(launch_env_reset(env, 0));                                  // reset!(env) before the first trajectory
    std::vector<int32_t> left((size_t)N);
    bool all = false;
    for (int64_t t = 0; t < Tmax && !all; ++t) {
        PPO_TRY(launch_env_observe(env, scratch->states.p, scratch->active.p));
        PPO_TRY(launch_policy_rollout(pol, env, scratch->states.p, scratch->active.p, scratch->actions.p, scratch->p_sel.p, nullptr));
        PPO_TRY(launch_env_step(env, scratch->actions.p, scratch->rewards.p, scratch->done.p, scratch->valid.p, 0, 1, &ev));
        if ((t & 7) == 7 || t + 1 == Tmax) {            // poll completion every 8 steps
            PPO_TRY(d2h(left.data(), env->episodes_left.p, (size_t)N));
            all = true;
            for (int64_t n = 0; n < N; ++n) if (left[n] > 0) { all = false; break; }
        }
    }
    ARG_CHECK(all, "evaluator: the episodes did not finish within max_actions steps each");
    scratch->T = 0; scratch->len = 0; scratch->adv_T = -1;
    values.resize((size_t)num_traj);
    PPO_TRY(d2h(values.data(), out.p, (size_t)num_traj));
    return ppo_env_check_errors(env, nullptr);
}

static void mean_std(const std::vector<double>& x, double* mean, double* std) {
    double m = 0.0;
    for (double v : x) m += v;
    m /= (double)x.size();
    double s = 0.0;
    for (double v : x) s += (v - m) * (v - m);
    *mean = m;
    *std = x.size() > 1 ? std::sqrt(s / (double)(x.size() - 1)) : NAN;   // Flux.std: corrected (n-1)
}

int32_t ppo_evaluate_trajectories(ppo_policy_t pol, ppo_env_t env, ppo_rollouts_t scratch, int64_t num_trajectories,
                                  int32_t kind, double* values) {
    ARG_CHECK(pol && env && scratch && values, "evaluator: null argument");
    std::vector<double> v;
    PPO_TRY(evaluate_impl(pol, env, scratch, num_trajectories, kind, v));
    std::memcpy(values, v.data(), v.size() * sizeof(double));
    return PPO_OK;
}

// average_best_returns(wrapper, policy, num_trajectories)         test/quad_game_utilities.jl:299-307
int32_t ppo_average_best_returns(ppo_policy_t pol, ppo_env_t env, ppo_rollouts_t scratch, int64_t num_trajectories,
                                 double* mean, double* std) {
    ARG_CHECK(pol && env && scratch && mean && std, "average_best_returns: null argument");
    std::vector<double> v;
    PPO_TRY(evaluate_impl(pol, env, scratch, num_trajectories, 2, v));
    mean_std(v, mean, std);
    return PPO_OK;
}

// average_normalized_returns(wrapper, policy, num_trajectories)   test/quad_game_utilities.jl:380-387
int32_t ppo_average_normalized_returns(ppo_policy_t pol, ppo_env_t env, ppo_rollouts_t scratch, int64_t num_trajectories,
                                       double* mean, double* std) {
    ARG_CHECK(pol && env && scratch && mean && std, "average_normalized_returns: null argument");
    std::vector<double> v;
    PPO_TRY(evaluate_impl(pol, env, scratch, num_trajectories, 3, v));
    mean_std(v, mean, std);
    return PPO_OK;
}

#define RO_GETTER(name, member, type, per)                                                     \
    int32_t name(ppo_rollouts_t ro, type* out) {                                               \
        ARG_CHECK(ro && out, #name ": null");                                                  \
        return d2h(out, ro->member.p, (size_t)ro->T * ro->N * (per));                          \
    }
RO_GETTER(ppo_rollouts_get_actions, actions, int32_t, 1)
RO_GETTER(ppo_rollouts_get_probs, p_sel, float, 1)
RO_GETTER(ppo_rollouts_get_returns, returns, float, 1)
RO_GETTER(ppo_rollouts_get_raw_rewards, rewards, float, 1)
RO_GETTER(ppo_rollouts_get_terminal, done, uint8_t, 1)
RO_GETTER(ppo_rollouts_get_valid, valid, uint8_t, 1)

int32_t ppo_rollouts_get_states(ppo_rollouts_t ro, int8_t* states, uint32_t* active) {
    ARG_CHECK(ro, "null");
    if (states && !ro->compact) PPO_TRY(d2h(states, ro->states.p, (size_t)ro->T * ro->N * ro->H * ro->F));
    if (states && ro->compact) {       // compact form: the observations are re-derived on the device, <= 256 MiB at a time
        const size_t per = (size_t)ro->H * ro->F, total = (size_t)ro->T * ro->N;
        const size_t chunk = std::max<size_t>(1, std::min(total, ((size_t)256 << 20) / per));
        PPO_TRY(ro->expand_tmp.alloc(chunk * per));
        for (size_t o = 0; o < total; o += chunk) {
            const size_t c = std::min(chunk, total - o);
            PPO_TRY(launch_expand_states(ro->cstate.p + o * 2 * ro->V, ro->active.p + o, ro->tmpl.p, (int64_t)c, ro->V / 4, ro->expand_tmp.p));
            PPO_TRY(d2h(states + o * per, ro->expand_tmp.p, c * per));
        }
    }
    if (active) PPO_TRY(d2h(active, ro->active.p, (size_t)ro->T * ro->N));
    return PPO_OK;
}
int32_t ppo_rollouts_get_full_probs(ppo_rollouts_t ro, float* probs) {
    ARG_CHECK(ro && probs, "null");
    ARG_CHECK(ro->full_probs.p && ro->full_probs.n >= (size_t)ro->T * ro->N * ro->A, "full probabilities were not recorded");
    return d2h(probs, ro->full_probs.p, (size_t)ro->T * ro->N * ro->A);
}
int32_t ppo_rollouts_get_index(ppo_rollouts_t ro, int64_t* idx) {
    ARG_CHECK(ro && idx, "null");
    std::vector<int32_t> tmp((size_t)ro->len);
    PPO_TRY(d2h(tmp.data(), ro->index.p, tmp.size()));
    for (size_t i = 0; i < tmp.size(); ++i) idx[i] = tmp[i];
    return PPO_OK;
}

int32_t ppo_rollouts_set(ppo_rollouts_t ro, int64_t T, const int8_t* states, const uint32_t* active,
                         const int32_t* actions0, const float* p_sel, const float* returns, const uint8_t* terminal) {
    ARG_CHECK(ro && T >= 1 && states && active && actions0 && p_sel && returns, "rollouts_set: bad argument");
    PPO_TRY(rollouts_reserve(ro, T, false));                  // host-supplied observations: always the expanded form
    const size_t n = (size_t)T * ro->N;
    PPO_TRY(h2d(ro->states.p, states, n * ro->H * ro->F)); PPO_TRY(h2d(ro->active.p, active, n));
    PPO_TRY(h2d(ro->actions.p, actions0, n)); PPO_TRY(h2d(ro->p_sel.p, p_sel, n)); PPO_TRY(h2d(ro->returns.p, returns, n));
    PPO_TRY(h2d(ro->rewards.p, returns, n));
    if (terminal) PPO_TRY(h2d(ro->done.p, terminal, n));
    ro->T = T; ro->adv_T = -1;
    return set_index_all(ro);
}

// batch_advantage plugin as GAE(gamma, lambda): the state values come from the host (the reference has no value head;
// a user's critic supplies them), the scan runs on the device over the buffer's raw rewards / terminal flags.
int32_t ppo_rollouts_compute_gae(ppo_rollouts_t ro, const float* values, double gamma, double lambda, float* adv_out,
                                 float* lambda_returns_out) {
    ARG_CHECK(ro && values, "compute_gae: null argument");
    ARG_CHECK(ro->T >= 1, "compute_gae: empty rollout buffer");
    const size_t n = (size_t)ro->T * ro->N;
    PPO_TRY(ro->values.alloc(n + ro->N)); PPO_TRY(ro->adv.alloc((size_t)ro->capT * ro->N)); PPO_TRY(ro->lam_ret.alloc((size_t)ro->capT * ro->N));
    PPO_TRY(h2d(ro->values.p, values, n + ro->N));
    PPO_TRY(launch_gae_tn(ro->rewards.p, ro->done.p, ro->values.p, ro->adv.p, ro->lam_ret.p, ro->T, ro->N, gamma, lambda));
    ro->adv_T = ro->T;
    if (adv_out) PPO_TRY(d2h(adv_out, ro->adv.p, n));
    if (lambda_returns_out) PPO_TRY(d2h(lambda_returns_out, ro->lam_ret.p, n));
    return PPO_OK;
}

// minibatches up to this many 32-row tiles take the three-product backward (ppo_policy_bwd_small.hip).  Measured on
// MI355X (HID = 256): 7.6 % faster per PPO iteration at 256 tiles, level with the fused kernel at
// 512, 6 % slower at 1024 -- so the default switches between the two at 384.  PPO_BWD_SMALL_MAX_TILES overrides (0 = never).
static int64_t g_bwd_small_max_tiles = [] { const char* v = std::getenv("PPO_BWD_SMALL_MAX_TILES"); return v ? (int64_t)atoll(v) : (int64_t)384; }();

int32_t ppo_set_bwd_small_max_tiles(int64_t tiles) { g_bwd_small_max_tiles = tiles < 0 ? 384 : tiles; return PPO_OK; }

// minibatches up to this many tiles run forward + loss + backward-data of each tile in one workgroup (k_policy_train_tile)
// and the weight gradients as a split-K product on operand-layout tiles.  PPO_TRAIN_TILE_MAX_TILES overrides (0 = never).
#ifndef PPO_TRAIN_TILE_DEFAULT
#define PPO_TRAIN_TILE_DEFAULT 0
#endif
static int64_t g_train_tile_max_tiles = [] { const char* v = std::getenv("PPO_TRAIN_TILE_MAX_TILES"); return v ? (int64_t)atoll(v) : (int64_t)PPO_TRAIN_TILE_DEFAULT; }();
int32_t ppo_set_train_tile_max_tiles(int64_t tiles) { g_train_tile_max_tiles = tiles < 0 ? PPO_TRAIN_TILE_DEFAULT : tiles; return PPO_OK; }

// fused backward: weight-gradient products as split-fp32 ("bf16x6") MFMAs (ppo_policy_bwd_x6.hip).  PPO_BWD_SPLIT_BF16 overrides.
#ifndef PPO_BWD_SPLIT_DEFAULT
#define PPO_BWD_SPLIT_DEFAULT 1
#endif
static int bwd_split_default() { const char* v = std::getenv("PPO_BWD_SPLIT_BF16"); return v ? (atoi(v) != 0) : PPO_BWD_SPLIT_DEFAULT; }
static int g_bwd_split = bwd_split_default();
int ppo_bwd_split_enabled() { return g_bwd_split; }
int32_t ppo_set_bwd_split_bf16(int32_t mode) { g_bwd_split = mode < 0 ? bwd_split_default() : (mode != 0); return PPO_OK; }

// ================================================================ training
// B = number of 32-row tiles of the minibatch (states * H/32)
static int32_t train_reserve(ppo_policy_s* p, int64_t B, bool compact = false) {
    if (compact) PPO_TRY(p->xs.alloc((size_t)std::max(B, p->cap_tiles) * 32 * p->F));
    if (B <= p->cap_tiles) return PPO_OK;
    const size_t NT = p->HID / 32;
    PPO_TRY(p->act1.alloc((size_t)B * NT * 1024));
    if (p->L >= 2) PPO_TRY(p->act2.alloc((size_t)B * NT * 1024));
    if (p->L > 2) PPO_TRY(p->actm.alloc((size_t)(p->L - 2) * B * NT * 1024));       // hidden layers between the first and the last
    PPO_TRY(p->dY.alloc((size_t)B * 128)); PPO_TRY(p->loss_terms.alloc((size_t)B * 2));
    // one gradient slab per backward workgroup: 256, or 512 where two workgroups share a CU (fp32 HID = 128, F = 72)
    PPO_TRY(p->slabs.alloc((size_t)((p->HID == 128 && p->F == 72) ? 512 : 256) * slab_floats(p->F, p->HID, p->L)));
    PPO_TRY(p->idx.alloc((size_t)B));
    p->cap_tiles = B;
    return PPO_OK;
}

// idx_dev: transition ids (already resolved through the dataset index)
static int32_t forward_backward_dev(ppo_policy_s* pol, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B,
                                    int64_t B_global, double eps, double ew, int32_t adv_mode, ppo_adam_s* fuse_opt = nullptr,
                                    float* fuse_hist2 = nullptr) {
    PPO_TRY(train_reserve(pol, B * (ro->H / 32), ro->compact));
    const float* adv = ro->returns.p;                       // batch_advantage = returns (reference-equivalent)
    if (adv_mode == PPO_ADV_GAE || adv_mode == PPO_ADV_GAE_NORMALISED) {
        ARG_CHECK(ro->adv.p && ro->adv_T == ro->T, "batch_advantage: GAE mode needs ppo_rollouts_compute_gae on these rollouts first");
        adv = ro->adv.p;
    }
    if (adv_mode == PPO_ADV_RETURNS_NORMALISED || adv_mode == PPO_ADV_GAE_NORMALISED) {   // normalised over this rank's minibatch
        PPO_TRY(pol->adv_col.alloc((size_t)ro->capT * ro->N));
        PPO_TRY(launch_adv_normalise(adv, idx_dev, B, pol->adv_col.p));
        adv = pol->adv_col.p;
    }
    // small minibatches: the whole training pass of a tile on one CU (ppo_policy_train_tile.hip), then the weight gradients
    if (B * (ro->H / 32) <= g_train_tile_max_tiles && pol->L == 2 && pol->dtype == PPO_DTYPE_F32) {
        const size_t frag = (size_t)pol->cap_tiles * (pol->HID / 32) * 1024;
        PPO_TRY(pol->dz1f.alloc(frag)); PPO_TRY(pol->dz2f.alloc(frag));
        const int32_t ts = launch_policy_train_tile(pol, ro, idx_dev, B, B_global, eps, ew, adv);
        if (ts != PPO_ERR_UNSUPPORTED) {
            if (ts != PPO_OK) return ts;
            PPO_TRY(launch_grad_reduce(pol, B, B_global, ew, fuse_opt, fuse_hist2));
            pol->last_B = B; pol->last_entropy_weight = ew;
            return PPO_OK;
        }
    }
    PPO_TRY(launch_policy_train_fwd(pol, ro, idx_dev, B, B_global, eps, ew, adv));
    // small minibatches: three-product backward (no per-workgroup gradient slabs); otherwise the fused kernel
    int32_t bs = PPO_ERR_UNSUPPORTED;
    // the fused kernel is the num_hidden_layers == 2 shape with all its weight gradients resident (and F = 216 at HID = 256
    // does not fit its LDS): every other policy takes the layer-looped three-product form at any minibatch size
    const bool fused_ok = pol->L == 2 && !(pol->F == 216 && pol->HID == 256);
    // with the split-fp32 fused backward (ppo_policy_bwd_x6.hip) the three-product form no longer wins at any size (measured,
    // gpurun_out/small1: 128 / 256 / 384 tiles 30.3 / 35.6 / 49.4 ms per iteration against 28.4 / 35.5 / 43.8): it is taken only
    // below PPO_BWD_SMALL_MAX_TILES_SPLIT (default 0) while the split form is on and covers the policy
    static const int64_t small_max_split = [] { const char* v = std::getenv("PPO_BWD_SMALL_MAX_TILES_SPLIT"); return v ? (int64_t)atoll(v) : (int64_t)0; }();
    const bool split_covers = ppo_bwd_split_enabled() && fused_ok && pol->F == 72 && pol->w2x.p != nullptr;
    const int64_t small_max = split_covers ? small_max_split : g_bwd_small_max_tiles;
    if ((B * (ro->H / 32) <= small_max || !fused_ok) && pol->dtype == PPO_DTYPE_F32) {
        const size_t frag = (size_t)pol->cap_tiles * (pol->HID / 32) * 1024;     // dZ in fragment order, like act1 / act2
        PPO_TRY(pol->dz1f.alloc(frag));
        if (pol->L >= 2) PPO_TRY(pol->dz2f.alloc(frag));
        if (pol->L > 2) PPO_TRY(pol->dzm.alloc((size_t)(pol->L - 2) * frag));
        bs = launch_policy_bwd_small(pol, ro, idx_dev, B);
        if (bs == PPO_ERR_UNSUPPORTED && !fused_ok) { ppo_set_error("step_batch!: no backward kernel for this policy / state shape"); return bs; }
    }
    if (bs != PPO_OK && bs != PPO_ERR_UNSUPPORTED) return bs;
    if (bs == PPO_ERR_UNSUPPORTED) PPO_TRY(launch_policy_bwd(pol, ro, idx_dev, B));
    PPO_TRY(launch_grad_reduce(pol, B, B_global, ew, fuse_opt, fuse_hist2));
    pol->last_B = B; pol->last_entropy_weight = ew;
    return PPO_OK;
}

__global__ void k_gather_index(const int32_t* __restrict__ index, const int64_t* __restrict__ pos, int64_t B,
                               int64_t len, int32_t* __restrict__ out, int32_t* __restrict__ err) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const int64_t p = pos[i];
    if (p < 0 || p >= len) { atomicOr(err, 16); out[i] = index[0]; return; }
    out[i] = index[p];
}

int32_t ppo_forward_backward(ppo_policy_t pol, ppo_rollouts_t ro, const int64_t* sample_idx, int64_t B,
                             int64_t B_global, double epsilon, double entropy_weight, int32_t adv_mode) {
    ARG_CHECK(pol && ro && sample_idx, "step_batch!: null argument");
    ARG_CHECK(B >= 1 && B <= ro->len, "step_batch!: 1 <= batch_size <= num_data (src/train.jl:88)");
    ARG_CHECK(B_global >= B, "step_batch!: B_global < B");
    ARG_CHECK(pol->F == ro->F && (ro->H == 32 || ro->H == 128), "step_batch!: shape mismatch");
    if (adv_mode < PPO_ADV_RETURNS || adv_mode > PPO_ADV_GAE_NORMALISED) { ppo_set_error("batch_advantage: unknown advantage mode"); return PPO_ERR_UNSUPPORTED; }
    for (int64_t i = 0; i < B; ++i) ARG_CHECK(sample_idx[i] >= 0 && sample_idx[i] < ro->len, "dataset index out of range (src/rollout_buffer.jl:105-106)");
    PPO_TRY(train_reserve(pol, B * (ro->H / 32)));
    DevBuf<int64_t> pos;
    PPO_TRY(pos.alloc(B));
    PPO_TRY(h2d(pos.p, sample_idx, (size_t)B));
    hipLaunchKernelGGL(k_gather_index, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, g_stream, ro->index.p, pos.p, B,
                       ro->len, pol->idx.p, pol->err.p);
    HIP_TRY(hipGetLastError());
    PPO_TRY(forward_backward_dev(pol, ro, pol->idx.p, B, B_global, epsilon, entropy_weight, adv_mode));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return PPO_OK;
}

int32_t ppo_adam_apply(ppo_adam_t opt, ppo_policy_t pol) {
    ARG_CHECK(opt && pol && opt->pol == pol, "update!: optimiser was created for another policy");
    return launch_adam(opt, nullptr);
}

int32_t ppo_last_losses(ppo_policy_t pol, double* ppoloss, double* entropyloss) {
    ARG_CHECK(pol, "null");
    float t[2];
    PPO_TRY(d2h(t, pol->grad.p + pol->np, 2));
    if (ppoloss) *ppoloss = t[0];
    if (entropyloss) *entropyloss = t[1];
    return PPO_OK;
}

int32_t ppo_step_batch(ppo_policy_t pol, ppo_adam_t opt, ppo_rollouts_t ro, const int64_t* sample_idx, int64_t B,
                       double epsilon, double entropy_weight, int32_t adv_mode, double* ppoloss, double* entropyloss) {
    PPO_TRY(ppo_forward_backward(pol, ro, sample_idx, B, B, epsilon, entropy_weight, adv_mode));
    PPO_TRY(ppo_last_losses(pol, ppoloss, entropyloss));
    return ppo_adam_apply(opt, pol);
}

__global__ void k_perm_index(const int32_t* __restrict__ index, const int64_t* __restrict__ perm, int64_t len,
                             int32_t* __restrict__ out, int32_t* __restrict__ err) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const int64_t p = perm[i];
    if (p < 0 || p >= len) { atomicOr(err, 16); out[i] = index[0]; return; }
    out[i] = index[p];
}

int32_t ppo_train(ppo_policy_t pol, ppo_adam_t opt, ppo_rollouts_t ro, double epsilon, int64_t batch_size,
                  int32_t num_epochs, double entropy_weight, int32_t adv_mode, const int64_t* perm, uint64_t seed,
                  int32_t rank, int32_t world, ppo_allreduce_fn allreduce, void* allreduce_ctx, double* ppo_hist,
                  double* entropy_hist, double* lr_hist) {
    ARG_CHECK(pol && opt && ro && opt->pol == pol, "ppo_train!: null/mismatched argument");
    const int64_t len = ro->len;
    ARG_CHECK(num_epochs >= 0 && world >= 1 && rank >= 0 && rank < world, "ppo_train!: bad epochs/rank/world");
    ARG_CHECK(world == 1 || allreduce, "ppo_train!: world > 1 needs an all-reduce hook");
    ARG_CHECK(pol->F == ro->F && (ro->H == 32 || ro->H == 128), "ppo_train!: shape mismatch");
    ARG_CHECK(batch_size >= 1, "1 <= batch_size <= num_data (src/train.jl:88)");
    if (adv_mode < PPO_ADV_RETURNS || adv_mode > PPO_ADV_GAE_NORMALISED) { ppo_set_error("batch_advantage: unknown advantage mode"); return PPO_ERR_UNSUPPORTED; }
    // Data-parallel shards may differ in length (remainder envs, episode-mode rollouts): every rank learns every
    // rank's dataset length once per call -- an all-gather spelled as the hook's sum all-reduce over a one-hot
    // [2*world] vector (length split into two exactly representable floats) -- and derives from them the SAME
    // number of optimiser steps and the exact global minibatch size of each step.  A rank whose shard is exhausted
    // contributes a zero gradient to the remaining steps, so all ranks issue the same collectives.
    std::vector<int64_t> lens((size_t)world, len);
    if (world > 1) {
        DevBuf<float> xch;
        PPO_TRY(xch.alloc((size_t)2 * world));
        std::vector<float> hx((size_t)2 * world, 0.0f);
        hx[2 * (size_t)rank] = (float)(len >> 12); hx[2 * (size_t)rank + 1] = (float)(len & 4095);
        PPO_TRY(h2d(xch.p, hx.data(), hx.size()));
        if (allreduce(allreduce_ctx, xch.p, 2 * (int64_t)world) != 0) { ppo_set_error("all-reduce hook failed (shard-length exchange)"); return PPO_ERR_ARG; }
        // a rank that fails locally from here on must not leave the others blocked in their next collective: every rank
        // carries its status into ONE more tiny all-reduce and all of them return together
        int32_t local = d2h(hx.data(), xch.p, hx.size());
        if (local == PPO_OK) {
            for (int32_t r = 0; r < world; ++r) lens[(size_t)r] = ((int64_t)hx[2 * (size_t)r] << 12) + (int64_t)hx[2 * (size_t)r + 1];
            if (lens[(size_t)rank] != len) {
                ppo_set_error("AssertionError: ppo_train!: shard-length exchange is inconsistent (two ranks with the same rank id, "
                              "or a hook that does not SUM the buffer it is handed?)");
                local = PPO_ERR_ARG;
            }
        }
        float bad = local == PPO_OK ? 0.0f : 1.0f;
        const int32_t hs = h2d(xch.p, &bad, 1);
        if (allreduce(allreduce_ctx, xch.p, 1) != 0) { ppo_set_error("all-reduce hook failed (status agreement)"); return PPO_ERR_ARG; }
        PPO_TRY(hs);
        PPO_TRY(d2h(&bad, xch.p, 1));
        if (local != PPO_OK) return local;
        if (bad != 0.0f) { ppo_set_error("ppo_train!: another data-parallel rank failed in the shard-length exchange"); return PPO_ERR_ARG; }
    }
    int64_t nb = 0, min_len = len;
    for (int64_t l : lens) { nb = std::max(nb, (l + batch_size - 1) / batch_size); min_len = std::min(min_len, l); }
    ARG_CHECK(batch_size <= min_len, "1 <= batch_size <= num_data (src/train.jl:88) on every data-parallel shard");
    PPO_TRY(train_reserve(pol, batch_size * (ro->H / 32)));
    DevBuf<int32_t> order; DevBuf<int64_t> permd; DevBuf<float> hist;
    PPO_TRY(order.alloc(len));
    if (perm) PPO_TRY(permd.alloc(len));
    PPO_TRY(hist.alloc((size_t)nb * 2));
    std::vector<float> hh((size_t)nb * 2);
    for (int32_t ep = 0; ep < num_epochs; ++ep) {
        if (perm) {                                                    // randperm(num_data)  src/train.jl:93
            PPO_TRY(h2d(permd.p, perm + (size_t)ep * len, (size_t)len));
            hipLaunchKernelGGL(k_perm_index, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, g_stream, ro->index.p,
                               permd.p, len, order.p, pol->err.p);
            HIP_TRY(hipGetLastError());
        } else {
            // keyed by (seed, epochs this optimiser has trained): no process-global state, so the same seed with a
            // fresh optimiser reproduces the run and a restored optimiser (ppo_adam_set_epoch_count) resumes it
            PPO_TRY(launch_feistel_index(ro->index.p, len, seed, (uint32_t)opt->epochs_done, order.p));
        }
        for (int64_t b = 0; b < nb; ++b) {                                          // :95-96 (last batch may be short)
            const int64_t start = b * batch_size;
            const int64_t B = std::max<int64_t>(0, std::min(batch_size, len - start));
            int64_t Bg = 0;
            for (int64_t l : lens) Bg += std::max<int64_t>(0, std::min(batch_size, l - start));
            // single-rank training: Adam and the re-pack ride in the slab-reduction launch (PPO_FUSE_REDUCE_ADAM=0: separate launches)
            static const bool fuse_ok = [] { const char* v = std::getenv("PPO_FUSE_REDUCE_ADAM"); return v ? atoi(v) != 0 : true; }();
            const bool fused = fuse_ok && !allreduce && B > 0;
            if (B > 0) PPO_TRY(forward_backward_dev(pol, ro, order.p + start, B, Bg, epsilon, entropy_weight, adv_mode, fused ? opt : nullptr, hist.p + 2 * b));
            else HIP_TRY(hipMemsetAsync(pol->grad.p, 0, (size_t)(pol->np + 2) * sizeof(float), g_stream));   // shard exhausted
            if (allreduce) {                     // every rank of a data-parallel run; a world of 1 may pass it too
                ProfScope ps("allreduce");
                const int32_t s = allreduce(allreduce_ctx, pol->grad.p, pol->np + 2);
                if (s != 0) { ppo_set_error("all-reduce hook failed"); return PPO_ERR_ARG; }
            }
            if (!fused) PPO_TRY(launch_adam(opt, hist.p + 2 * b));                  // Flux.update!  :81 (+ loss history)
        }
        opt->epochs_done += 1;
        PPO_TRY(d2h(hh.data(), hist.p, (size_t)nb * 2));
        double sp = 0.0, se = 0.0;
        for (int64_t i = 0; i < nb; ++i) { sp += hh[2 * i]; se += hh[2 * i + 1]; }
        if (ppo_hist) ppo_hist[ep] = sp / (double)nb;                               // unweighted mean over batches :127
        if (entropy_hist) entropy_hist[ep] = se / (double)nb;
        if (lr_hist) lr_hist[ep] = opt->eta;                                        // :144,155-158
    }
    int32_t f = 0;
    PPO_TRY(d2h(&f, pol->err.p, 1));
    if (f) { ppo_set_error("AssertionError (device flag): permutation / dataset index out of range"); return PPO_ERR_DEVICE_FLAG; }
    return PPO_OK;
}

}  // extern "C"
