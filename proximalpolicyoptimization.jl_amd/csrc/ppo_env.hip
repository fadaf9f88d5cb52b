// ppo_env.hip -- K1: batched synthetic rand-poly-shaped env (plugin contract
// src/ProximalPolicyOptimization.jl:16-20; tensor shapes test/quad_game_utilities.jl:39-59,95-110).
// QuadMeshGame's dynamics are not in the reference tree, so this is an honest synthetic stand-in
// with exactly specified integer dynamics (DESIGN.md "Synthetic env"): SoA state in HBM, one
// thread per env for the integer update, one thread per output dword for the observation.
#include "ppo_env_device.h"

struct EnvView {
    int8_t* score; int8_t* degree; uint32_t* active; int32_t* steps; float* reward; uint8_t* done;
    uint32_t* episode; uint32_t* tick; int32_t* err; int32_t* episodes_left; const int8_t* tmpl;
    int32_t Q, V, max_actions; float no_action_reward; int64_t N, global_offset; uint32_t k0, k1;
};

static EnvView view_of(ppo_env_s* e) {
    EnvView v;
    v.score = e->score.p; v.degree = e->degree.p; v.active = e->active.p; v.steps = e->steps.p;
    v.reward = e->reward.p; v.done = e->done.p; v.episode = e->episode.p; v.tick = e->tick.p; v.err = e->err.p;
    v.episodes_left = e->episodes_left.p; v.tmpl = e->tmpl.p;
    v.Q = e->Q; v.V = e->V; v.max_actions = e->max_actions; v.no_action_reward = e->no_action_reward;
    v.N = e->N; v.global_offset = e->global_offset; v.k0 = (uint32_t)e->seed; v.k1 = (uint32_t)(e->seed >> 32);
    return v;
}

static __device__ __forceinline__ EnvConst const_of(const EnvView& e) {
    EnvConst c;
    c.Q = e.Q; c.V = e.V; c.max_actions = e.max_actions; c.no_action_reward = e.no_action_reward; c.k0 = e.k0; c.k1 = e.k1;
    return c;
}
static __device__ __forceinline__ EnvRef ref_of(const EnvView& e, int64_t n) {
    EnvRef r;
    r.sc = e.score + n * e.V; r.dg = e.degree + n * e.V; r.active = e.active + n; r.steps = e.steps + n;
    r.reward = e.reward + n; r.done = e.done + n; r.episode = e.episode + n; r.tick = e.tick + n;
    return r;
}
__device__ __forceinline__ void env_reset_one(const EnvView& e, int64_t n) {
    env_reset_ref(const_of(e), ref_of(e, n), (uint32_t)(e.global_offset + n));
}

__global__ void k_env_reset(EnvView e, int only_done) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= e.N) return;
    if (only_done && !e.done[n]) return;
    env_reset_one(e, n);
}

// step!(env, a) for every env + record reward/is_terminal + optional auto-reset
// (call order src/collect_rollouts.jl:9-12; reset before the next episode src/rollout_buffer.jl:75).
// ev.kind != 0 (episodes mode only): the evaluator variants' per-episode values are tracked here, one thread per env --
// best_single_trajectory_return / single_trajectory_normalized_return (test/quad_game_utilities.jl:280-296,369-378) and
// single_trajectory_return (src/evaluate.jl:1-16).  env.current_score = sum |vertex score| over the active quads,
// env.opt_score = |sum of vertex scores| (the synthetic env's termination test uses the same two numbers).
__global__ void k_env_step(EnvView e, const int32_t* __restrict__ actions, float* __restrict__ reward_out,
                           uint8_t* __restrict__ done_out, uint8_t* __restrict__ valid_out, int auto_reset,
                           int episodes_mode, EvalView ev) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= e.N) return;
    if (episodes_mode && e.episodes_left[n] <= 0) {          // this env already played its episodes
        if (valid_out) valid_out[n] = 0;
        if (reward_out) reward_out[n] = 0.0f;
        if (done_out) done_out[n] = 1;
        return;
    }
    const EnvRef er = ref_of(e, n);
    int64_t slot = 0;
    if (ev.kind) {
        const int64_t q = ev.num_traj / e.N, rem = ev.num_traj % e.N;
        slot = n * q + (n < rem ? n : rem) + ev.ep_count[n];          // env n's episodes are consecutive in `out`
        if (*er.steps == 0) {                                        // first step! of an episode: the values at reset!
            const uint32_t act = *er.active;
            const int cur = env_total_abs(er.sc, act, e.Q), sum = env_total_sum(er.sc, act, e.Q);
            const int mr = cur - (sum < 0 ? -sum : sum);
            if (ev.kind == 3 && mr == 0) {
                // single_trajectory_normalized_return: maxreturn == 0 -> 1.0 and the episode is not played (:372-373).
                // The action sampled for this state is dropped; the next step sees the env's next episode.
                ev.out[slot] = 1.0;
                ev.ep_count[n] += 1;
                const int left = e.episodes_left[n] - 1;
                e.episodes_left[n] = left;
                if (left > 0) env_reset_one(e, n);
                if (valid_out) valid_out[n] = 0;
                if (reward_out) reward_out[n] = 0.0f;
                if (done_out) done_out[n] = 1;
                return;
            }
            ev.ep_ret[n] = 0.0; ev.ep_init[n] = cur; ev.ep_min[n] = cur; ev.ep_maxret[n] = mr;
        }
    }
    float rew; uint8_t dn;
    int score_after = 0;
    const int errf = env_step_ref(const_of(e), er, actions[n], rew, dn, &score_after);
    if (errf) atomicOr(e.err, errf);
    if (errf & 4) return;                          // step! on a terminated env: nothing recorded (as before)
    if (reward_out) reward_out[n] = rew;
    if (done_out) done_out[n] = dn;
    if (valid_out) valid_out[n] = 1;
    if (ev.kind) {
        const double ret = ev.ep_ret[n] + (double)rew;               // ret += reward(env)   src/evaluate.jl:13
        const int mn = min(ev.ep_min[n], score_after);               // minscore = min(minscore, env.current_score)
        ev.ep_ret[n] = ret; ev.ep_min[n] = mn;
        if (dn) {
            const int best = ev.ep_init[n] - mn;                     // initial_score - minscore
            ev.out[slot] = ev.kind == 1 ? ret : (ev.kind == 2 ? (double)best : (double)best / (double)ev.ep_maxret[n]);
            ev.ep_count[n] += 1;
        }
    }
    if (dn) {
        if (episodes_mode) {
            const int left = e.episodes_left[n] - 1;
            e.episodes_left[n] = left;
            if (left > 0) env_reset_one(e, n);
        } else if (auto_reset) {
            env_reset_one(e, n);
        }
    }
}

// state(env): obs[n][h][f] int8, f<36 template scores, f>=36 template degrees, 0 where the
// template entry is missing or its quad is inactive (test/quad_game_utilities.jl:35-37,46-59).
// One thread per output dword (4 features): coalesced 4-byte stores.
__global__ void k_env_observe(EnvView e, int8_t* __restrict__ obs, uint32_t* __restrict__ active_out) {
    const int H = 4 * e.Q, F = 2 * PPO_TPL, V = e.V;
    const int dw_per_env = H * F / 4;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = gid / dw_per_env;
    if (n >= e.N) return;
    const int rem = (int)(gid - n * dw_per_env);
    const int h = rem / (F / 4), f0 = (rem % (F / 4)) * 4;
    const uint32_t act = e.active[n];
    if (rem == 0 && active_out) active_out[n] = act;
    const int8_t* src = (f0 < PPO_TPL) ? (e.score + n * V) : (e.degree + n * V);
    const int t0 = (f0 < PPO_TPL) ? f0 : f0 - PPO_TPL;
    const bool own = (act >> (h >> 2)) & 1u;
    // four template vertex ids at once from the [H][36] table (36 % 4 == 0, t0 % 4 == 0: one aligned dword)
    const uint32_t ids = *reinterpret_cast<const uint32_t*>(e.tmpl + h * PPO_TPL + t0);
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = (int)(int8_t)(ids >> (8 * i));
        const bool ok = own && v >= 0 && ((act >> (v >> 2)) & 1u);
        const int8_t val = ok ? src[v] : (int8_t)0;
        packed |= ((uint32_t)(uint8_t)val) << (8 * i);
    }
    reinterpret_cast<uint32_t*>(obs)[gid] = packed;
}

// compact rollout storage: record [2V] = score[V] then degree[V] of every env (one thread per dword)
__global__ void k_env_snapshot(EnvView e, uint32_t* __restrict__ out) {
    const int dv = e.V / 4;                                    // dwords per array
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = gid / (2 * dv);
    if (n >= e.N) return;
    const int d = (int)(gid - n * 2 * dv);
    const int8_t* src = (d < dv) ? (e.score + n * e.V) : (e.degree + n * e.V);
    out[gid] = reinterpret_cast<const uint32_t*>(src)[d < dv ? d : d - dv];
}

// state(env) from stored snapshots (getters / exporter of the compact form): same arithmetic as k_env_observe with the
// score / degree bytes taken from record n of `cstate` instead of the live env arrays
// idx (optional): record n of the output is transition idx[n] (a minibatch in minibatch order) instead of record n
__global__ void k_expand_states(const int8_t* __restrict__ cstate, const uint32_t* __restrict__ active,
                                const int8_t* __restrict__ tmpl, int64_t count, int Q, int8_t* __restrict__ obs,
                                const int32_t* __restrict__ idx) {
    const int H = 4 * Q, F = 2 * PPO_TPL, V = 4 * Q;
    const int dw_per_env = H * F / 4;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = gid / dw_per_env;
    if (n >= count) return;
    const int rem = (int)(gid - n * dw_per_env);
    const int h = rem / (F / 4), f0 = (rem % (F / 4)) * 4;
    const int64_t rec = idx ? (int64_t)idx[n] : n;
    const uint32_t act = active[rec];
    const int8_t* src = cstate + rec * 2 * V + ((f0 < PPO_TPL) ? 0 : V);
    const int t0 = (f0 < PPO_TPL) ? f0 : f0 - PPO_TPL;
    const bool own = (act >> (h >> 2)) & 1u;
    const uint32_t ids = *reinterpret_cast<const uint32_t*>(tmpl + h * PPO_TPL + t0);
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = (int)(int8_t)(ids >> (8 * i));
        const bool ok = own && v >= 0 && ((act >> (v >> 2)) & 1u);
        const int8_t val = ok ? src[v] : (int8_t)0;
        packed |= ((uint32_t)(uint8_t)val) << (8 * i);
    }
    reinterpret_cast<uint32_t*>(obs)[gid] = packed;
}

int32_t launch_env_snapshot(ppo_env_s* e, int8_t* cstate_out) {
    const int64_t total = e->N * (int64_t)(e->V / 2);
    hipLaunchKernelGGL(k_env_snapshot, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ppo_stream(), view_of(e),
                       reinterpret_cast<uint32_t*>(cstate_out));
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

int32_t launch_expand_states(const int8_t* cstate, const uint32_t* active, const int8_t* tmpl, int64_t count, int32_t Q,
                             int8_t* obs_out, const int32_t* idx) {
    if (count <= 0) return PPO_OK;
    const int64_t total = count * (int64_t)(4 * Q * 2 * PPO_TPL / 4);
    hipLaunchKernelGGL(k_expand_states, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ppo_stream(), cstate, active,
                       tmpl, count, (int)Q, obs_out, idx);
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

int32_t launch_env_reset(ppo_env_s* e, int only_done) {
    dim3 grid((unsigned)((e->N + 255) / 256));
    hipLaunchKernelGGL(k_env_reset, grid, dim3(256), 0, ppo_stream(), view_of(e), only_done);
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

int32_t launch_env_step(ppo_env_s* e, const int32_t* actions_dev, float* reward_out, uint8_t* done_out,
                        uint8_t* valid_out, int auto_reset, int episodes_mode, const EvalView* ev) {
    ProfScope ps("k_env_step");
    dim3 grid((unsigned)((e->N + 63) / 64));
    hipLaunchKernelGGL(k_env_step, grid, dim3(64), 0, ppo_stream(), view_of(e), actions_dev, reward_out, done_out,
                       valid_out, auto_reset, episodes_mode, (ev && episodes_mode) ? *ev : EvalView());
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

int32_t launch_env_observe(ppo_env_s* e, int8_t* obs_out, uint32_t* active_out) {
    ProfScope ps("k_env_observe");
    const int64_t total = e->N * (int64_t)(e->H * e->F / 4);
    dim3 grid((unsigned)((total + 255) / 256));
    hipLaunchKernelGGL(k_env_observe, grid, dim3(256), 0, ppo_stream(), view_of(e), obs_out, active_out);
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}
