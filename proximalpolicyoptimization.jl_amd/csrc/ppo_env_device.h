// ppo_env_device.h -- the synthetic rand-poly-shaped env (DESIGN.md "Synthetic env") for ONE env instance, as device
// functions on a set of pointers: the per-step kernels (ppo_env.hip) pass pointers into the [N] global arrays, the
// persistent rollout kernel (ppo_policy_fwd.hip, MODE 3) pointers into the wave's LDS slot.  One definition of the
// integer dynamics, bit for bit the oracle's orc_env_step_one / orc_env_reset_one.
#pragma once
#include "ppo_internal.h"
#include "ppo_device.h"

struct EnvConst {                 // per-launch constants
    int32_t Q, V, max_actions;
    float no_action_reward;
    uint32_t k0, k1;              // Philox key (env seed)
};
// one env's state.  Two pointer flavours: plain (the [N] arrays in global memory) and LDS (address space 3: the
// persistent rollout keeps its envs in LDS slots; generic pointers into LDS compile to flat_* accesses with a global-
// memory-like round trip, 20x slower than ds_* for this pointer-chasing integer code)
#define PPO_LDS __attribute__((address_space(3)))
struct EnvRef {
    int8_t* sc; int8_t* dg;       // [V] vertex scores / degrees
    uint32_t* active; int32_t* steps; float* reward; uint8_t* done; uint32_t* episode; uint32_t* tick;
};
struct EnvRefLds {
    PPO_LDS int8_t* sc; PPO_LDS int8_t* dg;
    PPO_LDS uint32_t* active; PPO_LDS int32_t* steps; PPO_LDS float* reward; PPO_LDS uint8_t* done;
    PPO_LDS uint32_t* episode; PPO_LDS uint32_t* tick;
};

// reset!(env): new scores from Philox(global env id, episode, 1, quad)
template <typename REF>
__device__ __forceinline__ void env_reset_ref(const EnvConst& c, const REF& r, uint32_t g) {
    const int Q = c.Q, nact = (3 * Q) / 4;
    const uint32_t ep = *r.episode;
    for (int q = 0; q < Q; ++q) {
        uint32_t w[4];
        philox4x32_10(g, ep, 1u, (uint32_t)q, c.k0, c.k1, w);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = 4 * q + i;
            if (q < nact) {
                const int s = (int)(w[i] % 5u) - 2;
                const int desired = 3 + (int)((w[i] >> 8) & 1u);
                r.sc[v] = (int8_t)s; r.dg[v] = (int8_t)(desired - s);
            } else { r.sc[v] = 0; r.dg[v] = 0; }
        }
    }
    *r.active = (nact >= 32) ? 0xFFFFFFFFu : ((1u << nact) - 1u);
    *r.steps = 0; *r.reward = 0.0f; *r.done = 0;
    *r.episode = ep + 1u;
}

__device__ __forceinline__ bool env_deg_ok(int d) { return d >= 2 && d <= 7; }
template <typename P8>
__device__ __forceinline__ int env_total_abs(P8 sc, uint32_t act, int Q) {
    int s = 0;
    for (int q = 0; q < Q; ++q) if ((act >> q) & 1u)
        for (int i = 0; i < 4; ++i) { int x = sc[4 * q + i]; s += x < 0 ? -x : x; }
    return s;
}
template <typename P8>
__device__ __forceinline__ int env_total_sum(P8 sc, uint32_t act, int Q) {
    int s = 0;
    for (int q = 0; q < Q; ++q) if ((act >> q) & 1u)
        for (int i = 0; i < 4; ++i) s += sc[4 * q + i];
    return s;
}

// step!(env, a) + reward / is_terminal (call order src/collect_rollouts.jl:9-12).  Returns the error flags to OR into
// the device flag word (1 inactive quad, 2 index out of range, 4 step! on a terminated env: then nothing else changed).
// score_after (optional): the env's current score (sum of |vertex score| over the active quads) behind the step -- what
// the evaluator variants track (env.current_score, test/quad_game_utilities.jl:280-296).
template <typename REF>
__device__ __forceinline__ int env_step_ref(const EnvConst& c, const REF& r, int a, float& rew_out, uint8_t& done_out,
                                            int* score_after = nullptr) {
    const int Q = c.Q, A = 16 * Q;
    auto sc = r.sc;
    auto dg = r.dg;
    uint32_t act = *r.active;
    int errf = 0;
    *r.tick += 1u;
    if (*r.done) { rew_out = *r.reward; done_out = 1; return 4; }
    if (a < 0 || a >= A) { errf |= 2; a = 0; }
    const int q = a / 16, ed = (a % 16) / 4, type = a % 4;
    const int old_total = env_total_abs(sc, act, Q);
    bool valid = false;
    if (!((act >> q) & 1u)) {
        errf |= 1;
    } else {
        const int v0 = 4 * q + ed, v1 = 4 * q + ((ed + 1) & 3), v2 = 4 * q + ((ed + 2) & 3), v3 = 4 * q + ((ed + 3) & 3);
        const int nq = (q + 1 + ed) % Q;
        const int w0 = 4 * nq + ed, w1 = 4 * nq + ((ed + 1) & 3);
        const bool nq_ok = (nq != q) && ((act >> nq) & 1u);
        if (type == 0 || type == 1) {
            const int p = (type == 0) ? v3 : v2, rr = (type == 0) ? w0 : w1;
            if (nq_ok && env_deg_ok(dg[v0] - 1) && env_deg_ok(dg[v1] - 1) && env_deg_ok(dg[p] + 1) && env_deg_ok(dg[rr] + 1)) {
                dg[v0]--; sc[v0]++; dg[v1]--; sc[v1]++;
                dg[p]++; sc[p]--; dg[rr]++; sc[rr]--;
                valid = true;
            }
        } else if (type == 2) {
            int f = -1;
            for (int s = 0; s < Q; ++s) if (!((act >> s) & 1u)) { f = s; break; }
            if (f >= 0 && env_deg_ok(dg[v0] + 1) && env_deg_ok(dg[v2] + 1)) {
                dg[v0]++; sc[v0]--; dg[v2]++; sc[v2]--;
                for (int i = 0; i < 4; ++i) { sc[4 * f + i] = 0; dg[4 * f + i] = 4; }
                act |= (1u << f);
                valid = true;
            }
        } else {
            const int cnt = __popc(act);
            if (nq_ok && cnt > Q / 2 && env_deg_ok(dg[w0] - 1) && env_deg_ok(dg[w1] - 1)) {
                dg[w0]--; sc[w0]++; dg[w1]--; sc[w1]++;
                for (int i = 0; i < 4; ++i) { sc[4 * q + i] = 0; dg[4 * q + i] = 0; }
                act &= ~(1u << q);
                valid = true;
            }
        }
    }
    *r.active = act;
    const int new_total = env_total_abs(sc, act, Q);
    const float rew = valid ? (float)(old_total - new_total) : c.no_action_reward;
    const int st = *r.steps + 1;
    *r.steps = st;
    const int sum = env_total_sum(sc, act, Q);
    const int opt = sum < 0 ? -sum : sum;
    const uint8_t dn = (uint8_t)((new_total == opt) || (st >= c.max_actions));
    *r.reward = rew; *r.done = dn;
    rew_out = rew; done_out = dn;
    if (score_after) *score_after = new_total;
    return errf;
}

// state(env) for one lane of the forward kernel: features [36*half, 36*half + 36) of half-edge row `row`
// (half 0: template scores, half 1: template degrees), 0 where the template entry is missing or its quad inactive
// (test/quad_game_utilities.jl:35-37,46-59).  tid: the row's 36 template vertex ids, four per dword (they depend on the
// row only: the caller keeps them in registers across states and steps).  out: 9 packed dwords.
template <typename REF>
__device__ __forceinline__ void env_observe_lane(const REF& r, const uint32_t (&tid)[9], int row, int half, uint32_t (&out)[9]) {
    const uint32_t act = *r.active;
    auto src = half ? r.dg : r.sc;
    const bool own = (act >> (row >> 2)) & 1u;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = (int)(int8_t)(tid[k] >> (8 * i));
            const bool ok = own && v >= 0 && ((act >> (v >> 2)) & 1u);
            const int8_t val = ok ? src[v < 0 ? 0 : v] : (int8_t)0;
            packed |= ((uint32_t)(uint8_t)val) << (8 * i);
        }
        out[k] = packed;
    }
}

// ---------------------------------------------------------------- wavefront-parallel forms for V == 32 (Q == 8)
// The persistent rollout runs the env update inside the forward kernel, where a thread-per-env loop over 64 state
// bytes is a chain of ~100 dependent LDS round trips on one lane.  Here the 64 lanes hold one state byte each (lane l <
// 32: score of vertex l, lane l >= 32: degree of vertex l - 32), the totals are wave reductions and the handful of
// vertices an action touches are lane compares.  Same integer results as env_step_ref / env_reset_ref / env_observe_lane.
__device__ __forceinline__ int env_wave_sum(int x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// all 64 lanes call this; returns the error flags (wave-uniform)
template <typename REF>
__device__ __forceinline__ int env_step_wave32(const EnvConst& c, const REF& r, int a, int lane, float& rew_out, uint8_t& done_out) {
    constexpr int Q = 8, A = 128;
    const bool is_sc = lane < 32;
    const int vtx = lane & 31, myq = vtx >> 2;
    int ev = (int)r.sc[lane];                         // sc[32] and dg[32] are contiguous: lanes 32..63 read dg[lane - 32]
    uint32_t act = *r.active;
    int errf = 0;
    const uint32_t tick = *r.tick + 1u;
    if (lane == 0) *r.tick = tick;
    if (*r.done) { rew_out = *r.reward; done_out = 1; return 4; }
    if (a < 0 || a >= A) { errf |= 2; a = 0; }
    const int q = a / 16, ed = (a % 16) / 4, type = a % 4;
    auto DG = [&](int v) { return __shfl(ev, 32 + v); };           // degree of vertex v (wave-uniform argument)
    const int old_total = env_wave_sum((is_sc && ((act >> myq) & 1u)) ? (ev < 0 ? -ev : ev) : 0);
    bool valid = false;
    if (!((act >> q) & 1u)) {
        errf |= 1;
    } else {
        const int v0 = 4 * q + ed, v1 = 4 * q + ((ed + 1) & 3), v2 = 4 * q + ((ed + 2) & 3), v3 = 4 * q + ((ed + 3) & 3);
        const int nq = (q + 1 + ed) % Q;
        const int w0 = 4 * nq + ed, w1 = 4 * nq + ((ed + 1) & 3);
        const bool nq_ok = (nq != q) && ((act >> nq) & 1u);
        const int up = is_sc ? -1 : 1;                 // "degree + 1, score - 1" for this lane's array; negate for the opposite move
        if (type == 0 || type == 1) {
            const int p = (type == 0) ? v3 : v2, rr = (type == 0) ? w0 : w1;
            if (nq_ok && env_deg_ok(DG(v0) - 1) && env_deg_ok(DG(v1) - 1) && env_deg_ok(DG(p) + 1) && env_deg_ok(DG(rr) + 1)) {
                ev += (vtx == v0 ? -up : 0) + (vtx == v1 ? -up : 0) + (vtx == p ? up : 0) + (vtx == rr ? up : 0);
                valid = true;
            }
        } else if (type == 2) {
            const uint32_t freeq = ~act & 0xFFu;
            const int f = freeq ? (__ffs((int)freeq) - 1) : -1;
            if (f >= 0 && env_deg_ok(DG(v0) + 1) && env_deg_ok(DG(v2) + 1)) {
                ev += (vtx == v0 ? up : 0) + (vtx == v2 ? up : 0);
                if (myq == f) ev = is_sc ? 0 : 4;
                act |= (1u << f);
                valid = true;
            }
        } else {
            const int cnt = __popc(act);
            if (nq_ok && cnt > Q / 2 && env_deg_ok(DG(w0) - 1) && env_deg_ok(DG(w1) - 1)) {
                ev += (vtx == w0 ? -up : 0) + (vtx == w1 ? -up : 0);
                if (myq == q) ev = 0;
                act &= ~(1u << q);
                valid = true;
            }
        }
    }
    r.sc[lane] = (int8_t)ev;
    const bool on = is_sc && ((act >> myq) & 1u);
    const int new_total = env_wave_sum(on ? (ev < 0 ? -ev : ev) : 0);
    const int sum = env_wave_sum(on ? ev : 0);
    const float rew = valid ? (float)(old_total - new_total) : c.no_action_reward;
    const int st = *r.steps + 1;
    const int opt = sum < 0 ? -sum : sum;
    const uint8_t dn = (uint8_t)((new_total == opt) || (st >= c.max_actions));
    if (lane == 0) { *r.active = act; *r.steps = st; *r.reward = rew; *r.done = dn; }
    rew_out = rew; done_out = dn;
    return errf;
}

// reset!(env), all 64 lanes: lane l draws the Philox word of its own vertex
template <typename REF>
__device__ __forceinline__ void env_reset_wave32(const EnvConst& c, const REF& r, uint32_t g, int lane) {
    constexpr int Q = 8, nact = (3 * Q) / 4;
    const int vtx = lane & 31, qq = vtx >> 2, i = vtx & 3;
    const uint32_t ep = *r.episode;
    uint32_t w[4];
    philox4x32_10(g, ep, 1u, (uint32_t)qq, c.k0, c.k1, w);
    const uint32_t wi = i == 0 ? w[0] : i == 1 ? w[1] : i == 2 ? w[2] : w[3];
    const int s = (int)(wi % 5u) - 2;
    const int desired = 3 + (int)((wi >> 8) & 1u);
    const int ev = (qq < nact) ? (lane < 32 ? s : desired - s) : 0;
    r.sc[lane] = (int8_t)ev;
    if (lane == 0) {
        *r.active = (1u << nact) - 1u;
        *r.steps = 0; *r.reward = 0.0f; *r.done = 0;
        *r.episode = ep + 1u;
    }
}
