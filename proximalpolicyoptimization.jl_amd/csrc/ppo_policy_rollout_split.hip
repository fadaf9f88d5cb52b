// ppo_policy_rollout_split.hip -- the one-launch rollout (collect_step_data! looped over T steps, src/collect_rollouts.jl:
// 1-24; k_policy_fwd MODE 3) for FEW envs: S = 2 or 4 waves of a workgroup share an env.
//
// MODE 3 gives every env to one wave, so 512 envs use half of the chip's 1024 SIMDs and the reference's own regime (one
// env, BASELINE config 1) a thousandth of it, 47 us of dependent MFMAs per step (HID = 256).  Here the matrix work of a
// step is split over the S waves exactly as in k_policy_fwd_train_split -- wave s computes feature tiles [s NT/S, (s+1)
// NT/S) of layer 1, the tiles meet in LDS, every wave computes its output tiles of layer 2 -- but the results stay BIT
// FOR BIT those of the one-wave kernel (the parity tests compare both bit for bit): every 32x32 tile is still produced
// by one wave with the same MFMA sequence, and the layer-3 dot products, one fmaf chain over the tiles in the one-wave
// kernel, are CHAINED through the waves: wave s continues the four running sums wave s-1 left in LDS over its own tiles
// in the same order.  The last wave of the group then holds exactly the one-wave kernel's logits and runs its tail:
// masked softmax, sequential CDF walk, step!, reward, is_terminal, reset! (the env state lives in the group's LDS slot).
#include "ppo_policy_tail.h"
#include "ppo_env_device.h"

// ENV = 1: the persistent rollout (env state in LDS, T steps); ENV = 0: one step of the per-step form (k_policy_fwd MODE 1:
// rows, active-quad word and tick come from global arrays, the env is stepped by k_env_step afterwards) -- what
// collect_rollouts!(.., num_episodes, ..) and the evaluator run, usually on few envs
template <int F, int HID, int S, int ENV>
__global__ __launch_bounds__(256, 1) void k_rollout_split(FwdArgs a) {
    constexpr int NT = HID / 32, NTS = NT / S, G = 4 / S;
    constexpr int S41 = F / 8, S42 = NT * 4, XB = F / 2, XW = XB / 4;
    constexpr int PF = 8;
    static_assert(NT % S == 0 && (S == 2 || S == 4) && XW == 9, "shape (the built-in env has F = 72 features)");
    static_assert(PF * 64 * 4 <= PPO_PACK_PAD && S42 % PF == 0, "ring");
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wv / S, sw = wv % S;
    __shared__ __attribute__((aligned(16))) float4 sW3[2 * NT * 16];
    __shared__ __attribute__((aligned(16))) float4 sB1[NT * 2 * 4];
    __shared__ __attribute__((aligned(16))) float4 sB2[NT * 2 * 4];
    __shared__ __attribute__((aligned(16))) float4 sP[G * 64];                // running layer-3 sums handed wave to wave
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    float4* const sH = reinterpret_cast<float4*>(dyn_lds);                    // [G][NT][4][64]: layer-1 tiles of the groups' envs
    char* const env_lds = dyn_lds + (size_t)G * NT * 4 * 64 * sizeof(float4); // [G][env_slots] env state
    for (int i = threadIdx.x; i < 2 * NT * 16; i += 256) sW3[i] = a.w3p[i];
    for (int i = threadIdx.x; i < NT * 8; i += 256) { sB1[i] = a.b1p[i]; sB2[i] = a.b2p[i]; }
    const int slot_bytes = 2 * a.envV + 32;
    char* const my_slots = env_lds + (size_t)grp * a.env_slots * slot_bytes;
    auto slot_ref = [&](int slot) {
        PPO_LDS char* b = (PPO_LDS char*)(my_slots + (size_t)slot * slot_bytes);
        EnvRefLds r;
        r.sc = (PPO_LDS int8_t*)b; r.dg = r.sc + a.envV;
        PPO_LDS uint32_t* w = (PPO_LDS uint32_t*)(b + 2 * a.envV);
        r.active = w; r.steps = (PPO_LDS int32_t*)(w + 1); r.reward = (PPO_LDS float*)(w + 2);
        r.done = (PPO_LDS uint8_t*)(w + 3); r.episode = w + 4; r.tick = w + 5;
        return r;
    };
    EnvConst ec;
    ec.Q = a.envQ; ec.V = a.envV; ec.max_actions = a.env_max_actions; ec.no_action_reward = a.env_nar; ec.k0 = a.k0; ec.k1 = a.k1;
    const int64_t gid = (int64_t)blockIdx.x * G + grp, ngroups = (int64_t)gridDim.x * G;
    if (ENV && sw == S - 1) {                                    // the tail wave owns the group's env slots
        int slot = 0;
        for (int64_t n = gid; n < a.B; n += ngroups, ++slot) {
            const EnvRefLds r = slot_ref(slot);
            for (int v = lane; v < a.envV; v += 64) { r.sc[v] = a.env_score[n * a.envV + v]; r.dg[v] = a.env_degree[n * a.envV + v]; }
            if (lane == 0) {
                *r.active = a.env_active[n]; *r.steps = a.env_steps[n]; *r.reward = a.env_reward[n];
                *(PPO_LDS uint32_t*)r.done = a.env_done[n]; *r.episode = a.env_episode[n]; *r.tick = a.env_tick[n];
            }
        }
    }
    uint32_t tmpl_regs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (ENV) {
        const uint32_t* tp = reinterpret_cast<const uint32_t*>(a.env_tmpl + j * PPO_TPL);
#pragma unroll
        for (int k = 0; k < 9; ++k) tmpl_regs[k] = tp[k];
    }
    __syncthreads();
    const int64_t iters = (a.B + ngroups - 1) / ngroups;        // env slots per group: the same loop bounds for every wave
    const int64_t t_steps = ENV ? a.T : 1;
    for (int64_t tt = 0; tt < t_steps; ++tt) {
        for (int64_t it = 0; it < iters; ++it) {
            const int64_t state = it * ngroups + gid;
            const bool live = state < a.B;                         // uniform within the group
            EnvRefLds er = {};
            if (ENV) er = slot_ref((int)it);
            const int64_t out_index = ENV ? tt * a.B + state : state;
            int lane_o = lane, half_o = h;
            asm volatile("" : "+v"(lane_o), "+v"(half_o));         // per-step opaque offsets (see k_policy_fwd)
            uint32_t act = 0u, tick_val = 0u;
            if (live) {
                uint32_t ob[9];
                if (ENV) {
                    act = *er.active; tick_val = *er.tick;
                    // ---- state(env): every wave of the group needs the rows (layer-1 B operands); the tail wave records them
                    env_observe_lane(er, tmpl_regs, j, h, ob);
                } else {
                    act = a.active[state]; tick_val = a.tick[state];
                    const uint32_t* xr = reinterpret_cast<const uint32_t*>(a.states + (size_t)state * 32 * F + (size_t)j * F + (size_t)h * XB);
#pragma unroll
                    for (int k = 0; k < XW; ++k) ob[k] = xr[k];
                }
                if (ENV && sw == S - 1) {
                    if (a.states_out) {
                        uint32_t* so = reinterpret_cast<uint32_t*>(a.states_out + (size_t)out_index * 32 * F + (size_t)j * F + (size_t)h * XB);
#pragma unroll
                        for (int k = 0; k < XW; ++k) so[k] = ob[k];
                    }
                    if (a.cstate_out && lane < (a.envV >> 1))
                        reinterpret_cast<uint32_t*>(a.cstate_out)[(size_t)out_index * (a.envV >> 1) + lane] = reinterpret_cast<PPO_LDS uint32_t*>(er.sc)[lane];
                }
                float xf[XB];
#pragma unroll
                for (int k = 0; k < XW; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) xf[4 * k + i] = (float)(int)(int8_t)(ob[k] >> (8 * i));
                // ---- layer 1, this wave's feature tiles -> LDS
                const float4* wp = a.w1p + (size_t)(sw * NTS) * S41 * 64 + lane_o;
                float4 ring[PF];
#pragma unroll
                for (int g = 0; g < PF; ++g) ring[g] = wp[(size_t)g * 64];
#pragma unroll
                for (int oo = 0; oo < NTS; ++oo) {
                    const int o = sw * NTS + oo;
                    f32x16 acc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 b = sB1[(o * 2 + half_o) * 4 + q];
                        acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
                    }
#pragma unroll
                    for (int s4 = 0; s4 < S41; ++s4) {
                        const int g = oo * S41 + s4;
                        const float4 w = ring[g % PF];
                        ring[g % PF] = wp[(size_t)(g + PF) * 64];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, xf[4 * s4 + 0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, xf[4 * s4 + 1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, xf[4 * s4 + 2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, xf[4 * s4 + 3], acc, 0, 0, 0);
                    }
                    asm volatile("" : "+v"(acc));
                    lrelu16(acc);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        sH[((grp * NT + o) * 4 + q) * 64 + lane] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                }
            }
            __syncthreads();
            f32x16 h2[NTS];                                         // this wave's layer-2 tiles (after leakyrelu)
            if (live) {
                f32x16 h1[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v4 = sH[((grp * NT + t) * 4 + q) * 64 + lane];
                        h1[t][4 * q] = v4.x; h1[t][4 * q + 1] = v4.y; h1[t][4 * q + 2] = v4.z; h1[t][4 * q + 3] = v4.w;
                    }
                }
                const float4* wp = a.w2p + (size_t)(sw * NTS) * S42 * 64 + lane_o;
                float4 ring[PF];
#pragma unroll
                for (int g = 0; g < PF; ++g) ring[g] = wp[(size_t)g * 64];
#pragma unroll
                for (int oo = 0; oo < NTS; ++oo) {
                    const int o = sw * NTS + oo;
                    f32x16 acc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 b = sB2[(o * 2 + half_o) * 4 + q];
                        acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
                    }
                    const float4* wo = wp + (size_t)oo * S42 * 64;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const int s4 = t * 4 + r4;
                            const float4 w = ring[s4 % PF];
                            ring[s4 % PF] = wo[(size_t)(s4 + PF) * 64];
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, h1[t][4 * r4 + 0], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, h1[t][4 * r4 + 1], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, h1[t][4 * r4 + 2], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, h1[t][4 * r4 + 3], acc, 0, 0, 0);
                        }
                    }
                    asm volatile("" : "+v"(acc));
                    lrelu16(acc);
                    h2[oo] = acc;
                }
            }
            // ---- layer 3: ONE fmaf chain over the tiles in tile order, handed from wave to wave through LDS
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (live && sw == s) {
                    if (s > 0) { const float4 q4 = sP[grp * 64 + lane]; p0 = q4.x; p1 = q4.y; p2 = q4.z; p3 = q4.w; }
#pragma unroll
                    for (int oo = 0; oo < NTS; ++oo) {
                        const float4* w3 = sW3 + (half_o * NT + (s * NTS + oo)) * 16;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float4 w = w3[r];
                            p0 = fmaf(w.x, h2[oo][r], p0); p1 = fmaf(w.y, h2[oo][r], p1);
                            p2 = fmaf(w.z, h2[oo][r], p2); p3 = fmaf(w.w, h2[oo][r], p3);
                        }
                    }
                    if (s < S - 1) sP[grp * 64 + lane] = make_float4(p0, p1, p2, p3);
                }
                if (s < S - 1) __syncthreads();
            }
            // ---- tail on the last wave of the group: exactly k_policy_fwd's epilogue + MODE 3's env update
            if (live && sw == S - 1) {
                float l[1][4];
                l[0][0] = (p0 + __shfl_xor(p0, 32)) + a.b3[0];
                l[0][1] = (p1 + __shfl_xor(p1, 32)) + a.b3[1];
                l[0][2] = (p2 + __shfl_xor(p2, 32)) + a.b3[2];
                l[0][3] = (p3 + __shfl_xor(p3, 32)) + a.b3[3];
                const int sampled = policy_tail<(ENV ? 3 : 1), 1, false>(a, state, state, act, l, lane, j, h, tick_val, out_index);
                if (ENV) {
                asm volatile("" ::: "memory");
                float rew; uint8_t dn;
                // (an opaque copy of the lane id: the env update's lane constants -- vertex, array, quad -- are then
                // recomputed here instead of being hoisted out of the step loop as registers that spill)
                int lane_e = lane;
                asm volatile("" : "+v"(lane_e));
                const int errf = env_step_wave32(ec, er, sampled, lane_e, rew, dn);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) {
                    a.active_out[out_index] = act;
                    if (errf) atomicOr(a.err, errf);
                    a.rew_out[out_index] = rew; a.done_out[out_index] = dn;
                }
                if (dn) env_reset_wave32(ec, er, (uint32_t)(a.global_offset + state), lane_e);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
            __syncthreads();                                        // the other waves see the stepped env; sH / sP are free again
        }
    }
    if (ENV && sw == S - 1) {                                       // env state back to the [N] arrays
        int slot2 = 0;
        for (int64_t n = gid; n < a.B; n += ngroups, ++slot2) {
            const EnvRefLds r = slot_ref(slot2);
            for (int v = lane; v < a.envV; v += 64) { a.env_score[n * a.envV + v] = r.sc[v]; a.env_degree[n * a.envV + v] = r.dg[v]; }
            if (lane == 0) {
                a.env_active[n] = *r.active; a.env_steps[n] = *r.steps; a.env_reward[n] = *r.reward;
                a.env_done[n] = *r.done; a.env_episode[n] = *r.episode; a.env_tick[n] = *r.tick;
            }
        }
    }
}

template <int HID, int ENV>
static int32_t launch_rs(FwdArgs& a, int64_t N, int V) {
    constexpr int NT = HID / 32;
    const int S = (N <= 256) ? 4 : 2, G = 4 / S;
    const int64_t need = (N + G - 1) / G;
    const unsigned grid = (unsigned)(need < 256 ? need : 256);
    const int slots = ENV ? (int)((N + (int64_t)grid * G - 1) / ((int64_t)grid * G)) : 0;
    a.env_slots = slots;
    const size_t lds = (size_t)G * NT * 4096 + (size_t)G * slots * (2 * V + 32);
    if (lds > 140 * 1024) return PPO_ERR_UNSUPPORTED;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_rollout_split<72, HID, 4, ENV>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void*)k_rollout_split<72, HID, 2, ENV>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        attr_set = true;
    }
    if (S == 4) hipLaunchKernelGGL((k_rollout_split<72, HID, 4, ENV>), dim3(grid), dim3(256), lds, ppo_stream(), a);
    else hipLaunchKernelGGL((k_rollout_split<72, HID, 2, ENV>), dim3(grid), dim3(256), lds, ppo_stream(), a);
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

// `a` comes filled from launch_policy_rollout_persistent (env = 1) or launch_policy_rollout (env = 0, one step).
// PPO_ERR_UNSUPPORTED (no error text): shape not covered.
int32_t launch_rollout_split(ppo_policy_s* p, FwdArgs& a, int64_t N, int tps, int V, int env) {
    if (p->dtype != PPO_DTYPE_F32 || p->F != 72 || tps != 1 || V != 32 || p->L != 2) return PPO_ERR_UNSUPPORTED;
    if (p->HID == 256) return env ? launch_rs<256, 1>(a, N, V) : launch_rs<256, 0>(a, N, V);
    if (p->HID == 128) return env ? launch_rs<128, 1>(a, N, V) : launch_rs<128, 0>(a, N, V);
    return PPO_ERR_UNSUPPORTED;
}
