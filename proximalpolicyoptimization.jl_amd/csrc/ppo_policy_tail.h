// ppo_policy_tail.h -- the part of the policy forward that both compute modes (fp32 MFMA, bf16 MFMA) share:
// kernel arguments, and everything behind the 4 logits per lane of a state -- masked softmax
// (test/quad_game_utilities.jl:68-69,76-77), rand(Categorical) + ap[a] > 0 (src/collect_rollouts.jl:6-7), and the
// loss / dlogits of ppo_loss_with_entropy (src/train.jl:21-26,35-46).  All fp32 in both modes.
#pragma once
#include "ppo_internal.h"
#include "ppo_device.h"

#ifndef PPO_DPP_REDUCE
#define PPO_DPP_REDUCE 1
#endif

struct FwdArgs {
    // inputs
    const int8_t* states;      // MODE 0/1: [B][H][F]; MODE 2: rollout states base (gathered by idx)
    const uint32_t* active;    // same indexing as states
    const int32_t* idx;        // MODE 2: transition id per tile
    int64_t B;
    unsigned long long* stamps;   // diagnostic build only (-DPPO_FWD_STAMP)
    int wg_sync;               // 1: every wave of a workgroup runs the same number of tiles -> per-chunk barriers allowed
    const float4* w1p; const float4* w2p; const float4* b1p; const float4* b2p; const float4* w3p; const float* b3;
    // MODE 0
    float* probs_out;
    // MODE 1
    const uint32_t* tick; int64_t global_offset; uint32_t k0, k1;
    int32_t* actions_out; float* psel_out; float* full_probs; int32_t* err;
    // MODE 2
    float4* act1; float4* act2; float4* dY; double* loss_terms;
    const int32_t* actions; const float* p_old; const float* adv;
    double eps; float c_over_B; float inv_B;
    // MODE 3 (persistent rollout: T steps of every env in ONE launch; env state lives in the wave's LDS slots)
    int64_t T;                                                   // steps
    int8_t* env_score; int8_t* env_degree; uint32_t* env_active; int32_t* env_steps; float* env_reward;
    uint8_t* env_done; uint32_t* env_episode; uint32_t* env_tick; const int8_t* env_tmpl;
    int32_t envQ, envV, env_max_actions, env_slots; float env_nar;
    int8_t* states_out; uint32_t* active_out; float* rew_out; uint8_t* done_out;      // rollout columns [T][N]
    // compact rollout storage: an env snapshot (score[V] then degree[V], int8) per transition instead of the expanded
    // [H][F] observation.  MODE 3 writes cstate_out (states_out may then be null); MODE 4 = MODE 2 (train forward) reading
    // cstate through idx, re-deriving the rows like MODE 3 and leaving them in xs_out (minibatch order) for the backward
    int8_t* cstate_out; const int8_t* cstate; int8_t* xs_out;
    // deep form (num_hidden_layers != 2, k_policy_fwd<.., DEEP = 1>): nl2 = hidden->hidden layers (L - 1: 0, 2 or 3), w2p /
    // b2p hold them back to back; act_mid[l] = saved output of hidden->hidden layer l < nl2 - 1 (the last one goes to act2,
    // with nl2 == 0 only act1 exists); park_off = byte offset of the waves' activation park in dynamic LDS
    int32_t nl2; float4* act_mid[2]; uint32_t park_off;
    // bf16 compute mode (ppo_policy_bf16.hip): bf16 fragment streams and bf16 saved activations
    const uint4* w1b; const uint4* w2b; const uint4* w3c; uint4* act1b; uint4* act2b;
};

// value of lane (lane ^ OFF), OFF < 32.  Same lane mapping as __shfl_xor (which hipcc lowers to ds_bpermute_b32: an
// LDS round trip of ~100+ cycles per step, fully exposed in the one-or-two-waves-per-SIMD kernels here), but as DPP
// moves where the xor is a DPP pattern (1, 2: quad_perm; 8: row_ror:8; 4: row_shl:4 / row_shr:4 on alternate banks)
// and ds_swizzle (no address VGPR) for 16.  Bit-identical results: only the transport changes.
template <int OFF>
__device__ __forceinline__ float xor_lane(float v) {
#if PPO_DPP_REDUCE
    const int x = __float_as_int(v);
    if (OFF == 1) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));          // quad_perm:[1,0,3,2]
    if (OFF == 2) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));          // quad_perm:[2,3,0,1]
    if (OFF == 4) {
        const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);                            // row_shl:4 -> banks 0, 2
        return __int_as_float(__builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false));                   // row_shr:4 -> banks 1, 3
    }
    if (OFF == 8) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false));         // row_ror:8
    if (OFF == 16) return __int_as_float(__builtin_amdgcn_ds_swizzle(x, 0x401F));                            // bitmode: xor 16
#endif
    return __shfl_xor(v, OFF);
}
__device__ __forceinline__ float wave32_max(float v) {
    v = fmaxf(v, xor_lane<16>(v)); v = fmaxf(v, xor_lane<8>(v)); v = fmaxf(v, xor_lane<4>(v));
    v = fmaxf(v, xor_lane<2>(v)); v = fmaxf(v, xor_lane<1>(v));
    return v;
}
__device__ __forceinline__ float wave32_sum(float v) {
    v = v + xor_lane<16>(v); v = v + xor_lane<8>(v); v = v + xor_lane<4>(v);
    v = v + xor_lane<2>(v); v = v + xor_lane<1>(v);
    return v;
}
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}


// bf16 compute mode hands dL/dlogits to the backward MFMAs as bf16: round once here (RNE), keep the fp32 container
template <bool DYBF16>
__device__ __forceinline__ float dy_round(float x) {
    if (!DYBF16) return x;
    return (float)(__bf16)x;
}

// MODE 2 inputs of one transition, when the caller fetched them ahead of its own stores (vmcnt counts loads and stores
// in one in-order queue on gfx9: a load issued behind a store cannot be waited for without waiting for the store)
struct TailPre { int ab; float po; float adv; };

// l[ts][i]: logit of action 128*ts + 4*j + i of the state, present in both lane halves (j = lane & 31).
// MODE 2 also serves the train forward from compact states (kernel MODE 4).
// MODE 1 / 3 (rollout): tick_val = the env's tick (Philox counter word), out_index = position of the transition in
// the output columns (state for one step, t*N + state for the persistent rollout); returns the sampled action.
// dy_copy (MODE 2, TPS == 1, optional): a second destination for the state's 32 dL/dlogits rows, e.g. an LDS tile of the
// calling workgroup (k_policy_train_tile keeps the whole training pass of a tile on the CU).
template <int MODE, int TPS, bool DYBF16>
__device__ __forceinline__ int policy_tail(const FwdArgs& a, const int64_t state, const int64_t sid, const uint32_t act,
                                           float (&l)[TPS][4], const int lane, const int j, const int h,
                                           const uint32_t tick_val = 0u, const int64_t out_index = 0,
                                           const TailPre* pre = nullptr, float4* dy_copy = nullptr) {
    int sampled = 0;
    constexpr int A = 128 * TPS;
    // ---- masked softmax over the A = 128*TPS logits of the state (quad of row 32ts+j = 8ts + j/4)
    bool on[TPS];
    float m = -INFINITY;
#pragma unroll
    for (int ts = 0; ts < TPS; ++ts) {
        on[ts] = (act >> (8 * ts + (j >> 2))) & 1u;
        if (on[ts]) m = fmaxf(m, fmaxf(fmaxf(l[ts][0], l[ts][1]), fmaxf(l[ts][2], l[ts][3])));
    }
    m = wave32_max(m);
    float p[TPS][4];
    float ssum = 0.0f;
#pragma unroll
    for (int ts = 0; ts < TPS; ++ts) {
#pragma unroll
        for (int i = 0; i < 4; ++i) p[ts][i] = on[ts] ? exp_dev(l[ts][i] - m) : 0.0f;
        const float st = ((p[ts][0] + p[ts][1]) + p[ts][2]) + p[ts][3];
        ssum = (ts == 0) ? st : ssum + st;                 // tile partials in tile order, then the butterfly
    }
    ssum = wave32_sum(ssum);
#pragma unroll
    for (int ts = 0; ts < TPS; ++ts)
#pragma unroll
        for (int i = 0; i < 4; ++i) p[ts][i] = p[ts][i] / ssum;

    if (MODE == 0) {
        if (h == 0) {
#pragma unroll
            for (int ts = 0; ts < TPS; ++ts)
                reinterpret_cast<float4*>(a.probs_out)[((size_t)state * TPS + ts) * 32 + j] =
                    make_float4(p[ts][0], p[ts][1], p[ts][2], p[ts][3]);
        }
    }
    if (MODE == 1 || MODE == 3) {
        // rand(Categorical(p)): sequential fp32 inverse-CDF walk, same uniform as the oracle
        uint32_t rnd[4];
        philox4x32_10((uint32_t)(a.global_offset + state), tick_val, 0u, 0u, a.k0, a.k1, rnd);
        const float u = u01_from_u32(rnd[0]);
        float cp = readlane_f(p[0][0], 0);
        int ia = 0;
#pragma unroll
        for (int q = 1; q < A; ++q) {
            const float pa = readlane_f(p[q >> 7][q & 3], (q & 127) >> 2);
            const bool take = cp <= u;
            cp = take ? cp + pa : cp;
            ia = take ? q : ia;
        }
        float cand = 0.0f;
#pragma unroll
        for (int ts = 0; ts < TPS; ++ts)
#pragma unroll
            for (int i = 0; i < 4; ++i) cand = ((ia >> 7) == ts && (ia & 3) == i) ? p[ts][i] : cand;
        float psel = __shfl(cand, (ia & 127) >> 2);
        if (!(psel > 0.0f)) {                            // wave-uniform, about 3 in 10^8 samples
            // The walk ran off the end: the sequential fp32 sum of the probabilities fell short of u (u within a few
            // 2^-24 of 1).  Distributions.jl lets the LAST bin absorb that rounding residue; when the last action is
            // masked its probability is 0 and the reference's `@assert ap[a] > 0.0` throws -- at 10^6 samples per
            // iteration that would abort a run every few iterations.  The engine hands the residue to the last action
            // with non-zero probability instead (flag bit 32, informational); a distribution without any positive
            // entry still raises (bit 8).
            int best = -1;
#pragma unroll
            for (int ts = 0; ts < TPS; ++ts)
#pragma unroll
                for (int i = 0; i < 4; ++i) best = (p[ts][i] > 0.0f) ? 128 * ts + 4 * j + i : best;    // ascending: last wins
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) best = max(best, __shfl_xor(best, off));
            if (best >= 0) {
                ia = best;
                cand = 0.0f;
#pragma unroll
                for (int ts = 0; ts < TPS; ++ts)
#pragma unroll
                    for (int i = 0; i < 4; ++i) cand = ((ia >> 7) == ts && (ia & 3) == i) ? p[ts][i] : cand;
                psel = __shfl(cand, (ia & 127) >> 2);
            }
            if (lane == 0) atomicOr(a.err, best >= 0 ? 32 : 8);
        }
        if (lane == 0) {
            a.actions_out[out_index] = ia;
            a.psel_out[out_index] = psel;
        }
        sampled = ia;
        if (a.full_probs && h == 0) {
#pragma unroll
            for (int ts = 0; ts < TPS; ++ts)
                reinterpret_cast<float4*>(a.full_probs)[((size_t)out_index * TPS + ts) * 32 + j] =
                    make_float4(p[ts][0], p[ts][1], p[ts][2], p[ts][3]);
        }
    }
    if (MODE == 2) {
        const int ab = pre ? pre->ab : a.actions[sid];
        const float po = pre ? pre->po : a.p_old[sid];
        const float adv = pre ? pre->adv : a.adv[sid];
        float cand = 0.0f;
#pragma unroll
        for (int ts = 0; ts < TPS; ++ts)
#pragma unroll
            for (int i = 0; i < 4; ++i) cand = ((ab >> 7) == ts && (ab & 3) == i) ? p[ts][i] : cand;
        const float ps = __shfl(cand, (ab & 127) >> 2);
        const float gain = ps / po * adv;                                    // src/train.jl:39 (Float32)
        const double clip = adv >= 0.0f ? (1.0 + a.eps) * (double)adv : (1.0 - a.eps) * (double)adv;   // :1-7
        const bool unclipped = (double)gain < clip;
        const double minval = unclipped ? (double)gain : clip;
        const float sA = 1e-8f / (float)A;                                   // smooth/size(probs,1)  :22
        float lg[TPS][4], hl = 0.0f;
#pragma unroll
        for (int ts = 0; ts < TPS; ++ts)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float sp = p[ts][i] + sA; lg[ts][i] = logf(sp); hl += sp * lg[ts][i]; }
        hl = wave32_sum(hl);
        float dp[TPS][4], dot = 0.0f;
        const float dsel = unclipped ? -(a.inv_B * adv / po) : 0.0f;
#pragma unroll
        for (int ts = 0; ts < TPS; ++ts)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dp[ts][i] = a.c_over_B * (lg[ts][i] + 1.0f) + ((128 * ts + 4 * j + i == ab) ? dsel : 0.0f);
                dot += p[ts][i] * dp[ts][i];
            }
        dot = wave32_sum(dot);
        if (h == 0) {
#pragma unroll
            for (int ts = 0; ts < TPS; ++ts) {
                const float4 dyv =
                    make_float4(dy_round<DYBF16>(p[ts][0] * (dp[ts][0] - dot)), dy_round<DYBF16>(p[ts][1] * (dp[ts][1] - dot)),
                                dy_round<DYBF16>(p[ts][2] * (dp[ts][2] - dot)), dy_round<DYBF16>(p[ts][3] * (dp[ts][3] - dot)));
                a.dY[((size_t)state * TPS + ts) * 32 + j] = dyv;
                if (TPS == 1 && dy_copy) dy_copy[j] = dyv;
            }
        }
        if (lane == 0) { a.loss_terms[state * 2] = minval; a.loss_terms[state * 2 + 1] = (double)(-hl); }
    }
    return sampled;
}
