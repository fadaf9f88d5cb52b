// ppo_policy_fwd.hip -- K2/K3/K4 (rollout) and K8/K9/K10 (train forward + loss + dlogits).
//
// Reference: action_probabilities / batch_action_probabilities (test/quad_game_utilities.jl:65-79)
// over SimplePolicy (test/policy.jl:9-31): per half-edge MLP F -> HID -> HID -> 4 (leakyrelu 0.01),
// vec, + mask, softmax; rand(Categorical) + ap[a] > 0 (src/collect_rollouts.jl:5-7); loss
// src/train.jl:35-46.
//
// gfx950 mapping: one wave owns one state = 32 half-edge rows.  The transposed product
// Y^T = W * X^T puts the rows on the MFMA lane axis (B operand, D columns) and the features on the
// accumulator-register axis, so the 32x32 accumulator tiles of layer 1 ARE the B operands of
// layer 2 with no LDS round trip or lane movement (v_mfma_f32_32x32x2_f32, exact fp32).  Weights
// stream as A operands from HBM/L2 in a pre-packed fragment order (1 KiB contiguous per wave
// load).  The 128 logits of a state live 4-per-lane, so masked softmax and the entropy/loss
// reductions are wave shuffles; the categorical sample is the reference's sequential fp32 CDF walk
// (bit-exact action indices).  Two waves per SIMD (<=256 VGPRs) hide the weight-load latency.
//
// MFMA-bound: 2*(F*HID + HID*HID)*32 flop per state on the matrix pipe; layer 3 (HID x 4) is a VALU
// dot-product epilogue.
#include "ppo_internal.h"
#include "ppo_device.h"

struct FwdArgs {
    // inputs
    const int8_t* states;      // MODE 0/1: [B][32][F]; MODE 2: rollout states base (gathered by idx)
    const uint32_t* active;    // same indexing as states
    const int32_t* idx;        // MODE 2: transition id per tile
    int64_t B;
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
};

__device__ __forceinline__ float wave32_max(float v) {
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave32_sum(float v) {
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) v = v + __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// waves per SIMD: the HID=256 / F=216 instantiations need more than 256 VGPRs (128 for the layer-1
// accumulators + operands in flight), so they run one wave per SIMD with the whole 512-entry file.
template <int F, int HID>
struct FwdCfg { static constexpr int WPS = (HID >= 256 || F > 128) ? 1 : 2; };

template <int F, int HID, int MODE>
__global__ __launch_bounds__(256, (FwdCfg<F, HID>::WPS)) void k_policy_fwd(FwdArgs a) {
    constexpr int NT = HID / 32;       // 32-feature tiles
    constexpr int S41 = F / 8;         // float4 groups of layer-1 k-steps
    constexpr int XB = F / 2;          // bytes of the state row held by one lane
    constexpr int XW = XB / 4;
    constexpr int PF = (FwdCfg<F, HID>::WPS == 1) ? 8 : 4;   // weight-fragment groups kept in flight per wave
    static_assert(F % 8 == 0 && HID % 32 == 0, "shape");
    const int lane = threadIdx.x & 63;
    const int j = lane & 31;           // half-edge row
    const int h = lane >> 5;           // lane half = k parity of the MFMA step
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;

    // the state rows of the NEXT tile are fetched while the current tile computes (the gather through idx is
    // two dependent HBM round trips that one wave per SIMD cannot hide otherwise)
    uint32_t xw[XW];
    uint32_t act_next = 0;
    int64_t sid_next = 0;
    auto fetch_state = [&](int64_t t) {
        const int64_t sidn = (MODE == 2) ? (int64_t)a.idx[t] : t;
        const uint32_t* xr = reinterpret_cast<const uint32_t*>(a.states + (size_t)sidn * 32 * F + (size_t)j * F + (size_t)h * XB);
#pragma unroll
        for (int k = 0; k < XW; ++k) xw[k] = xr[k];
        act_next = a.active[sidn];
        sid_next = sidn;
    };
    constexpr bool PFX = (FwdCfg<F, HID>::WPS == 1);    // with two waves per SIMD the partner wave hides it instead
    if (PFX && wave < a.B) fetch_state(wave);

    for (int64_t tile = wave; tile < a.B; tile += nwaves) {
        int64_t sid;
        uint32_t act;
        // ---- state row -> B operands of layer 1: lane half 0 holds features [0,F/2), half 1 [F/2,F)
        float xf[XB];
        if (PFX) {
            sid = sid_next; act = act_next;
#pragma unroll
            for (int k = 0; k < XW; ++k) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[4 * k + i] = (float)(int)(int8_t)(xw[k] >> (8 * i));
            }
        } else {
            sid = (MODE == 2) ? (int64_t)a.idx[tile] : tile;
            act = a.active[sid];
            const uint32_t* xr = reinterpret_cast<const uint32_t*>(a.states + (size_t)sid * 32 * F + (size_t)j * F + (size_t)h * XB);
#pragma unroll
            for (int k = 0; k < XW; ++k) {
                const uint32_t w = xr[k];
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[4 * k + i] = (float)(int)(int8_t)(w >> (8 * i));
            }
        }
        if (PFX) {
            const int64_t nt = tile + nwaves;
            fetch_state(nt < a.B ? nt : tile);                  // unconditional: lands under the MFMA chains below
        }

        // ---- layer 1: H1^T[o-tile] = W1[o-tile,:] * X^T  (accumulator initialised with the bias)
        // The weight stream of a layer is one linear run of 1 KiB fragment groups (4 MFMA k-steps each);
        // a PF-deep register ring keeps PF groups in flight so the single wave of a SIMD never waits on L2.
        f32x16 h1[NT];
        {
            const float4* wp = a.w1p + lane;
            float4 ring[PF];
#pragma unroll
            for (int g = 0; g < PF; ++g) ring[g] = wp[(size_t)g * 64];
#pragma unroll
            for (int o = 0; o < NT; ++o) {
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = a.b1p[(o * 2 + h) * 4 + q];
                    acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
                }
#pragma unroll
                for (int s4 = 0; s4 < S41; ++s4) {
                    const int g = o * S41 + s4;
                    const float4 w = ring[g % PF];
                    ring[g % PF] = wp[(size_t)(g + PF) * 64];      // the packed buffers carry PF groups of tail padding
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, xf[4 * s4 + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, xf[4 * s4 + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, xf[4 * s4 + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, xf[4 * s4 + 3], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = lrelu(acc[r]);
                h1[o] = acc;
                if (MODE == 2) {
                    float4* dst = a.act1 + ((size_t)tile * NT + o) * 4 * 64;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        dst[q * 64 + lane] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                }
            }
        }

        // ---- layer 2 (MFMA, B operands = layer-1 accumulators) + layer 3 (VALU dot epilogue)
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
        {
            constexpr int S42 = NT * 4;                  // groups per output tile
            static_assert(S42 % PF == 0, "ring depth must divide the groups per tile");
            const float4* wp = a.w2p + lane;
            float4 ring[PF];
#pragma unroll
            for (int g = 0; g < PF; ++g) ring[g] = wp[(size_t)g * 64];
#pragma unroll 1
            for (int o = 0; o < NT; ++o) {
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = a.b2p[(o * 2 + h) * 4 + q];
                    acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
                }
                const float4* wo = wp + (size_t)o * S42 * 64;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int s4 = t * 4 + r4;
                        const float4 w = ring[s4 % PF];
                        // next group PF ahead in the linear stream (tail padding covers the last tile's over-read)
                        ring[s4 % PF] = wo[(size_t)(s4 + PF) * 64];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, h1[t][4 * r4 + 0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, h1[t][4 * r4 + 1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, h1[t][4 * r4 + 2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, h1[t][4 * r4 + 3], acc, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = lrelu(acc[r]);
                if (MODE == 2) {
                    float4* dst = a.act2 + ((size_t)tile * NT + o) * 4 * 64;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        dst[q * 64 + lane] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                }
                const float4* w3 = a.w3p + (size_t)(h * NT + o) * 16;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float4 w = w3[r];
                    p0 = fmaf(w.x, acc[r], p0); p1 = fmaf(w.y, acc[r], p1);
                    p2 = fmaf(w.z, acc[r], p2); p3 = fmaf(w.w, acc[r], p3);
                }
            }
        }
        float l[4];
        l[0] = (p0 + __shfl_xor(p0, 32)) + a.b3[0];
        l[1] = (p1 + __shfl_xor(p1, 32)) + a.b3[1];
        l[2] = (p2 + __shfl_xor(p2, 32)) + a.b3[2];
        l[3] = (p3 + __shfl_xor(p3, 32)) + a.b3[3];

        // ---- masked softmax over the 128 logits of the state (quad = row/4)
        const bool on = (act >> (j >> 2)) & 1u;
        float m = on ? fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3])) : -INFINITY;
        m = wave32_max(m);
        float e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = on ? exp_dev(l[i] - m) : 0.0f;
        float ssum = ((e[0] + e[1]) + e[2]) + e[3];
        ssum = wave32_sum(ssum);
        float p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = e[i] / ssum;

        if (MODE == 0) {
            if (h == 0) reinterpret_cast<float4*>(a.probs_out)[(size_t)tile * 32 + j] = make_float4(p[0], p[1], p[2], p[3]);
        }
        if (MODE == 1) {
            // rand(Categorical(p)): sequential fp32 inverse-CDF walk, same uniform as the oracle
            uint32_t rnd[4];
            philox4x32_10((uint32_t)(a.global_offset + tile), a.tick[tile], 0u, 0u, a.k0, a.k1, rnd);
            const float u = u01_from_u32(rnd[0]);
            float cp = readlane_f(p[0], 0);
            int ia = 0;
#pragma unroll
            for (int q = 1; q < 128; ++q) {
                const float pa = readlane_f(p[q & 3], q >> 2);
                const bool take = cp <= u;
                cp = take ? cp + pa : cp;
                ia = take ? q : ia;
            }
            const int sel_lane = ia >> 2;
            const float cand = (ia & 3) == 0 ? p[0] : (ia & 3) == 1 ? p[1] : (ia & 3) == 2 ? p[2] : p[3];
            const float psel = __shfl(cand, sel_lane);
            if (lane == 0) {
                if (!(psel > 0.0f)) atomicOr(a.err, 8);     // @assert ap[a] > 0.0
                a.actions_out[tile] = ia;
                a.psel_out[tile] = psel;
            }
            if (a.full_probs && h == 0)
                reinterpret_cast<float4*>(a.full_probs)[(size_t)tile * 32 + j] = make_float4(p[0], p[1], p[2], p[3]);
        }
        if (MODE == 2) {
            const int ab = a.actions[sid];
            const float po = a.p_old[sid];
            const float adv = a.adv[sid];
            const float cand = (ab & 3) == 0 ? p[0] : (ab & 3) == 1 ? p[1] : (ab & 3) == 2 ? p[2] : p[3];
            const float ps = __shfl(cand, ab >> 2);
            const float gain = ps / po * adv;                                    // src/train.jl:39 (Float32)
            const double clip = adv >= 0.0f ? (1.0 + a.eps) * (double)adv : (1.0 - a.eps) * (double)adv;   // :1-7
            const bool unclipped = (double)gain < clip;
            const double minval = unclipped ? (double)gain : clip;
            const float sA = 1e-8f / 128.0f;                                     // smooth/size(probs,1)  :22
            float lg[4], hl = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float sp = p[i] + sA; lg[i] = logf(sp); hl += sp * lg[i]; }
            hl = wave32_sum(hl);
            float dp[4], dot = 0.0f;
            const float dsel = unclipped ? -(a.inv_B * adv / po) : 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dp[i] = a.c_over_B * (lg[i] + 1.0f) + ((4 * j + i == ab) ? dsel : 0.0f);
                dot += p[i] * dp[i];
            }
            dot = wave32_sum(dot);
            if (h == 0) a.dY[(size_t)tile * 32 + j] = make_float4(p[0] * (dp[0] - dot), p[1] * (dp[1] - dot),
                                                                  p[2] * (dp[2] - dot), p[3] * (dp[3] - dot));
            if (lane == 0) { a.loss_terms[tile * 2] = minval; a.loss_terms[tile * 2 + 1] = (double)(-hl); }
        }
    }
}

// rand(Categorical) on given probabilities (parity entry point, src/collect_rollouts.jl:6-7):
// one thread per row walks the CDF sequentially in fp32.
__global__ void k_categorical(const float* __restrict__ probs, const float* __restrict__ u, int64_t B, int64_t A,
                              int32_t* __restrict__ actions, float* __restrict__ psel, int32_t* __restrict__ err) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* p = probs + b * A;
    const float uu = u[b];
    float cp = p[0];
    int64_t i = 0;
    while (cp <= uu && i < A - 1) { i += 1; cp = cp + p[i]; }
    actions[b] = (int32_t)i;
    psel[b] = p[i];
    err[b] = !(p[i] > 0.0f);
}

template <int MODE>
static int32_t dispatch_fwd(ppo_policy_s* p, const FwdArgs& args, int64_t B) {
    const int64_t need = (B + 3) / 4;
    // persistent waves: 256 CUs x WPS blocks of 4 waves (one per SIMD)
#define LAUNCH(FF, HH)                                                                                   \
    do {                                                                                                 \
        const int64_t cap = 256 * FwdCfg<FF, HH>::WPS;                                                   \
        const unsigned grid = (unsigned)(need < cap ? need : cap);                                       \
        hipLaunchKernelGGL((k_policy_fwd<FF, HH, MODE>), dim3(grid), dim3(256), 0, ppo_stream(), args);  \
    } while (0)
    if (p->F == 72 && p->HID == 256) LAUNCH(72, 256);
    else if (p->F == 72 && p->HID == 128) LAUNCH(72, 128);
    else if (p->F == 216 && p->HID == 128) LAUNCH(216, 128);
    else { ppo_set_error("unsupported policy shape (F,HID) for the gfx950 kernels"); return PPO_ERR_UNSUPPORTED; }
#undef LAUNCH
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

static void fill_weights(ppo_policy_s* p, FwdArgs& a) {
    a.w1p = (const float4*)p->w1p.p; a.w2p = (const float4*)p->w2p.p; a.b1p = (const float4*)p->b1p.p;
    a.b2p = (const float4*)p->b2p.p; a.w3p = (const float4*)p->w3p.p; a.b3 = p->b3.p;
}

int32_t launch_policy_probs(ppo_policy_s* p, const int8_t* states_dev, const uint32_t* active_dev, int64_t B,
                            float* probs_dev) {
    if (B <= 0) return PPO_OK;
    FwdArgs a = {};
    fill_weights(p, a);
    a.states = states_dev; a.active = active_dev; a.B = B; a.probs_out = probs_dev;
    ProfScope ps("k_policy_fwd_probs");
    return dispatch_fwd<0>(p, a, B);
}

int32_t launch_policy_rollout(ppo_policy_s* p, ppo_env_s* e, const int8_t* states_dev, const uint32_t* active_dev,
                              int32_t* actions_out, float* psel_out, float* full_probs_or_null) {
    FwdArgs a = {};
    fill_weights(p, a);
    a.states = states_dev; a.active = active_dev; a.B = e->N;
    a.tick = e->tick.p; a.global_offset = e->global_offset; a.k0 = (uint32_t)e->seed; a.k1 = (uint32_t)(e->seed >> 32);
    a.actions_out = actions_out; a.psel_out = psel_out; a.full_probs = full_probs_or_null; a.err = e->err.p;
    ProfScope ps("k_policy_fwd_rollout");
    return dispatch_fwd<1>(p, a, e->N);
}

int32_t launch_policy_train_fwd(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B,
                                int64_t B_global, double eps, double entropy_weight) {
    FwdArgs a = {};
    fill_weights(p, a);
    a.states = ro->states.p; a.active = ro->active.p; a.idx = idx_dev; a.B = B;
    a.act1 = (float4*)p->act1.p; a.act2 = (float4*)p->act2.p; a.dY = (float4*)p->dY.p; a.loss_terms = p->loss_terms.p;
    a.actions = ro->actions.p; a.p_old = ro->p_sel.p; a.adv = ro->returns.p;
    a.eps = eps; a.c_over_B = (float)(entropy_weight / (double)B_global); a.inv_B = (float)(1.0 / (double)B_global);
    ProfScope ps("k_policy_fwd_train");
    return dispatch_fwd<2>(p, a, B);
}

int32_t launch_categorical(const float* probs, const float* u, int64_t B, int64_t A, int32_t* actions, float* psel,
                           int32_t* err) {
    if (B <= 0) return PPO_OK;
    hipLaunchKernelGGL(k_categorical, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, ppo_stream(), probs, u, B, A,
                       actions, psel, err);
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}
