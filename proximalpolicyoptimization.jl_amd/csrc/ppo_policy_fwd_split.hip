// ppo_policy_fwd_split.hip -- train forward (K8/K9/K10: batch_action_probabilities + ppo_loss_with_entropy,
// test/quad_game_utilities.jl:73-79, src/train.jl:35-46) for SMALL minibatches.
//
// k_policy_fwd gives every state to ONE wave, 1312 dependent-chain MFMAs (~47 us for HID = 256): a minibatch of 512
// states occupies half of the chip's 1024 SIMDs for that long, one of 256 a quarter.  Here S = 2 or 4 waves of a
// workgroup share a state: wave s computes feature tiles [s NT/S, (s+1) NT/S) of layer 1, the tiles meet in LDS
// (fragment order, 4 KiB each), every wave takes all of them as B operands and computes ITS output tiles of layer 2
// and its share of the layer-3 dot products; the four partial logits per row meet in LDS again and wave 0 of the group
// runs the softmax / loss tail.  Saved activations, dY and the loss terms are the same buffers in the same layout as
// k_policy_fwd's MODE 2, so both backward kernels follow unchanged.
// Numerics: identical operations except the order in which the layer-3 partial sums are added (S partial chains instead
// of one): logits agree with k_policy_fwd to fp32 rounding, which is why only the TRAIN forward (tested against a float64
// restatement with a tolerance) has this form and the bit-exact rollout kernels do not.
#include "ppo_policy_tail.h"
#include "ppo_env_device.h"

static __device__ __forceinline__ void act_store_nt4(float4* p, float4 v) {
    f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
}

// CS: 1 = the states are env snapshots (compact rollouts): rows re-derived in LDS like k_policy_fwd MODE 4
template <int F, int HID, int S, int CS>
__global__ __launch_bounds__(256, 1) void k_policy_fwd_train_split(FwdArgs a) {
    constexpr int NT = HID / 32, NTS = NT / S, G = 4 / S;      // tiles per wave, state groups per workgroup
    constexpr int S41 = F / 8, S42 = NT * 4, XB = F / 2, XW = XB / 4;
    constexpr int PF = 8;
    static_assert(NT % S == 0 && (S == 2 || S == 4) && F % 8 == 0, "shape");
    static_assert(PF * 64 * 4 <= PPO_PACK_PAD, "ring over-read is covered by the stream padding");
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wv / S, sw = wv % S;
    __shared__ __attribute__((aligned(16))) float4 sW3[2 * NT * 16];
    __shared__ __attribute__((aligned(16))) float4 sB1[NT * 2 * 4];
    __shared__ __attribute__((aligned(16))) float4 sB2[NT * 2 * 4];
    __shared__ __attribute__((aligned(16))) float4 sP[4 * 64];                // layer-3 partial logits per wave
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    float4* const sH = reinterpret_cast<float4*>(dyn_lds);                    // [G][NT][4][64]: layer-1 tiles of the groups' states
    char* const env_lds = dyn_lds + (size_t)G * NT * 4 * 64 * sizeof(float4); // CS: one snapshot slot per wave
    for (int i = threadIdx.x; i < 2 * NT * 16; i += 256) sW3[i] = a.w3p[i];
    for (int i = threadIdx.x; i < NT * 8; i += 256) { sB1[i] = a.b1p[i]; sB2[i] = a.b2p[i]; }
    __syncthreads();
    EnvRefLds er = {};
    uint32_t tmpl_regs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (CS) {
        PPO_LDS char* b = (PPO_LDS char*)(env_lds + (size_t)wv * (2 * a.envV + 32));
        er.sc = (PPO_LDS int8_t*)b; er.dg = er.sc + a.envV;
        er.active = (PPO_LDS uint32_t*)(b + 2 * a.envV);
        const uint32_t* tp = reinterpret_cast<const uint32_t*>(a.env_tmpl + j * PPO_TPL);
#pragma unroll
        for (int k = 0; k < 9; ++k) tmpl_regs[k] = tp[k];
    }
    const int64_t gstride = (int64_t)gridDim.x * G;
    const int64_t iters = (a.B + gstride - 1) / gstride;       // the same for every wave of the grid: barriers stay matched
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t state = it * gstride + (int64_t)blockIdx.x * G + grp;
        const bool live = state < a.B;                         // uniform within the group of S waves
        int lane_o = lane, half_o = h;
        asm volatile("" : "+v"(lane_o), "+v"(half_o));         // per-state opaque offsets (see k_policy_fwd)
        int64_t sid = 0;
        uint32_t act = 0u;
        float xf[XB];
        if (live) {
            sid = a.idx[state];
            act = a.active[sid];
            uint32_t xw[XW];
            if (CS) {
                static_assert(!CS || XW == 9, "the built-in env has F = 72 features");
                const int nd = a.envV >> 1;
                if (lane < nd) reinterpret_cast<PPO_LDS uint32_t*>(er.sc)[lane] = reinterpret_cast<const uint32_t*>(a.cstate)[(size_t)sid * nd + lane];
                if (lane == 0) *er.active = act;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                uint32_t ob[9];
                env_observe_lane(er, tmpl_regs, j, h, ob);
#pragma unroll
                for (int k = 0; k < XW; ++k) xw[k] = ob[k < 9 ? k : 0];
                if (sw == 0) {                                  // rows for the backward, minibatch order
                    uint32_t* so = reinterpret_cast<uint32_t*>(a.xs_out + (size_t)state * 32 * F + (size_t)j * F + (size_t)h * XB);
#pragma unroll
                    for (int k = 0; k < XW; ++k) so[k] = xw[k];
                }
            } else {
                const uint32_t* xr = reinterpret_cast<const uint32_t*>(a.states + (size_t)sid * 32 * F + (size_t)j * F + (size_t)h * XB);
#pragma unroll
                for (int k = 0; k < XW; ++k) xw[k] = xr[k];
            }
#pragma unroll
            for (int k = 0; k < XW; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[4 * k + i] = (float)(int)(int8_t)(xw[k] >> (8 * i));
            // ---- layer 1, this wave's feature tiles
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
                float4* dst = a.act1 + ((size_t)state * NT + o) * 4 * 64;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v4 = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                    act_store_nt4(dst + q * 64 + lane, v4);
                    sH[((grp * NT + o) * 4 + q) * 64 + lane] = v4;
                }
            }
        }
        __syncthreads();
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
        if (live) {
            // ---- all layer-1 tiles of the state as B operands
            f32x16 h1[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v4 = sH[((grp * NT + t) * 4 + q) * 64 + lane];
                    h1[t][4 * q] = v4.x; h1[t][4 * q + 1] = v4.y; h1[t][4 * q + 2] = v4.z; h1[t][4 * q + 3] = v4.w;
                }
            }
            // ---- layer 2 (this wave's output tiles) + layer-3 partial dot products
            const float4* wp = a.w2p + (size_t)(sw * NTS) * S42 * 64 + lane_o;
            float4 ring[PF];
#pragma unroll
            for (int g = 0; g < PF; ++g) ring[g] = wp[(size_t)g * 64];
#pragma unroll 1
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
                float4* dst = a.act2 + ((size_t)state * NT + o) * 4 * 64;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    act_store_nt4(dst + q * 64 + lane, make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]));
                const float4* w3 = sW3 + (half_o * NT + o) * 16;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float4 w = w3[r];
                    p0 = fmaf(w.x, acc[r], p0); p1 = fmaf(w.y, acc[r], p1);
                    p2 = fmaf(w.z, acc[r], p2); p3 = fmaf(w.w, acc[r], p3);
                }
            }
            sP[wv * 64 + lane] = make_float4(p0, p1, p2, p3);
        }
        __syncthreads();
        if (live && sw == 0) {
            // ---- partial logits of the group in wave order, then exactly k_policy_fwd's epilogue
#pragma unroll
            for (int s = 1; s < S; ++s) {
                const float4 q4 = sP[(grp * S + s) * 64 + lane];
                p0 += q4.x; p1 += q4.y; p2 += q4.z; p3 += q4.w;
            }
            float l[1][4];
            l[0][0] = (p0 + __shfl_xor(p0, 32)) + a.b3[0];
            l[0][1] = (p1 + __shfl_xor(p1, 32)) + a.b3[1];
            l[0][2] = (p2 + __shfl_xor(p2, 32)) + a.b3[2];
            l[0][3] = (p3 + __shfl_xor(p3, 32)) + a.b3[3];
            policy_tail<2, 1, false>(a, state, sid, act, l, lane, j, h);
        }
    }
}

template <int F, int HID, int CS>
static int32_t launch_split(FwdArgs& a, int64_t B) {
    constexpr int NT = HID / 32;
    const size_t slots = CS ? (size_t)4 * (2 * a.envV + 32) : 0;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_policy_fwd_train_split<F, HID, 4, CS>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void*)k_policy_fwd_train_split<F, HID, 2, CS>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    if (B <= 256) {
        const unsigned grid = (unsigned)B;                                   // one state per workgroup
        hipLaunchKernelGGL((k_policy_fwd_train_split<F, HID, 4, CS>), dim3(grid), dim3(256), (size_t)1 * NT * 4096 + slots, ppo_stream(), a);
    } else {
        const int64_t need = (B + 1) / 2;
        const unsigned grid = (unsigned)(need < 256 ? need : 256);           // two states per workgroup
        hipLaunchKernelGGL((k_policy_fwd_train_split<F, HID, 2, CS>), dim3(grid), dim3(256), (size_t)2 * NT * 4096 + slots, ppo_stream(), a);
    }
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

// `a` comes filled like MODE 2 / MODE 4 of k_policy_fwd.  PPO_ERR_UNSUPPORTED (no error text): shape not covered.
int32_t launch_policy_train_fwd_split(ppo_policy_s* p, FwdArgs& a, int64_t B, int tps, bool compact) {
    if (p->dtype != PPO_DTYPE_F32 || p->F != 72 || tps != 1 || p->L != 2) return PPO_ERR_UNSUPPORTED;
    if (p->HID == 256) return compact ? launch_split<72, 256, 1>(a, B) : launch_split<72, 256, 0>(a, B);
    if (p->HID == 128) return compact ? launch_split<72, 128, 1>(a, B) : launch_split<72, 128, 0>(a, B);
    return PPO_ERR_UNSUPPORTED;
}
