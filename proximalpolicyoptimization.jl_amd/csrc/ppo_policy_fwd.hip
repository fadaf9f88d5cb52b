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
#include "ppo_policy_tail.h"
#include "ppo_env_device.h"
#include <cstdlib>

// activation stores of the train forward: tuning knobs for A/B builds (defaults are the shipped configuration)
#ifndef PPO_FWD_STORE
#define PPO_FWD_STORE 1          // 0: timing-only build without activation stores (results are wrong)
#endif
// Non-temporal stores for the 268 MB of saved activations: they are written once and read once by the backward
// kernel much later, so they should not churn the L2; measured -4 % on the train forward (A/B in one process).
#ifndef PPO_FWD_NT
#define PPO_FWD_NT 1
#endif
#if PPO_FWD_NT
static __device__ __forceinline__ void act_store_nt(float4* p, float4 v) {
    f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
}
#define ACT_STORE(ptr, val) act_store_nt((ptr), (val))
#else
#define ACT_STORE(ptr, val) (*(ptr) = (val))
#endif

// waves per SIMD: the HID=256 / F=216 instantiations need more than 256 VGPRs (128 for the layer-1
// accumulators + operands in flight), so they run one wave per SIMD with the whole 512-entry file.
// HID = 128, F = 72: two waves per SIMD with the deep (16-group) ring in the train forward.  Re-measured in round 2
// (gpurun_out/r2l, train forward / 128-step rollout): 2 waves + 16 groups 0.0685 ms / 8.34 ms; 2 + 8: 0.0817 / 8.37;
// 3 waves + 4 groups: 0.0717 / 8.87 (the rollout instantiation spills at 168 registers); 3 + 8: 0.0889 / 8.88.
#ifndef PPO_FWD_WPS_SMALL
#define PPO_FWD_WPS_SMALL 2          // waves per SIMD of the HID = 128, F = 72 instantiations (A/B knob)
#endif
#ifndef PPO_FWD_PF_SMALL_TRAIN
#define PPO_FWD_PF_SMALL_TRAIN PPO_FWD_PF   // their train-forward ring depth
#endif
template <int F, int HID>
struct FwdCfg { static constexpr int WPS = (HID >= 256 || F > 128) ? 1 : PPO_FWD_WPS_SMALL; };

// TPS = 32-row tiles per state (H = 32*TPS half-edges, A = 128*TPS actions): the wave walks the tiles of its
// state one after the other, keeps the 4*TPS logits per lane, and runs softmax / sampling / loss once per state.
// DEEP = 1: Policy(F, HID, num_hidden_layers != 2, 4) (test/policy.jl:9-19).  The hidden->hidden block becomes a loop over
// a.nl2 layers: the output tiles of every layer but the last are parked in a wave-private LDS region (lane-linear 1 KiB
// rows, no barrier: only the wave itself reads them back) and re-loaded as the next layer's B operands; the last one feeds
// the layer-3 dot epilogue as before (nl2 == 0: the dot runs on the layer-1 tiles).  Every layer walks its contraction in
// the same accumulator-register order as layer 2, so the fp32 chain order per layer is the one the oracle's device-order
// mode mirrors.  DEEP = 0 is the num_hidden_layers == 2 kernel, unchanged.  MODE 3 (persistent rollout) has no deep form:
// its env slots and the park would share the LDS; deep policies roll out with the per-step launches.
template <int F, int HID, int MODE, int TPS, int DEEP = 0>
__global__ __launch_bounds__(256, (FwdCfg<F, HID>::WPS)) void k_policy_fwd(FwdArgs a) {
    constexpr int NT = HID / 32;       // 32-feature tiles
    constexpr int S41 = F / 8;         // float4 groups of layer-1 k-steps
    constexpr int XB = F / 2;          // bytes of the state row held by one lane
    constexpr int XW = XB / 4;
#ifndef PPO_FWD_PF
#define PPO_FWD_PF 16
#endif
#ifndef PPO_FWD_WGSYNC
#define PPO_FWD_WGSYNC 0
#endif
#ifndef PPO_FWD_OUNROLL
#define PPO_FWD_OUNROLL 1
#endif
    constexpr int PFW = (PPO_FWD_PF < HID / 8) ? PPO_FWD_PF : HID / 8;     // at most the groups of one output tile
    // two waves per SIMD (HID = 128, F = 72): 4 groups cover the L2 latency; the train forward, whose activation stores
    // sit in the same vmcnt queue, wants the deep ring here too (0.072 -> 0.069 ms), the rollout does not (8.3 -> 8.7 ms)
    constexpr int PFT = (PPO_FWD_PF_SMALL_TRAIN < HID / 8) ? PPO_FWD_PF_SMALL_TRAIN : HID / 8;
    constexpr int PF = (FwdCfg<F, HID>::WPS == 1) ? PFW : ((MODE == 2 || MODE == 4) ? PFT : 4);   // weight-fragment groups kept in flight per wave
    constexpr bool TRAIN = (MODE == 2 || MODE == 4);    // train forward: saves activations, loss tail
    constexpr bool OBS = (MODE == 3 || MODE == 4);      // the state rows are re-derived from an env snapshot in LDS
    constexpr int TMODE = TRAIN ? 2 : MODE;             // policy_tail's mode
    static_assert(F % 8 == 0 && HID % 32 == 0, "shape");
    static_assert(PF * 64 * 4 <= PPO_PACK_PAD, "the ring reads PF groups past the end of a packed weight stream: padding must cover it");
    const int lane = threadIdx.x & 63;
    const int j = lane & 31;           // half-edge row inside the tile
    const int h = lane >> 5;           // lane half = k parity of the MFMA step
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;

    // W3 fragments and the two bias packs are read once per output tile by every wave: staged in LDS once per
    // workgroup so the per-tile epilogues see an LDS round trip instead of a global (L1/L2) one
    __shared__ __attribute__((aligned(16))) float4 sW3[2 * NT * 16];     // [half][tile][reg]  (HID*4 floats)
    __shared__ __attribute__((aligned(16))) float4 sB1[NT * 2 * 4];      // [tile][half][4]    (HID floats)
    constexpr int NB2L = DEEP ? 3 : 1;                                   // bias packs of up to three hidden->hidden layers
    __shared__ __attribute__((aligned(16))) float4 sB2[NB2L * NT * 2 * 4];
    static_assert(!(DEEP && MODE == 3), "no deep persistent rollout");
    for (int i = threadIdx.x; i < 2 * NT * 16; i += 256) sW3[i] = a.w3p[i];
    for (int i = threadIdx.x; i < NT * 8; i += 256) sB1[i] = a.b1p[i];
    for (int i = threadIdx.x; i < NT * 8 * (DEEP ? a.nl2 : 1); i += 256) sB2[i] = a.b2p[i];
    __syncthreads();

    // the rows of the NEXT 32-row tile are fetched while the current one computes (the gather through idx is two
    // dependent HBM round trips that one wave per SIMD cannot hide otherwise)
    constexpr bool PFX = (FwdCfg<F, HID>::WPS == 1) && !OBS;      // with two waves per SIMD the partner wave hides it instead;
                                                                  // MODE 3 / 4 compute the rows themselves (no fetch)
    uint32_t xw[XW];
    auto fetch_rows = [&](int64_t state, int ts) {
        const int64_t sidn = (MODE == 2) ? (int64_t)a.idx[state] : state;
        const uint32_t* xr = reinterpret_cast<const uint32_t*>(a.states + ((size_t)sidn * TPS + ts) * 32 * F +
                                                               (size_t)j * F + (size_t)h * XB);
#pragma unroll
        for (int k = 0; k < XW; ++k) xw[k] = xr[k];
    };
    if (PFX && wave < a.B) fetch_rows(wave, 0);

#ifdef PPO_FWD_STAMP
    unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0}, st_t = clock64();
#define FSTAMP(i) do { unsigned long long _n = clock64(); st_sum[i] += _n - st_t; st_t = _n; } while (0)
#else
#define FSTAMP(i) do {} while (0)
#endif
    // ---- MODE 3: persistent rollout.  Envs are independent, so every wave walks ITS envs (state = wave, wave + nwaves,
    // ...) through all T steps without any grid-wide synchronisation: observe -> MLP -> sample -> step!, with the env
    // state (scores, degrees, counters) resident in the wave's LDS slots and written back once at the end.
    extern __shared__ __attribute__((aligned(16))) char env_lds[];
    const int slot_bytes = 2 * a.envV + 32;
    char* const my_slots = env_lds + (size_t)(threadIdx.x >> 6) * a.env_slots * slot_bytes;
    EnvConst ec;
    auto slot_ref = [&](int slot) {                               // LDS-typed pointers: ds_* instead of flat_* accesses
        PPO_LDS char* b = (PPO_LDS char*)(my_slots + (size_t)slot * slot_bytes);
        EnvRefLds r;
        r.sc = (PPO_LDS int8_t*)b; r.dg = r.sc + a.envV;
        PPO_LDS uint32_t* w = (PPO_LDS uint32_t*)(b + 2 * a.envV);
        r.active = w; r.steps = (PPO_LDS int32_t*)(w + 1); r.reward = (PPO_LDS float*)(w + 2);
        r.done = (PPO_LDS uint8_t*)(w + 3); r.episode = w + 4; r.tick = w + 5;
        return r;
    };
    if (MODE == 3) {
        ec.Q = a.envQ; ec.V = a.envV; ec.max_actions = a.env_max_actions; ec.no_action_reward = a.env_nar; ec.k0 = a.k0; ec.k1 = a.k1;
        int slot = 0;
        for (int64_t n = wave; n < a.B; n += nwaves, ++slot) {
            const EnvRefLds r = slot_ref(slot);
            for (int v = lane; v < a.envV; v += 64) { r.sc[v] = a.env_score[n * a.envV + v]; r.dg[v] = a.env_degree[n * a.envV + v]; }
            if (lane == 0) {
                *r.active = a.env_active[n]; *r.steps = a.env_steps[n]; *r.reward = a.env_reward[n];
                *(PPO_LDS uint32_t*)r.done = a.env_done[n]; *r.episode = a.env_episode[n]; *r.tick = a.env_tick[n];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // slots are wave-private: program order + in-order LDS suffice
    }
    uint32_t tmpl_regs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};       // MODE 3 / 4, TPS == 1: this lane's 36 template vertex ids
    if (OBS && TPS == 1) {
        const uint32_t* tp = reinterpret_cast<const uint32_t*>(a.env_tmpl + j * PPO_TPL);
#pragma unroll
        for (int k = 0; k < 9; ++k) tmpl_regs[k] = tp[k];
    }
    // MODE 4: the env snapshot of a transition (2V bytes + the active-quad word) is fetched through idx one state ahead
    // (two dependent HBM round trips) into registers and parked in the wave's LDS slot at the top of its state
    // (one dword per lane: 2V/4 <= 64 dwords for V <= 128)
    uint32_t cs_next = 0u, act_next = 0u;
    int32_t sid_next = 0;
    auto fetch_snapshot = [&](int64_t state) {
        sid_next = a.idx[state];
        act_next = a.active[sid_next];
        const int nd = a.envV >> 1;                              // dwords of one snapshot
        cs_next = (lane < nd) ? reinterpret_cast<const uint32_t*>(a.cstate)[(size_t)sid_next * nd + lane] : 0u;
    };
    if (MODE == 4 && wave < a.B) fetch_snapshot(wave);
    const int64_t t_steps = (MODE == 3) ? a.T : 1;
    for (int64_t tt = 0; tt < t_steps; ++tt) {
    int slot = 0;
    for (int64_t state = wave; state < a.B; state += nwaves, ++slot) {
        const int64_t sid = (MODE == 2) ? (int64_t)a.idx[state] : ((MODE == 4) ? (int64_t)sid_next : state);
        EnvRefLds er = {};
        if (MODE == 3) er = slot_ref(slot);
        if (MODE == 4) {
            er = slot_ref(0);
            if (lane < (a.envV >> 1)) reinterpret_cast<PPO_LDS uint32_t*>(er.sc)[lane] = cs_next;
            if (lane == 0) *er.active = act_next;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private slot: program order + in-order LDS suffice
        }
        const uint32_t act = (MODE == 3) ? *er.active : ((MODE == 4) ? act_next : a.active[sid]);
        if (MODE == 4) fetch_snapshot(state + nwaves < a.B ? state + nwaves : state);      // lands under this state's MFMAs
        const uint32_t tick_val = (MODE == 3) ? *er.tick : ((MODE == 1) ? a.tick[state] : 0u);
        const int64_t out_index = (MODE == 3) ? tt * a.B + state : state;
        float l[TPS][4];
        // multi-tile states: the tile loop stays a real loop (no 4x code blow-up, no cross-tile hoisting that
        // would spill); the 4 logits per lane of each tile are parked in LDS and re-read after the loop
        __shared__ float4 sL[(TPS > 1) ? 4 * TPS * 64 : 1];

#pragma unroll 1
        for (int ts = 0; ts < TPS; ++ts) {
            const int64_t tile = state * TPS + ts;
            // per-tile opaque lane offsets: the weight/bias fragment loads below must be re-issued for every tile
            // (they are L2 hits); without this LICM hoists ~330 loop-invariant float4 loads out of the tile loop
            int lane_o = lane, half_o = h;
            asm volatile("" : "+v"(lane_o), "+v"(half_o));
            // ---- state rows -> B operands of layer 1: lane half 0 holds features [0,F/2), half 1 [F/2,F)
            if (OBS) {
                // state(env): this lane's 36 features of half-edge row 32*ts + j, recorded into the rollout buffer
                // (MODE 3, expanded storage) or into the minibatch-ordered row scratch the backward reads (MODE 4)
                static_assert(!OBS || XW == 9, "the built-in env has F = 72 features");
                uint32_t ob[9], tid[9];
                if (TPS == 1) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) tid[k] = tmpl_regs[k];          // row j's ids, loaded once per kernel
                } else {
                    const uint32_t* tp = reinterpret_cast<const uint32_t*>(a.env_tmpl + (32 * ts + j) * PPO_TPL);
#pragma unroll
                    for (int k = 0; k < 9; ++k) tid[k] = tp[k];
                }
                env_observe_lane(er, tid, 32 * ts + j, h, ob);
#pragma unroll
                for (int k = 0; k < XW; ++k) xw[k] = ob[k < 9 ? k : 0];
                int8_t* const rows_out = (MODE == 4) ? a.xs_out : a.states_out;
                if (rows_out) {                                  // wave-uniform
                    uint32_t* so = reinterpret_cast<uint32_t*>(rows_out + ((size_t)(MODE == 4 ? state : out_index) * TPS + ts) * 32 * F +
                                                               (size_t)j * F + (size_t)h * XB);
#pragma unroll
                    for (int k = 0; k < XW; ++k) so[k] = xw[k];
                }
                if (MODE == 3 && ts == 0 && a.cstate_out && lane < (a.envV >> 1))      // compact storage: the env snapshot itself
                    reinterpret_cast<uint32_t*>(a.cstate_out)[(size_t)out_index * (a.envV >> 1) + lane] =
                        reinterpret_cast<PPO_LDS uint32_t*>(er.sc)[lane];
            } else if (!PFX) fetch_rows(state, ts);
            float xf[XB];
#pragma unroll
            for (int k = 0; k < XW; ++k) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[4 * k + i] = (float)(int)(int8_t)(xw[k] >> (8 * i));
            }
            if (PFX) {                                          // unconditional: lands under the MFMA chains below
                if (ts + 1 < TPS) fetch_rows(state, ts + 1);
                else fetch_rows(state + nwaves < a.B ? state + nwaves : state, 0);
            }

            FSTAMP(0);
            // ---- layer 1: H1^T[o-tile] = W1[o-tile,:] * X^T  (accumulator initialised with the bias)
            // The weight stream of a layer is one linear run of 1 KiB fragment groups (4 MFMA k-steps each);
            // a PF-deep register ring keeps PF groups in flight so the single wave of a SIMD never waits on L2.
            f32x16 h1[NT];
            {
                const float4* wp = a.w1p + lane_o;
                float4 ring[PF];
#pragma unroll
                for (int g = 0; g < PF; ++g) ring[g] = wp[(size_t)g * 64];
#pragma unroll
                for (int o = 0; o < NT; ++o) {
                    f32x16 acc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 b = sB1[(o * 2 + half_o) * 4 + q];
                        acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
                    }
#pragma unroll
                    for (int s4 = 0; s4 < S41; ++s4) {
                        const int g = o * S41 + s4;
                        const float4 w = ring[g % PF];
                        ring[g % PF] = wp[(size_t)(g + PF) * 64];      // the packed buffers carry tail padding
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, xf[4 * s4 + 0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, xf[4 * s4 + 1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, xf[4 * s4 + 2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, xf[4 * s4 + 3], acc, 0, 0, 0);
                    }
                    // the epilogue reads the tile with VALU instructions: ask for VGPRs (hipcc otherwise parks the
                    // accumulator in AGPRs and copies it out with 16 v_accvgpr_read per tile)
                    asm volatile("" : "+v"(acc));
                    lrelu16(acc);
                    h1[o] = acc;
                    if (TRAIN && PPO_FWD_STORE) {
                        float4* dst = a.act1 + ((size_t)tile * NT + o) * 4 * 64;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            ACT_STORE(dst + q * 64 + lane, make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]));
                    }
                }
            }

            FSTAMP(1);
            // ---- layer 2 (MFMA, B operands = layer-1 accumulators) + layer 3 (VALU dot epilogue)
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
            if constexpr (DEEP) {
                constexpr int S42 = NT * 4;
                static_assert(S42 % PF == 0, "ring depth must divide the groups per tile");
                // this wave's park: [NT tiles][4 quarter-tiles][64 lanes] float4
                float4* const park = reinterpret_cast<float4*>(env_lds + a.park_off) + (size_t)(threadIdx.x >> 6) * NT * 256 + lane;
                const int nl2 = a.nl2;
                auto dot3 = [&](int o, const f32x16& acc) {
                    const float4* w3 = sW3 + (half_o * NT + o) * 16;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float4 w = w3[r];
                        p0 = fmaf(w.x, acc[r], p0); p1 = fmaf(w.y, acc[r], p1);
                        p2 = fmaf(w.z, acc[r], p2); p3 = fmaf(w.w, acc[r], p3);
                    }
                };
#pragma unroll 1
                for (int l = 0; l < nl2; ++l) {
                    const bool last = (l + 1 == nl2);                   // wave-uniform
                    const float4* wp = a.w2p + (size_t)l * (HID * HID / 4) + lane_o;
                    float4* const act_out = last ? a.act2 : a.act_mid[l < 2 ? l : 0];
                    float4 ring[PF];
#pragma unroll
                    for (int g = 0; g < PF; ++g) ring[g] = wp[(size_t)g * 64];
#pragma unroll 1
                    for (int o = 0; o < NT; ++o) {
                        f32x16 acc;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 b = sB2[((l * NT + o) * 2 + half_o) * 4 + q];
                            acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
                        }
                        const float4* wo = wp + (size_t)o * S42 * 64;
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
#pragma unroll
                            for (int r4 = 0; r4 < 4; ++r4) {
                                const int s4 = t * 4 + r4;
                                const float4 w = ring[s4 % PF];
                                ring[s4 % PF] = wo[(size_t)(s4 + PF) * 64];      // the next layer's head / the tail padding
                                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, h1[t][4 * r4 + 0], acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, h1[t][4 * r4 + 1], acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, h1[t][4 * r4 + 2], acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, h1[t][4 * r4 + 3], acc, 0, 0, 0);
                            }
                        }
                        asm volatile("" : "+v"(acc));
                        lrelu16(acc);
                        if (TRAIN && PPO_FWD_STORE) {
                            float4* dst = act_out + ((size_t)tile * NT + o) * 4 * 64;
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                ACT_STORE(dst + q * 64 + lane, make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]));
                        }
                        if (last) dot3(o, acc);
                        else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                park[(o * 4 + q) * 64] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                        }
                    }
                    if (!last) {                                        // this layer's output becomes the next layer's input
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private: program order + in-order LDS suffice
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4 v = park[(t * 4 + q) * 64];
                                h1[t][4 * q + 0] = v.x; h1[t][4 * q + 1] = v.y; h1[t][4 * q + 2] = v.z; h1[t][4 * q + 3] = v.w;
                            }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read back before the next layer overwrites the park
                    }
                }
                if (nl2 == 0) {                                         // num_hidden_layers == 1: layer 3 on the layer-1 tiles
#pragma unroll
                    for (int o = 0; o < NT; ++o) dot3(o, h1[o]);
                }
            } else {
                constexpr int S42 = NT * 4;                  // groups per output tile
                static_assert(S42 % PF == 0, "ring depth must divide the groups per tile");
                const float4* wp = a.w2p + lane_o;
                float4 ring[PF];
#pragma unroll
                for (int g = 0; g < PF; ++g) ring[g] = wp[(size_t)g * 64];
#pragma unroll PPO_FWD_OUNROLL
                for (int o = 0; o < NT; ++o) {
                    // keep the 4 waves of the workgroup on the same 32 KiB weight chunk: their 1 KiB fragment loads then
                    // hit in the CU's L1 for three of the four waves (speed only; skipped when trip counts differ)
                    if (PPO_FWD_WGSYNC && a.wg_sync) __builtin_amdgcn_s_barrier();
                    f32x16 acc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 b = sB2[(o * 2 + half_o) * 4 + q];
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
                    FSTAMP(2);
                    asm volatile("" : "+v"(acc));
                    lrelu16(acc);
                    if (TRAIN && PPO_FWD_STORE) {
                        float4* dst = a.act2 + ((size_t)tile * NT + o) * 4 * 64;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            ACT_STORE(dst + q * 64 + lane, make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]));
                    }
                    const float4* w3 = sW3 + (half_o * NT + o) * 16;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float4 w = w3[r];
                        p0 = fmaf(w.x, acc[r], p0); p1 = fmaf(w.y, acc[r], p1);
                        p2 = fmaf(w.z, acc[r], p2); p3 = fmaf(w.w, acc[r], p3);
                    }
                    FSTAMP(3);
                }
            }
            const float l0 = (p0 + __shfl_xor(p0, 32)) + a.b3[0];
            const float l1 = (p1 + __shfl_xor(p1, 32)) + a.b3[1];
            const float l2 = (p2 + __shfl_xor(p2, 32)) + a.b3[2];
            const float l3 = (p3 + __shfl_xor(p3, 32)) + a.b3[3];
            if (TPS == 1) { l[0][0] = l0; l[0][1] = l1; l[0][2] = l2; l[0][3] = l3; }
            else sL[((threadIdx.x >> 6) * TPS + ts) * 64 + lane] = make_float4(l0, l1, l2, l3);
        }
        if (TPS > 1) {
#pragma unroll
            for (int ts = 0; ts < TPS; ++ts) {
                const float4 v = sL[((threadIdx.x >> 6) * TPS + ts) * 64 + lane];   // own lane's values: no barrier needed
                l[ts][0] = v.x; l[ts][1] = v.y; l[ts][2] = v.z; l[ts][3] = v.w;
            }
        }

        FSTAMP(4);
        const int sampled = policy_tail<TMODE, TPS, false>(a, state, sid, act, l, lane, j, h, tick_val, out_index);
        if (MODE == 3) {
            // update!: the observed mask, then step!(env, a), reward, is_terminal (src/collect_rollouts.jl:9-14) and the
            // reset! before the next episode (src/rollout_buffer.jl:75) -- one lane, on the LDS slot
            asm volatile("" ::: "memory");
            if (TPS == 1) {                                  // Q == 8: wavefront-parallel env update (one state byte per lane)
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
            } else if (lane == 0) {
                a.active_out[out_index] = act;
                float rew; uint8_t dn;
                const int errf = env_step_ref(ec, er, sampled, rew, dn);
                if (errf) atomicOr(a.err, errf);
                a.rew_out[out_index] = rew; a.done_out[out_index] = dn;
                if (dn) env_reset_ref(ec, er, (uint32_t)(a.global_offset + state));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        FSTAMP(5);
    }
    }
    if (MODE == 3) {                                            // env state back to the [N] arrays
        int slot2 = 0;
        for (int64_t n = wave; n < a.B; n += nwaves, ++slot2) {
            const EnvRefLds r = slot_ref(slot2);
            for (int v = lane; v < a.envV; v += 64) { a.env_score[n * a.envV + v] = r.sc[v]; a.env_degree[n * a.envV + v] = r.dg[v]; }
            if (lane == 0) {
                a.env_active[n] = *r.active; a.env_steps[n] = *r.steps; a.env_reward[n] = *r.reward;
                a.env_done[n] = *r.done; a.env_episode[n] = *r.episode; a.env_tick[n] = *r.tick;
            }
        }
    }
#ifdef PPO_FWD_STAMP
    if (a.stamps && lane == 0 && wave < 1024)
        for (int i = 0; i < 6; ++i) a.stamps[wave * 6 + i] = st_sum[i];
#endif
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

// num_hidden_layers != 2: the layer-looped instantiations (DEEP = 1).  Dynamic LDS = [MODE 4: one env-snapshot slot per
// wave][park: 4 waves x HID/32 tiles x 4 KiB].
template <int MODE>
static int32_t dispatch_fwd_deep(ppo_policy_s* p, const FwdArgs& args, int64_t B, int tps) {
    const int64_t need = (B + 3) / 4;
    FwdArgs a = args;
    a.nl2 = p->L - 1;
    const size_t snap = (MODE == 4) ? (((size_t)4 * (2 * a.envV + 32) + 15) & ~(size_t)15) : 0;
    a.park_off = (uint32_t)snap;
#define LAUNCHD(FF, HH, TT)                                                                              \
    do {                                                                                                 \
        const int64_t cap = 256 * FwdCfg<FF, HH>::WPS;                                                   \
        const unsigned grid = (unsigned)(need < cap ? need : cap);                                       \
        a.wg_sync = 0;                                                                                   \
        if constexpr (MODE == 4 && FF != 72) {                                                           \
            ppo_set_error("compact rollouts need the built-in env's F = 72"); return PPO_ERR_UNSUPPORTED; \
        } else {                                                                                         \
            const size_t dlds = snap + (size_t)4 * (HH / 32) * 4096;                                     \
            static thread_local size_t attr_lds = 0;                                                                  \
            if (dlds > attr_lds) {                                                                       \
                HIP_TRY(hipFuncSetAttribute((const void*)k_policy_fwd<FF, HH, MODE, TT, 1>,              \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds));     \
                attr_lds = dlds;                                                                         \
            }                                                                                            \
            hipLaunchKernelGGL((k_policy_fwd<FF, HH, MODE, TT, 1>), dim3(grid), dim3(256), dlds, ppo_stream(), a); \
        }                                                                                                \
    } while (0)
    if (p->F == 72 && p->HID == 256 && tps == 1) LAUNCHD(72, 256, 1);
    else if (p->F == 72 && p->HID == 256 && tps == 4) LAUNCHD(72, 256, 4);
    else if (p->F == 72 && p->HID == 128 && tps == 1) LAUNCHD(72, 128, 1);
    else if (p->F == 72 && p->HID == 128 && tps == 4) LAUNCHD(72, 128, 4);
    else if (p->F == 216 && p->HID == 128 && tps == 1) LAUNCHD(216, 128, 1);
    else if (p->F == 216 && p->HID == 256 && tps == 1) LAUNCHD(216, 256, 1);
    else { ppo_set_error("unsupported policy/state shape (F,HID,H) for the gfx950 kernels"); return PPO_ERR_UNSUPPORTED; }
#undef LAUNCHD
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

template <int MODE>
static int32_t dispatch_fwd(ppo_policy_s* p, const FwdArgs& args, int64_t B, int tps) {
    if (p->L != 2) return dispatch_fwd_deep<MODE>(p, args, B, tps);
    const int64_t need = (B + 3) / 4;
    // persistent waves: 256 CUs x WPS blocks of 4 waves (one per SIMD)
#define LAUNCH(FF, HH, TT)                                                                               \
    do {                                                                                                 \
        const int64_t cap = 256 * FwdCfg<FF, HH>::WPS;                                                   \
        const unsigned grid = (unsigned)(need < cap ? need : cap);                                       \
        const_cast<FwdArgs&>(args).wg_sync = (B % ((int64_t)grid * 4) == 0) ? 1 : 0;                     \
        if constexpr (MODE == 4 && FF != 72) {                                                             \
            ppo_set_error("compact rollouts need the built-in env's F = 72"); return PPO_ERR_UNSUPPORTED;    \
        } else {                                                                                             \
            const size_t dlds = (MODE == 4) ? (size_t)4 * (2 * args.envV + 32) : 0;   /* one snapshot slot per wave */ \
            hipLaunchKernelGGL((k_policy_fwd<FF, HH, MODE, TT>), dim3(grid), dim3(256), dlds, ppo_stream(), args); \
        }                                                                                                    \
    } while (0)
    if (p->F == 72 && p->HID == 256 && tps == 1) LAUNCH(72, 256, 1);
    else if (p->F == 72 && p->HID == 256 && tps == 4) LAUNCH(72, 256, 4);
    else if (p->F == 72 && p->HID == 128 && tps == 1) LAUNCH(72, 128, 1);
    else if (p->F == 72 && p->HID == 128 && tps == 4) LAUNCH(72, 128, 4);
    else if (p->F == 216 && p->HID == 128 && tps == 1) LAUNCH(216, 128, 1);
    else if (p->F == 216 && p->HID == 256 && tps == 1) LAUNCH(216, 256, 1);
    else { ppo_set_error("unsupported policy/state shape (F,HID,H) for the gfx950 kernels"); return PPO_ERR_UNSUPPORTED; }
#undef LAUNCH
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

#ifdef PPO_FWD_STAMP
static unsigned long long* g_fwd_stamps = nullptr;
extern "C" int32_t ppo_debug_fwd_stamps(unsigned long long* out) {
    if (!g_fwd_stamps) return -1;
    (void)hipDeviceSynchronize();
    return hipMemcpy(out, g_fwd_stamps, 1024 * 6 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

static void fill_weights(ppo_policy_s* p, FwdArgs& a) {
    a.stamps = nullptr;
#ifdef PPO_FWD_STAMP
    if (!g_fwd_stamps) (void)hipMalloc((void**)&g_fwd_stamps, 1024 * 6 * 8);
    a.stamps = g_fwd_stamps;
#endif
    a.w1p = (const float4*)p->w1p.p; a.w2p = (const float4*)p->w2p.p; a.b1p = (const float4*)p->b1p.p;
    a.b2p = (const float4*)p->b2p.p; a.w3p = (const float4*)p->w3p.p; a.b3 = p->b3.p;
}

int32_t launch_policy_probs(ppo_policy_s* p, const int8_t* states_dev, const uint32_t* active_dev, int64_t B,
                            int32_t H, float* probs_dev) {
    if (B <= 0) return PPO_OK;
    FwdArgs a = {};
    fill_weights(p, a);
    a.states = states_dev; a.active = active_dev; a.B = B; a.probs_out = probs_dev;
    ProfScope ps("k_policy_fwd_probs");
    if (p->dtype == PPO_DTYPE_BF16) return launch_policy_fwd_bf16(p, a, 0, B, H / 32);
    return dispatch_fwd<0>(p, a, B, H / 32);
}

// rollouts of up to this many envs give every env to 2 or 4 waves; ppo_set_rollout_split_max_envs / PPO_ROLLOUT_SPLIT_MAX_ENVS
static int64_t g_rollout_split_max_envs = [] { const char* v = std::getenv("PPO_ROLLOUT_SPLIT_MAX_ENVS"); return v ? (int64_t)atoll(v) : (int64_t)512; }();
extern "C" int32_t ppo_set_rollout_split_max_envs(int64_t envs) { g_rollout_split_max_envs = envs < 0 ? 512 : envs; return PPO_OK; }

int32_t launch_policy_rollout(ppo_policy_s* p, ppo_env_s* e, const int8_t* states_dev, const uint32_t* active_dev,
                              int32_t* actions_out, float* psel_out, float* full_probs_or_null) {
    FwdArgs a = {};
    fill_weights(p, a);
    a.states = states_dev; a.active = active_dev; a.B = e->N;
    a.tick = e->tick.p; a.global_offset = e->global_offset; a.k0 = (uint32_t)e->seed; a.k1 = (uint32_t)(e->seed >> 32);
    a.actions_out = actions_out; a.psel_out = psel_out; a.full_probs = full_probs_or_null; a.err = e->err.p;
    ProfScope ps("k_policy_fwd_rollout");
    if (p->dtype == PPO_DTYPE_BF16) return launch_policy_fwd_bf16(p, a, 1, e->N, e->H / 32);
    if (e->N <= g_rollout_split_max_envs) {   // few envs: 2 or 4 waves per env, same bits
        const int32_t rs = launch_rollout_split(p, a, e->N, e->H / 32, e->V, 0);
        if (rs != PPO_ERR_UNSUPPORTED) return rs;
    }
    return dispatch_fwd<1>(p, a, e->N, e->H / 32);
}

// Persistent rollout (MODE 3): T steps of all N envs in one launch.  Returns PPO_ERR_UNSUPPORTED (without setting an
// error) when the shape is not covered, so the caller falls back to the per-step launches.
// t0: first row of the rollout columns this launch writes (streaming collects a long rollout as a chain of launches)
int32_t launch_policy_rollout_persistent(ppo_policy_s* p, ppo_env_s* e, ppo_rollouts_s* ro, int64_t T, int record_probs,
                                         int64_t t0) {
    if (p->F != 72 || e->F != 72 || p->L != 2) return PPO_ERR_UNSUPPORTED;
    const int tps = e->H / 32;
    const int64_t N = e->N;
    const int64_t need = (N + 3) / 4;
    const int wps = (p->HID >= 256) ? 1 : PPO_FWD_WPS_SMALL;     // FwdCfg<72, HID>::WPS
    const int64_t cap = 256 * wps;
    const unsigned grid = (unsigned)(need < cap ? need : cap);
    const int slots = (int)((N + (int64_t)grid * 4 - 1) / ((int64_t)grid * 4));
    const size_t lds = (size_t)4 * slots * (2 * e->V + 32);
    if (p->dtype == PPO_DTYPE_F32 && lds > 96 * 1024) return PPO_ERR_UNSUPPORTED;
    FwdArgs a = {};
    fill_weights(p, a);
    a.B = N; a.T = T;
    a.global_offset = e->global_offset; a.k0 = (uint32_t)e->seed; a.k1 = (uint32_t)(e->seed >> 32);
    a.env_score = e->score.p; a.env_degree = e->degree.p; a.env_active = e->active.p; a.env_steps = e->steps.p;
    a.env_reward = e->reward.p; a.env_done = e->done.p; a.env_episode = e->episode.p; a.env_tick = e->tick.p;
    a.env_tmpl = e->tmpl.p; a.envQ = e->Q; a.envV = e->V; a.env_max_actions = e->max_actions; a.env_slots = slots;
    a.env_nar = e->no_action_reward; a.err = e->err.p;
    const size_t r0 = (size_t)t0 * N;                                   // transitions in front of this launch's rows
    a.states_out = ro->compact ? nullptr : ro->states.p + r0 * e->H * e->F;
    a.cstate_out = ro->compact ? ro->cstate.p + r0 * 2 * e->V : nullptr;
    a.active_out = ro->active.p + r0; a.actions_out = ro->actions.p + r0; a.psel_out = ro->p_sel.p + r0;
    a.rew_out = ro->rewards.p + r0; a.done_out = ro->done.p + r0;
    a.full_probs = record_probs ? ro->full_probs.p + r0 * e->A : nullptr;
    ProfScope ps("k_rollout_persistent");
    if (p->dtype == PPO_DTYPE_BF16) return launch_policy_rollout_persistent_bf16(p, a, N, tps, e->V);
    if (N <= g_rollout_split_max_envs) {      // few envs: 2 or 4 waves per env, same bits (ppo_policy_rollout_split.hip)
        const int32_t rs = launch_rollout_split(p, a, N, tps, e->V, 1);
        if (rs != PPO_ERR_UNSUPPORTED) return rs;
        a.env_slots = slots;
    }
#define LAUNCH3(HH, TT)                                                                                      \
    do {                                                                                                     \
        static thread_local size_t attr_lds = 0;                                                                          \
        if (lds > attr_lds) {                                                                                \
            HIP_TRY(hipFuncSetAttribute((const void*)k_policy_fwd<72, HH, 3, TT>,                            \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));              \
            attr_lds = lds;                                                                                  \
        }                                                                                                    \
        hipLaunchKernelGGL((k_policy_fwd<72, HH, 3, TT>), dim3(grid), dim3(256), lds, ppo_stream(), a);      \
    } while (0)
    if (p->HID == 256 && tps == 1) LAUNCH3(256, 1);
    else if (p->HID == 256 && tps == 4) LAUNCH3(256, 4);
    else if (p->HID == 128 && tps == 1) LAUNCH3(128, 1);
    else if (p->HID == 128 && tps == 4) LAUNCH3(128, 4);
    else return PPO_ERR_UNSUPPORTED;
#undef LAUNCH3
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

// minibatches of up to this many states take the split train forward; ppo_set_fwd_split_max_states / PPO_FWD_SPLIT_MAX_STATES
static int64_t g_fwd_split_max_states = [] { const char* v = std::getenv("PPO_FWD_SPLIT_MAX_STATES"); return v ? (int64_t)atoll(v) : (int64_t)512; }();
extern "C" int32_t ppo_set_fwd_split_max_states(int64_t states) { g_fwd_split_max_states = states < 0 ? 512 : states; return PPO_OK; }

int32_t launch_policy_train_fwd(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B,
                                int64_t B_global, double eps, double entropy_weight, const float* adv_col) {
    FwdArgs a = {};
    fill_weights(p, a);
    a.states = ro->states.p; a.active = ro->active.p; a.idx = idx_dev; a.B = B;
    a.act1 = (float4*)p->act1.p; a.act2 = (float4*)p->act2.p; a.dY = (float4*)p->dY.p; a.loss_terms = p->loss_terms.p;
    for (int l = 0; l < 2; ++l)          // deep policies: the hidden layers between the first and the last
        a.act_mid[l] = (p->L > 2 && l < p->L - 2) ? (float4*)p->actm.p + (size_t)l * p->cap_tiles * (p->HID / 32) * 256 : nullptr;
    a.actions = ro->actions.p; a.p_old = ro->p_sel.p; a.adv = adv_col;
    a.eps = eps; a.c_over_B = (float)(entropy_weight / (double)B_global); a.inv_B = (float)(1.0 / (double)B_global);
    ProfScope ps("k_policy_fwd_train");
    if (ro->compact) {      // env snapshots instead of observation rows (MODE 4 / the CS form of the split kernel)
        a.states = nullptr; a.cstate = ro->cstate.p; a.xs_out = p->xs.p;
        a.env_tmpl = ro->tmpl.p; a.envV = ro->V; a.envQ = ro->V / 4; a.env_slots = 1;
    }
    {   // Dense products as split-fp32 MFMAs on the bf16 pipe (ppo_policy_fwd_x6.hip: fp32 Policy(72, h, 2, 4), expanded states)
        const int32_t rx = launch_policy_train_fwd_x6(p, a, B, ro->H / 32, ro->compact);
        if (rx != PPO_ERR_UNSUPPORTED) return rx;
    }
    if (B <= g_fwd_split_max_states) {        // small minibatch: 2 or 4 waves per state (ppo_policy_fwd_split.hip)
        const int32_t rs = launch_policy_train_fwd_split(p, a, B, ro->H / 32, ro->compact);
        if (rs != PPO_ERR_UNSUPPORTED) return rs;
    }
    if (ro->compact) {      // rows are re-derived from the env snapshots and left in p->xs (minibatch order) for the backward
        if (p->dtype == PPO_DTYPE_BF16) return launch_policy_fwd_bf16(p, a, 4, B, ro->H / 32);
        return dispatch_fwd<4>(p, a, B, ro->H / 32);
    }
    if (p->dtype == PPO_DTYPE_BF16) return launch_policy_fwd_bf16(p, a, 2, B, ro->H / 32);
    return dispatch_fwd<2>(p, a, B, ro->H / 32);
}

int32_t launch_categorical(const float* probs, const float* u, int64_t B, int64_t A, int32_t* actions, float* psel,
                           int32_t* err) {
    if (B <= 0) return PPO_OK;
    hipLaunchKernelGGL(k_categorical, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, ppo_stream(), probs, u, B, A,
                       actions, psel, err);
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}
