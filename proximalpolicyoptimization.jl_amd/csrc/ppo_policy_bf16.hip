// ppo_policy_bf16.hip -- bf16 compute mode of the policy MLP (BASELINE config 5: "bf16 MLP on MFMA + fp32 GAE").
//
// Same reference ops as ppo_policy_fwd.hip / ppo_policy_bwd.hip (action_probabilities / batch_action_probabilities,
// test/quad_game_utilities.jl:65-79 over SimplePolicy, test/policy.jl:9-31; Zygote gradient of step_batch!,
// src/train.jl:54-84), with the three Dense contractions on v_mfma_f32_32x32x16_bf16:
//   weights and layer inputs are bfloat16 (RNE from the fp32 master parameters / fp32 activations), products are
//   exact, accumulation, bias, leakyrelu, softmax, sampling, loss and Adam are fp32 as in the fp32 mode.
//   Saved activations are bf16; the backward signals (dY, dZ2, dZ1) are rounded to bf16 before they enter an MFMA.
// The reference is Float32 only: this mode has no reference counterpart beyond "same function at lower precision";
// the tests check it against a CPU restatement with the same rounding points and float64 accumulation (tolerances
// stated there).
//
// gfx950 mapping, forward: one wave per 32-row tile, Y^T = W * X^T as in the fp32 kernel, so the 32x32 accumulators
// of a layer (row on the lane, 16 features in registers) become the B operands of the next layer after one
// v_cvt_pk_bf16_f32 per register pair (k order permuted: element j of lane half h of k-step s is feature
// 16s + 8(j>>2) + 4h + (j&3); the packed weight fragments carry the same permutation).  A bf16 MFMA retires
// 16x the flops of the fp32 one per cycle, so the weight stream has to come from LDS: W2 (128 KiB as bf16 for
// HID = 256) is staged once per workgroup, one ds_read_b128 per MFMA; W1 (36 KiB) streams through L1; layer 3
// (HID x 4) is two more MFMAs per feature tile with a zero-padded A operand instead of 64 VALU FMAs.
// Backward (two kernels, see BwdB below): a workgroup of 4 waves walks tiles; wave w owns feature tiles w*FT.. of dW2
// (accumulators resident all launch).  The lane-is-row tiles (dZ2, H1, dZ1, H2) are written once as row-major
// bf16 images into LDS and come back TRANSPOSED through ds_read_b64_tr_b16 as the operands of the products that
// contract over the 32 rows (dW2, dW3; dZ1 and X are emitted as fragments for the dW1 kernel); dH1 = W2^T dZ2
// takes dZ2 lane-linear in accumulator-fragment order.
#include "ppo_policy_tail.h"
#include "ppo_env_device.h"

#ifndef PPO_BF16_ACT_NT
#define PPO_BF16_ACT_NT 1
#endif
#ifndef PPO_BF16_TMPL_LDS
#define PPO_BF16_TMPL_LDS 1
#endif
#ifndef PPO_BF16_RING_SPREAD
#define PPO_BF16_RING_SPREAD 1
#endif
#ifndef PPO_BF16_STORE_LATE
#define PPO_BF16_STORE_LATE 1
#endif
#ifndef PPO_BF16_ACT_NT_LOAD
#define PPO_BF16_ACT_NT_LOAD 1
#endif
// dW1 += dZ1^T X inside k_policy_bwd_bf16 (NI more accumulator tiles per feature tile of the wave; the dZ1 / X operand
// fragments never leave the CU) where the registers allow it: HID = 128 (168 + 112 registers, no scratch).  At HID = 256
// the wave already holds 256 dW2 accumulator registers and a 236-register working set: with the 96 dW1 registers hipcc
// spills 67 (268 B of scratch per lane), so that width keeps the round-1 form: fragments to HBM + k_policy_dw1_bf16.
#ifndef PPO_BF16_DW1_FUSED_MAX_HID
#define PPO_BF16_DW1_FUSED_MAX_HID 128
#endif
#define PPO_BF16_DW1_FUSED (HID <= PPO_BF16_DW1_FUSED_MAX_HID)
// The backward kernel recomputes H1 = lrelu(W1 X + b1) of its feature tiles (KS1 MFMAs per tile from the 2.3 KB of
// state rows + L2-resident W1 fragments) instead of reading what the train forward saved: 16 KB per tile less written
// by the forward and 16 KB less read here -- a third of the training step's HBM bytes.  Same operand order as the
// forward, so the recomputed tile is bit-identical to the one that was saved.
#ifndef PPO_BF16_H1_RECOMPUTE
#define PPO_BF16_H1_RECOMPUTE 1
#endif
// Where dW1 is NOT accumulated in the backward kernel (HID = 256), the two wave-private transposes -- H2 for the dW3 sums,
// dZ1 for the fragments handed to k_policy_dw1_bf16 -- run as identity MFMAs on the (70 % idle) matrix pipe instead of an LDS
// image write + transposed read: D = P * I puts the rows of the lane-is-row fragment P on the accumulator registers of
// the lane that owns the feature.  No H2 / dZ1 image: 18 KB of LDS back (two more resident W2^T k-steps).
#ifndef PPO_BF16_MFMA_TRANSPOSE
#define PPO_BF16_MFMA_TRANSPOSE 1
#endif
// identity-MFMA form: the dH1 epilogue in front of phase C (0) or inside it, two accumulator registers behind the dW2 MFMAs
// of each k-tile (1).  Measured 2 % SLOWER (0.940 -> 0.958 ms at 65,536 states, gpurun_out/r2b14, r2b15; with and without
// scheduling pins): unlike the forward's chains, phase C already overlaps its transposed reads with its MFMAs, and
// carrying the dH1 accumulator through it costs more than the vector work it hides.  Kept as an A/B switch.
#ifndef PPO_BF16_BWD_EPI_IN_C
#define PPO_BF16_BWD_EPI_IN_C 0
#endif
#ifndef PPO_BF16_BWD_DL8
#define PPO_BF16_BWD_DL8 2            // LDS operand queue depth of the dH1 chain in the 8-wave form (the partner wave covers the rest)
#endif
#ifndef PPO_BF16_BWD_WAVES
#define PPO_BF16_BWD_WAVES 8          // waves per backward workgroup at HID = 256 (4 = one per SIMD, the round-1 form)
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {          // v_cvt_pk_bf16_f32 (RNE)
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(uint32_t d) { return __uint_as_float(d << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t d) { return __uint_as_float(d & 0xFFFF0000u); }
__device__ __forceinline__ f32x16 mfma_bf16(const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// leakyrelu(x) = max(x, 0.01x) for slope < 1 (same value as x > 0 ? x : 0.01x for every finite x)
__device__ __forceinline__ float lrelu_max(float x) { return fmaxf(x, 0.01f * x); }
// accumulator tile -> two B/A operand fragments (k-steps 0 and 1) of the next product
__device__ __forceinline__ void pack_tile(const f32x16& acc, uint4 (&out)[2]) {
    out[0] = make_uint4(pack_bf16(acc[0], acc[1]), pack_bf16(acc[2], acc[3]), pack_bf16(acc[4], acc[5]), pack_bf16(acc[6], acc[7]));
    out[1] = make_uint4(pack_bf16(acc[8], acc[9]), pack_bf16(acc[10], acc[11]), pack_bf16(acc[12], acc[13]), pack_bf16(acc[14], acc[15]));
}
// leakyrelu'(h) for the two bf16 values of a packed dword: h > 0 <=> the bit pattern read as int16 is > 0
__device__ __forceinline__ float slope_lo(uint32_t d) { return (int16_t)(d & 0xFFFFu) > 0 ? 1.0f : 0.01f; }
__device__ __forceinline__ float slope_hi(uint32_t d) { return (int32_t)d >= 0x10000 ? 1.0f : 0.01f; }
// saved activations are written once and read once by the backward pass much later: non-temporal stores keep them from
// churning the L2 (same as the fp32 forward kernel)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void act_store_nt_u4(uint4* p, const uint4& v) {
#if PPO_BF16_ACT_NT
    const u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ uint32_t dw(const uint4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// ================================================================ forward
// waves per forward workgroup (one workgroup per CU: W2 takes 128 KiB of its LDS): 8 = two per SIMD with 256 registers,
// 12 = three per SIMD with 168
#ifndef PPO_BF16_FWD_WQ
#define PPO_BF16_FWD_WQ 4             // W2 fragments in flight (LDS -> registers) ahead of the layer-2 MFMAs
#endif
// layer-2 epilogues software-pipelined into the next feature tile's MFMA chain: 1 = where the second accumulator costs at
// most 4 registers of scratch (every H = 32 instantiation -- the persistent rollout and the compact-source forward spill
// 16 bytes and still gain 4-6 % -- and the H = 128 ones without a loss tail), 2 = everywhere (A/B), 0 = off
#ifndef PPO_BF16_FWD_PIPE
#define PPO_BF16_FWD_PIPE 1
#endif
#ifndef PPO_BF16_FWD_WAVES
#define PPO_BF16_FWD_WAVES 8
#endif
#define FWB_W PPO_BF16_FWD_WAVES
#define FWB_T (PPO_BF16_FWD_WAVES * 64)
template <int F, int HID>
struct FwdB {
    static constexpr int NT = HID / 32, NS = HID / 16, KS1 = (F + 15) / 16;
    static constexpr int W2_U4 = NT * NS * 64;            // uint4 elements of the staged W2 fragments
    static constexpr int W3_U4 = NS * 8 + 1;              // compact layer-3 rows + one zero block
    static constexpr size_t lds_bytes = (size_t)(W2_U4 + W3_U4) * 16 + (size_t)NT * 8 * 16 * 2;
};

template <int F, int HID, int MODE, int TPS>
__global__ __launch_bounds__(FWB_T, (FWB_W / 4)) void k_policy_fwd_bf16(FwdArgs a) {
    using C = FwdB<F, HID>;
    constexpr int NT = C::NT, NS = C::NS, KS1 = C::KS1;
    constexpr bool TRAIN = (MODE == 2 || MODE == 4);    // train forward: saves activations, loss tail
    constexpr bool OBS = (MODE == 3 || MODE == 4);      // rows re-derived from an env snapshot in LDS (MODE 4: compact rollouts)
    constexpr int TMODE = TRAIN ? 2 : MODE;
    static_assert(F % 8 == 0 && HID % 32 == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) uint4 smem_u4[];
    uint4* const sW2 = smem_u4;
    uint4* const sW3 = sW2 + C::W2_U4;
    float4* const sB1 = reinterpret_cast<float4*>(sW3 + C::W3_U4);
    float4* const sB2 = sB1 + NT * 8;
    char* const env_lds = reinterpret_cast<char*>(sB2 + NT * 8);     // MODE 3: env slots of the 8 waves
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PPO_BF16_STAMP
    unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0}, st_t = clock64();
#define FBSTAMP(i) do { unsigned long long _n = clock64(); st_sum[i] += _n - st_t; st_t = _n; } while (0)
#else
#define FBSTAMP(i) do {} while (0)
#endif
    {   // unrolled: all the loads are in flight before the first LDS write waits for one (the rolled loop was 4 us slower)
        constexpr int NCH = (C::W2_U4 + FWB_T - 1) / FWB_T;
#pragma unroll
        for (int k = 0; k < NCH; ++k) if (C::W2_U4 % FWB_T == 0 || k * FWB_T + tid < C::W2_U4) sW2[k * FWB_T + tid] = a.w2b[k * FWB_T + tid];
    }
    for (int i = tid; i < NS * 8; i += FWB_T) sW3[i] = a.w3c[i];
    if (tid == 0) sW3[NS * 8] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < NT * 8; i += FWB_T) { sB1[i] = a.b1p[i]; sB2[i] = a.b2p[i]; }
    __syncthreads();
    // layer-3 A operand: rows 0..3 of the 32-row operand tile are W3, the rest zero (lanes j >= 4 read the zero block)
    const int w3_lane = (j < 4) ? (h * 4 + j) : -1;

    // ---- MODE 3: persistent rollout (see k_policy_fwd): every wave walks its envs through all T steps; W2 is staged in
    // LDS once per ROLLOUT instead of once per step.  Env state lives in the wave's LDS slots.
    const int64_t wave0 = (int64_t)blockIdx.x * FWB_W + w, nwaves = (int64_t)gridDim.x * FWB_W;
    const int slot_bytes = 2 * a.envV + 32;
    char* const my_slots = env_lds + (size_t)w * a.env_slots * slot_bytes;
    EnvConst ec = {};
    auto slot_ref = [&](int slot) {
        PPO_LDS char* b = (PPO_LDS char*)(my_slots + (size_t)slot * slot_bytes);
        EnvRefLds r;
        r.sc = (PPO_LDS int8_t*)b; r.dg = r.sc + a.envV;
        PPO_LDS uint32_t* ww = (PPO_LDS uint32_t*)(b + 2 * a.envV);
        r.active = ww; r.steps = (PPO_LDS int32_t*)(ww + 1); r.reward = (PPO_LDS float*)(ww + 2);
        r.done = (PPO_LDS uint8_t*)(ww + 3); r.episode = ww + 4; r.tick = ww + 5;
        return r;
    };
    uint32_t tmpl_regs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (MODE == 3) {
        ec.Q = a.envQ; ec.V = a.envV; ec.max_actions = a.env_max_actions; ec.no_action_reward = a.env_nar; ec.k0 = a.k0; ec.k1 = a.k1;
        int slot = 0;
        for (int64_t n = wave0; n < a.B; n += nwaves, ++slot) {
            const EnvRefLds r = slot_ref(slot);
            for (int v = lane; v < a.envV; v += 64) { r.sc[v] = a.env_score[n * a.envV + v]; r.dg[v] = a.env_degree[n * a.envV + v]; }
            if (lane == 0) {
                *r.active = a.env_active[n]; *r.steps = a.env_steps[n]; *r.reward = a.env_reward[n];
                *(PPO_LDS uint32_t*)r.done = a.env_done[n]; *r.episode = a.env_episode[n]; *r.tick = a.env_tick[n];
            }
        }
    }
    if (OBS) {
        if (TPS == 1) {
#if PPO_BF16_TMPL_LDS
            // the 32 template rows (36 ids each) sit behind the env slots in LDS: nine registers less to carry through
            // the tile loop (they spilled, and every scratch reload waits on vmcnt behind the rollout-column stores)
            uint32_t* tl = reinterpret_cast<uint32_t*>(env_lds + (size_t)FWB_W * a.env_slots * slot_bytes);
            for (int i = tid; i < 32 * PPO_TPL / 4; i += FWB_T) tl[i] = reinterpret_cast<const uint32_t*>(a.env_tmpl)[i];
            __syncthreads();
#else
            const uint32_t* tp = reinterpret_cast<const uint32_t*>(a.env_tmpl + j * PPO_TPL);
#pragma unroll
            for (int k = 0; k < 9; ++k) tmpl_regs[k] = tp[k];
#endif
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // MODE 4: env snapshot of a transition fetched through idx one state ahead (see k_policy_fwd)
    uint32_t cs_next = 0u, act_next = 0u;
    int32_t sid_next = 0;
    auto fetch_snapshot = [&](int64_t state) {
        sid_next = a.idx[state];
        act_next = a.active[sid_next];
        const int nd = a.envV >> 1;
        cs_next = (lane < nd) ? reinterpret_cast<const uint32_t*>(a.cstate)[(size_t)sid_next * nd + lane] : 0u;
    };
    if (MODE == 4 && wave0 < a.B) fetch_snapshot(wave0);
    const int64_t t_steps = (MODE == 3) ? a.T : 1;
    FBSTAMP(0);
    for (int64_t tstep = 0; tstep < t_steps; ++tstep) {
    int slot = 0;
    for (int64_t state = wave0; state < a.B; state += nwaves, ++slot) {
        const int64_t sid = (MODE == 2) ? (int64_t)a.idx[state] : ((MODE == 4) ? (int64_t)sid_next : state);
        EnvRefLds er = {};
        if (MODE == 3) er = slot_ref(slot);
        if (MODE == 4) {
            er = slot_ref(0);
            if (lane < (a.envV >> 1)) reinterpret_cast<PPO_LDS uint32_t*>(er.sc)[lane] = cs_next;
            if (lane == 0) *er.active = act_next;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        const uint32_t act = (MODE == 3) ? *er.active : ((MODE == 4) ? act_next : a.active[sid]);
        if (MODE == 4) fetch_snapshot(state + nwaves < a.B ? state + nwaves : state);
        const uint32_t tick_val = (MODE == 3) ? *er.tick : ((MODE == 1) ? a.tick[state] : 0u);
        const int64_t out_index = (MODE == 3) ? tstep * a.B + state : state;
        float l[TPS][4];
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) l[tt][0] = l[tt][1] = l[tt][2] = l[tt][3] = 0.0f;
        TailPre tpre = {0, 0.0f, 0.0f};
        if (TRAIN && PPO_BF16_STORE_LATE) { tpre.ab = a.actions[sid]; tpre.po = a.p_old[sid]; tpre.adv = a.adv[sid]; }
#pragma unroll 1
        for (int ts = 0; ts < TPS; ++ts) {
            const int64_t tile = state * TPS + ts;
            // ---- state rows -> layer-1 B operands: lane (row j, half h) holds features 16s + 8h .. +7 of k-step s
            const int8_t* row = OBS ? nullptr : a.states + ((size_t)sid * TPS + ts) * 32 * F + (size_t)j * F;
            uint32_t ob[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};       // MODE 3 / 4: this lane's 36 observed features (half h of row j)
            if (OBS) {
                static_assert(!OBS || F == 72, "the built-in env has F = 72 features");
                uint32_t tid9[9];
                if (TPS == 1) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) tid9[k] = tmpl_regs[k];
#if PPO_BF16_TMPL_LDS
                    const uint32_t* tl = reinterpret_cast<const uint32_t*>(env_lds + (size_t)FWB_W * a.env_slots * slot_bytes) + j * (PPO_TPL / 4);
#pragma unroll
                    for (int k = 0; k < 9; ++k) tid9[k] = tl[k];
#endif
                } else {
                    const uint32_t* tp = reinterpret_cast<const uint32_t*>(a.env_tmpl + (32 * ts + j) * PPO_TPL);
#pragma unroll
                    for (int k = 0; k < 9; ++k) tid9[k] = tp[k];
                }
                env_observe_lane(er, tid9, 32 * ts + j, h, ob);
                int8_t* const rows_out = (MODE == 4) ? a.xs_out : a.states_out;
                if (rows_out) {                                  // wave-uniform
                    uint32_t* so = reinterpret_cast<uint32_t*>(rows_out + ((size_t)(MODE == 4 ? state : out_index) * TPS + ts) * 32 * F + (size_t)j * F + (size_t)h * 36);
#pragma unroll
                    for (int k = 0; k < 9; ++k) so[k] = ob[k];
                }
                if (MODE == 3 && ts == 0 && a.cstate_out && lane < (a.envV >> 1))      // compact storage: the env snapshot itself
                    reinterpret_cast<uint32_t*>(a.cstate_out)[(size_t)out_index * (a.envV >> 1) + lane] =
                        reinterpret_cast<PPO_LDS uint32_t*>(er.sc)[lane];
            }
            uint4 xs[KS1];
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                const int off = 16 * s + 8 * h;
                uint2 d = make_uint2(0u, 0u);
                if (OBS) {
                    // layer-1 k-step s wants features [16s + 8h, +8) of row j; the row's features sit 36 per lane half
                    // (lane j: 0..35, lane j + 32: 36..71) as 9 dwords each: pull the two dwords from the lane half that
                    // observed them.  Source half and dword index are compile-time per (s, destination half).
                    uint32_t lo[2] = {0u, 0u}, hi[2] = {0u, 0u};
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int o0 = 16 * s + 8 * hh, o4 = o0 + 4;
                        if (o0 < F) {
                            const int lh = o0 >= 36 ? 1 : 0, hh4 = o4 >= 36 ? 1 : 0;
                            lo[hh] = __shfl(ob[(o0 - 36 * lh) >> 2], j + 32 * lh);
                            hi[hh] = __shfl(ob[(o4 - 36 * hh4) >> 2], j + 32 * hh4);
                        }
                    }
                    d = make_uint2(h ? lo[1] : lo[0], h ? hi[1] : hi[0]);
                } else if (off < F) d = *reinterpret_cast<const uint2*>(row + off);
                float f[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f[i] = (float)(int)(int8_t)(d.x >> (8 * i));
                    f[4 + i] = (float)(int)(int8_t)(d.y >> (8 * i));
                }
                xs[s] = make_uint4(pack_bf16(f[0], f[1]), pack_bf16(f[2], f[3]), pack_bf16(f[4], f[5]), pack_bf16(f[6], f[7]));
            }
            FBSTAMP(1);
            // ---- layer 1 (W1 fragments through L1: 1 KiB per wave load, PF1 of them in flight in a register ring;
            // sched_barrier keeps hipcc from hoisting the whole 40-fragment stream into registers)
            uint4 h1p[NT][2];
            {
                constexpr int NG = NT * KS1, PF1 = (NG < 6) ? NG : 6;      // 12 measured slower
                const uint4* wp = a.w1b + lane;
                uint4 ring[PF1];
#pragma unroll
                for (int g = 0; g < PF1; ++g) ring[g] = wp[(size_t)g * 64];
                auto bias1 = [&](f32x16& acc, int o) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 b = sB1[(o * 2 + h) * 4 + q];
                        acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
                    }
                };
                auto step1 = [&](f32x16& acc, int o, int s) {
                    const int g = o * KS1 + s;
                    const uint4 wv = ring[g % PF1];
                    if (g + PF1 < NG) ring[g % PF1] = wp[(size_t)(g + PF1) * 64];
                    acc = mfma_bf16(wv, xs[s], acc);
                };
                // a quarter of a tile's epilogue: leakyrelu + RNE pack of accumulator registers 4p .. 4p+3 -> two dwords of
                // the next layer's B operand (same values and order as lrelu16 + pack_tile)
                auto epi1 = [&](f32x16& acc, int o, int p4) {
                    const pk2f x0 = {acc[4 * p4], acc[4 * p4 + 1]}, x1 = {acc[4 * p4 + 2], acc[4 * p4 + 3]};
                    const pk2f y0 = x0 * pk2f{0.01f, 0.01f}, y1 = x1 * pk2f{0.01f, 0.01f};
                    float r0, r1, r2, r3;
                    asm("v_max_f32 %0, %1, %2" : "=v"(r0) : "v"(x0.x), "v"(y0.x));
                    asm("v_max_f32 %0, %1, %2" : "=v"(r1) : "v"(x0.y), "v"(y0.y));
                    asm("v_max_f32 %0, %1, %2" : "=v"(r2) : "v"(x1.x), "v"(y1.x));
                    asm("v_max_f32 %0, %1, %2" : "=v"(r3) : "v"(x1.y), "v"(y1.y));
                    const uint32_t d0 = pack_bf16(r0, r1), d1 = pack_bf16(r2, r3);
                    if (p4 & 1) { h1p[o][p4 >> 1].z = d0; h1p[o][p4 >> 1].w = d1; }
                    else        { h1p[o][p4 >> 1].x = d0; h1p[o][p4 >> 1].y = d1; }
                };
                static_assert(PPO_BF16_STORE_LATE, "the layer-1 stores leave behind the loop");
                if constexpr (PPO_BF16_FWD_PIPE && KS1 >= 4) {
                    // the epilogue of tile o in four pieces behind the first four MFMAs of tile o + 1 (a bf16 MFMA leaves the
                    // vector ALU free for the 32 clocks it occupies the matrix pipe)
                    f32x16 accs[2];
                    bias1(accs[0], 0);
#pragma unroll
                    for (int s = 0; s < KS1; ++s) step1(accs[0], 0, s);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int o = 0; o + 1 < NT; ++o) {
                        bias1(accs[(o + 1) & 1], o + 1);
#pragma unroll
                        for (int s = 0; s < KS1; ++s) {
                            step1(accs[(o + 1) & 1], o + 1, s);
                            if (s < 4) epi1(accs[o & 1], o, s);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#pragma unroll
                    for (int p4 = 0; p4 < 4; ++p4) epi1(accs[(NT - 1) & 1], NT - 1, p4);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
#pragma unroll
                for (int o = 0; o < NT; ++o) {
                    f32x16 acc;
                    bias1(acc, o);
#pragma unroll
                    for (int s = 0; s < KS1; ++s) step1(acc, o, s);
#pragma unroll
                    for (int p4 = 0; p4 < 4; ++p4) epi1(acc, o, p4);
                    __builtin_amdgcn_sched_barrier(0);
                }
                }
                // saved layer-1 activations leave AFTER the last W1 fragment has been waited for: loads and stores share
                // the in-order vmcnt queue, so a store between two ring loads puts its HBM round trip on the MFMA chain
                // (layer 2 takes its operands from LDS and never waits on vmcnt)
                if (TRAIN && PPO_BF16_STORE_LATE && a.act1b) {        // (null: the backward kernel recomputes H1)
#pragma unroll
                    for (int o = 0; o < NT; ++o) {
                        act_store_nt_u4(a.act1b + ((size_t)tile * NT + o) * 128 + lane, h1p[o][0]);
                        act_store_nt_u4(a.act1b + ((size_t)tile * NT + o) * 128 + 64 + lane, h1p[o][1]);
                    }
                }
            }
            FBSTAMP(2);
            // ---- layer 2 (W2 fragments from LDS) + layer 3 (two MFMAs per feature tile)
            f32x16 acc3;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[r] = 0.0f;
            // W2 fragments are read from LDS WQ k-steps ahead of their MFMAs through an explicit queue that runs across the
            // feature tiles (left alone hipcc keeps two fragments in flight and waits for each one a single 32-clock MFMA
            // after issuing its read; sched_barrier pins the order).  The last tile's look-ahead reads the first
            // fragments of what follows W2 in LDS (W3 / bias packs: in bounds, never used).
            constexpr int WQ = (MODE == 4 && TPS == 4) ? 2 : PPO_BF16_FWD_WQ;      // (that instantiation has no registers to spare)
            uint4 wq[WQ];
#pragma unroll
            for (int k = 0; k < WQ; ++k) wq[k] = sW2[k * 64 + lane];
            // One step of a feature tile's chain: k-step k of tile o into `acc` (queue slot refilled WQ steps ahead).
            auto chain_step = [&](f32x16& acc, int o, int k) {
                const uint4* wo = sW2 + (size_t)o * NS * 64 + lane;
                const uint4 wv = wq[k % WQ];
                wq[k % WQ] = wo[(k + WQ) * 64];
                __builtin_amdgcn_sched_barrier(0);
                acc = mfma_bf16(wv, h1p[k >> 1][k & 1], acc);
            };
            auto bias_init = [&](f32x16& acc, int o) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = sB2[(o * 2 + h) * 4 + q];
                    acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
                }
            };
            // The epilogue of a feature tile (leakyrelu, RNE pack, activation store, the two layer-3 MFMAs) cut into 16
            // slices, one per k-step of the NEXT tile's chain: a bf16 MFMA occupies the matrix pipe for 32 clocks and, unlike
            // the fp32 one, leaves the vector ALU free, so a slice issued right behind an MFMA runs in its shadow.  Values and
            // operation order per value are those of the one-piece epilogue.
            uint32_t hp[8];                                       // the packed H2 tile being assembled (two B operands)
            uint4 w3q;                                            // layer-3 A operand, read two slices ahead of its MFMA
            auto epi_slice = [&](f32x16& acc, int o, int k) {
                if (k & 1) {
                    const pk2f x = {acc[k - 1], acc[k]};
                    const pk2f y = x * pk2f{0.01f, 0.01f};
                    float r0, r1;
                    asm("v_max_f32 %0, %1, %2" : "=v"(r0) : "v"(x.x), "v"(y.x));
                    asm("v_max_f32 %0, %1, %2" : "=v"(r1) : "v"(x.y), "v"(y.y));
                    hp[k >> 1] = pack_bf16(r0, r1);
                }
                if (k == 5 || k == 13) {
                    const int sh = k >> 3;
                    const int idx = (w3_lane >= 0) ? ((2 * o + sh) * 8 + w3_lane) : NS * 8;
                    w3q = sW3[idx];
                }
                if (k == 7 || k == 15) {
                    const int sh = k >> 3;
                    const uint4 hv = make_uint4(hp[4 * sh], hp[4 * sh + 1], hp[4 * sh + 2], hp[4 * sh + 3]);
                    if (TRAIN) act_store_nt_u4(a.act2b + ((size_t)tile * NT + o) * 128 + 64 * sh + lane, hv);
                    acc3 = mfma_bf16(w3q, hv, acc3);
                }
            };
            // (one epilogue slice per k-step: the interleaved form needs 16 of each, i.e. HID = 256)
            constexpr bool PIPE2 = PPO_BF16_FWD_PIPE && (NT % 2 == 0) && (NS == 16) && (TPS == 1 || MODE <= 1 || PPO_BF16_FWD_PIPE > 1);
            if constexpr (PIPE2) {
                f32x16 acc0, acc1;
                bias_init(acc0, 0);
#pragma unroll
                for (int k = 0; k < NS; ++k) { chain_step(acc0, 0, k); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll 1
                for (int o = 0; o + 2 < NT; o += 2) {            // chains o+1, o+2 beside the epilogues of o, o+1
                    bias_init(acc1, o + 1);
#pragma unroll
                    for (int k = 0; k < NS; ++k) { chain_step(acc1, o + 1, k); epi_slice(acc0, o, k); __builtin_amdgcn_sched_barrier(0); }
                    bias_init(acc0, o + 2);
#pragma unroll
                    for (int k = 0; k < NS; ++k) { chain_step(acc0, o + 2, k); epi_slice(acc1, o + 1, k); __builtin_amdgcn_sched_barrier(0); }
                }
                bias_init(acc1, NT - 1);
#pragma unroll
                for (int k = 0; k < NS; ++k) { chain_step(acc1, NT - 1, k); epi_slice(acc0, NT - 2, k); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int k = 0; k < 16; ++k) epi_slice(acc1, NT - 1, k);
            } else {
#pragma unroll 1
            for (int o = 0; o < NT; ++o) {
                f32x16 acc;
                bias_init(acc, o);
#pragma unroll
                for (int k = 0; k < NS; ++k) { chain_step(acc, o, k); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int k = 0; k < 16; ++k) epi_slice(acc, o, k);
            }
            }
            FBSTAMP(3);
            // logits of row j sit in accumulator registers 0..3 of lane j (lane half 0): hand them to both halves
            float lg[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) lg[i] = __shfl(acc3[i], j) + a.b3[i];
#pragma unroll
            for (int tt = 0; tt < TPS; ++tt)
#pragma unroll
                for (int i = 0; i < 4; ++i) l[tt][i] = (tt == ts) ? lg[i] : l[tt][i];
        }
        const int sampled = policy_tail<TMODE, TPS, true>(a, state, sid, act, l, lane, j, h, tick_val, out_index,
                                                          (TRAIN && PPO_BF16_STORE_LATE) ? &tpre : nullptr);
        FBSTAMP(4);
        if (MODE == 3) {
            asm volatile("" ::: "memory");
            if (TPS == 1) {                                  // Q == 8: wavefront-parallel env update
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
        FBSTAMP(5);
    }
    }
#ifdef PPO_BF16_STAMP
    if (a.stamps && lane == 0 && wave0 < 2048)
        for (int i = 0; i < 6; ++i) a.stamps[wave0 * 6 + i] = st_sum[i];
#endif
    if (MODE == 3) {                                            // env state back to the [N] arrays
        int slot2 = 0;
        for (int64_t n = wave0; n < a.B; n += nwaves, ++slot2) {
            const EnvRefLds r = slot_ref(slot2);
            for (int v = lane; v < a.envV; v += 64) { a.env_score[n * a.envV + v] = r.sc[v]; a.env_degree[n * a.envV + v] = r.dg[v]; }
            if (lane == 0) {
                a.env_active[n] = *r.active; a.env_steps[n] = *r.steps; a.env_reward[n] = *r.reward;
                a.env_done[n] = *r.done; a.env_episode[n] = *r.episode; a.env_tick[n] = *r.tick;
            }
        }
    }
}

#ifdef PPO_BF16_STAMP
static unsigned long long* g_bf16_fstamps = nullptr;
extern "C" int32_t ppo_debug_bf16_fwd_stamps(unsigned long long* out) {
    if (!g_bf16_fstamps) return -1;
    (void)hipDeviceSynchronize();
    return hipMemcpy(out, g_bf16_fstamps, 2048 * 6 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
static void set_fwd_stamps(FwdArgs& a) {
    if (!g_bf16_fstamps) (void)hipMalloc((void**)&g_bf16_fstamps, 2048 * 6 * 8);
    a.stamps = g_bf16_fstamps;
}
#else
static void set_fwd_stamps(FwdArgs&) {}
#endif

template <int MODE>
static int32_t dispatch_fwd_bf16(ppo_policy_s* p, const FwdArgs& args, int64_t B, int tps) {
    set_fwd_stamps(const_cast<FwdArgs&>(args));
    const int64_t need = (B + FWB_W - 1) / FWB_W;
    const unsigned grid = (unsigned)(need < 256 ? need : 256);
#define LAUNCHB(FF, HH, TT)                                                                                          \
    do {                                                                                                             \
        /* MODE 4: one env-snapshot slot per wave + the template rows behind them */                               \
        /* + the W2 queue's look-ahead past the last feature tile (the persistent form's env slots cover it there) */ \
        const size_t lds = FwdB<FF, HH>::lds_bytes + PPO_BF16_FWD_WQ * 1024 + (MODE == 4 ? (size_t)FWB_W * (2 * args.envV + 32) + 32 * PPO_TPL : 0); \
        static thread_local size_t attr_lds = 0;                                                                                  \
        if (lds > attr_lds) {                                                                                        \
            HIP_TRY(hipFuncSetAttribute((const void*)k_policy_fwd_bf16<FF, HH, MODE, TT>,                            \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
            attr_lds = lds;                                                                                          \
        }                                                                                                            \
        hipLaunchKernelGGL((k_policy_fwd_bf16<FF, HH, MODE, TT>), dim3(grid), dim3(FWB_T), lds, ppo_stream(), args); \
    } while (0)
    if (p->F == 72 && p->HID == 256 && tps == 1) LAUNCHB(72, 256, 1);
    else if (p->F == 72 && p->HID == 256 && tps == 4) LAUNCHB(72, 256, 4);
    else if (p->F == 72 && p->HID == 128 && tps == 1) LAUNCHB(72, 128, 1);
    else if (p->F == 72 && p->HID == 128 && tps == 4) LAUNCHB(72, 128, 4);
    else { ppo_set_error("unsupported policy/state shape (F,HID,H) for the gfx950 bf16 kernels"); return PPO_ERR_UNSUPPORTED; }
#undef LAUNCHB
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

// persistent rollout in bf16 mode: `a` comes filled from launch_policy_rollout_persistent (ppo_policy_fwd.hip)
int32_t launch_policy_rollout_persistent_bf16(ppo_policy_s* p, FwdArgs& a, int64_t N, int tps, int V) {
    const int64_t need = (N + FWB_W - 1) / FWB_W;
    const unsigned grid = (unsigned)(need < 256 ? need : 256);
    const int slots = (int)((N + (int64_t)grid * FWB_W - 1) / ((int64_t)grid * FWB_W));
    a.env_slots = slots;
    set_fwd_stamps(a);
    a.w1b = (const uint4*)p->w1b.p; a.w2b = (const uint4*)p->w2b.p; a.w3c = (const uint4*)p->w3c.p;
    const size_t env_bytes = (size_t)FWB_W * slots * (2 * V + 32) + 32 * PPO_TPL;  // env slots + the template rows
#define LAUNCHP(HH, TT)                                                                                              \
    do {                                                                                                             \
        /* the W2 queue's look-ahead past the last feature tile needs PPO_BF16_FWD_WQ KiB behind W2: env slots cover it, */ \
        /* few of them are topped up */                                                                              \
        size_t lds = FwdB<72, HH>::lds_bytes + env_bytes;                                                            \
        if (lds > 160 * 1024) return PPO_ERR_UNSUPPORTED;                                                            \
        lds = std::max(lds, (size_t)FwdB<72, HH>::W2_U4 * 16 + PPO_BF16_FWD_WQ * 1024);                              \
        static thread_local size_t attr_lds = 0;                                                                                  \
        if (lds > attr_lds) {                                                                                        \
            HIP_TRY(hipFuncSetAttribute((const void*)k_policy_fwd_bf16<72, HH, 3, TT>,                               \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
            attr_lds = lds;                                                                                          \
        }                                                                                                            \
        hipLaunchKernelGGL((k_policy_fwd_bf16<72, HH, 3, TT>), dim3(grid), dim3(FWB_T), lds, ppo_stream(), a);       \
    } while (0)
    if (p->HID == 256 && tps == 1) LAUNCHP(256, 1);
    else if (p->HID == 128 && tps == 1) LAUNCHP(128, 1);
    else return PPO_ERR_UNSUPPORTED;
#undef LAUNCHP
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

int32_t launch_policy_fwd_bf16(ppo_policy_s* p, FwdArgs& a, int mode, int64_t B, int tps) {
    a.w1b = (const uint4*)p->w1b.p; a.w2b = (const uint4*)p->w2b.p; a.w3c = (const uint4*)p->w3c.p;
    a.act1b = PPO_BF16_H1_RECOMPUTE ? nullptr : (uint4*)p->act1.p; a.act2b = (uint4*)p->act2.p;
    if (mode == 0) return dispatch_fwd_bf16<0>(p, a, B, tps);
    if (mode == 1) return dispatch_fwd_bf16<1>(p, a, B, tps);
    if (mode == 4) return dispatch_fwd_bf16<4>(p, a, B, tps);
    return dispatch_fwd_bf16<2>(p, a, B, tps);
}

// ================================================================ backward
struct BwdBArgs {
    const int8_t* states; const int32_t* idx; int32_t B;   // B = number of 32-row tiles (states * tps), < 2^31
    int tps_shift;                                          // tiles per state = 1 << tps_shift (H = 32 or 128)
    int x_by_tile;                                          // 1: `states` is the forward's row scratch in minibatch order (compact rollouts)
    int nwg;                                                // == gridDim.x (as an argument: no dispatch-packet reload in the loop)
    const uint4* act1b; const uint4* act2b; const float4* dY;
    const uint4* w2tb; const uint2* w3tb;
    const uint4* w1b; const float4* b1p;                    // layer-1 fragments / bias pack (H1 recomputed here)
    uint4* z1f; uint4* xf;                                  // operand fragments handed to k_policy_dw1_bf16
    float* slabs; size_t slab_stride;
    unsigned long long* stamps;                             // diagnostic build only (-DPPO_BF16_STAMP)
};
#ifdef PPO_BF16_STAMP
#define BSTAMP(i) do { unsigned long long _n = clock64(); st_sum[i] += _n - st_t; st_t = _n; } while (0)
#else
#define BSTAMP(i) do {} while (0)
#endif

// The weight-gradient accumulators decide the shape of the backward pass: dW2 (HID x HID) and dW1 (HID x F) need
// 22 accumulator tiles = 352 registers per SIMD for HID = 256 -- two waves per SIMD cannot hold both next to their
// working set, and one wave with > 256 accumulator registers spills.  So the pass is two kernels:
//   k_policy_bwd_bf16  workgroup of 4 waves, one per SIMD with the whole register file; wave w owns feature tiles
//                      w*FT .. w*FT+FT-1 of dW2 (256 accumulator registers = the AGPR half for HID = 256) and the
//                      small grads.  It also emits, per tile, dZ1^T as ready-made MFMA operand fragments.
//   k_policy_dw1_bf16  dW1 += dZ1^T X from the dZ1^T fragments (64 MB per 4096-state minibatch, written and re-read
//                      through the 256 MB Infinity Cache) and the state rows themselves (9 MB).
template <int F, int HID>
struct BwdB {
    static constexpr int NT = HID / 32, NS = HID / 16, FP = ((F + 31) / 32) * 32, NI = FP / 32;
    // waves per workgroup and feature tiles per wave.  HID = 256: 8 waves, two per SIMD, one feature tile each (128 dW2
    // accumulator registers + a working set that fits the other 128 now that only 4 k-steps of W2^T stream through
    // registers): the partner wave covers the LDS / L2 round trips a single wave per SIMD sat through
    static constexpr int NW = (HID >= 256) ? PPO_BF16_BWD_WAVES : 4, FT = NT / NW;
    static_assert(NT % NW == 0, "feature tiles per wave");
    static constexpr int ST = 2 * HID + 64;               // image row stride (bytes): 16 dwords mod 64 -> the 4 rows of a
    static constexpr int STX = 2 * FP;                    //   transposed-read block sit on different bank quarters
    static_assert((ST / 4) % 64 == 16 || (ST / 4) % 64 == 48, "image stride");
    static_assert((STX / 4) % 64 == 16 || (STX / 4) % 64 == 48, "X image stride");
    static constexpr int IMG = 32 * ST;
    // k-steps of W2^T resident in LDS for the whole launch (the rest streams from L2 once per tile into a register ring):
    // three images (dZ1 reuses the H2 image, see phase B) + the X image leave room for 12 of the 16 at HID = 256
    static constexpr bool FUSE = (HID <= PPO_BF16_DW1_FUSED_MAX_HID);
    static constexpr bool TRN = PPO_BF16_MFMA_TRANSPOSE && !FUSE;       // identity-MFMA transposes, no H2 / dZ1 image
    static constexpr int NSL = (HID >= 256) ? (TRN ? 14 : 12) : NS, NSR = NS - NSL;
    static constexpr int oZ2 = 0, oH1 = IMG, oH2 = 2 * IMG, oX = TRN ? 2 * IMG : 3 * IMG,
                         oDY = oX + 32 * STX, oDB3 = oDY + 512, oB1 = oDB3 + 512, oID = oB1 + NT * 128,
                         oW = oID + (TRN ? 2048 : 0), wEnd = oW + NT * NSL * 1024;
    static constexpr int KS1 = (F + 15) / 16;             // layer-1 k-steps
    // layer-1 B operands (X as bf16 fragments, KS1 KiB) of the tile about to be processed: where the X image would be
    // when dW1 is not accumulated here, behind the weights otherwise
    static constexpr int oXF = FUSE ? wEnd : oX, total = FUSE ? wEnd + (KS1 + 1) * 1024 : wEnd;     // + one spare slot
    static_assert((KS1 + 1) * 1024 <= 32 * STX, "X fragments fit the X image's place");
    static_assert(total <= 160 * 1024, "LDS budget");
};

// global accesses as wave-uniform base (SGPR pair) + 32-bit per-lane byte offset: saddr-form instructions, no per-lane
// 64-bit pointers (sixteen of those for the W2^T ring alone would spill, and hipcc waits vmcnt(0) at every reload)
__device__ __forceinline__ uint4 ldg16(const void* sbase, unsigned voff) {
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sbase) + voff);
}
// read-once streams (the saved activations in the backward pass): non-temporal, so they do not evict the W2^T
// fragments and the dW1 operand fragments that ARE re-read
__device__ __forceinline__ uint4 ldg16_nt(const void* sbase, unsigned voff) {
#if PPO_BF16_ACT_NT_LOAD
    typedef uint32_t u32x4l __attribute__((ext_vector_type(4)));
    const u32x4l t = __builtin_nontemporal_load(reinterpret_cast<const u32x4l*>(reinterpret_cast<const char*>(sbase) + voff));
    return make_uint4(t.x, t.y, t.z, t.w);
#else
    return ldg16(sbase, voff);
#endif
}
__device__ __forceinline__ float4 ldg16f(const void* sbase, unsigned voff) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sbase) + voff);
}
__device__ __forceinline__ uint2 ldg8(const void* sbase, unsigned voff) {
    return *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(sbase) + voff);
}
__device__ __forceinline__ void stg16(void* sbase, unsigned voff, const uint4& v) {
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(sbase) + voff) = v;
}

// two transposed 4x16 block reads -> one 32x32x16 operand fragment (8 consecutive rows of the lane's column)
__device__ __forceinline__ uint4 tr_frag(const char* p0, const char* p1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 u0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
    const s16x4 u1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
    const uint2 a = __builtin_bit_cast(uint2, u0), b = __builtin_bit_cast(uint2, u1);
    return make_uint4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ float sum_frag(const uint4& v) {
    return ((bf16_lo(v.x) + bf16_hi(v.x)) + (bf16_lo(v.y) + bf16_hi(v.y))) + ((bf16_lo(v.z) + bf16_hi(v.z)) + (bf16_lo(v.w) + bf16_hi(v.w)));
}

template <int F, int HID>
__global__ __launch_bounds__((BwdB<F, HID>::NW * 64), 1) void k_policy_bwd_bf16(BwdBArgs a) {
    using C = BwdB<F, HID>;
    constexpr int NT = C::NT, NS = C::NS, NI = C::NI, FT = C::FT, ST = C::ST, STX = C::STX, NTHR = C::NW * 64;
    constexpr int XDW = 32 * F / 4, XPD = (XDW + NTHR - 1) / NTHR;
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* const imgZ2 = smem_c + C::oZ2;
    char* const imgH1 = smem_c + C::oH1;
    char* const imgH2 = smem_c + C::oH2;
    char* const imgZ1 = imgH2;      // dZ1 overwrites H2: a wave's H2 columns are read by that wave only (dW3 sums, phase B) before it writes them
    char* const imgX = smem_c + C::oX;
    float* const sDY = reinterpret_cast<float*>(smem_c + C::oDY);
    uint4* const sW = reinterpret_cast<uint4*>(smem_c + C::oW);
    float4* const sB1 = reinterpret_cast<float4*>(smem_c + C::oB1);
    u32x4* const sXF = reinterpret_cast<u32x4*>(smem_c + C::oXF);
    constexpr bool RC1 = PPO_BF16_H1_RECOMPUTE;
    constexpr int KS1 = C::KS1;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (RC1) for (int i = tid; i < NT * 8; i += C::NW * 64) sB1[i] = a.b1p[i];
    constexpr bool TRN = C::TRN;
    // identity B operands of the transposing MFMAs: k-slot (step s, lane half hB, element e) carries feature
    // 16s + 8(e>>2) + 4hB + (e&3) of the packed fragment, so lane (n, hB) holds a single 1.0 -- in step n>>4, element
    // 4((n>>3)&1) + (n&3), and only if hB == (n>>2)&1
    u32x4* const sID = reinterpret_cast<u32x4*>(smem_c + C::oID);
    if (TRN && tid < 64) {
        const int n = tid & 31, hB = tid >> 5, e = 4 * ((n >> 3) & 1) + (n & 3);
#pragma unroll
        for (int s1 = 0; s1 < 2; ++s1) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (hB == ((n >> 2) & 1) && s1 == (n >> 4)) {
                const uint32_t one = (e & 1) ? 0x3F800000u : 0x00003F80u;
                if ((e >> 1) == 0) v.x = one; else if ((e >> 1) == 1) v.y = one; else if ((e >> 1) == 2) v.z = one; else v.w = one;
            }
            sID[s1 * 64 + tid] = v;
        }
    }

    // k-steps 0 .. NSL-1 of this wave's W2^T fragments stay in LDS for the whole launch (all that is left of the 160 KiB
    // next to the images); the other NSR stream from L2 once per tile, issued through phase A, so the dH1 chain of
    // phase B never waits for a refill
    constexpr int NSL = C::NSL, NSR = C::NSR;
#pragma unroll
    for (int i = 0; i < FT; ++i)
#pragma unroll
        for (int s = 0; s < NSL; ++s)
            sW[((w * FT + i) * NSL + s) * 64 + lane] = a.w2tb[((size_t)(w * FT + i) * NS + s) * 64 + lane];

    constexpr bool FUSE1 = PPO_BF16_DW1_FUSED;
    // the state rows are needed here only when dW1 is accumulated in this kernel (k_policy_dw1_bf16 stages its own)
    // zero the padded X image once (columns >= F are never written afterwards)
    if (FUSE1) for (int i = tid; i < 32 * STX / 4; i += NTHR) reinterpret_cast<uint32_t*>(imgX)[i] = 0u;

    f32x16 accW2[FT][NT];
#pragma unroll
    for (int i = 0; i < FT; ++i)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) accW2[i][kt][r] = 0.0f;
    f32x16 accW1[FUSE1 ? FT : 1][FUSE1 ? NI : 1];        // dW1[k-tiles of this wave][all input tiles] (columns >= F: zero padding)
    if (FUSE1) {
#pragma unroll
        for (int i = 0; i < FT; ++i)
#pragma unroll
            for (int it = 0; it < NI; ++it)
#pragma unroll
                for (int r = 0; r < 16; ++r) accW1[FUSE1 ? i : 0][FUSE1 ? it : 0][r] = 0.0f;
    }
    float db1[FT], db2[FT], dw3[FT][4];
    // per-row partial sums of dY (combined over the 32 rows at the end): wave 0 keeps them in LDS, not in four registers
    // that every wave would carry through the whole tile loop
    float4* const sDB3 = reinterpret_cast<float4*>(smem_c + C::oDB3);
    if (w == 0 && h == 0) sDB3[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < FT; ++i) { db1[i] = db2[i] = 0.f; dw3[i][0] = dw3[i][1] = dw3[i][2] = dw3[i][3] = 0.f; }

    // A operands of dH2^T = W3^T dY^T for this wave's feature tiles: k = output index o (4 of the 16 k-slots used)
    constexpr bool DIET = C::NW > 4;                     // two waves per SIMD: 128 registers beside the accumulators
    constexpr bool A3_RELOAD = DIET;                     // W3^T operand (4 registers per feature tile) back per tile from L2
    uint4 a3[FT];
#pragma unroll
    for (int i = 0; i < FT; ++i) {
        a3[i] = make_uint4(0u, 0u, 0u, 0u);
        if (!A3_RELOAD && h == 0) { const uint2 t = a.w3tb[32 * (w * FT + i) + j]; a3[i].x = t.x; a3[i].y = t.y; }
    }

    // image addressing.  Row-major [32 rows][cols] bf16, 8-byte chunk cc = col/4 stored at chunk cc ^ ((row>>1)&7):
    // the 16 lanes of a ds_write_b64 group (16 consecutive rows, same chunk) then hit 16 different bank pairs.
    // write side: lane (row j, half h) owns chunks 8*tile + 2g + h, g = 0..3
    // (re-derived from an opaque copy of j at the top of every phase that uses them: hoisted out of the tile loop the
    // swizzled addresses are a dozen registers that spill, and a scratch reload waits vmcnt(0))
    int jt = j;
#define BF16_SWZ() asm volatile("" : "+v"(jt)); const int wsw = (jt >> 1) & 7, wrow = jt * ST
    // transposed-read side (operand lane l: column l&31, rows 8h + 4u + q of k-step s; q = (l&15)>>2 supplies the row,
    // p = l&3 the 4-column chunk, (l>>4)&1 the 16-column block of the 32-column tile)
    const int tq = (lane & 15) >> 2, tcc = 4 * ((lane >> 4) & 1) + (lane & 3);
    int tro[2], trx[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = 8 * h + 4 * u + tq;
        tro[u] = r * ST + 8 * (tcc ^ ((r >> 1) & 7));     // + 16*s*ST + 64*tile   (tile = 32-column tile index)
        trx[u] = r * STX + 8 * tcc;                        // X image: no swizzle
    }

    // next tile's inputs, fetched one tile ahead
    uint4 nh2[FT][2], nh1[RC1 ? 1 : FT][2];
    float4 ndy;
    constexpr int KPW = (KS1 + C::NW - 1) / C::NW;          // layer-1 k-steps a wave converts: w, w + NW, ...
    uint2 nxr[KPW];                                         // RC1: this lane's 8 state bytes of those k-steps
#pragma unroll
    for (int q = 0; q < KPW; ++q) nxr[q] = make_uint2(0u, 0u);
    uint32_t nx[XPD];
    int xoff[XPD];                                          // X-image byte offset of state dword tid + i*256 (tile-independent)
#pragma unroll
    for (int i = 0; i < XPD; ++i) {
        const int d = tid + i * NTHR;
        xoff[i] = (d < XDW) ? (d / (F / 4)) * STX + (d % (F / 4)) * 8 : -1;
    }
    unsigned lo16 = (unsigned)lane * 16u;
    auto rows_of = [&](int t, int sidx) {                   // first byte of tile t's 32 state rows (wave-uniform)
        return reinterpret_cast<const char*>(a.states + (a.x_by_tile ? (size_t)t : (((size_t)sidx << a.tps_shift) + (size_t)(t & ((1 << a.tps_shift) - 1)))) * 32 * F);
    };
    // layer-1 B operand of k-step s = w: lane (row j, half h) holds features 16w + 8h .. +7 of its row -- exactly 8
    // consecutive state bytes (none for the half that lies beyond F)
    auto load_xraw = [&](int t, int sidx) {
        // branch-free (a wave-uniform branch inside the unrolled dH1 chain splits its scheduling region): a wave without a
        // k-step loads the last one again and never stores it; the half beyond F loads its row's first bytes and is zeroed
        // at the store
#pragma unroll
        for (int q = 0; q < KPW; ++q) {
            const int s0 = w + C::NW * q, ws = s0 < KS1 ? s0 : KS1 - 1;
            const bool in = 16 * ws + 8 * h + 8 <= F;
            const unsigned off = (lo16 & 0x1F0u) / 16u * (unsigned)F + (in ? 16u * (unsigned)ws + ((lo16 >> 9) & 1u) * 8u : 0u);
            nxr[q] = ldg8(rows_of(t, sidx), off);
        }
    };
    auto store_xfrag = [&]() {                              // nxr -> bf16 fragments -> sXF[k-step]
#pragma unroll
        for (int q = 0; q < KPW; ++q) {
            // branch-free as well: a wave without a k-step writes the spare slot KS1 (nobody reads it)
            const int s0 = w + C::NW * q, sl = s0 < KS1 ? s0 : KS1;
            uint2 d = nxr[q];
            if (16 * sl + 8 * h + 8 > F) d = make_uint2(0u, 0u);
            float f[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) { f[i] = (float)(int)(int8_t)(d.x >> (8 * i)); f[4 + i] = (float)(int)(int8_t)(d.y >> (8 * i)); }
            const u32x4 v = {pack_bf16(f[0], f[1]), pack_bf16(f[2], f[3]), pack_bf16(f[4], f[5]), pack_bf16(f[6], f[7])};
            sXF[sl * 64 + lane] = v;
        }
    };
    // sidx: transition id of tile t's state.  It is itself a global load, so it is fetched one tile earlier still
    // (idx_next below): a prefetch that first had to wait for its own index stalled phase B for an HBM round trip.
    auto prefetch = [&](int t, int sidx_v) {
        const int sidx = __builtin_amdgcn_readfirstlane(sidx_v);
#pragma unroll
        for (int i = 0; i < FT; ++i) {
            const size_t base = ((size_t)t * NT + (w * FT + i)) * 128;          // wave-uniform
            nh2[i][0] = ldg16_nt(a.act2b + base, lo16); nh2[i][1] = ldg16_nt(a.act2b + base + 64, lo16);
            if constexpr (!RC1) { nh1[i][0] = ldg16_nt(a.act1b + base, lo16); nh1[i][1] = ldg16_nt(a.act1b + base + 64, lo16); }
        }
        ndy = ldg16f(a.dY + (size_t)t * 32, lo16 & 0x1F0u);                     // row j: (lane & 31) * 16 bytes
        if constexpr (FUSE1) {
            const char* xs = rows_of(t, sidx);
#pragma unroll
            for (int i = 0; i < XPD; ++i) {
                const unsigned d = (unsigned)tid + (unsigned)i * NTHR;
                nx[i] = d < (unsigned)XDW ? *reinterpret_cast<const uint32_t*>(xs + d * 4u) : 0u;
            }
        }
    };
    // the same loads one at a time, to be spread between the MFMAs of phase B: issued in one burst they stall the wave
    // for ~1200 cycles (the CU's miss queue back-pressures the issue), and nothing overlaps.  Piece 0 is the state
    // bytes (the first thing consumed: converted at the end of this tile's phase C), then H2 (and H1), dY, X image
    constexpr int NPA = RC1 ? 2 * FT : 4 * FT;              // activation pieces
    constexpr int NPIECE = 1 + NPA + 2;
    auto prefetch_piece = [&](int k0, int t, int sidx) {
        if (k0 == 0) { if constexpr (RC1) load_xraw(t, sidx); return; }
        const int k = k0 - 1;
        if (k < NPA) {
            const int i = RC1 ? (k >> 1) : (k >> 2);
            const size_t base = ((size_t)t * NT + (w * FT + i)) * 128 + ((k & 1) ? 64 : 0);
            if (!RC1 && (k & 2)) nh1[RC1 ? 0 : i][k & 1] = ldg16_nt(a.act1b + base, lo16);
            else                 nh2[i][k & 1] = ldg16_nt(a.act2b + base, lo16);
        } else if (k == NPA) {
            ndy = ldg16f(a.dY + (size_t)t * 32, lo16 & 0x1F0u);                     // row j: (lane & 31) * 16 bytes
        } else if (FUSE1 && k == NPA + 1) {
            const char* xs = rows_of(t, sidx);
#pragma unroll
            for (int i = 0; i < XPD; ++i) {
                const unsigned d = (unsigned)tid + (unsigned)i * NTHR;
                nx[i] = d < (unsigned)XDW ? *reinterpret_cast<const uint32_t*>(xs + d * 4u) : 0u;
            }
        }
    };
    // RC1: this wave's W1 fragments (the same KS1 KiB every tile, L2-resident).  Fetched at the end of the previous tile --
    // register pressure is lowest there -- so that the layer-1 chain at the top of phase A does not start with an L2
    // round trip (measured: +0.1 ms per 65,536 tiles when it did)
    uint4 w1f[RC1 ? FT : 1][RC1 ? KS1 : 1];
    auto load_w1 = [&]() {
#pragma unroll
        for (int i = 0; i < FT; ++i)
#pragma unroll
            for (int s1 = 0; s1 < KS1; ++s1) w1f[RC1 ? i : 0][RC1 ? s1 : 0] = ldg16(a.w1b + ((size_t)(w * FT + i) * KS1 + s1) * 64, lo16);
    };
    if constexpr (RC1) load_w1();
    auto tile_or_last = [&](int t) { return t < a.B ? t : a.B - 1; };
    int idx_next = 0;
    if ((int)blockIdx.x < a.B) {
        const int sidx0 = (FUSE1 || RC1) && !a.x_by_tile ? a.idx[(int)blockIdx.x >> a.tps_shift] : 0;
        prefetch((int)blockIdx.x, sidx0);
        if ((FUSE1 || RC1) && !a.x_by_tile) idx_next = a.idx[tile_or_last((int)blockIdx.x + a.nwg) >> a.tps_shift];
        if constexpr (RC1) { load_xraw((int)blockIdx.x, __builtin_amdgcn_readfirstlane(sidx0)); store_xfrag(); }   // first tile: latency exposed once
    }
    __syncthreads();

#ifdef PPO_BF16_STAMP
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = clock64();
#endif
    for (int tile = blockIdx.x; tile < a.B; tile += a.nwg) {
        // ================= phase A: dZ2 = (W3^T dY) . lrelu'(H2), images (inputs were fetched one tile ahead)
#ifdef PPO_BF16_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BSTAMP(0);
#endif
        // per-tile opaque lane offset: every global address below is re-formed from it (SGPR base + this VGPR) instead
        // of being hoisted out of the tile loop as ~30 loop-invariant 64-bit per-lane pointers that then spill
        asm volatile("" : "+v"(lo16));
        BF16_SWZ();
        const float4 dy = ndy;
#pragma unroll
        for (int i = 0; i < (FUSE1 ? XPD : 0); ++i) {
            if (xoff[i] >= 0) {
                const uint32_t v = nx[i];
                *reinterpret_cast<uint2*>(imgX + xoff[i]) =
                    make_uint2(pack_bf16((float)(int)(int8_t)(v), (float)(int)(int8_t)(v >> 8)),
                               pack_bf16((float)(int)(int8_t)(v >> 16), (float)(int)(int8_t)(v >> 24)));
            }
        }
        // second half of this wave's W2^T fragments (L2-resident): issued at the top of the tile, right behind the wait
        // for the prefetched inputs, so they land under phase A (sched_barrier: hipcc otherwise sinks them to the barrier)
        uint4 ring[NSR > 0 ? NSR : 1][FT];
        const uint4* wt = a.w2tb + (size_t)(w * FT) * NS * 64;       // wave-uniform base of this wave's W2^T tiles
        // (k-step order = the order the chain consumes them in: vmcnt retires in order).  PPO_BF16_RING_SPREAD: not in
        // one burst -- 8 KiB per wave through a 64 B/clk L1 stalls the issuing wave ~500 cycles -- but a few at a
        // time between the pieces of phase A's VALU/LDS work (2 here, then 3 per feature tile)
        constexpr int NRL = NSR * FT;
        auto ring_load = [&](int q) { if (q < NRL) ring[q / FT][q % FT] = ldg16(wt + (size_t)((q % FT) * NS + NSL + q / FT) * 64, lo16); };
        constexpr bool SPREAD = PPO_BF16_RING_SPREAD && (NRL <= 2 + 3 * FT);
#pragma unroll
        for (int q = 0; q < (SPREAD ? 2 : NRL); ++q) ring_load(q);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RC1) {
            // H1 of this wave's feature tiles, recomputed exactly as the forward kernel computes it (bias in the accumulator,
            // KS1 k-steps in order, leakyrelu, RNE pack) and laid into the H1 image.  Its own segment in front of the dZ2
            // work: the W1 fragments (L2-resident; the partner wave covers their round trip) and the accumulator are dead
            // again before the dZ2 registers come alive
#pragma unroll
            for (int i = 0; i < FT; ++i) {
                const int ft = w * FT + i;
                // LDS addresses re-formed from the per-tile opaque lane offset (hoisted out of the tile loop they spill)
                const unsigned lds0 = (unsigned)(size_t)(PPO_LDS void*)smem_c;
                const PPO_LDS f32x4* const b1l = (const PPO_LDS f32x4*)(size_t)(lds0 + (unsigned)C::oB1 + (unsigned)ft * 128u + ((lo16 >> 3) & 64u));
                const PPO_LDS u32x4* const xfl = (const PPO_LDS u32x4*)(size_t)(lds0 + (unsigned)C::oXF + lo16);
                f32x16 a1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b = b1l[q];
                    a1[4 * q + 0] = b.x; a1[4 * q + 1] = b.y; a1[4 * q + 2] = b.z; a1[4 * q + 3] = b.w;
                }
#pragma unroll
                for (int s1 = 0; s1 < KS1; ++s1) {
                    const u32x4 xb = xfl[s1 * 64];
                    a1 = mfma_bf16(w1f[i][s1], make_uint4(xb.x, xb.y, xb.z, xb.w), a1);
                }
                lrelu16(a1);
                uint4 h1p[2];
                pack_tile(a1, h1p);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<uint2*>(imgH1 + wrow + 64 * ft + 8 * ((2 * g + h) ^ wsw)) = make_uint2(dw(h1p[g >> 1], 2 * (g & 1)), dw(h1p[g >> 1], 2 * (g & 1) + 1));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const uint32_t dy01 = pack_bf16(dy.x, dy.y), dy23 = pack_bf16(dy.z, dy.w);    // exact: dY is stored bf16-rounded
        if (w == 0 && h == 0) *reinterpret_cast<float4*>(sDY + j * 4) = dy;
        if (w == 0 && h == 0) { float4 t = sDB3[j]; t.x += dy.x; t.y += dy.y; t.z += dy.z; t.w += dy.w; sDB3[j] = t; }
        const uint4 bdy = (h == 0) ? make_uint4(dy01, dy23, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int i = 0; i < FT; ++i) {
            const int ft = w * FT + i;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            uint4 a3i = make_uint4(0u, 0u, 0u, 0u);       // A operand of dH2^T = W3^T dY^T (k = output index: 4 of the 16 k-slots used)
            if constexpr (A3_RELOAD) { if (h == 0) { const uint2 t = ldg8(a.w3tb + 32 * ft, (lo16 & 0x1F0u) >> 1); a3i.x = t.x; a3i.y = t.y; } }
            else a3i = a3[i];
            acc = mfma_bf16(a3i, bdy, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t d = dw(nh2[i][r >> 3], (r >> 1) & 3);
                acc[r] = acc[r] * ((r & 1) ? slope_hi(d) : slope_lo(d));
            }
            if constexpr (SPREAD) {
                __builtin_amdgcn_sched_barrier(0);
                ring_load(2 + 3 * i);
                __builtin_amdgcn_sched_barrier(0);
            }
            uint4 zf[2];
            pack_tile(acc, zf);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int off = wrow + 64 * ft + 8 * ((2 * g + h) ^ wsw);
                *reinterpret_cast<uint2*>(imgZ2 + off) = make_uint2(dw(zf[g >> 1], 2 * (g & 1)), dw(zf[g >> 1], 2 * (g & 1) + 1));
                if constexpr (!TRN) *reinterpret_cast<uint2*>(imgH2 + off) = make_uint2(dw(nh2[i][g >> 1], 2 * (g & 1)), dw(nh2[i][g >> 1], 2 * (g & 1) + 1));
                if constexpr (!RC1) *reinterpret_cast<uint2*>(imgH1 + off) = make_uint2(dw(nh1[RC1 ? 0 : i][g >> 1], 2 * (g & 1)), dw(nh1[RC1 ? 0 : i][g >> 1], 2 * (g & 1) + 1));
                if constexpr (SPREAD) {
                    if (g < 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        ring_load(3 + 3 * i + g);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        BSTAMP(1);
        __syncthreads();
        BSTAMP(2);
        // ================= phase B: dH1^T[k-tiles of this wave] = W2^T dZ2^T, dZ1 = dH1 . lrelu'(H1) -> image
        f32x16 accH[FT];                // dH1 of this wave's feature tiles (EPI_C: its epilogue runs inside phase C)
        constexpr bool EPI_C = TRN && PPO_BF16_BWD_EPI_IN_C && (16 % NT == 0);
        {
            // the next tile's inputs start their HBM round trip here (their registers were consumed in phase A).  They
            // are issued BEHIND the streamed W2^T half: vmcnt retires in order, so the chain below waits only for
            // fragments that are older than these loads
            static_assert(NPIECE <= NS, "one prefetch piece per k-step");
            constexpr int PF_T0 = 0;
            const int pf_tile = tile_or_last(tile + a.nwg), pf_sidx = __builtin_amdgcn_readfirstlane(idx_next);
            if ((FUSE1 || RC1) && !a.x_by_tile) idx_next = a.idx[tile_or_last(tile + 2 * a.nwg) >> a.tps_shift];
            auto& acc = accH;
#pragma unroll
            for (int i = 0; i < FT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
            // dW3[o][f] = sum_rows dY[row][o] H2[row][f] on the VALU (4 accumulators per lane instead of a 16-register
            // MFMA tile that would be 7/8 zero padding): lane (f, h) holds rows 16s + 8h + e of column f.  Done HERE, in
            // front of the chain: afterwards this wave's columns of the H2 image are free and take dZ1 (epilogue below)
            if constexpr (TRN) {
                // H2 of the wave's feature tiles is still in the registers it was prefetched into (lane = row): two identity
                // MFMAs turn it (D[row][feature] = sum_slot P[row][slot] I[slot][feature]): lane (feature, h) then holds rows
                // (r&3) + 8(r>>2) + 4h in accumulator register r, as exact fp32 copies of the bf16 values
                const unsigned lds0 = (unsigned)(size_t)(PPO_LDS void*)smem_c;
                const PPO_LDS u32x4* const idl = (const PPO_LDS u32x4*)(size_t)(lds0 + (unsigned)C::oID + lo16);
                const u32x4 i0 = idl[0], i1 = idl[64];
#pragma unroll
                for (int i = 0; i < FT; ++i) {
                    f32x16 d;
#pragma unroll
                    for (int r = 0; r < 16; ++r) d[r] = 0.0f;
                    d = mfma_bf16(nh2[i][0], make_uint4(i0.x, i0.y, i0.z, i0.w), d);
                    d = mfma_bf16(nh2[i][1], make_uint4(i1.x, i1.y, i1.z, i1.w), d);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float4 y = *reinterpret_cast<const float4*>(sDY + ((r & 3) + 8 * (r >> 2) + 4 * h) * 4);
                        dw3[i][0] = fmaf(y.x, d[r], dw3[i][0]); dw3[i][1] = fmaf(y.y, d[r], dw3[i][1]);
                        dw3[i][2] = fmaf(y.z, d[r], dw3[i][2]); dw3[i][3] = fmaf(y.w, d[r], dw3[i][3]);
                    }
                }
            } else {
#pragma unroll
            for (int i = 0; i < FT; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const uint4 hb = tr_frag(imgH2 + tro[0] + 16 * s * ST + 64 * (w * FT + i), imgH2 + tro[1] + 16 * s * ST + 64 * (w * FT + i));
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const uint32_t d = dw(hb, e >> 1);
                        const float hv = (e & 1) ? bf16_hi(d) : bf16_lo(d);
                        const float4 y = *reinterpret_cast<const float4*>(sDY + (16 * s + 8 * h + e) * 4);
                        dw3[i][0] = fmaf(y.x, hv, dw3[i][0]); dw3[i][1] = fmaf(y.y, hv, dw3[i][1]);
                        dw3[i][2] = fmaf(y.z, hv, dw3[i][2]); dw3[i][3] = fmaf(y.w, hv, dw3[i][3]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // The k-steps whose W2^T fragments were streamed into registers run first (their registers free up), then
            // the LDS-resident ones.  LDS operands are read DL steps ahead of their MFMAs (explicit queue; sched_barrier
            // pins the order): one step ahead covers 64 cycles of MFMA, a ds_read_b128 round trip is longer than that.
            // B operand of k-step s: dZ2[row j][features 16s + 8h .. +7] straight from the row-major image -- chunks
            // 4(s&1) + 2h and the next one of feature tile s>>1, each where the row's swizzle put it (two 8-byte reads;
            // W2^T is packed in this natural contraction order)
            constexpr int DL = DIET ? PPO_BF16_BWD_DL8 : 3;
            // this wave's resident W2^T fragments behind ONE opaque per-lane base: the k-step offsets then fit the ds_read
            // immediate (with the array base folded in they pass 64 KiB, and hipcc hoists one address register per k-step
            // out of the tile loop -- which spill, and every scratch reload waits vmcnt(0), i.e. for the whole prefetch)
            unsigned swl_a = (unsigned)(size_t)(PPO_LDS void*)(sW + (size_t)(w * FT) * NSL * 64 + lane);
            asm volatile("" : "+v"(swl_a));
            const PPO_LDS u32x4* const swl = (const PPO_LDS u32x4*)(size_t)swl_a;
            uint4 bzq[DL], wlq[DL][FT];
            BF16_SWZ();
            const char* const zrow = imgZ2 + wrow;
            const int zc0 = 8 * ((2 * h) ^ wsw), zc1 = 8 * ((2 * h + 1) ^ wsw);        // byte offsets of the two chunks for even s; odd s: ^ 32
            auto kstep = [&](int t) { return t < NSR ? t + NSL : t - NSR; };
            auto issue = [&](int t) {
                const int s = kstep(t);
                const char* zr = zrow + 64 * (s >> 1);
                const uint2 lo = *reinterpret_cast<const uint2*>(zr + ((s & 1) ? (zc0 ^ 32) : zc0));
                const uint2 hi = *reinterpret_cast<const uint2*>(zr + ((s & 1) ? (zc1 ^ 32) : zc1));
                bzq[t % DL] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                if (s < NSL) {
#pragma unroll
                    for (int i = 0; i < FT; ++i) { const u32x4 t4 = swl[(i * NSL + s) * 64]; wlq[t % DL][i] = make_uint4(t4.x, t4.y, t4.z, t4.w); }
                }
            };
#pragma unroll
            for (int t = 0; t < DL; ++t) issue(t);
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                const int s = kstep(t);
                const uint4 bz = bzq[t % DL];
                uint4 wv[FT];
#pragma unroll
                for (int i = 0; i < FT; ++i) wv[i] = (s < NSL) ? wlq[t % DL][i] : ring[s >= NSL ? s - NSL : 0][i];
                if (t + DL < NS) issue(t + DL);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < FT; ++i) acc[i] = mfma_bf16(wv[i], bz, acc[i]);
                __builtin_amdgcn_sched_barrier(0);
                if (t >= PF_T0 && t - PF_T0 < NPIECE) prefetch_piece(t - PF_T0, pf_tile, pf_sidx);   // harmless re-load behind the last tile
                __builtin_amdgcn_sched_barrier(0);
            }
            BSTAMP(3);
#pragma unroll
            for (int i = 0; i < (EPI_C ? 0 : FT); ++i) {
                const int ft = w * FT + i;
                BF16_SWZ();          // (shadows the chain's copies: those die with the chain)
                uint2 hc[4];         // H1 of this feature tile, back from the image written in phase A (lane = row again)
#pragma unroll
                for (int g = 0; g < 4; ++g) hc[g] = *reinterpret_cast<const uint2*>(imgH1 + wrow + 64 * ft + 8 * ((2 * g + h) ^ wsw));
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t d = (r & 2) ? hc[r >> 2].y : hc[r >> 2].x;
                    acc[i][r] = acc[i][r] * ((r & 1) ? slope_hi(d) : slope_lo(d));
                }
                uint4 z1[2];
                pack_tile(acc[i], z1);
                if constexpr (TRN) {
                    // dZ1^T of this feature tile as the A-operand fragments of k_policy_dw1_bf16, by the same identity MFMAs:
                    // lane (k, h) gets rows (r&3) + 8(r>>2) + 4h -- the dW1 kernel reads its X operands in that row order
                    const unsigned lds0 = (unsigned)(size_t)(PPO_LDS void*)smem_c;
                    const PPO_LDS u32x4* const idl = (const PPO_LDS u32x4*)(size_t)(lds0 + (unsigned)C::oID + lo16);
                    const u32x4 i0 = idl[0], i1 = idl[64];
                    f32x16 d;
#pragma unroll
                    for (int r = 0; r < 16; ++r) d[r] = 0.0f;
                    d = mfma_bf16(z1[0], make_uint4(i0.x, i0.y, i0.z, i0.w), d);
                    d = mfma_bf16(z1[1], make_uint4(i1.x, i1.y, i1.z, i1.w), d);
                    float sd = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sd += d[r];
                    db1[i] += sd;
                    uint4 zt[2];
                    pack_tile(d, zt);
                    stg16(a.z1f + (((size_t)tile * NT + ft) * 2 + 0) * 64, lo16, zt[0]);
                    stg16(a.z1f + (((size_t)tile * NT + ft) * 2 + 1) * 64, lo16, zt[1]);
                } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int off = wrow + 64 * ft + 8 * ((2 * g + h) ^ wsw);
                    *reinterpret_cast<uint2*>(imgZ1 + off) = make_uint2(dw(z1[g >> 1], 2 * (g & 1)), dw(z1[g >> 1], 2 * (g & 1) + 1));
                }
                }
            }
        }
        // a wave's dZ1 columns are read back (transposed) by that wave only: its own LDS writes just have to land.
        // Phase C writes no LDS, so the other waves' phase-B reads need no barrier here; the one at the end of the
        // tile keeps the next phase A from overwriting the images early.
        // (identity-MFMA form: no dZ1 image, nothing to wait for -- and without the wait hipcc may run the epilogue's vector
        // work beside phase C's transposed reads and MFMAs)
        if constexpr (!TRN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BSTAMP(4);
        // ================= phase C: products that contract over the 32 rows (operands: transposed image reads)
        {
            // (interleaving these MFMAs with the dH1 chain to hide its W2^T refills was measured slower: the chain's
            // two working accumulators then compete with the 256 resident ones for the AGPR half and hipcc parks two
            // dW2 tiles in scratch)
            auto emit_z = [&](int i, int s) {
                const uint4 z = tr_frag(imgZ1 + tro[0] + 16 * s * ST + 64 * (w * FT + i), imgZ1 + tro[1] + 16 * s * ST + 64 * (w * FT + i));
                db1[i] += sum_frag(z);
                stg16(a.z1f + (((size_t)tile * NT + (w * FT + i)) * 2 + s) * 64, lo16, z);
            };
            auto emit_x = [&](int i) {
                const int it = w * FT + i;                     // wave-uniform
                if (it < NI) {
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        stg16(a.xf + (((size_t)tile * NI + it) * 2 + s) * 64, lo16,
                              tr_frag(imgX + trx[0] + 16 * s * STX + 64 * it, imgX + trx[1] + 16 * s * STX + 64 * it));
                }
            };
            uint4 az[FT][2];
#pragma unroll
            for (int i = 0; i < FT; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    az[i][s] = tr_frag(imgZ2 + tro[0] + 16 * s * ST + 64 * (w * FT + i), imgZ2 + tro[1] + 16 * s * ST + 64 * (w * FT + i));
                    db2[i] += sum_frag(az[i][s]);
                }
            // EPI_C: the dH1 epilogue (dZ1 = dH1 . lrelu'(H1), two accumulator registers per step) issues behind the dW2 MFMAs
            // of each k-tile -- the chain that produced dH1 is done, the two are independent, and a bf16 MFMA leaves the
            // vector ALU free for the 32 clocks it holds the matrix pipe
            uint2 hcq[FT][4];
            if constexpr (EPI_C) {
                BF16_SWZ();
#pragma unroll
                for (int i = 0; i < FT; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) hcq[i][g] = *reinterpret_cast<const uint2*>(imgH1 + wrow + 64 * (w * FT + i) + 8 * ((2 * g + h) ^ wsw));
            }
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                uint4 b[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) b[s] = tr_frag(imgH1 + tro[0] + 16 * s * ST + 64 * kt, imgH1 + tro[1] + 16 * s * ST + 64 * kt);
#pragma unroll
                for (int i = 0; i < FT; ++i)
#pragma unroll
                    for (int s = 0; s < 2; ++s) accW2[i][kt] = mfma_bf16(az[i][s], b[s], accW2[i][kt]);
                if constexpr (EPI_C) {
                    constexpr int RPS = 16 / NT;
#pragma unroll
                    for (int i = 0; i < FT; ++i)
#pragma unroll
                        for (int rr = 0; rr < RPS; ++rr) {
                            const int r = RPS * kt + rr;
                            const uint32_t d = (r & 2) ? hcq[i][r >> 2].y : hcq[i][r >> 2].x;
                            accH[i][r] = accH[i][r] * ((r & 1) ? slope_hi(d) : slope_lo(d));
                        }
                }
            }
            if constexpr (EPI_C) {
                // the rest of the epilogue: RNE pack, the identity-MFMA transpose, db1, the fragments for the dW1 kernel
                const unsigned lds0 = (unsigned)(size_t)(PPO_LDS void*)smem_c;
                const PPO_LDS u32x4* const idl = (const PPO_LDS u32x4*)(size_t)(lds0 + (unsigned)C::oID + lo16);
                const u32x4 i0 = idl[0], i1 = idl[64];
#pragma unroll
                for (int i = 0; i < FT; ++i) {
                    const int ft = w * FT + i;
                    uint4 z1[2];
                    pack_tile(accH[i], z1);
                    f32x16 d;
#pragma unroll
                    for (int r = 0; r < 16; ++r) d[r] = 0.0f;
                    d = mfma_bf16(z1[0], make_uint4(i0.x, i0.y, i0.z, i0.w), d);
                    d = mfma_bf16(z1[1], make_uint4(i1.x, i1.y, i1.z, i1.w), d);
                    float sd = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sd += d[r];
                    db1[i] += sd;
                    uint4 zt[2];
                    pack_tile(d, zt);
                    stg16(a.z1f + (((size_t)tile * NT + ft) * 2 + 0) * 64, lo16, zt[0]);
                    stg16(a.z1f + (((size_t)tile * NT + ft) * 2 + 1) * 64, lo16, zt[1]);
                }
            }
            BSTAMP(5);
            BSTAMP(6);
            if constexpr (FUSE1) {
            // dW1[k][i] += sum_rows dZ1[k][row] X[row][i]: both operands come back transposed from their LDS images (the
            // fragments the HID = 256 form sends through HBM to a second kernel), NI x 2 MFMAs per feature tile
            (void)emit_z; (void)emit_x;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 xb[NI];
#pragma unroll
                for (int it = 0; it < NI; ++it) xb[it] = tr_frag(imgX + trx[0] + 16 * s * STX + 64 * it, imgX + trx[1] + 16 * s * STX + 64 * it);
#pragma unroll
                for (int i = 0; i < FT; ++i) {
                    const uint4 z = tr_frag(imgZ1 + tro[0] + 16 * s * ST + 64 * (w * FT + i), imgZ1 + tro[1] + 16 * s * ST + 64 * (w * FT + i));
                    db1[i] += sum_frag(z);
#pragma unroll
                    for (int it = 0; it < NI; ++it) accW1[FUSE1 ? i : 0][FUSE1 ? it : 0] = mfma_bf16(z, xb[it], accW1[FUSE1 ? i : 0][FUSE1 ? it : 0]);
                }
            }
            } else {
            // dZ1^T of this wave's feature tiles and (first NI slots) the X column tiles, as MFMA operand fragments
            // for k_policy_dw1_bf16
            if constexpr (!TRN) {
#pragma unroll
            for (int i = 0; i < FT; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) emit_z(i, s);
            }
            (void)emit_x;                                      // the dW1 kernel builds its X operands from the state rows itself
            }
        }
        // layer-1 B operands of the NEXT tile (its state bytes were the first piece fetched during phase B) into sXF: read
        // by everybody in that tile's phase A, behind the barrier below; this tile's readers are all past barrier 1
        if constexpr (RC1) { store_xfrag(); load_w1(); }
        __syncthreads();
        BSTAMP(7);
    }
#ifdef PPO_BF16_STAMP
    if (a.stamps && lane == 0 && w < 4)
        for (int i = 0; i < 8; ++i) a.stamps[((size_t)blockIdx.x * 4 + w) * 8 + i] = st_sum[i];
#endif

    // ================= slab (same fragment order as the fp32 kernel: k_grad_reduce maps it to Flux order);
    // the dW1 region is written by k_policy_dw1_bf16
    float* slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
    float* sW2 = slab;
    float* sW1 = sW2 + (size_t)HID * HID;
    float* sb1 = sW1 + (size_t)HID * C::FP;
    float* sb2 = sb1 + HID;
    float* sw3 = sb2 + HID;
    float* sb3 = sw3 + HID * 4;
#pragma unroll
    for (int i = 0; i < FT; ++i) {
        const int ft = w * FT + i;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sW2[((size_t)(ft * NT + kt) * 16 + r) * 64 + lane] = accW2[i][kt][r];
        if constexpr (FUSE1) {
#pragma unroll
            for (int it = 0; it < NI; ++it)
#pragma unroll
                for (int r = 0; r < 16; ++r) sW1[((size_t)(ft * NI + it) * 16 + r) * 64 + lane] = accW1[FUSE1 ? i : 0][FUSE1 ? it : 0][r];
        }
        const float b1 = db1[i] + __shfl_xor(db1[i], 32), b2 = db2[i] + __shfl_xor(db2[i], 32);
        float d3[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) d3[o] = dw3[i][o] + __shfl_xor(dw3[i][o], 32);
        if (h == 0) {
            const int f = 32 * ft + j;
            sb1[f] = b1; sb2[f] = b2;
            *reinterpret_cast<float4*>(&sw3[f * 4]) = make_float4(d3[0], d3[1], d3[2], d3[3]);
        }
    }
    if (w == 0) {
        const float4 db3 = sDB3[j];                           // (lanes 32..63 read the same rows: the butterfly below stays inside a half)
        float t4[4] = {db3.x, db3.y, db3.z, db3.w};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) t4[o] += __shfl_xor(t4[o], off);
        }
        if (lane == 0) { sb3[0] = t4[0]; sb3[1] = t4[1]; sb3[2] = t4[2]; sb3[3] = t4[3]; }
    }
}

// dW1[k][i] += sum_rows dZ1[k][row] X[row][i].  dZ1^T comes as ready-made operand fragments from k_policy_bwd_bf16
// (16 KB per tile through the Infinity Cache); the X operands are built HERE from the tile's 2.3 KB of int8 state rows
// (round 1 had the backward kernel emit them as fragments too: 6 KB written + 6 KB read per tile): the workgroup lays
// the rows down as a row-major bf16 image in LDS, double-buffered so that one barrier per tile is enough, and every
// wave reads its B operands back transposed (ds_read_b64_tr_b16).  Same tile -> workgroup assignment as the backward
// kernel, so workgroup g fills the dW1 region of slab g.  Wave w owns k-tile w (NI accumulator tiles).
template <int F, int HID>
__global__ __launch_bounds__(HID * 2) void k_policy_dw1_bf16(BwdBArgs a) {
    using C = BwdB<F, HID>;
    constexpr int NT = C::NT, NI = C::NI, STX = C::STX, NTHR = HID * 2;
    constexpr int XQW = 32 * F / 8;                              // 8-byte units of one tile's rows: thread u stages unit u
    constexpr int XPD = (XQW + NTHR - 1) / NTHR;                 // staging passes (1 for HID = 256, F = 72)
    static_assert(F % 8 == 0, "state rows are staged in 8-byte units");
    __shared__ __attribute__((aligned(16))) char imgX[2][32 * STX];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 32 * STX / 4; i += NTHR) reinterpret_cast<uint32_t*>(&imgX[0][0])[i] = 0u;   // pad columns stay zero
    f32x16 acc[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[it][r] = 0.0f;
    // transposed-read addressing of the X image (no swizzle; see k_policy_bwd_bf16)
    const int tq = (lane & 15) >> 2, tcc = 4 * ((lane >> 4) & 1) + (lane & 3);
    int trx[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)     // rows of k-slot element e: 8h + e in the natural order, (e&3) + 8(e>>2) + 4h behind the identity-MFMA transposes
        trx[u] = (C::TRN ? (4 * h + 8 * u + tq) : (8 * h + 4 * u + tq)) * STX + 8 * tcc;
    int xw_off[XPD];                                             // image byte offset of this thread's 8 features
#pragma unroll
    for (int i = 0; i < XPD; ++i) { const int u = tid + i * NTHR; xw_off[i] = (u / (F / 8)) * STX + (u % (F / 8)) * 16; }
    const unsigned lo16 = (unsigned)lane * 16u;
    // next tile's inputs are fetched while the current tile's MFMAs run (two waves per SIMD, HBM/MALL-bound); the
    // transition id of the tile after that is fetched one tile earlier still (it is itself a global load)
    uint4 z[2];
    uint2 nx[XPD];
#pragma unroll
    for (int i = 0; i < XPD; ++i) nx[i] = make_uint2(0u, 0u);
    auto tile_or_last = [&](int t) { return t < a.B ? t : a.B - 1; };
    auto fetch = [&](int t, int sidx_v) {
#pragma unroll
        for (int s = 0; s < 2; ++s) z[s] = ldg16(a.z1f + (((size_t)t * NT + w) * 2 + s) * 64, lo16);
        const int sidx = __builtin_amdgcn_readfirstlane(sidx_v);
        const char* xs = reinterpret_cast<const char*>(a.states + (a.x_by_tile ? (size_t)t : (((size_t)sidx << a.tps_shift) + (size_t)(t & ((1 << a.tps_shift) - 1)))) * 32 * F);
#pragma unroll
        for (int i = 0; i < XPD; ++i)
            if (tid + i * NTHR < XQW) nx[i] = *reinterpret_cast<const uint2*>(xs + (unsigned)(tid + i * NTHR) * 8u);
    };
    int idx_next = 0;
    if ((int)blockIdx.x < a.B) {
        fetch((int)blockIdx.x, a.x_by_tile ? 0 : a.idx[(int)blockIdx.x >> a.tps_shift]);
        if (!a.x_by_tile) idx_next = a.idx[tile_or_last((int)blockIdx.x + (int)gridDim.x) >> a.tps_shift];
    }
    __syncthreads();                                             // images zeroed
    int buf = 0;
    for (int tile = blockIdx.x; tile < a.B; tile += gridDim.x, buf ^= 1) {
        char* const img = &imgX[buf][0];
#pragma unroll
        for (int p = 0; p < XPD; ++p) {
            if (tid + p * NTHR < XQW) {
                float f[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { f[i] = (float)(int)(int8_t)(nx[p].x >> (8 * i)); f[4 + i] = (float)(int)(int8_t)(nx[p].y >> (8 * i)); }
                *reinterpret_cast<uint4*>(img + xw_off[p]) = make_uint4(pack_bf16(f[0], f[1]), pack_bf16(f[2], f[3]), pack_bf16(f[4], f[5]), pack_bf16(f[6], f[7]));
            }
        }
        uint4 cz[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) cz[s] = z[s];
        // one barrier per tile: image `buf` is complete behind it, and nobody still reads the image the NEXT tile writes
        // (its readers, one tile back, are all in front of this barrier)
        __syncthreads();
        {
            const int nt = tile_or_last(tile + (int)gridDim.x);   // harmless re-load on the last tile
            const int sidx = idx_next;
            if (!a.x_by_tile) idx_next = a.idx[tile_or_last(tile + 2 * (int)gridDim.x) >> a.tps_shift];
            fetch(nt, sidx);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 xb[NI];
#pragma unroll
            for (int it = 0; it < NI; ++it) xb[it] = tr_frag(img + trx[0] + 16 * s * STX + 64 * it, img + trx[1] + 16 * s * STX + 64 * it);
#pragma unroll
            for (int it = 0; it < NI; ++it) acc[it] = mfma_bf16(cz[s], xb[it], acc[it]);
        }
    }
    float* sW1 = a.slabs + (size_t)blockIdx.x * a.slab_stride + (size_t)HID * HID;
#pragma unroll
    for (int it = 0; it < NI; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) sW1[((size_t)(w * NI + it) * 16 + r) * 64 + lane] = acc[it][r];
}

#ifdef PPO_BF16_STAMP
static unsigned long long* g_bf16_stamps = nullptr;
extern "C" int32_t ppo_debug_bf16_stamps(unsigned long long* out) {
    if (!g_bf16_stamps) return -1;
    (void)hipDeviceSynchronize();
    return hipMemcpy(out, g_bf16_stamps, 256 * 4 * 8 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

int32_t launch_policy_bwd_bf16(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B) {
    BwdBArgs a;
    const int tps = ro->H / 32;
    ARG_CHECK((tps == 1 || tps == 4) && B * tps < ((int64_t)1 << 30), "bf16 backward: H must be 32 or 128 and the minibatch below 2^30 tiles");
    a.tps_shift = (tps == 4) ? 2 : 0;
    a.states = ro->compact ? p->xs.p : ro->states.p; a.x_by_tile = ro->compact ? 1 : 0;
    a.idx = idx_dev; a.B = (int32_t)(B * tps);
    a.act1b = (const uint4*)p->act1.p; a.act2b = (const uint4*)p->act2.p; a.dY = (const float4*)p->dY.p;
    a.w2tb = (const uint4*)p->w2tb.p; a.w3tb = (const uint2*)p->w3tb.p;
    a.w1b = (const uint4*)p->w1b.p; a.b1p = (const float4*)p->b1p.p;
    // the fp32 mode's activation buffers are twice the size the bf16 activations need: the operand fragments for the
    // dW1 kernel live in their upper halves (z1f behind act1b, xf behind act2b; NI <= NT)
    a.z1f = (uint4*)p->act1.p + (size_t)a.B * (p->HID / 32) * 128;
    a.xf = (uint4*)p->act2.p + (size_t)a.B * (p->HID / 32) * 128;
    ARG_CHECK(p->act1.n >= (size_t)a.B * (p->HID / 32) * 1024 && p->act2.n >= (size_t)a.B * (p->HID / 32) * 1024,
              "bf16 backward: activation workspace smaller than the minibatch (train_reserve)");
    a.slabs = p->slabs.p; a.slab_stride = slab_floats(p->F, p->HID);
    a.stamps = nullptr;
#ifdef PPO_BF16_STAMP
    if (!g_bf16_stamps) (void)hipMalloc((void**)&g_bf16_stamps, 256 * 4 * 8 * 8);
    a.stamps = g_bf16_stamps;
#endif
    const int nwg = (int)(a.B < 256 ? a.B : 256);
    p->nwg_bwd = nwg; p->nwg_small = 0;
    a.nwg = nwg;
#define LAUNCHB(FF, HH)                                                                                           \
    do {                                                                                                          \
        static_assert(BwdB<FF, HH>::NI <= BwdB<FF, HH>::NT, "X fragments are emitted by waves 0..NI-1");          \
        const size_t lds = BwdB<FF, HH>::total;                                                                   \
        static thread_local bool attr_set = false;                                                                             \
        if (!attr_set) {                                                                                          \
            HIP_TRY(hipFuncSetAttribute((const void*)k_policy_bwd_bf16<FF, HH>,                                   \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
            attr_set = true;                                                                                      \
        }                                                                                                         \
        {                                                                                                         \
            ProfScope ps("k_policy_bwd");                                                                         \
            hipLaunchKernelGGL((k_policy_bwd_bf16<FF, HH>), dim3(nwg), dim3(BwdB<FF, HH>::NW * 64), lds, ppo_stream(), a); \
        }                                                                                                         \
        if (HH > PPO_BF16_DW1_FUSED_MAX_HID) {                                                                    \
            ProfScope ps("k_policy_dw1");                                                                         \
            hipLaunchKernelGGL((k_policy_dw1_bf16<FF, HH>), dim3(nwg), dim3(HH * 2), 0, ppo_stream(), a);         \
        }                                                                                                         \
    } while (0)
    if (p->F == 72 && p->HID == 256) LAUNCHB(72, 256);
    else if (p->F == 72 && p->HID == 128) LAUNCHB(72, 128);
    else { ppo_set_error("unsupported policy shape (F,HID) for the gfx950 bf16 kernels"); return PPO_ERR_UNSUPPORTED; }
#undef LAUNCHB
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}
