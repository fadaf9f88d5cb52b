// ppo_policy_fwd_x6.hip -- the TRAIN forward (policy + ppo_loss_with_entropy terms + dL/dlogits: src/train.jl:35-46,65-79
// forward half; test/policy.jl:21-31) with its two Dense products on the bf16 matrix pipe as split-fp32 ("bf16x6")
// products, like the backward in ppo_policy_bwd_x6.hip.  Rollouts and the probability entry points keep the fp32-MFMA
// kernel (their sampled actions are pinned bit for bit); the train forward's outputs -- saved activations, dL/dlogits, the
// two loss sums -- feed tolerance-checked quantities only (gradient within 2e-5 max|g| of the float64 restatement).
//
// Shape of the kernel: a workgroup of HID/32 waves owns a 32-row tile (one state), wave w owns feature tile w of both hidden
// layers (the k_policy_train_tile arrangement):
//   layer 1   H1^T[f, row] = lrelu(W1 X^T + b1): A = W1 pieces streamed from L2 (3 x 5 k-steps), B = the state rows, exact
//             in bf16 (converted in registers by every wave: 2.3 KB from L1/L2) -> 15 MFMAs; the tile is stored for the
//             backward (accumulator-fragment order, as k_policy_fwd does) and, split in three, written to LDS as the
//             B-operand fragments of layer 2 (the packed accumulator registers ARE that operand)
//   layer 2   H2^T = lrelu(W2 H1^T + b2): A = W2 pieces from L2 (lo, mid, hi per k-step: 1 + 2 + 3 MFMAs), B = the H1
//             fragments of all feature tiles from LDS -> 96 MFMAs; stored; layer-3 partial dots on the VALU (fp32, as in
//             every forward) -> LDS; wave 0 adds the partials in wave order and runs the shared loss tail (policy_tail)
// 128 registers per wave and 62 KB of LDS per workgroup: TWO workgroups share a CU at HID = 256 (four at 128) and run out
// of phase, so one workgroup's conversions / splits / tail sit beside the other's MFMAs.
// The weights stream from L2 once per TILE (63 KB per wave, 0.5 MB per tile and CU: 2 GB per 4096-state launch, which the
// eight L2s deliver).  Measured against the one-wave-per-state fp32-MFMA forward (gpurun_out/x6h, same box): 4096 states
// 0.211 -> 0.142 ms, 2048 0.110 -> 0.075, 1024 0.059 -> 0.039, 512 0.031 -> 0.023.  ppo_set_bwd_split_bf16(0) or
// PPO_FWD_SPLIT_MAX_TILES=0 select the fp32-MFMA forward.
#include "ppo_policy_tail.h"
#include <type_traits>
#include "ppo_x6.h"
#include <cstdlib>

#ifndef PPO_FX6_RING
#define PPO_FX6_RING 6                // W2 piece fragments in flight ahead of layer 2 (3 per k-step)
#endif
#ifndef PPO_FX6_ACC3
#define PPO_FX6_ACC3 1                // 1: one accumulator per term level (3); 0: leading + small (2)
#endif
#define X6F_LANE() unsigned ln = (unsigned)lane; asm volatile("" : "+v"(ln)); const int j = (int)(ln & 31u), h = (int)(ln >> 5); (void)j; (void)h

// A/B knob (make -C csrc fsidx): the one- and two-tile train forwards read their transition ids (the two-tile one also the active word of its tail) through the
// scalar cache instead of vector loads + readfirstlane, each of which waits for vmcnt(0) -- every store and load in flight.
// Not measured yet: off.
#ifndef PPO_FX6_SIDX
#define PPO_FX6_SIDX 0
#endif
// A/B knob (make -C csrc fzpipe): layer 2 of the two-tile train forward reads the H1 pieces one (k-step, tile) ahead of their MFMAs
// (today each set is read right in front of its six MFMAs: two LDS latencies per k-step that only the SIMD partner can hide).
// Not measured yet: off.
#ifndef PPO_FX6_ZPIPE
#define PPO_FX6_ZPIPE 0
#endif
// A/B knob (make -C csrc fxearly; implies the scalar-cache ids): the next pass's state rows are requested right after this pass's are
// converted, behind a W1 ring that holds all 15 pieces -- vmcnt counts in order, so where they are issued today (in front of barrier 1,
// behind the W2 ring fill) the first MFMA of layer 2 waits for them: one HBM latency per pass on all eight waves.  Not measured yet: off.
#ifndef PPO_FX6_XEARLY
#define PPO_FX6_XEARLY 0
#endif
#if PPO_FX6_XEARLY
#undef PPO_FX6_SIDX
#define PPO_FX6_SIDX 1
#endif
// A/B knob (make -C csrc fnodangle): the last ring round of layer 2 of the two-tile train forward does not issue the reloads that run
// past the stream.  Not measured yet: off.
#ifndef PPO_FX6_NODANGLE
#define PPO_FX6_NODANGLE 0
#endif

template <int HID>
struct FXCfg {
    static constexpr int F = 72, NT = HID / 32, KS = HID / 16, K1 = 5;     // layer-1 k-steps: 72 inputs zero padded to 80
    static constexpr size_t oFr = 0, szFr = (size_t)NT * 6 * 1024;         // H1 fragments [feature tile][k-step 2][piece 3][64 lanes][16 B]
    static constexpr size_t oP = oFr + szFr, oW3 = oP + (size_t)NT * 1024, oB1 = oW3 + (size_t)2 * NT * 256,
                            oB2 = oB1 + (size_t)NT * 128, total = oB2 + (size_t)NT * 128;
    static constexpr int WG_PER_CU = HID == 256 ? 2 : 4;
    static_assert(total * WG_PER_CU <= 160 * 1024, "LDS budget");
};

template <int HID>
__global__ __launch_bounds__(HID * 2, 4) void k_policy_fwd_train_x6(FwdArgs a, const uint4* __restrict__ w1x, const uint4* __restrict__ w2fx, int x_by_tile) {
    using C = FXCfg<HID>;
    constexpr int F = C::F, NT = C::NT, KS = C::KS, K1 = C::K1;
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* const frag = smem_c + C::oFr;
    float4* const sP = reinterpret_cast<float4*>(smem_c + C::oP);          // [NT][64] layer-3 partial dots
    float4* const sW3p = reinterpret_cast<float4*>(smem_c + C::oW3);       // [2][NT][16] the forward's layer-3 pack
    float4* const sB1 = reinterpret_cast<float4*>(smem_c + C::oB1);        // [NT][2][4]
    float4* const sB2 = reinterpret_cast<float4*>(smem_c + C::oB2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PPO_X6_ZERO_LDS
    for (int i = tid; i < (int)(C::total / 4); i += NT * 64) reinterpret_cast<uint32_t*>(smem_c)[i] = 0u;
    __syncthreads();
#endif
    for (int i = tid; i < 2 * NT * 16; i += NT * 64) sW3p[i] = a.w3p[i];
    for (int i = tid; i < NT * 8; i += NT * 64) { sB1[i] = a.b1p[i]; sB2[i] = a.b2p[i]; }
    __syncthreads();
    const char* const w1s = reinterpret_cast<const char*>(w1x + (size_t)w * K1 * 3 * 64);     // [k-step][piece lo, mid, hi][64][8]
    const char* const w2s = reinterpret_cast<const char*>(w2fx + (size_t)w * KS * 3 * 64);
    char* const fown = frag + (size_t)w * 6 * 1024;

    // the state rows of a tile as this lane's B-operand source: 8 int8 per k-step, inputs 16s + 8h .. +7 of row j
    uint2 xr[K1];
    // (x_by_tile: `states` holds the minibatch's rows in minibatch order -- compact rollouts, expanded by k_expand_states)
    auto load_x = [&](int64_t rec, unsigned ln) {
        const char* row = reinterpret_cast<const char*>(a.states) + (size_t)rec * 32 * F + (ln & 31u) * (unsigned)F + (ln >> 5) * 8u;
#pragma unroll
        for (int s = 0; s < K1; ++s) {
            // k-step 4 covers inputs 64 .. 79: the upper lane half (72 .. 79) is padding (and would read past the row)
            const bool pad = (s == K1 - 1) && (ln >> 5);
            const uint2 v = *reinterpret_cast<const uint2*>(row + (pad ? 0 : 16 * s));
            xr[s] = pad ? make_uint2(0u, 0u) : v;
        }
    };
    int32_t sid = 0;
    if ((int64_t)blockIdx.x < a.B) { sid = __builtin_amdgcn_readfirstlane(a.idx[blockIdx.x]); load_x(x_by_tile ? (int64_t)blockIdx.x : (int64_t)sid, (unsigned)lane); }

    for (int64_t tile = blockIdx.x; tile < a.B; tile += gridDim.x) {
        const uint32_t act = a.active[sid];
        const int32_t sid_cur = sid;
        // ================= layer 1: H1 tile w
        {
            X6F_LANE();
            unsigned lo16 = ln * 16u;
            constexpr int R1 = 8;                                   // W1 piece fragments in flight
            uint4 ring[R1];
#pragma unroll
            for (int g = 0; g < R1; ++g) ring[g] = *reinterpret_cast<const uint4*>(w1s + (lo16 + (unsigned)g * 1024u));
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = sB1[(w * 2 + h) * 4 + q];
                acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
            }
            // int8 -> bf16 (exact): the float of the byte, upper 16 bits
            uint4 xb[K1];
#pragma unroll
            for (int s = 0; s < K1; ++s) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (float)(int)(int8_t)((e < 4 ? xr[s].x : xr[s].y) >> (8 * (e & 3)));
                xb[s] = make_uint4(x_perm(v[0], v[1]), x_perm(v[2], v[3]), x_perm(v[4], v[5]), x_perm(v[6], v[7]));
            }
            f32x16 accs;                                            // the mid and lo pieces of W1 (2^-8, 2^-16 of the leading terms)
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[r] = 0.0f;
#pragma unroll
            for (int st = 0; st < 3 * K1; ++st) {
                if (st % 3 == 2) acc = x_mfma(ring[st % R1], xb[st / 3], acc);
                else accs = x_mfma(ring[st % R1], xb[st / 3], accs);
                __builtin_amdgcn_sched_barrier(0);
                if (st + R1 < 3 * K1) ring[st % R1] = *reinterpret_cast<const uint4*>(w1s + (lo16 + (unsigned)(st + R1) * 1024u));
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = acc[r] + accs[r];
            asm volatile("" : "+v"(acc));
            lrelu16(acc);
            float4* dst = a.act1 + ((size_t)tile * NT + w) * 4 * 64;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                typedef float f32x4l __attribute__((ext_vector_type(4)));
                const f32x4l t = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                __builtin_nontemporal_store(t, reinterpret_cast<f32x4l*>(dst + q * 64 + ln));
            }
            // registers 8s .. 8s+7, split and packed = the B-operand fragment of k-step (w, s) of layer 2
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint2 zh[2], zm[2], zl[2];
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int q = 2 * s + qq;
                    const float hv[4] = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                    x_split4(hv, zh[qq], zm[qq], zl[qq]);
                }
                *reinterpret_cast<uint4*>(fown + (s * 3 + 0) * 1024 + ln * 16) = make_uint4(zh[0].x, zh[0].y, zh[1].x, zh[1].y);
                *reinterpret_cast<uint4*>(fown + (s * 3 + 1) * 1024 + ln * 16) = make_uint4(zm[0].x, zm[0].y, zm[1].x, zm[1].y);
                *reinterpret_cast<uint4*>(fown + (s * 3 + 2) * 1024 + ln * 16) = make_uint4(zl[0].x, zl[0].y, zl[1].x, zl[1].y);
            }
        }
        // the W2 piece ring of layer 2 is in flight across the barrier; so are the next tile's state rows
        constexpr int RD = PPO_FX6_RING;
        static_assert(RD % 3 == 0 && KS % (RD / 3) == 0, "ring rounds");
        uint4 ring[RD];
        {
            unsigned lo = (unsigned)lane * 16u;
            asm volatile("" : "+v"(lo));
#pragma unroll
            for (int g = 0; g < RD; ++g) ring[g] = *reinterpret_cast<const uint4*>(w2s + (lo + (unsigned)g * 1024u));
        }
        const int64_t ntile = (tile + gridDim.x < a.B) ? tile + gridDim.x : tile;
        {
            unsigned ln2 = (unsigned)lane;
            asm volatile("" : "+v"(ln2));
#if PPO_FX6_SIDX
            {
                const int32_t* const ip = a.idx + ntile;
                asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sid) : "s"(ip) : "memory");
            }
#else
            sid = __builtin_amdgcn_readfirstlane(a.idx[ntile]);
#endif
            load_x(x_by_tile ? ntile : (int64_t)sid, ln2);
        }
        __syncthreads();                                                // (1) every layer-1 tile is in LDS
        // ================= layer 2: H2 tile w from all layer-1 tiles; layer-3 partial dots
        {
            X6F_LANE();
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = sB2[(w * 2 + h) * 4 + q];
                acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
            }
            // one accumulator per term level (h h | h m + m h | h l + m m + l h): the 2^-8 and 2^-16 terms are summed among
            // themselves and meet the leading sum in two fp32 additions at the end.  Inside one MFMA the 16 products and the
            // accumulator are aligned to the largest of them before they are added (tools/microbench/mfma_bf16_accumulate.hip):
            // small terms fed into the leading accumulator would lose their low bits 96 times per output
            f32x16 accm, accl_;
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[r] = 0.0f; accl_[r] = 0.0f; }
            f32x16& accl = PPO_FX6_ACC3 ? accl_ : accm;
            const unsigned lo16 = ln * 16u;
            const char* zp = frag + lo16;
            const char* wn = w2s + (size_t)RD * 1024;
#pragma unroll 1
            for (int k0 = 0; k0 < KS; k0 += RD / 3) {
#pragma unroll
                for (int u = 0; u < RD / 3; ++u) {
                    const uint4 z_h = *reinterpret_cast<const uint4*>(zp + (u * 3 + 0) * 1024);
                    const uint4 z_m = *reinterpret_cast<const uint4*>(zp + (u * 3 + 1) * 1024);
                    const uint4 z_l = *reinterpret_cast<const uint4*>(zp + (u * 3 + 2) * 1024);
                    accl = x_mfma(ring[3 * u + 0], z_h, accl);
                    __builtin_amdgcn_sched_barrier(0);
                    ring[3 * u + 0] = *reinterpret_cast<const uint4*>(wn + lo16);          // the last round reads RD KiB ahead (padding / next wave's stream)
                    __builtin_amdgcn_sched_barrier(0);
                    accl = x_mfma(ring[3 * u + 1], z_m, accl);
                    accm = x_mfma(ring[3 * u + 1], z_h, accm);
                    __builtin_amdgcn_sched_barrier(0);
                    ring[3 * u + 1] = *reinterpret_cast<const uint4*>(wn + 1024 + lo16);
                    __builtin_amdgcn_sched_barrier(0);
                    accl = x_mfma(ring[3 * u + 2], z_l, accl);
                    accm = x_mfma(ring[3 * u + 2], z_m, accm);
                    acc = x_mfma(ring[3 * u + 2], z_h, acc);
                    __builtin_amdgcn_sched_barrier(0);
                    ring[3 * u + 2] = *reinterpret_cast<const uint4*>(wn + 2048 + lo16);
                    wn += 3 * 1024;
                    __builtin_amdgcn_sched_barrier(0);
                }
                zp += (RD / 3) * 3 * 1024;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = acc[r] + (PPO_FX6_ACC3 ? accm[r] + accl_[r] : accm[r]);
            asm volatile("" : "+v"(acc));
            lrelu16(acc);
            float4* dst = a.act2 + ((size_t)tile * NT + w) * 4 * 64;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                typedef float f32x4l __attribute__((ext_vector_type(4)));
                const f32x4l t = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                __builtin_nontemporal_store(t, reinterpret_cast<f32x4l*>(dst + q * 64 + ln));
            }
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
            const float4* w3 = sW3p + (h * NT + w) * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 wv = w3[r];
                p0 = fmaf(wv.x, acc[r], p0); p1 = fmaf(wv.y, acc[r], p1);
                p2 = fmaf(wv.z, acc[r], p2); p3 = fmaf(wv.w, acc[r], p3);
            }
            sP[w * 64 + ln] = make_float4(p0, p1, p2, p3);
        }
        __syncthreads();                                                // (2) every wave's partial logits are in LDS; the H1 fragments are free
        if (w == 0) {
            // partial logits in wave (= feature tile) order, the two lane halves, b3; then the epilogue every forward shares
            X6F_LANE();
            float4 s = sP[ln];
#pragma unroll
            for (int u = 1; u < NT; ++u) { const float4 q4 = sP[u * 64 + ln]; s.x += q4.x; s.y += q4.y; s.z += q4.z; s.w += q4.w; }
            float l[1][4];
            l[0][0] = (s.x + __shfl_xor(s.x, 32)) + a.b3[0];
            l[0][1] = (s.y + __shfl_xor(s.y, 32)) + a.b3[1];
            l[0][2] = (s.z + __shfl_xor(s.z, 32)) + a.b3[2];
            l[0][3] = (s.w + __shfl_xor(s.w, 32)) + a.b3[3];
            policy_tail<2, 1, false>(a, tile, sid_cur, act, l, (int)ln, j, h);
        }
    }
}

// ---------------------------------------------------------------- T state tiles per workgroup pass (large minibatches)
// The one-tile form streams 63 KB of weight pieces per wave and TILE: 2 GB per 4096-state launch = 15 TB/s at its 0.134 ms, i.e.
// the eight L2s' limit (MI355X_MICROARCH.md: 16.8-18.8 TB/s for lines every workgroup shares).  Here a workgroup takes T tiles
// through both layers against ONE pass over its weight stream: every W1 / W2 piece fragment feeds T MFMAs.  T x 48 KB of H1
// fragments: one workgroup per CU, two waves per SIMD; the T independent accumulator chains per wave stand in for the second
// workgroup's latency hiding.
template <int HID, int T>
struct FXTCfg {
    static constexpr int F = 72, NT = HID / 32, KS = HID / 16, K1 = 5;
    static constexpr size_t oFr = 0, szFr = (size_t)T * NT * 6 * 1024;
    static constexpr size_t oP = oFr + szFr, oW3 = oP + (size_t)T * NT * 1024, oB1 = oW3 + (size_t)2 * NT * 256,
                            oB2 = oB1 + (size_t)NT * 128, total = oB2 + (size_t)NT * 128;
    static_assert(total <= 160 * 1024, "LDS budget");
};

template <int HID, int T>
__global__ __launch_bounds__(HID * 2, 2) void k_policy_fwd_train_x6t(FwdArgs a, const uint4* __restrict__ w1x, const uint4* __restrict__ w2fx, int x_by_tile) {
    using C = FXTCfg<HID, T>;
    constexpr int F = C::F, NT = C::NT, KS = C::KS, K1 = C::K1;
    static_assert(T <= NT, "one tail wave per tile");
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* const frag = smem_c + C::oFr;                                    // [T][feature tile][k-step 2][piece 3][64 lanes][16 B]
    float4* const sP = reinterpret_cast<float4*>(smem_c + C::oP);          // [T][NT][64] layer-3 partial dots
    float4* const sW3p = reinterpret_cast<float4*>(smem_c + C::oW3);
    float4* const sB1 = reinterpret_cast<float4*>(smem_c + C::oB1);
    float4* const sB2 = reinterpret_cast<float4*>(smem_c + C::oB2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PPO_FX6_STAMP
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = clock64();
#define FXSTAMP(i) do { unsigned long long _n = clock64(); st_sum[i] += _n - st_t; st_t = _n; } while (0)
#else
#define FXSTAMP(i) do { } while (0)
#endif
    for (int i = tid; i < 2 * NT * 16; i += NT * 64) sW3p[i] = a.w3p[i];
    for (int i = tid; i < NT * 8; i += NT * 64) { sB1[i] = a.b1p[i]; sB2[i] = a.b2p[i]; }
    __syncthreads();
    const char* const w1s = reinterpret_cast<const char*>(w1x + (size_t)w * K1 * 3 * 64);
    const char* const w2s = reinterpret_cast<const char*>(w2fx + (size_t)w * KS * 3 * 64);

    uint2 xr[T][K1];
    auto load_x = [&](int i, int64_t rec, unsigned ln) {
        const char* row = reinterpret_cast<const char*>(a.states) + (size_t)rec * 32 * F + (ln & 31u) * (unsigned)F + (ln >> 5) * 8u;
#if PPO_FX6_XEARLY
        // every load unconditional (the padding lanes of the last k-step re-read the row's first bytes and mask them with a value
        // the compiler cannot see through): a load under an exec mask counts as "maybe not issued" in the compiler's vmcnt
        // arithmetic, and the waits of layer 1 would then include the first state-row loads
        unsigned keep = (ln >> 5) ? 0u : 0xFFFFFFFFu;
        asm volatile("" : "+v"(keep));
#pragma unroll
        for (int s = 0; s < K1; ++s) {
            const bool last = (s == K1 - 1);
            unsigned off = (last && (ln >> 5)) ? 0u : (unsigned)(16 * s);
            if (last) asm volatile("" : "+v"(off));
            const uint2 v = *reinterpret_cast<const uint2*>(row + off);
            xr[i][s] = last ? make_uint2(v.x & keep, v.y & keep) : v;
        }
#else
#pragma unroll
        for (int s = 0; s < K1; ++s) {
            const bool pad = (s == K1 - 1) && (ln >> 5);
            const uint2 v = *reinterpret_cast<const uint2*>(row + (pad ? 0 : 16 * s));
            xr[i][s] = pad ? make_uint2(0u, 0u) : v;
        }
#endif
    };
    // tile i of group g is g*T + i; a group that runs past the minibatch re-does the last tile and discards it
    auto tile_of = [&](int64_t g, int i) { const int64_t t = g * T + i; return t < a.B ? t : a.B - 1; };
#if PPO_FX6_SIDX
    // transition ids of a pass through the scalar cache, both in one wait (a vector load + readfirstlane waits on vmcnt(0), i.e.
    // on every activation store and operand load the wave has in flight, twice per pass)
    static_assert(T == 2, "two scalar loads per pass");
    int cid[T] = {0, 0}, nid[T] = {0, 0};
    auto sload_ids = [&](int64_t g, int (&id)[T]) {
        const int32_t* const ip0 = a.idx + tile_of(g, 0);
        const int32_t* const ip1 = a.idx + tile_of(g, 1);
        asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(id[0]), "=&s"(id[1]) : "s"(ip0), "s"(ip1) : "memory");
    };
    if ((int64_t)blockIdx.x * T < a.B) {
        sload_ids(blockIdx.x, cid);
#pragma unroll
        for (int i = 0; i < T; ++i) load_x(i, x_by_tile ? tile_of(blockIdx.x, i) : (int64_t)cid[i], (unsigned)lane);
    }
#else
    if ((int64_t)blockIdx.x * T < a.B) {
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int64_t t = tile_of(blockIdx.x, i);
            load_x(i, x_by_tile ? t : (int64_t)__builtin_amdgcn_readfirstlane(a.idx[t]), (unsigned)lane);
        }
    }
#endif
    for (int64_t g = blockIdx.x; g * T < a.B; g += gridDim.x) {
        // ================= layer 1: H1 tile w of the T states
        {
            X6F_LANE();
            unsigned lo16 = ln * 16u;
            constexpr int R1 = PPO_FX6_XEARLY ? 3 * K1 : 8;          // XEARLY: every W1 piece up front, no reload younger than the X loads below
            uint4 ring[R1];
#pragma unroll
            for (int q = 0; q < R1; ++q) ring[q] = *reinterpret_cast<const uint4*>(w1s + (lo16 + (unsigned)q * 1024u));
            f32x16 acc[T], accs[T];
            uint4 xb[T][K1];
#pragma unroll
            for (int i = 0; i < T; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = sB1[(w * 2 + h) * 4 + q];
                    acc[i][4 * q + 0] = b.x; acc[i][4 * q + 1] = b.y; acc[i][4 * q + 2] = b.z; acc[i][4 * q + 3] = b.w;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[i][r] = 0.0f;
#pragma unroll
                for (int s = 0; s < K1; ++s) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (float)(int)(int8_t)((e < 4 ? xr[i][s].x : xr[i][s].y) >> (8 * (e & 3)));
                    xb[i][s] = make_uint4(x_perm(v[0], v[1]), x_perm(v[2], v[3]), x_perm(v[4], v[5]), x_perm(v[6], v[7]));
                }
            }
#if PPO_FX6_XEARLY
            // the next pass's state rows leave for HBM here, as soon as this pass's are converted: they are younger than every W1
            // piece (all loaded above), so no wait in layer 1 includes them, and layer 1 + the H1 epilogue (3-4 k clocks) pass before
            // the first wait that does (the W2 ring in layer 2).  Issued in front of barrier 1, as before, that wait came
            // a few hundred clocks after them.
            {
                __builtin_amdgcn_sched_barrier(0);
                const int64_t gn = ((g + gridDim.x) * T < a.B) ? g + gridDim.x : g;
                sload_ids(gn, nid);
#pragma unroll
                for (int i = 0; i < T; ++i) load_x(i, x_by_tile ? tile_of(gn, i) : (int64_t)nid[i], ln);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
#pragma unroll
            for (int st = 0; st < 3 * K1; ++st) {
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    if (st % 3 == 2) acc[i] = x_mfma(ring[st % R1], xb[i][st / 3], acc[i]);
                    else accs[i] = x_mfma(ring[st % R1], xb[i][st / 3], accs[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (st + R1 < 3 * K1) ring[st % R1] = *reinterpret_cast<const uint4*>(w1s + (lo16 + (unsigned)(st + R1) * 1024u));
                __builtin_amdgcn_sched_barrier(0);
            }
            FXSTAMP(0);
#pragma unroll
            for (int i = 0; i < T; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = acc[i][r] + accs[i][r];
                asm volatile("" : "+v"(acc[i]));
                lrelu16(acc[i]);
                const int64_t tile = g * T + i;
                if (tile < a.B) {
                    float4* dst = a.act1 + ((size_t)tile * NT + w) * 4 * 64;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        typedef float f32x4l __attribute__((ext_vector_type(4)));
                        const f32x4l t = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
                        __builtin_nontemporal_store(t, reinterpret_cast<f32x4l*>(dst + q * 64 + ln));
                    }
                }
                char* const fown = frag + ((size_t)i * NT + w) * 6 * 1024;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    uint2 zh[2], zm[2], zl[2];
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int q = 2 * s + qq;
                        const float hv[4] = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
                        x_split4(hv, zh[qq], zm[qq], zl[qq]);
                    }
                    *reinterpret_cast<uint4*>(fown + (s * 3 + 0) * 1024 + ln * 16) = make_uint4(zh[0].x, zh[0].y, zh[1].x, zh[1].y);
                    *reinterpret_cast<uint4*>(fown + (s * 3 + 1) * 1024 + ln * 16) = make_uint4(zm[0].x, zm[0].y, zm[1].x, zm[1].y);
                    *reinterpret_cast<uint4*>(fown + (s * 3 + 2) * 1024 + ln * 16) = make_uint4(zl[0].x, zl[0].y, zl[1].x, zl[1].y);
                }
            }
        }
        FXSTAMP(1);
        constexpr int RD = 6;
        static_assert(KS % (RD / 3) == 0, "ring rounds");
        uint4 ring[RD];
        {
            unsigned lo = (unsigned)lane * 16u;
            asm volatile("" : "+v"(lo));
#pragma unroll
            for (int q = 0; q < RD; ++q) ring[q] = *reinterpret_cast<const uint4*>(w2s + (lo + (unsigned)q * 1024u));
        }
#if !PPO_FX6_XEARLY
        {
            const int64_t gn = ((g + gridDim.x) * T < a.B) ? g + gridDim.x : g;
            unsigned ln2 = (unsigned)lane;
            asm volatile("" : "+v"(ln2));
#if PPO_FX6_SIDX
            sload_ids(gn, nid);
#pragma unroll
            for (int i = 0; i < T; ++i) load_x(i, x_by_tile ? tile_of(gn, i) : (int64_t)nid[i], ln2);
#else
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const int64_t t = tile_of(gn, i);
                load_x(i, x_by_tile ? t : (int64_t)__builtin_amdgcn_readfirstlane(a.idx[t]), ln2);
            }
#endif
        }
#endif
        FXSTAMP(2);
        __syncthreads();                                                // (1) every layer-1 tile of the T states is in LDS
        FXSTAMP(3);
        // ================= layer 2
        {
            X6F_LANE();
            f32x16 acc[T], accs[T];
#pragma unroll
            for (int i = 0; i < T; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = sB2[(w * 2 + h) * 4 + q];
                    acc[i][4 * q + 0] = b.x; acc[i][4 * q + 1] = b.y; acc[i][4 * q + 2] = b.z; acc[i][4 * q + 3] = b.w;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[i][r] = 0.0f;
            }
            const unsigned lo16 = ln * 16u;
            const char* zp = frag + lo16;
            const char* wn = w2s + (size_t)RD * 1024;
#if PPO_FX6_ZPIPE
            // the H1 pieces of the NEXT (k-step, tile) are read from the LDS in front of the six MFMAs of the current one (one
            // set of lookahead, pinned), instead of right in front of the MFMAs that use them; the last set reads one k-step
            // past this tile's fragments (the next tile's, or the partial-dot buffer: inside the LDS block, never used)
            uint4 zc[3], zn[3];
            auto load_z = [&](const char* zb, uint4 (&z)[3]) {
                z[0] = *reinterpret_cast<const uint4*>(zb);
                z[1] = *reinterpret_cast<const uint4*>(zb + 1024);
                z[2] = *reinterpret_cast<const uint4*>(zb + 2048);
            };
            load_z(zp, zc);
#endif
#if PPO_FX6_NODANGLE
            // one ring round (RD / 3 k-steps).  LAST (PPO_FX6_NODANGLE): the round whose reloads would run past the wave's stream does
            // not issue them -- the ring registers are reused right behind the loop, and overwriting a register with a load in flight
            // costs a vmcnt(0) wait there, which then also waits for the H2 stores of the first tile
            auto l2_round = [&](auto last_c) {
                constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
                for (int u = 0; u < RD / 3; ++u) {
                    const uint4 wl = ring[3 * u + 0], wm = ring[3 * u + 1], wh = ring[3 * u + 2];
#pragma unroll
                    for (int i = 0; i < T; ++i) {
#if PPO_FX6_ZPIPE
                        load_z(i + 1 < T ? zp + (size_t)(i + 1) * NT * 6 * 1024 + u * 3 * 1024 : zp + (u + 1) * 3 * 1024, zn);
                        __builtin_amdgcn_sched_barrier(0);
                        const uint4 z_h = zc[0], z_m = zc[1], z_l = zc[2];
#else
                        const char* zi = zp + (size_t)i * NT * 6 * 1024;
                        const uint4 z_h = *reinterpret_cast<const uint4*>(zi + (u * 3 + 0) * 1024);
                        const uint4 z_m = *reinterpret_cast<const uint4*>(zi + (u * 3 + 1) * 1024);
                        const uint4 z_l = *reinterpret_cast<const uint4*>(zi + (u * 3 + 2) * 1024);
#endif
                        accs[i] = x_mfma(wl, z_h, accs[i]);
                        accs[i] = x_mfma(wm, z_m, accs[i]);
                        accs[i] = x_mfma(wm, z_h, accs[i]);
                        accs[i] = x_mfma(wh, z_l, accs[i]);
                        accs[i] = x_mfma(wh, z_m, accs[i]);
                        acc[i] = x_mfma(wh, z_h, acc[i]);
#if PPO_FX6_ZPIPE
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < 3; ++q) zc[q] = zn[q];
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (!LAST) {
                        ring[3 * u + 0] = *reinterpret_cast<const uint4*>(wn + lo16);          // (without PPO_FX6_NODANGLE the last round reads RD KiB ahead: padding / next wave's stream)
                        ring[3 * u + 1] = *reinterpret_cast<const uint4*>(wn + 1024 + lo16);
                        ring[3 * u + 2] = *reinterpret_cast<const uint4*>(wn + 2048 + lo16);
                        wn += 3 * 1024;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                zp += (RD / 3) * 3 * 1024;
            };
#pragma unroll 1
            for (int k0 = 0; k0 < KS - RD / 3; k0 += RD / 3) l2_round(std::false_type{});
            l2_round(std::true_type{});
#else
#pragma unroll 1
            for (int k0 = 0; k0 < KS; k0 += RD / 3) {
#pragma unroll
                for (int u = 0; u < RD / 3; ++u) {
                    const uint4 wl = ring[3 * u + 0], wm = ring[3 * u + 1], wh = ring[3 * u + 2];
#pragma unroll
                    for (int i = 0; i < T; ++i) {
#if PPO_FX6_ZPIPE
                        load_z(i + 1 < T ? zp + (size_t)(i + 1) * NT * 6 * 1024 + u * 3 * 1024 : zp + (u + 1) * 3 * 1024, zn);
                        __builtin_amdgcn_sched_barrier(0);
                        const uint4 z_h = zc[0], z_m = zc[1], z_l = zc[2];
#else
                        const char* zi = zp + (size_t)i * NT * 6 * 1024;
                        const uint4 z_h = *reinterpret_cast<const uint4*>(zi + (u * 3 + 0) * 1024);
                        const uint4 z_m = *reinterpret_cast<const uint4*>(zi + (u * 3 + 1) * 1024);
                        const uint4 z_l = *reinterpret_cast<const uint4*>(zi + (u * 3 + 2) * 1024);
#endif
                        accs[i] = x_mfma(wl, z_h, accs[i]);
                        accs[i] = x_mfma(wm, z_m, accs[i]);
                        accs[i] = x_mfma(wm, z_h, accs[i]);
                        accs[i] = x_mfma(wh, z_l, accs[i]);
                        accs[i] = x_mfma(wh, z_m, accs[i]);
                        acc[i] = x_mfma(wh, z_h, acc[i]);
#if PPO_FX6_ZPIPE
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < 3; ++q) zc[q] = zn[q];
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    ring[3 * u + 0] = *reinterpret_cast<const uint4*>(wn + lo16);          // the last round reads RD KiB ahead (padding / next wave's stream)
                    ring[3 * u + 1] = *reinterpret_cast<const uint4*>(wn + 1024 + lo16);
                    ring[3 * u + 2] = *reinterpret_cast<const uint4*>(wn + 2048 + lo16);
                    wn += 3 * 1024;
                    __builtin_amdgcn_sched_barrier(0);
                }
                zp += (RD / 3) * 3 * 1024;
            }
#endif
            FXSTAMP(4);
#pragma unroll
            for (int i = 0; i < T; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = acc[i][r] + accs[i][r];
                asm volatile("" : "+v"(acc[i]));
                lrelu16(acc[i]);
                const int64_t tile = g * T + i;
                if (tile < a.B) {
                    float4* dst = a.act2 + ((size_t)tile * NT + w) * 4 * 64;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        typedef float f32x4l __attribute__((ext_vector_type(4)));
                        const f32x4l t = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
                        __builtin_nontemporal_store(t, reinterpret_cast<f32x4l*>(dst + q * 64 + ln));
                    }
                }
                float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
                const float4* w3 = sW3p + (h * NT + w) * 16;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float4 wv = w3[r];
                    p0 = fmaf(wv.x, acc[i][r], p0); p1 = fmaf(wv.y, acc[i][r], p1);
                    p2 = fmaf(wv.z, acc[i][r], p2); p3 = fmaf(wv.w, acc[i][r], p3);
                }
                sP[(i * NT + w) * 64 + ln] = make_float4(p0, p1, p2, p3);
            }
        }
        FXSTAMP(5);
        __syncthreads();                                                // (2) partial logits in LDS; the H1 fragments are free
        FXSTAMP(6);
        if (w < T && g * T + w < a.B) {                                 // wave i runs the loss tail of tile i
            X6F_LANE();
            const int64_t tile = g * T + w;
#if PPO_FX6_SIDX
            const int32_t sidw = w == 0 ? cid[0] : cid[1];
            uint32_t act;
            {
                const uint32_t* const ap = a.active + sidw;
                asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(act) : "s"(ap) : "memory");
            }
#else
            const int32_t sidw = __builtin_amdgcn_readfirstlane(a.idx[tile]);
            const uint32_t act = a.active[sidw];
#endif
            const float4* sPi = sP + (size_t)w * NT * 64;
            float4 s = sPi[ln];
#pragma unroll
            for (int u = 1; u < NT; ++u) { const float4 q4 = sPi[u * 64 + ln]; s.x += q4.x; s.y += q4.y; s.z += q4.z; s.w += q4.w; }
            float l[1][4];
            l[0][0] = (s.x + __shfl_xor(s.x, 32)) + a.b3[0];
            l[0][1] = (s.y + __shfl_xor(s.y, 32)) + a.b3[1];
            l[0][2] = (s.z + __shfl_xor(s.z, 32)) + a.b3[2];
            l[0][3] = (s.w + __shfl_xor(s.w, 32)) + a.b3[3];
            policy_tail<2, 1, false>(a, tile, sidw, act, l, (int)ln, j, h);
        }
        FXSTAMP(7);
#if PPO_FX6_SIDX
        cid[0] = nid[0]; cid[1] = nid[1];
#endif
    }
#ifdef PPO_FX6_STAMP
    if (a.stamps && lane == 0 && (w == 0 || w == NT - 1))
        for (int i = 0; i < 8; ++i) a.stamps[((size_t)blockIdx.x * 2 + (w ? 1 : 0)) * 8 + i] = st_sum[i];
#endif
#undef FXSTAMP
}

template <int HID, int TPS>
struct FXSCfg {                                   // H1 fragments of the two tiles of a pass, partial logits of all TPS tiles
    static constexpr int F = 72, NT = HID / 32, KS = HID / 16, K1 = 5;
    static constexpr size_t oFr = 0, szFr = (size_t)2 * NT * 6 * 1024;
    static constexpr size_t oP = oFr + szFr, oW3 = oP + (size_t)TPS * NT * 1024, oB1 = oW3 + (size_t)2 * NT * 256,
                            oB2 = oB1 + (size_t)NT * 128, total = oB2 + (size_t)NT * 128;
    static_assert(total <= 160 * 1024, "LDS budget");
};

// ---------------------------------------------------------------- states of TPS tiles (H = 32 TPS half-edges: Q = 32 -> TPS = 4)
// One workgroup per STATE: its TPS tiles go through both layers two at a time (TPS / 2 passes over the weight stream), the
// layer-3 partial dots of all tiles wait in LDS, then wave 0 runs the state's loss tail over its 128 TPS actions.
template <int HID, int TPS>
__global__ __launch_bounds__(HID * 2, 2) void k_policy_fwd_train_x6s(FwdArgs a, const uint4* __restrict__ w1x, const uint4* __restrict__ w2fx, int x_by_tile) {
    constexpr int T = 2;
    static_assert(TPS % T == 0, "tiles of a state in pairs");
    using C = FXSCfg<HID, TPS>;
    constexpr int F = C::F, NT = C::NT, KS = C::KS, K1 = C::K1;
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* const frag = smem_c + C::oFr;                                    // [T][feature tile][k-step 2][piece 3][64 lanes][16 B]
    float4* const sP = reinterpret_cast<float4*>(smem_c + C::oP);          // [T][NT][64] layer-3 partial dots
    float4* const sW3p = reinterpret_cast<float4*>(smem_c + C::oW3);
    float4* const sB1 = reinterpret_cast<float4*>(smem_c + C::oB1);
    float4* const sB2 = reinterpret_cast<float4*>(smem_c + C::oB2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * NT * 16; i += NT * 64) sW3p[i] = a.w3p[i];
    for (int i = tid; i < NT * 8; i += NT * 64) { sB1[i] = a.b1p[i]; sB2[i] = a.b2p[i]; }
    __syncthreads();
    const char* const w1s = reinterpret_cast<const char*>(w1x + (size_t)w * K1 * 3 * 64);
    const char* const w2s = reinterpret_cast<const char*>(w2fx + (size_t)w * KS * 3 * 64);

    uint2 xr[T][K1];
    auto load_x = [&](int i, int64_t rec, unsigned ln) {
        const char* row = reinterpret_cast<const char*>(a.states) + (size_t)rec * 32 * F + (ln & 31u) * (unsigned)F + (ln >> 5) * 8u;
#pragma unroll
        for (int s = 0; s < K1; ++s) {
            const bool pad = (s == K1 - 1) && (ln >> 5);
            const uint2 v = *reinterpret_cast<const uint2*>(row + (pad ? 0 : 16 * s));
            xr[i][s] = pad ? make_uint2(0u, 0u) : v;
        }
    };
    // pass q = state * (TPS / T) + p covers tiles ts = p*T + i of its state; the record of tile (state, ts) is state*TPS + ts
    constexpr int NP = TPS / T;
    auto rec_of = [&](int64_t state, int ts) {
        return (x_by_tile ? state : (int64_t)__builtin_amdgcn_readfirstlane(a.idx[state])) * TPS + ts;
    };
    if ((int64_t)blockIdx.x < a.B) {
#pragma unroll
        for (int i = 0; i < T; ++i) load_x(i, rec_of(blockIdx.x, i), (unsigned)lane);
    }
    for (int64_t st8 = blockIdx.x; st8 < a.B; st8 += gridDim.x) {
    for (int p = 0; p < NP; ++p) {
        const int64_t g = st8 * NP + p;                                 // global pass index: tiles g*T + i
        // ================= layer 1: H1 tile w of the T states
        {
            X6F_LANE();
            unsigned lo16 = ln * 16u;
            constexpr int R1 = 8;
            uint4 ring[R1];
#pragma unroll
            for (int q = 0; q < R1; ++q) ring[q] = *reinterpret_cast<const uint4*>(w1s + (lo16 + (unsigned)q * 1024u));
            f32x16 acc[T], accs[T];
            uint4 xb[T][K1];
#pragma unroll
            for (int i = 0; i < T; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = sB1[(w * 2 + h) * 4 + q];
                    acc[i][4 * q + 0] = b.x; acc[i][4 * q + 1] = b.y; acc[i][4 * q + 2] = b.z; acc[i][4 * q + 3] = b.w;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[i][r] = 0.0f;
#pragma unroll
                for (int s = 0; s < K1; ++s) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (float)(int)(int8_t)((e < 4 ? xr[i][s].x : xr[i][s].y) >> (8 * (e & 3)));
                    xb[i][s] = make_uint4(x_perm(v[0], v[1]), x_perm(v[2], v[3]), x_perm(v[4], v[5]), x_perm(v[6], v[7]));
                }
            }
#pragma unroll
            for (int st = 0; st < 3 * K1; ++st) {
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    if (st % 3 == 2) acc[i] = x_mfma(ring[st % R1], xb[i][st / 3], acc[i]);
                    else accs[i] = x_mfma(ring[st % R1], xb[i][st / 3], accs[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (st + R1 < 3 * K1) ring[st % R1] = *reinterpret_cast<const uint4*>(w1s + (lo16 + (unsigned)(st + R1) * 1024u));
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < T; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = acc[i][r] + accs[i][r];
                asm volatile("" : "+v"(acc[i]));
                lrelu16(acc[i]);
                const int64_t tile = g * T + i;
                {
                    float4* dst = a.act1 + ((size_t)tile * NT + w) * 4 * 64;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        typedef float f32x4l __attribute__((ext_vector_type(4)));
                        const f32x4l t = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
                        __builtin_nontemporal_store(t, reinterpret_cast<f32x4l*>(dst + q * 64 + ln));
                    }
                }
                char* const fown = frag + ((size_t)i * NT + w) * 6 * 1024;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    uint2 zh[2], zm[2], zl[2];
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int q = 2 * s + qq;
                        const float hv[4] = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
                        x_split4(hv, zh[qq], zm[qq], zl[qq]);
                    }
                    *reinterpret_cast<uint4*>(fown + (s * 3 + 0) * 1024 + ln * 16) = make_uint4(zh[0].x, zh[0].y, zh[1].x, zh[1].y);
                    *reinterpret_cast<uint4*>(fown + (s * 3 + 1) * 1024 + ln * 16) = make_uint4(zm[0].x, zm[0].y, zm[1].x, zm[1].y);
                    *reinterpret_cast<uint4*>(fown + (s * 3 + 2) * 1024 + ln * 16) = make_uint4(zl[0].x, zl[0].y, zl[1].x, zl[1].y);
                }
            }
        }
        constexpr int RD = 6;
        static_assert(KS % (RD / 3) == 0, "ring rounds");
        uint4 ring[RD];
        {
            unsigned lo = (unsigned)lane * 16u;
            asm volatile("" : "+v"(lo));
#pragma unroll
            for (int q = 0; q < RD; ++q) ring[q] = *reinterpret_cast<const uint4*>(w2s + (lo + (unsigned)q * 1024u));
        }
        {
            // the next pass: the state's next pair of tiles, or the first pair of this workgroup's next state (the last pass re-loads itself)
            const bool more = p + 1 < NP;
            const int64_t sn = more ? st8 : (st8 + gridDim.x < a.B ? st8 + gridDim.x : st8);
            const int pn = more ? p + 1 : (st8 + gridDim.x < a.B ? 0 : p);
            unsigned ln2 = (unsigned)lane;
            asm volatile("" : "+v"(ln2));
#pragma unroll
            for (int i = 0; i < T; ++i) load_x(i, rec_of(sn, pn * T + i), ln2);
        }
        __syncthreads();                                                // (1) every layer-1 tile of the T states is in LDS
        // ================= layer 2
        {
            X6F_LANE();
            f32x16 acc[T], accs[T];
#pragma unroll
            for (int i = 0; i < T; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = sB2[(w * 2 + h) * 4 + q];
                    acc[i][4 * q + 0] = b.x; acc[i][4 * q + 1] = b.y; acc[i][4 * q + 2] = b.z; acc[i][4 * q + 3] = b.w;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[i][r] = 0.0f;
            }
            const unsigned lo16 = ln * 16u;
            const char* zp = frag + lo16;
            const char* wn = w2s + (size_t)RD * 1024;
#pragma unroll 1
            for (int k0 = 0; k0 < KS; k0 += RD / 3) {
#pragma unroll
                for (int u = 0; u < RD / 3; ++u) {
                    const uint4 wl = ring[3 * u + 0], wm = ring[3 * u + 1], wh = ring[3 * u + 2];
#pragma unroll
                    for (int i = 0; i < T; ++i) {
                        const char* zi = zp + (size_t)i * NT * 6 * 1024;
                        const uint4 z_h = *reinterpret_cast<const uint4*>(zi + (u * 3 + 0) * 1024);
                        const uint4 z_m = *reinterpret_cast<const uint4*>(zi + (u * 3 + 1) * 1024);
                        const uint4 z_l = *reinterpret_cast<const uint4*>(zi + (u * 3 + 2) * 1024);
                        accs[i] = x_mfma(wl, z_h, accs[i]);
                        accs[i] = x_mfma(wm, z_m, accs[i]);
                        accs[i] = x_mfma(wm, z_h, accs[i]);
                        accs[i] = x_mfma(wh, z_l, accs[i]);
                        accs[i] = x_mfma(wh, z_m, accs[i]);
                        acc[i] = x_mfma(wh, z_h, acc[i]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    ring[3 * u + 0] = *reinterpret_cast<const uint4*>(wn + lo16);          // the last round reads RD KiB ahead (padding / next wave's stream)
                    ring[3 * u + 1] = *reinterpret_cast<const uint4*>(wn + 1024 + lo16);
                    ring[3 * u + 2] = *reinterpret_cast<const uint4*>(wn + 2048 + lo16);
                    wn += 3 * 1024;
                    __builtin_amdgcn_sched_barrier(0);
                }
                zp += (RD / 3) * 3 * 1024;
            }
#pragma unroll
            for (int i = 0; i < T; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = acc[i][r] + accs[i][r];
                asm volatile("" : "+v"(acc[i]));
                lrelu16(acc[i]);
                const int64_t tile = g * T + i;
                {
                    float4* dst = a.act2 + ((size_t)tile * NT + w) * 4 * 64;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        typedef float f32x4l __attribute__((ext_vector_type(4)));
                        const f32x4l t = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
                        __builtin_nontemporal_store(t, reinterpret_cast<f32x4l*>(dst + q * 64 + ln));
                    }
                }
                float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
                const float4* w3 = sW3p + (h * NT + w) * 16;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float4 wv = w3[r];
                    p0 = fmaf(wv.x, acc[i][r], p0); p1 = fmaf(wv.y, acc[i][r], p1);
                    p2 = fmaf(wv.z, acc[i][r], p2); p3 = fmaf(wv.w, acc[i][r], p3);
                }
                sP[((p * T + i) * NT + w) * 64 + ln] = make_float4(p0, p1, p2, p3);
            }
        }
        __syncthreads();                                                // (2) partial logits in LDS; the H1 fragments are free
        if (p == NP - 1 && w == 0) {                                    // all tiles of the state are through: its loss tail
            X6F_LANE();
            const int32_t sidw = __builtin_amdgcn_readfirstlane(a.idx[st8]);
            const uint32_t act = a.active[sidw];
            float l[TPS][4];
#pragma unroll
            for (int ts = 0; ts < TPS; ++ts) {
                const float4* sPi = sP + (size_t)ts * NT * 64;
                float4 s = sPi[ln];
#pragma unroll
                for (int u = 1; u < NT; ++u) { const float4 q4 = sPi[u * 64 + ln]; s.x += q4.x; s.y += q4.y; s.z += q4.z; s.w += q4.w; }
                l[ts][0] = (s.x + __shfl_xor(s.x, 32)) + a.b3[0];
                l[ts][1] = (s.y + __shfl_xor(s.y, 32)) + a.b3[1];
                l[ts][2] = (s.z + __shfl_xor(s.z, 32)) + a.b3[2];
                l[ts][3] = (s.w + __shfl_xor(s.w, 32)) + a.b3[3];
            }
            policy_tail<2, TPS, false>(a, st8, sidw, act, l, (int)ln, j, h);
        }
    }
    }
}

// minibatches of up to this many 32-row tiles take the split-fp32 train forward (when ppo_set_bwd_split_bf16 is on);
// PPO_FWD_SPLIT_MAX_TILES overrides (0 = never)
#ifndef PPO_FWD_X6_DEFAULT_MAX_TILES
#define PPO_FWD_X6_DEFAULT_MAX_TILES (1 << 30)
#endif
// HID = 256 minibatches of at least this many tiles take the two-tiles-per-pass form (PPO_FWD_SPLIT_T2_MIN_TILES; 0 = never)
#ifndef PPO_FWD_X6_T2_DEFAULT_MIN_TILES
#define PPO_FWD_X6_T2_DEFAULT_MIN_TILES 1536
#endif
// (read at every launch: the tests move the switch point to sizes their float64 checker finishes in seconds)
static int64_t fwd_x6_t2_min_tiles() { const char* v = std::getenv("PPO_FWD_SPLIT_T2_MIN_TILES"); return v ? (int64_t)atoll(v) : (int64_t)PPO_FWD_X6_T2_DEFAULT_MIN_TILES; }
// the same switch for HID = 128 (PPO_FWD_SPLIT_T2_MIN_TILES_128; 0 = never).  Measured (gpurun_out/h128_t2, alternating): train forward
// 0.0570 -> 0.0548 ms at 4096 states, 6.00 -> 6.09 M env-steps/s
static int64_t fwd_x6_t2_min_tiles_128() { const char* v = std::getenv("PPO_FWD_SPLIT_T2_MIN_TILES_128"); return v ? (int64_t)atoll(v) : (int64_t)1024; }
static int64_t g_fwd_x6_max_tiles = [] { const char* v = std::getenv("PPO_FWD_SPLIT_MAX_TILES"); return v ? (int64_t)atoll(v) : (int64_t)PPO_FWD_X6_DEFAULT_MAX_TILES; }();

#ifdef PPO_FX6_STAMP
// diagnostic build (make -C csrc fxstamp, tools/fx6_stamps.py): per-phase clocks of k_policy_fwd_train_x6t, [workgroup][wave 0 / last][8]
static unsigned long long* g_fx6_stamps = nullptr;
extern "C" int32_t ppo_debug_fx6_stamps(unsigned long long* out) {
    if (!g_fx6_stamps) return -1;
    (void)hipDeviceSynchronize();
    return hipMemcpy(out, g_fx6_stamps, 512 * 2 * 8 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

int32_t launch_policy_train_fwd_x6(ppo_policy_s* p, FwdArgs& a, int64_t B, int tps, bool compact) {
    if (!ppo_bwd_split_enabled() || B > g_fwd_x6_max_tiles) return PPO_ERR_UNSUPPORTED;
#ifdef PPO_FX6_STAMP
    if (!g_fx6_stamps) { (void)hipMalloc((void**)&g_fx6_stamps, 512 * 2 * 8 * 8); (void)hipMemset(g_fx6_stamps, 0, 512 * 2 * 8 * 8); }
    a.stamps = g_fx6_stamps;
#endif
    if (p->dtype != PPO_DTYPE_F32 || p->L != 2 || p->F != 72 || !(tps == 1 || (tps == 4 && p->HID == 256)) || !p->w1x.p || !p->w2fx.p) return PPO_ERR_UNSUPPORTED;
    if (compact) {
        // env snapshots: the minibatch's observation rows are re-derived first (the arithmetic of state(env), ppo_env.hip) into
        // the scratch the backward reads in this storage form anyway; both kernels then see the rows the expanded form holds,
        // so the storage form does not change a bit of the result
        PPO_TRY(launch_expand_states(a.cstate, a.active, a.env_tmpl, B, a.envQ, a.xs_out, a.idx));
        a.states = a.xs_out;
    }
#define LAUNCH(HH)                                                                                           \
    do {                                                                                                     \
        const int64_t cap = 256 * FXCfg<HH>::WG_PER_CU;                                                      \
        const int nwg = (int)(B < cap ? B : cap);                                                            \
        const size_t lds = FXCfg<HH>::total;                                                                 \
        static thread_local bool attr_set = false;                                                           \
        if (!attr_set) {                                                                                     \
            HIP_TRY(hipFuncSetAttribute((const void*)k_policy_fwd_train_x6<HH>,                              \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));              \
            attr_set = true;                                                                                 \
        }                                                                                                    \
        hipLaunchKernelGGL((k_policy_fwd_train_x6<HH>), dim3(nwg), dim3(HH * 2), lds, ppo_stream(), a,       \
                           (const uint4*)p->w1x.p, (const uint4*)p->w2fx.p, compact ? 1 : 0);                                 \
    } while (0)
    if (tps == 4) {                                               // Q = 32 states: one workgroup per state, its four tiles in two passes
        const int nwg = (int)(B < 256 ? B : 256);
        const size_t lds = FXSCfg<256, 4>::total;
        static thread_local bool attr_set4 = false;
        if (!attr_set4) {
            HIP_TRY(hipFuncSetAttribute((const void*)k_policy_fwd_train_x6s<256, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set4 = true;
        }
        hipLaunchKernelGGL((k_policy_fwd_train_x6s<256, 4>), dim3(nwg), dim3(512), lds, ppo_stream(), a,
                           (const uint4*)p->w1x.p, (const uint4*)p->w2fx.p, compact ? 1 : 0);
    }
    else if (p->HID == 256 && fwd_x6_t2_min_tiles() > 0 && B >= fwd_x6_t2_min_tiles()) {
        constexpr int T = 2;
        const int64_t groups = (B + T - 1) / T;
        const int nwg = (int)(groups < 256 ? groups : 256);
        const size_t lds = FXTCfg<256, T>::total;
        static thread_local bool attr_set2 = false;
        if (!attr_set2) {
            HIP_TRY(hipFuncSetAttribute((const void*)k_policy_fwd_train_x6t<256, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set2 = true;
        }
        hipLaunchKernelGGL((k_policy_fwd_train_x6t<256, T>), dim3(nwg), dim3(512), lds, ppo_stream(), a,
                           (const uint4*)p->w1x.p, (const uint4*)p->w2fx.p, compact ? 1 : 0);
    }
    else if (p->HID == 128 && fwd_x6_t2_min_tiles_128() > 0 && B >= fwd_x6_t2_min_tiles_128()) {
        constexpr int T = 2;                                       // 54 KB of LDS per workgroup: two (four-wave) workgroups per CU
        const int64_t groups = (B + T - 1) / T;
        const int nwg = (int)(groups < 512 ? groups : 512);
        const size_t lds = FXTCfg<128, T>::total;
        static thread_local bool attr_set3 = false;
        if (!attr_set3) {
            HIP_TRY(hipFuncSetAttribute((const void*)k_policy_fwd_train_x6t<128, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set3 = true;
        }
        hipLaunchKernelGGL((k_policy_fwd_train_x6t<128, T>), dim3(nwg), dim3(256), lds, ppo_stream(), a,
                           (const uint4*)p->w1x.p, (const uint4*)p->w2fx.p, compact ? 1 : 0);
    }
    else if (p->HID == 256) LAUNCH(256);
    else if (p->HID == 128) LAUNCH(128);
    else return PPO_ERR_UNSUPPORTED;
#undef LAUNCH
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}
