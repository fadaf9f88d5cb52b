// ppo_policy_train_tile.hip -- step_batch! (src/train.jl:54-84) for SMALL minibatches (the per-GPU shard of a strong-
// scaling run, the reference's own batch sizes): train forward, loss and the whole backward-data pass of a 32-row tile
// in ONE workgroup, nothing but the operands of the weight-gradient products leaves the CU.
//
// Why: below ~1000 tiles an optimiser step is a chain of launches that are each mostly ramp-up, drain and first-touch
// latency (train forward, backward, slab reduction, all-reduce, Adam), and the fused backward's per-workgroup gradient
// slabs (87 MB written + 87 MB re-read at any batch size) are a fixed 30 us of it.  Here
//   k_policy_train_tile  workgroup = HID/32 waves = one tile (state); wave w owns feature tile w of every layer:
//       layer 1 tile w (MFMA, Y^T = W X^T like k_policy_fwd)            -> LDS in fragment order (all waves need all tiles)
//       layer 2 tile w from all layer-1 tiles (B operands streamed from LDS), layer-3 partial dots -> LDS
//       wave 0: logits, masked softmax, ppo_loss_with_entropy terms, dL/dlogits (policy_tail, shared with every forward)
//       dZ2 tile w = (W3^T dY) . lrelu'(H2)  [H2 never left the wave's registers]   -> LDS (B operand of the next product)
//       dH1 tile w = W2^T dZ2 (MFMA, W2^T fragments streamed from L2), dZ1 = dH1 . lrelu'(H1)  [H1: registers]
//       bias / W3 gradients on the VALU (fixed order); dZ2^T, dZ1^T and H1^T of the tile leave in the OPERAND LAYOUT of
//       the row contraction (lane = feature, four consecutive rows per 16-byte element), 1 KiB coalesced stores
//   k_policy_wgrad<.., TR = 1> (ppo_policy_bwd_small.hip)  dW2 += dZ2 H1^T, dW1 += dZ1 X^T as an output-stationary split-K
//       product whose A / B operands are plain 1 KiB wave loads of those tiles: no transposes in the MFMA-critical kernel
//   k_grad_reduce unchanged (fixed-order sum of the K-slice partials: bitwise reproducible gradients).
// Numerics: the same operations as k_policy_fwd_train_split + k_policy_bwd_data; the layer-3 partial sums are added in
// wave order (like the split forward), so logits agree with the one-wave kernel to fp32 rounding -- train path only.
#include "ppo_policy_tail.h"

#define TT_LD 36        // leading dimension (rows) of the transposed LDS tiles, as in k_policy_bwd

struct TrainTileArgs {
    FwdArgs f;                         // forward + loss inputs (states, idx, weights, actions, p_old, adv, eps ...), dY / loss_terms out
    const float4* w2tp;                // packed W2^T (backward A operand)
    float4* h1t; float4* dz2t; float4* dz1t;   // [tile][feature tile][4][64] float4, operand layout of k_policy_wgrad<TR = 1>
    float* slabs; size_t slab_stride;  // small-gradient tails (db1, db2, dW3, db3), one per workgroup
};

template <int F, int HID>
__global__ __launch_bounds__(HID * 2, 2) void k_policy_train_tile(TrainTileArgs ta) {
    const FwdArgs& a = ta.f;
    constexpr int NT = HID / 32, LD = TT_LD, S41 = F / 8, S42 = NT * 4, S4 = HID / 8, XB = F / 2, XW = XB / 4;
    constexpr int FP = ((F + 31) / 32) * 32;
    constexpr bool Z2R = PPO_BWD_Z2ROW_AT(HID);          // dZ2 row-major [32][RS] (must match the W2^T packing, ppo_internal.h)
    constexpr int RS = HID + 4, Z2SZ = Z2R ? 32 * RS : HID * LD;
    static_assert(F % 8 == 0 && S4 % 8 == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* const sHf = reinterpret_cast<float4*>(smem);      // [NT][4][64] float4: layer-1 tiles, fragment order
    float* const sZ2 = smem + NT * 1024;                      // dZ2 in the B-operand form of dH1 = W2^T dZ2
    float* const sH2 = sZ2 + Z2SZ;                            // [HID][LD] H2^T (dW3 sums), then H1^T (own rows, emit)
    float* const sZ1 = sH2 + HID * LD;                        // [HID][LD] dZ1^T; its head doubles as the layer-3 partials
    float4* const sP = reinterpret_cast<float4*>(sZ1);        // [NT][64] float4 (dead before dZ1 is written)
    float* const sDY = sZ1 + HID * LD;                        // [32][4]
    float* const sW3 = sDY + 32 * 4;                          // [HID][4]  W3[:,f] per feature
    float4* const sW3p = reinterpret_cast<float4*>(sW3 + HID * 4);   // [2][NT][16] float4: the forward's layer-3 pack
    float4* const sB1 = sW3p + 2 * NT * 16;                   // [NT][2][4] float4
    float4* const sB2 = sB1 + NT * 8;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < HID) {                                          // w3p is [h][tile][r][4]: un-permute to [f][4]
        const int kk = tid & 31, hh = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3);
        *reinterpret_cast<float4*>(&sW3[tid * 4]) = a.w3p[(size_t)(hh * NT + (tid >> 5)) * 16 + r];
    }
    for (int i = tid; i < 2 * NT * 16; i += NT * 64) sW3p[i] = a.w3p[i];
    for (int i = tid; i < NT * 8; i += NT * 64) { sB1[i] = a.b1p[i]; sB2[i] = a.b2p[i]; }
    __syncthreads();
    float db1 = 0.f, db2 = 0.f, db3 = 0.f, dw3[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned fb = (unsigned)(32 * w + 4 * h);
    const float4* const w1t = a.w1p + (size_t)w * S41 * 64;              // this wave's tile of each packed stream
    const float4* const w2f = a.w2p + (size_t)w * S42 * 64;
    const float4* const w2t = ta.w2tp + (size_t)w * S4 * 64;

    for (int64_t tile = blockIdx.x; tile < a.B; tile += gridDim.x) {
        int lane_o = lane, half_o = h;
        asm volatile("" : "+v"(lane_o), "+v"(half_o));                   // per-tile opaque offsets (see k_policy_fwd)
        unsigned lb = fb * LD + j;
        asm volatile("" : "+v"(lb));
        const int32_t sid = a.idx[tile];
        const uint32_t act = a.active[sid];
        // ================= forward, layer 1: H1 tile w
        f32x16 h1o;
        {
            float xf[XB];
            const uint32_t* xr = reinterpret_cast<const uint32_t*>(a.states + (size_t)sid * 32 * F + (size_t)j * F + (size_t)h * XB);
            uint32_t xw[XW];
#pragma unroll
            for (int k = 0; k < XW; ++k) xw[k] = xr[k];
            float4 wr[S41];
#pragma unroll
            for (int g = 0; g < S41; ++g) wr[g] = w1t[(size_t)g * 64 + lane_o];
#pragma unroll
            for (int k = 0; k < XW; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[4 * k + i] = (float)(int)(int8_t)(xw[k] >> (8 * i));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = sB1[(w * 2 + half_o) * 4 + q];
                h1o[4 * q + 0] = b.x; h1o[4 * q + 1] = b.y; h1o[4 * q + 2] = b.z; h1o[4 * q + 3] = b.w;
            }
#pragma unroll
            for (int s4 = 0; s4 < S41; ++s4) {
                h1o = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[s4].x, xf[4 * s4 + 0], h1o, 0, 0, 0);
                h1o = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[s4].y, xf[4 * s4 + 1], h1o, 0, 0, 0);
                h1o = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[s4].z, xf[4 * s4 + 2], h1o, 0, 0, 0);
                h1o = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[s4].w, xf[4 * s4 + 3], h1o, 0, 0, 0);
            }
            asm volatile("" : "+v"(h1o));
            lrelu16(h1o);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                sHf[(w * 4 + q) * 64 + lane] = make_float4(h1o[4 * q], h1o[4 * q + 1], h1o[4 * q + 2], h1o[4 * q + 3]);
        }
        // first fragment groups of this wave's W2 tile: in flight across the barrier
        constexpr int PF = 8;
        static_assert(S42 % PF == 0 && PF * 64 * 4 <= PPO_PACK_PAD, "ring");
        float4 ring[PF];
#pragma unroll
        for (int g = 0; g < PF; ++g) ring[g] = w2f[(size_t)g * 64 + lane_o];
        __syncthreads();                                                // (1) every layer-1 tile is in LDS
        // ================= forward, layer 2: H2 tile w from all layer-1 tiles; layer-3 partial dots
        f32x16 h2o;
        {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = sB2[(w * 2 + half_o) * 4 + q];
                h2o[4 * q + 0] = b.x; h2o[4 * q + 1] = b.y; h2o[4 * q + 2] = b.z; h2o[4 * q + 3] = b.w;
            }
            float4 hb[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) hb[0][q] = sHf[(0 * 4 + q) * 64 + lane];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t + 1 < NT) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) hb[(t + 1) & 1][q] = sHf[((t + 1) * 4 + q) * 64 + lane];
                }
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int s4 = t * 4 + r4;
                    const float4 wv = ring[s4 % PF];
                    ring[s4 % PF] = w2f[(size_t)(s4 + PF) * 64 + lane_o];        // next tile's head / tail padding covers the over-read
                    const float4 b4 = hb[t & 1][r4];
                    h2o = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, b4.x, h2o, 0, 0, 0);
                    h2o = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, b4.y, h2o, 0, 0, 0);
                    h2o = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, b4.z, h2o, 0, 0, 0);
                    h2o = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, b4.w, h2o, 0, 0, 0);
                }
            }
            asm volatile("" : "+v"(h2o));
            lrelu16(h2o);
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
            const float4* w3 = sW3p + (half_o * NT + w) * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 wv = w3[r];
                p0 = fmaf(wv.x, h2o[r], p0); p1 = fmaf(wv.y, h2o[r], p1);
                p2 = fmaf(wv.z, h2o[r], p2); p3 = fmaf(wv.w, h2o[r], p3);
            }
            sP[w * 64 + lane] = make_float4(p0, p1, p2, p3);
        }
        // first fragment groups of this wave's W2^T tile: in flight across the loss tail
        constexpr int PB = 4;
        static_assert(S4 % (2 * PB) == 0, "ring sets");
        float4 ringA[PB], ringB[PB];
#pragma unroll
        for (int g = 0; g < PB; ++g) ringA[g] = w2t[(size_t)g * 64 + lane_o];
        __syncthreads();                                                // (2) every wave's partial logits are in LDS
        if (w == 0) {
            // partial logits in wave (= tile) order, then exactly k_policy_fwd's epilogue; dL/dlogits also into sDY
            float4 s = sP[lane];
#pragma unroll
            for (int u = 1; u < NT; ++u) { const float4 q4 = sP[u * 64 + lane]; s.x += q4.x; s.y += q4.y; s.z += q4.z; s.w += q4.w; }
            float l[1][4];
            l[0][0] = (s.x + __shfl_xor(s.x, 32)) + a.b3[0];
            l[0][1] = (s.y + __shfl_xor(s.y, 32)) + a.b3[1];
            l[0][2] = (s.z + __shfl_xor(s.z, 32)) + a.b3[2];
            l[0][3] = (s.w + __shfl_xor(s.w, 32)) + a.b3[3];
            policy_tail<2, 1, false>(a, tile, sid, act, l, lane, j, h, 0u, 0, nullptr, reinterpret_cast<float4*>(sDY));
        }
        __syncthreads();                                                // (3) dY of the tile is in LDS
        // ================= backward, phase A: dZ2 tile w = (W3^T dY) . lrelu'(H2)
        {
            const float4 dy = *reinterpret_cast<const float4*>(&sDY[j * 4]);
            f32x16 dh;
#pragma unroll
            for (int r = 0; r < 16; ++r) dh[r] = 0.0f;
            dh = __builtin_amdgcn_mfma_f32_32x32x2f32(sW3[(32 * w + j) * 4 + h], h ? dy.y : dy.x, dh, 0, 0, 0);     // register 4q + e <-> feature e + 8q (+ 4h)
            dh = __builtin_amdgcn_mfma_f32_32x32x2f32(sW3[(32 * w + j) * 4 + 2 + h], h ? dy.w : dy.z, dh, 0, 0, 0);
            float* const z2b = sZ2 + lb;
            float* const h2b = sH2 + lb;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float z[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int fo = e + 8 * q;
                    const float hv = h2o[4 * q + e];
                    z[e] = dh[4 * q + e] * (hv > 0.0f ? 1.0f : 0.01f);
                    if constexpr (!Z2R) z2b[fo * LD] = z[e];
                    h2b[fo * LD] = hv;
                }
                if constexpr (Z2R) *reinterpret_cast<float4*>(sZ2 + j * RS + fb + 8 * q) = make_float4(z[0], z[1], z[2], z[3]);
            }
        }
        __syncthreads();                                                // (4) dZ2 of every feature tile is in LDS
        // ================= phase B: small gradients of feature 32w + j (rows 16h .. 16h+15), dH1 tile w, dZ1
        {
            const float* gz = Z2R ? sZ2 + (16 * h) * RS + 32 * w + j : sZ2 + (32 * w + j) * LD + 16 * h;
            const float* gh = sH2 + (32 * w + j) * LD + 16 * h;
            const float* gy = sDY + 64 * h;
            float s2 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
            float4 zt[4];                                               // the same 16 rows, kept for the operand-layout store
#pragma unroll
            for (int rc = 0; rc < 16; rc += 4) {
                float4 z4;
                if constexpr (Z2R) z4 = make_float4(gz[rc * RS], gz[(rc + 1) * RS], gz[(rc + 2) * RS], gz[(rc + 3) * RS]);
                else z4 = *reinterpret_cast<const float4*>(gz + rc);
                zt[rc >> 2] = z4;
                const float4 h4 = *reinterpret_cast<const float4*>(gh + rc);
                const float z[4] = {z4.x, z4.y, z4.z, z4.w}, hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 y = *reinterpret_cast<const float4*>(gy + (rc + i) * 4);
                    s2 += z[i]; d0 = fmaf(y.x, hv[i], d0); d1 = fmaf(y.y, hv[i], d1); d2 = fmaf(y.z, hv[i], d2); d3 = fmaf(y.w, hv[i], d3);
                }
            }
            db2 += s2; dw3[0] += d0; dw3[1] += d1; dw3[2] += d2; dw3[3] += d3;
            if (tid < 4) {
                float s = 0.f;
                for (int r = 0; r < 32; ++r) s += sDY[r * 4 + tid];
                db3 += s;
            }
            // dZ2^T of (feature 32w + j, rows 16h + 4q .. + 3): the A operand of dW2 += dZ2 H1^T, as it will be loaded
            float4* const zo = ta.dz2t + ((size_t)tile * NT + w) * 4 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) zo[q * 64] = zt[q];
        }
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            // B operand: Z2R: dZ2[row j][f = 8g + 4h + e], one 16-byte read per fragment group; else dZ2^T[f = 8g + 2e + h][row j]
            const float* bz = Z2R ? sZ2 + j * RS + 4 * h : sZ2 + h * LD + j;
            auto mfma_set = [&](const float4 (&rg)[PB]) {
#pragma unroll
                for (int u = 0; u < PB; ++u) {
                    float b[4];
                    if constexpr (Z2R) {
                        const float4 b4 = *reinterpret_cast<const float4*>(bz + 8 * u);
                        b[0] = b4.x; b[1] = b4.y; b[2] = b4.z; b[3] = b4.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) b[e] = bz[(8 * u + 2 * e) * LD];
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rg[u].x, b[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rg[u].y, b[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rg[u].z, b[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rg[u].w, b[3], acc, 0, 0, 0);
                }
                bz += Z2R ? 8 * PB : 8 * PB * LD;
            };
            const float4* wn = w2t + (size_t)PB * 64 + lane_o;
#pragma unroll 1
            for (int s0 = 0; s0 < S4; s0 += 2 * PB) {
#pragma unroll
                for (int u = 0; u < PB; ++u) ringB[u] = wn[(size_t)u * 64];
                wn += (size_t)PB * 64;
                __builtin_amdgcn_sched_barrier(0);
                mfma_set(ringA);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < PB; ++u) ringA[u] = wn[(size_t)u * 64];      // tail padding covers the over-read
                wn += (size_t)PB * 64;
                __builtin_amdgcn_sched_barrier(0);
                mfma_set(ringB);
                __builtin_amdgcn_sched_barrier(0);
            }
            // dZ1 = dH1 . lrelu'(H1) -> own rows of sZ1; H1 -> own rows of sH2 (this wave's small-gradient reads of them are done)
            float* const z1b = sZ1 + lb;
            float* const h1b = sH2 + lb;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int fo = (r & 3) + 8 * (r >> 2);
                const float hv = h1o[r];
                z1b[fo * LD] = acc[r] * (hv > 0.0f ? 1.0f : 0.01f);
                h1b[fo * LD] = hv;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // wave-private rows: program order + in-order LDS suffice
        {   // db1 and the operand-layout stores of dZ1^T / H1^T (feature 32w + j, rows 16h + 4q .. + 3)
            const float* g1 = sZ1 + (32 * w + j) * LD + 16 * h;
            const float* gh = sH2 + (32 * w + j) * LD + 16 * h;
            float4* const z1o = ta.dz1t + ((size_t)tile * NT + w) * 4 * 64 + lane;
            float4* const h1o_g = ta.h1t + ((size_t)tile * NT + w) * 4 * 64 + lane;
            float s1 = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 z4 = *reinterpret_cast<const float4*>(g1 + 4 * q);
                const float4 h4 = *reinterpret_cast<const float4*>(gh + 4 * q);
                s1 += z4.x; s1 += z4.y; s1 += z4.z; s1 += z4.w;
                z1o[q * 64] = z4;
                h1o_g[q * 64] = h4;
            }
            db1 += s1;
        }
        // no barrier here: the next tile's layer 1 writes sHf only (last read before barrier (2)); sP / sZ1, sZ2, sH2 and sDY
        // are rewritten behind the next tile's barriers (1)-(3), which every wave reaches only after finishing this tile
    }
    // small-gradient tail of slab blockIdx.x (same places as k_policy_bwd's slab)
    float* slab = ta.slabs + (size_t)blockIdx.x * ta.slab_stride + (size_t)HID * HID + (size_t)HID * FP;
    float* sb1 = slab; float* sb2 = sb1 + HID; float* sw3 = sb2 + HID; float* sb3 = sw3 + HID * 4;
    db1 += __shfl_xor(db1, 32); db2 += __shfl_xor(db2, 32);
#pragma unroll
    for (int i = 0; i < 4; ++i) dw3[i] += __shfl_xor(dw3[i], 32);
    if (h == 0) {
        const int f = 32 * w + j;
        sb1[f] = db1; sb2[f] = db2;
        *reinterpret_cast<float4*>(&sw3[f * 4]) = make_float4(dw3[0], dw3[1], dw3[2], dw3[3]);
    }
    if (tid < 4) sb3[tid] = db3;
}

// weight-gradient half of the pass (ppo_policy_bwd_small.hip)
int32_t launch_policy_wgrad_tr(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B, int nwg_tail);

template <int F, int HID>
static int32_t launch_tt(ppo_policy_s* p, TrainTileArgs& ta, int64_t B, int* nwg_out) {
    constexpr int NT = HID / 32;
    const size_t z2 = PPO_BWD_Z2ROW_AT(HID) ? (size_t)32 * (HID + 4) : (size_t)HID * TT_LD;
    const size_t lds = sizeof(float) * ((size_t)NT * 1024 + z2 + (size_t)2 * HID * TT_LD + 32 * 4 + (size_t)HID * 4 + (size_t)2 * NT * 16 * 4 + (size_t)2 * NT * 8 * 4);
    static thread_local bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_policy_train_tile<F, HID>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int64_t cap = (lds <= 80 * 1024) ? 512 : 256;                 // two workgroups per CU where the LDS allows
    const int nwg = (int)(B < cap ? B : cap);
    *nwg_out = nwg;
    ProfScope ps("k_policy_train_tile");
    hipLaunchKernelGGL((k_policy_train_tile<F, HID>), dim3(nwg), dim3(HID * 2), lds, ppo_stream(), ta);
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

// PPO_ERR_UNSUPPORTED (no error text): shape not covered -> the caller runs the separate forward / backward kernels.
// On success the flat-gradient inputs of k_grad_reduce (slabs, loss terms) are complete.
int32_t launch_policy_train_tile(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B, int64_t B_global,
                                 double eps, double entropy_weight, const float* adv_col) {
    if (p->dtype != PPO_DTYPE_F32 || p->F != 72 || p->L != 2 || ro->H != 32 || ro->compact) return PPO_ERR_UNSUPPORTED;
    if (!p->dz1f.p || !p->dz2f.p) return PPO_ERR_UNSUPPORTED;
    TrainTileArgs ta = {};
    FwdArgs& a = ta.f;
    a.w1p = (const float4*)p->w1p.p; a.w2p = (const float4*)p->w2p.p; a.b1p = (const float4*)p->b1p.p;
    a.b2p = (const float4*)p->b2p.p; a.w3p = (const float4*)p->w3p.p; a.b3 = p->b3.p;
    a.states = ro->states.p; a.active = ro->active.p; a.idx = idx_dev; a.B = B;
    a.dY = (float4*)p->dY.p; a.loss_terms = p->loss_terms.p;
    a.actions = ro->actions.p; a.p_old = ro->p_sel.p; a.adv = adv_col;
    a.eps = eps; a.c_over_B = (float)(entropy_weight / (double)B_global); a.inv_B = (float)(1.0 / (double)B_global);
    ta.w2tp = (const float4*)p->w2tp.p;
    ta.h1t = (float4*)p->act1.p; ta.dz2t = (float4*)p->dz2f.p; ta.dz1t = (float4*)p->dz1f.p;   // operand layout, not fragment order
    ta.slabs = p->slabs.p; ta.slab_stride = slab_floats(p->F, p->HID, p->L);
    int nwg = 0;
    int32_t s;
    if (p->HID == 256) s = launch_tt<72, 256>(p, ta, B, &nwg);
    else if (p->HID == 128) s = launch_tt<72, 128>(p, ta, B, &nwg);
    else return PPO_ERR_UNSUPPORTED;
    if (s != PPO_OK) return s;
    return launch_policy_wgrad_tr(p, ro, idx_dev, B, nwg);
}
