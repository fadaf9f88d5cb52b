// ppo_policy_bwd_small.hip -- K11 for SMALL minibatches (a few hundred 32-row tiles: the per-GPU shard of a strong-
// scaling run, or the reference's own batch sizes): the same gradient as k_policy_bwd (Zygote's Flux.gradient of
// step_batch!, src/train.jl:65-79; math: SURVEY.md Appendix A) organised as the classic three products instead of one
// fused pass.
//
// Why: the fused kernel keeps a full weight-gradient accumulator set per workgroup and ends with one 341 KB slab per
// workgroup -- 87 MB written and 87 MB re-read by k_grad_reduce whatever the batch size (17 + 17 us).  At 4096 tiles
// that is 6 % of the optimiser step; at 512 tiles (two tiles per workgroup) it is 27 %.  Here the weight gradients are
// output-stationary: every 64 x 64 block of dW2 / 64 x 96 block of dW1 has a few owners that split the ROWS between
// them (split-K), so the partials are 7 MB at any batch size and nothing is accumulated that is not also reduced.
//   k_policy_bwd_data   per tile: dZ2 = (W3^T dY) . lrelu'(H2), dH1 = W2^T dZ2 (MFMA), dZ1 = dH1 . lrelu'(H1), the
//                       small gradients (db1, db2, dW3, db3); dZ2 / dZ1 leave in the forward's accumulator-fragment
//                       order (coalesced 1 KiB stores), 67 MB at 512 tiles: L2 / Infinity-Cache resident
//   k_policy_wgrad      dW2 += dZ2 H1^T, dW1 += dZ1 X^T: workgroup = (output block, K-slice), 4 waves interleave the row
//                       tiles of the slice; operands come back with coalesced fragment loads and are transposed through
//                       wave-private LDS ([feature][36], the layout of k_policy_bwd's phases C / D); the four waves' sums
//                       meet in LDS (fixed order) and go to the block's region of "virtual slab" `slice`
//   k_grad_reduce       unchanged: fixed-order sum of the virtual slabs -> flat Flux-order gradient (bitwise reproducible)
#include "ppo_internal.h"
#include "ppo_device.h"
#include <cstdlib>

#define SB_LD 36        // LDS leading dimension (rows) of the transposed tiles: see k_policy_bwd

struct BwdSmallArgs {
    const int8_t* states; const int32_t* idx; int64_t B;   // B = 32-row tiles
    int tps, x_by_tile;
    const float4* act1; const float4* act2; const float4* dY;
    const float4* w2tp; const float4* w3p;
    float4* dz2f; float4* dz1f;                             // [tile][feature tile][4][64] float4, like act1 / act2
    float* slabs; size_t slab_stride;
    int ksplit;                                             // K-slices of k_policy_wgrad
    int xcd_map;                                            // 1: workgroup -> (block, slice) keeps a K-slice on ONE XCD (ksplit % 8 == 0)
    // any num_hidden_layers (test/policy.jl:9-19): L hidden layers, actl[l] / dzl[l] = saved output / dZ of hidden layer l
    // (l = 0 first, L - 1 last; L == 2: {act1, act2} / {dz1f, dz2f}), w2tp = the L - 1 packed W^T streams back to back
    int L;
    const float4* actl[4]; float4* dzl[4];
};

// ---------------------------------------------------------------------------------------------- backward-data
template <int F, int HID>
__global__ __launch_bounds__(HID * 2, 2) void k_policy_bwd_data(BwdSmallArgs a) {
    constexpr int NT = HID / 32, LD = SB_LD, S4 = HID / 8, FP = ((F + 31) / 32) * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool Z2R = PPO_BWD_Z2ROW_AT(HID);         // dZ2 row-major [32][RS] (16-byte B-operand reads, see k_policy_bwd)
    constexpr int RS = HID + 4, Z2SZ = Z2R ? 32 * RS : HID * LD;
    float* sZ2 = smem;                          // Z2R: dZ2 [32][RS]; else [HID][LD] dZ2^T
    float* sH2 = sZ2 + Z2SZ;                    // [HID][LD] H2^T
    float* sZ1 = sH2 + HID * LD;                // [HID][LD] dZ1^T
    float* sDY = sZ1 + HID * LD;                // [32][4]
    float* sW3 = sDY + 32 * 4;                  // [HID][4]
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float db1 = 0.f, db2 = 0.f, db3 = 0.f, dw3[4] = {0.f, 0.f, 0.f, 0.f};
    if (tid < HID) {                                            // w3p is [h][tile][r][4]: un-permute to [f][4]
        const int kk = tid & 31, hh = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3);
        *reinterpret_cast<float4*>(&sW3[tid * 4]) = a.w3p[(size_t)(hh * NT + (tid >> 5)) * 16 + r];
    }
    __syncthreads();
    const float4* const w2t = a.w2tp + (size_t)w * S4 * 64 + lane;     // this wave's W2^T tile
    const unsigned fb = (unsigned)(32 * w + 4 * h);
    // the tile's inputs are fetched one tile ahead (first tile: before the loop): a workgroup walks only a few tiles and
    // an HBM / L2 round trip per tile in front of phase A is a third of the tile's time
    float4 n1[4], n2[4], ndy = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch_tile = [&](int64_t t) {
        const float4* s1 = a.act1 + ((size_t)t * NT + w) * 4 * 64 + lane;
        const float4* s2 = a.act2 + ((size_t)t * NT + w) * 4 * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) { n2[q] = s2[q * 64]; n1[q] = s1[q * 64]; }
        ndy = a.dY[(size_t)t * 32 + j];
    };
    if ((int64_t)blockIdx.x < a.B) fetch_tile(blockIdx.x);
    for (int64_t tile = blockIdx.x; tile < a.B; tile += gridDim.x) {
        unsigned lb = fb * LD + j;
        asm volatile("" : "+v"(lb));                                   // per-tile opaque base (see k_policy_bwd)
        float* const z2b = sZ2 + lb;
        float* const h2b = sH2 + lb;
        float* const z1b = sZ1 + lb;
        const float* const w3b = sW3 + fb * 4;
        // ---- phase A: dZ2 of feature tile w
        float4 v1[4], v2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { v2[q] = n2[q]; v1[q] = n1[q]; }
        const float4 dy = ndy;
        if (w == 0 && h == 0) *reinterpret_cast<float4*>(&sDY[j * 4]) = dy;
        float4* const zo = a.dz2f + ((size_t)tile * NT + w) * 4 * 64 + lane;
        // dH2^T[f, row] = sum_o W3[o, f] dY[row, o] of this wave's feature tile as two MFMAs (k = the 4 outputs): the
        // accumulator is the lane-is-row fragment, register 4q + e <-> feature e + 8q (+ 4h) (see k_policy_bwd)
        f32x16 dh2;
#pragma unroll
        for (int r = 0; r < 16; ++r) dh2[r] = 0.0f;
        dh2 = __builtin_amdgcn_mfma_f32_32x32x2f32(sW3[(32 * w + j) * 4 + h], h ? dy.y : dy.x, dh2, 0, 0, 0);
        dh2 = __builtin_amdgcn_mfma_f32_32x32x2f32(sW3[(32 * w + j) * 4 + 2 + h], h ? dy.w : dy.z, dh2, 0, 0, 0);
        (void)w3b;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float h2v[4] = {v2[q].x, v2[q].y, v2[q].z, v2[q].w};
            float z[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int fo = e + 8 * q;
                const float dh = dh2[4 * q + e];
                z[e] = dh * (h2v[e] > 0.0f ? 1.0f : 0.01f);
                if constexpr (!Z2R) z2b[fo * LD] = z[e];
                h2b[fo * LD] = h2v[e];
            }
            if constexpr (Z2R) *reinterpret_cast<float4*>(sZ2 + j * RS + fb + 8 * q) = make_float4(z[0], z[1], z[2], z[3]);
            zo[q * 64] = make_float4(z[0], z[1], z[2], z[3]);
        }
        __syncthreads();
        // ---- phase B: small grads of feature 32w + j (rows 16h .. 16h+15), dH1 tile w = W2^T dZ2, dZ1
        // W2^T stream (L2-resident): two register sets of PF fragment groups, ping-ponged -- the loads of one set are
        // issued BEFORE the MFMAs that consume the other (sched_barrier pins that order, see k_policy_bwd); the first
        // set and the next tile's inputs go out in front of the small VALU gradient sums
        constexpr int PF = 4;
        static_assert(S4 % (2 * PF) == 0, "ring sets");
        float4 ringA[PF], ringB[PF];
#pragma unroll
        for (int g = 0; g < PF; ++g) ringA[g] = w2t[(size_t)g * 64];
        if (tile + gridDim.x < a.B) fetch_tile(tile + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
        {
            const float* gz = Z2R ? sZ2 + (16 * h) * RS + 32 * w + j : sZ2 + (32 * w + j) * LD + 16 * h;
            const float* gh = sH2 + (32 * w + j) * LD + 16 * h;
            const float* gy = sDY + 64 * h;
            float s2 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll 1
            for (int rc = 0; rc < 16; rc += 4) {
                float4 z4;
                if constexpr (Z2R) z4 = make_float4(gz[rc * RS], gz[(rc + 1) * RS], gz[(rc + 2) * RS], gz[(rc + 3) * RS]);
                else z4 = *reinterpret_cast<const float4*>(gz + rc);
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
        }
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            // B operand: Z2R: dZ2[row j][f = 8g + 4h + e], one 16-byte read per fragment group; else dZ2^T[f = 8g + 2e + h][row j]
            const float* bz = Z2R ? sZ2 + j * RS + 4 * h : sZ2 + h * LD + j;
            auto mfma_set = [&](const float4 (&rg)[PF]) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
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
                bz += Z2R ? 8 * PF : 8 * PF * LD;
            };
            const float4* wn = w2t + (size_t)PF * 64;
#pragma unroll 1
            for (int s0 = 0; s0 < S4; s0 += 2 * PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) ringB[u] = wn[(size_t)u * 64];
                wn += (size_t)PF * 64;
                __builtin_amdgcn_sched_barrier(0);
                mfma_set(ringA);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < PF; ++u) ringA[u] = wn[(size_t)u * 64];      // tail padding covers the over-read
                wn += (size_t)PF * 64;
                __builtin_amdgcn_sched_barrier(0);
                mfma_set(ringB);
                __builtin_amdgcn_sched_barrier(0);
            }
            float4* const z1o = a.dz1f + ((size_t)tile * NT + w) * 4 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float h1v[4] = {v1[q].x, v1[q].y, v1[q].z, v1[q].w};
                float z[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    z[e] = acc[4 * q + e] * (h1v[e] > 0.0f ? 1.0f : 0.01f);
                    z1b[(e + 8 * q) * LD] = z[e];
                }
                z1o[q * 64] = make_float4(z[0], z[1], z[2], z[3]);
            }
        }
        __syncthreads();
        // ---- db1 of feature 32w + j
        {
            const float* g1 = sZ1 + (32 * w + j) * LD + 16 * h;
            float s1 = 0.f;
#pragma unroll
            for (int rc = 0; rc < 16; rc += 4) {
                const float4 z4 = *reinterpret_cast<const float4*>(g1 + rc);
                s1 += z4.x; s1 += z4.y; s1 += z4.z; s1 += z4.w;
            }
            db1 += s1;
        }
        // no barrier here: the next tile's phase A writes sZ2 / sH2 / sDY only, which every wave finished reading before
        // the barrier above; sZ1 is rewritten behind the next tile's first barrier
    }
    // small-gradient tail of slab blockIdx.x (same places as k_policy_bwd's slab)
    float* slab = a.slabs + (size_t)blockIdx.x * a.slab_stride + (size_t)HID * HID + (size_t)HID * FP;
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

// ---------------------------------------------------------------------------------------------- backward-data, any depth
// The same pass for Policy(F, HID, L, 4) with L = 1, 3 or 4 hidden layers (test/policy.jl:9-19): per tile
//   dZ_{L-1} = (W3^T dY) . lrelu'(H_{L-1});   for l = L-2 .. 0:  dH_l = W_{l+1}^T dZ_{l+1} (MFMA),  dZ_l = dH_l . lrelu'(H_l)
// with dZ ping-ponged between two LDS tiles in the B-operand form of the next product (one workgroup barrier per layer:
// a layer reads tile `cur` from every wave and writes its own feature tile of `cur ^ 1`), every dZ_l stored in fragment
// order for k_policy_wgrad, and the bias / W3 gradients summed on the VALU as in k_policy_bwd_data.  Correctness first:
// the per-layer activation fetch is not hidden behind the previous layer's MFMAs.
template <int F, int HID>
__global__ __launch_bounds__(HID * 2, 2) void k_policy_bwd_data_deep(BwdSmallArgs a) {
    constexpr int NT = HID / 32, LD = SB_LD, S4 = HID / 8, FP = ((F + 31) / 32) * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool Z2R = PPO_BWD_Z2ROW_AT(HID);
    constexpr int RS = HID + 4, Z2SZ = Z2R ? 32 * RS : HID * LD;
    float* const sZ0 = smem;                    // dZ tile A  (Z2R: [32][RS] row-major; else [HID][LD] transposed)
    float* const sZ1 = sZ0 + Z2SZ;              // dZ tile B
    float* const sH = sZ1 + Z2SZ;               // [HID][LD] H_{L-1}^T (dW3 sums)
    float* const sDY = sH + HID * LD;           // [32][4]
    float* const sW3 = sDY + 32 * 4;            // [HID][4]
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.L;
    float db[4] = {0.f, 0.f, 0.f, 0.f}, db3 = 0.f, dw3[4] = {0.f, 0.f, 0.f, 0.f};
    if (tid < HID) {                                            // w3p is [h][tile][r][4]: un-permute to [f][4]
        const int kk = tid & 31, hh = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3);
        *reinterpret_cast<float4*>(&sW3[tid * 4]) = a.w3p[(size_t)(hh * NT + (tid >> 5)) * 16 + r];
    }
    __syncthreads();
    const unsigned fb = (unsigned)(32 * w + 4 * h);
    // write this lane's 16 dZ values (register 4q + e <-> feature fb + e + 8q, row j) into a dZ tile in B-operand form
    auto put_z = [&](float* sZ, int q, const float (&z)[4]) {
        if constexpr (Z2R) *reinterpret_cast<float4*>(sZ + j * RS + fb + 8 * q) = make_float4(z[0], z[1], z[2], z[3]);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) sZ[(fb + e + 8 * q) * LD + j] = z[e];
        }
    };
    // sum over rows [16h, 16h + 16) of feature 32w + j of a dZ tile
    auto row_sum = [&](const float* sZ) {
        float s = 0.f;
        if constexpr (Z2R) {
            const float* g = sZ + (16 * h) * RS + 32 * w + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += g[r * RS];
        } else {
            const float* g = sZ + (32 * w + j) * LD + 16 * h;
#pragma unroll
            for (int rc = 0; rc < 16; rc += 4) { const float4 z4 = *reinterpret_cast<const float4*>(g + rc); s += z4.x; s += z4.y; s += z4.z; s += z4.w; }
        }
        return s;
    };
    auto add_db = [&](int l, float s) {
#pragma unroll
        for (int k = 0; k < 4; ++k) db[k] += (k == l) ? s : 0.0f;
    };
    for (int64_t tile = blockIdx.x; tile < a.B; tile += gridDim.x) {
        // ---- top layer: dZ_{L-1} of feature tile w
        float4 vt[4];
        {
            const float4* st = a.actl[L - 1] + ((size_t)tile * NT + w) * 4 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) vt[q] = st[q * 64];
        }
        const float4 dy = a.dY[(size_t)tile * 32 + j];
        if (w == 0 && h == 0) *reinterpret_cast<float4*>(&sDY[j * 4]) = dy;
        f32x16 dh;
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = 0.0f;
        dh = __builtin_amdgcn_mfma_f32_32x32x2f32(sW3[(32 * w + j) * 4 + h], h ? dy.y : dy.x, dh, 0, 0, 0);
        dh = __builtin_amdgcn_mfma_f32_32x32x2f32(sW3[(32 * w + j) * 4 + 2 + h], h ? dy.w : dy.z, dh, 0, 0, 0);
        {
            float4* const zo = a.dzl[L - 1] + ((size_t)tile * NT + w) * 4 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float hv[4] = {vt[q].x, vt[q].y, vt[q].z, vt[q].w};
                float z[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    z[e] = dh[4 * q + e] * (hv[e] > 0.0f ? 1.0f : 0.01f);
                    sH[(fb + e + 8 * q) * LD + j] = hv[e];
                }
                put_z(sZ0, q, z);
                zo[q * 64] = make_float4(z[0], z[1], z[2], z[3]);
            }
        }
        __syncthreads();
        {   // small gradients of the top layer: db_{L-1}, dW3, db3 (feature 32w + j, rows 16h .. 16h+15)
            add_db(L - 1, row_sum(sZ0));
            const float* gh = sH + (32 * w + j) * LD + 16 * h;
            const float* gy = sDY + 64 * h;
            float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll 1
            for (int rc = 0; rc < 16; rc += 4) {
                const float4 h4 = *reinterpret_cast<const float4*>(gh + rc);
                const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 y = *reinterpret_cast<const float4*>(gy + (rc + i) * 4);
                    d0 = fmaf(y.x, hv[i], d0); d1 = fmaf(y.y, hv[i], d1); d2 = fmaf(y.z, hv[i], d2); d3 = fmaf(y.w, hv[i], d3);
                }
            }
            dw3[0] += d0; dw3[1] += d1; dw3[2] += d2; dw3[3] += d3;
            if (tid < 4) {
                float s = 0.f;
                for (int r = 0; r < 32; ++r) s += sDY[r * 4 + tid];
                db3 += s;
            }
        }
        // ---- the layers below: dH_l = W_{l+1}^T dZ_{l+1}, dZ_l = dH_l . lrelu'(H_l)
        int cur = 0;
#pragma unroll 1
        for (int l = L - 2; l >= 0; --l) {
            const float* const sZc = cur ? sZ1 : sZ0;
            float* const sZn = cur ? sZ0 : sZ1;
            float4 vl[4];
            {
                const float4* sl = a.actl[l] + ((size_t)tile * NT + w) * 4 * 64 + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) vl[q] = sl[q * 64];
            }
            const float4* const w2t = a.w2tp + (size_t)l * (HID * HID / 4) + (size_t)w * S4 * 64 + lane;   // wave w's tile of W_{l+1}^T
            constexpr int PF = 4;
            static_assert(S4 % (2 * PF) == 0, "ring sets");
            float4 ringA[PF], ringB[PF];
#pragma unroll
            for (int g = 0; g < PF; ++g) ringA[g] = w2t[(size_t)g * 64];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const float* bz = Z2R ? sZc + j * RS + 4 * h : sZc + h * LD + j;
            auto mfma_set = [&](const float4 (&rg)[PF]) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
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
                bz += Z2R ? 8 * PF : 8 * PF * LD;
            };
            const float4* wn = w2t + (size_t)PF * 64;
#pragma unroll 1
            for (int s0 = 0; s0 < S4; s0 += 2 * PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) ringB[u] = wn[(size_t)u * 64];
                wn += (size_t)PF * 64;
                __builtin_amdgcn_sched_barrier(0);
                mfma_set(ringA);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < PF; ++u) ringA[u] = wn[(size_t)u * 64];      // next stream's head / tail padding covers the over-read
                wn += (size_t)PF * 64;
                __builtin_amdgcn_sched_barrier(0);
                mfma_set(ringB);
                __builtin_amdgcn_sched_barrier(0);
            }
            float4* const zo = a.dzl[l] + ((size_t)tile * NT + w) * 4 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float hv[4] = {vl[q].x, vl[q].y, vl[q].z, vl[q].w};
                float z[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = acc[4 * q + e] * (hv[e] > 0.0f ? 1.0f : 0.01f);
                put_z(sZn, q, z);
                zo[q * 64] = make_float4(z[0], z[1], z[2], z[3]);
            }
            __syncthreads();
            add_db(l, row_sum(sZn));
            cur ^= 1;
        }
        __syncthreads();                                              // the next tile's top layer rewrites sZ0 / sH / sDY
    }
    // small-gradient tail of slab blockIdx.x: [db of hidden layer 0][db of layers 1 .. L-1][dW3][db3]
    float* slab = a.slabs + (size_t)blockIdx.x * a.slab_stride + (size_t)(L - 1) * HID * HID + (size_t)HID * FP;
    float* sw3 = slab + (size_t)L * HID; float* sb3 = sw3 + HID * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) db[k] += __shfl_xor(db[k], 32);
#pragma unroll
    for (int i = 0; i < 4; ++i) dw3[i] += __shfl_xor(dw3[i], 32);
    if (h == 0) {
        const int f = 32 * w + j;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < L) slab[(size_t)k * HID + f] = db[k];
        *reinterpret_cast<float4*>(&sw3[f * 4]) = make_float4(dw3[0], dw3[1], dw3[2], dw3[3]);
    }
    if (tid < 4) sb3[tid] = db3;
}

// ---------------------------------------------------------------------------------------------- weight gradients
// wave-private transposed tile: fragment (lane = row, registers = 16 features) -> [feature][LD rows]
__device__ __forceinline__ void frag_to_lds(float* buf, const float4 (&v)[4], int j, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) buf[(4 * h + e + 8 * q) * SB_LD + j] = x[e];
    }
}

// TR = 1: dzl[] / actl[] hold the tiles in the OPERAND LAYOUT of the row contraction (lane = feature, element q of the
// tile = rows 16 * half + 4q .. + 3: what k_policy_train_tile stores) -- the A / B operands are plain wave loads and no
// LDS transposes are left in this kernel; TR = 0: fragment order (lane = row), transposed through wave-private LDS tiles.
template <int F, int HID, bool TR>
__global__ __launch_bounds__(256, 2) void k_policy_wgrad(BwdSmallArgs a) {
    constexpr int NT = HID / 32, LD = SB_LD, FP = ((F + 31) / 32) * 32, NI = FP / 32;
    constexpr int NB2 = (NT / 2) * (NT / 2);            // 64 x 64 blocks of dW2 (2 A tiles + 2 B tiles)
    constexpr int NB1 = NT;                             // 32 x FP blocks of dW1 (1 A tile + NI B tiles)
    constexpr int TILE = 32 * LD;                       // floats of one transposed tile
    constexpr int WT = (1 + NI > 4) ? 1 + NI : 4;       // transposed tiles per wave (4 -> 73.7 KB per workgroup: two per CU)
    constexpr int XDW = 32 * F / 4, XPL = (XDW + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int v = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* const bufA = smem + (size_t)v * WT * TILE;                  // wave-private: A tiles first, then the B tiles
    const int NBH = (a.L - 1) * NB2;                                   // dW blocks of the L - 1 hidden->hidden layers
    // Workgroup -> (output block, K-slice).  All blocks of a K-slice read the SAME row tiles (every dZ / H operand tile is
    // used by NT/2 blocks), and the hardware deals consecutive workgroup ids round-robin over the 8 XCDs, each with its own
    // L2: the plain mapping (block fastest) spreads a slice's blocks over all eight L2s, so every operand tile is filled
    // eight times from the Infinity Cache.  XCD-aware mapping (xcd_map, opt-in): the workgroups that land on XCD x (ids x,
    // x + 8, ...) walk the blocks of slices x, x + 8, ... one slice after the other, so a slice's tiles are fetched into
    // ONE L2 once.  Measured slower (launch_small): the kernel is not bound by L2 fills -- with every workgroup of an XCD on
    // the same few tiles at the same time their loads queue on the same L2 channels instead of spreading over all of them.
    int block, slice;
    if (a.xcd_map) {
        const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
        slice = x + 8 * (i / (NBH + NB1)); block = i % (NBH + NB1);
    } else { block = blockIdx.x % (NBH + NB1); slice = blockIdx.x / (NBH + NB1); }
    const int64_t per = (a.B + a.ksplit - 1) / a.ksplit;
    const int64_t t0 = (int64_t)slice * per, t1 = (t0 + per < a.B) ? t0 + per : a.B;
    float* slab = a.slabs + (size_t)slice * a.slab_stride;             // one partial per K-slice: the four waves' sums meet in LDS
    const float* const red0 = smem;                                    // wave u's accumulators at red0 + u * WT * TILE
    // fixed-order in-workgroup sum: every wave parks its accumulators [reg][lane] in its own LDS tiles, then wave v adds
    // registers [v R/4, (v+1) R/4) of the four waves in wave order (bitwise reproducible)
    auto wg_sum = [&](int r) {
        return ((red0[(size_t)0 * WT * TILE + r * 64 + lane] + red0[(size_t)1 * WT * TILE + r * 64 + lane]) +
                red0[(size_t)2 * WT * TILE + r * 64 + lane]) + red0[(size_t)3 * WT * TILE + r * 64 + lane];
    };
    float* sW1 = slab + (size_t)(a.L - 1) * HID * HID;
    if (block < NBH) {
        // ---------------- dW block of hidden->hidden layer m (hidden layer m -> m + 1): dW = dZ_{m+1} H_m^T,
        // f-tiles 2fb, 2fb+1 x k-tiles 2kb, 2kb+1
        const int m = block / NB2, blk = block % NB2;
        float* sW2 = slab + (size_t)m * HID * HID;
        const float4* const dz_hi = a.dzl[m + 1];
        const float4* const act_lo = a.actl[m];
        const int fbk = blk / (NT / 2), kbk = blk % (NT / 2);
        f32x16 acc[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
        float* const bufB = bufA + 2 * TILE;
        float4 fa[2][4], fbv[2][4];
        auto fetch = [&](int64_t t) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    fa[x][q] = dz_hi[((size_t)t * NT + (2 * fbk + x)) * 4 * 64 + q * 64 + lane];
                    fbv[x][q] = act_lo[((size_t)t * NT + (2 * kbk + x)) * 4 * 64 + q * 64 + lane];
                }
        };
        if (t0 + v < t1) fetch(t0 + v);
        for (int64_t t = t0 + v; t < t1; t += 4) {
            if constexpr (TR) {
                float4 ca[2][4], cb[2][4];                             // element q of a tile IS the operand of k-steps 4q .. 4q+3
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { ca[x][q] = fa[x][q]; cb[x][q] = fbv[x][q]; }
                if (t + 4 < t1) fetch(t + 4);                          // next row tile lands under the MFMAs
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int y = 0; y < 2; ++y) {
                            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[x][q].x, cb[y][q].x, acc[x][y], 0, 0, 0);
                            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[x][q].y, cb[y][q].y, acc[x][y], 0, 0, 0);
                            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[x][q].z, cb[y][q].z, acc[x][y], 0, 0, 0);
                            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[x][q].w, cb[y][q].w, acc[x][y], 0, 0, 0);
                        }
                continue;
            }
#pragma unroll
            for (int x = 0; x < 2; ++x) { frag_to_lds(bufA + x * TILE, fa[x], j, h); frag_to_lds(bufB + x * TILE, fbv[x], j, h); }
            if (t + 4 < t1) fetch(t + 4);                              // next row tile lands under the MFMAs
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // wave-private tiles: own writes have landed
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                float4 a4[2], b4[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    a4[x] = *reinterpret_cast<const float4*>(bufA + x * TILE + j * LD + 16 * h + 4 * q);
                    b4[x] = *reinterpret_cast<const float4*>(bufB + x * TILE + j * LD + 16 * h + 4 * q);
                }
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) {
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[x].x, b4[y].x, acc[x][y], 0, 0, 0);
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[x].y, b4[y].y, acc[x][y], 0, 0, 0);
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[x].z, b4[y].z, acc[x][y], 0, 0, 0);
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[x].w, b4[y].w, acc[x][y], 0, 0, 0);
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // reads done before the next tile overwrites the buffers
        }
        static_assert(64 * 64 <= WT * TILE, "accumulators fit the wave's LDS tiles");
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) bufA[((x * 2 + y) * 16 + r) * 64 + lane] = acc[x][y][r];
        __syncthreads();
        {   // wave v owns tile (x, y) = (v >> 1, v & 1) of the block
            const int x = v >> 1, y = v & 1;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sW2[((size_t)((2 * fbk + x) * NT + (2 * kbk + y)) * 16 + r) * 64 + lane] = wg_sum(v * 16 + r);
        }
    } else {
        // ---------------- dW1 block: k-tile kb x all NI input tiles (columns >= F are zero padding)
        const int kbk = block - NBH;
        float* const bufB = bufA + TILE;
        f32x16 acc[NI];
#pragma unroll
        for (int y = 0; y < NI; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[y][r] = 0.0f;
        for (int i = lane; i < NI * TILE; i += 64) bufB[i] = 0.0f;    // padded input features stay zero
        float4 fa[4];
        uint32_t xd[XPL];
        auto fetch = [&](int64_t t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) fa[q] = a.dzl[0][((size_t)t * NT + kbk) * 4 * 64 + q * 64 + lane];
            const int sidx = a.x_by_tile ? 0 : a.idx[t / a.tps];
            const uint32_t* xs = reinterpret_cast<const uint32_t*>(
                a.states + (a.x_by_tile ? (size_t)t : ((size_t)sidx * a.tps + (size_t)(t % a.tps))) * 32 * F);
#pragma unroll
            for (int i = 0; i < XPL; ++i) { const int d = lane + 64 * i; xd[i] = d < XDW ? xs[d] : 0u; }
        };
        if (t0 + v < t1) fetch(t0 + v);
        for (int64_t t = t0 + v; t < t1; t += 4) {
            float4 ca[4];
            if constexpr (TR) {
#pragma unroll
                for (int q = 0; q < 4; ++q) ca[q] = fa[q];
            } else frag_to_lds(bufA, fa, j, h);
#pragma unroll
            for (int i = 0; i < XPL; ++i) {
                const int d = lane + 64 * i;                          // dword d = row * (F/4) + c: features 4c .. 4c+3
                if (d < XDW) {
                    const int row = d / (F / 4), c = d % (F / 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bufB[(4 * c + e) * LD + row] = (float)(int)(int8_t)(xd[i] >> (8 * e));
                }
            }
            if (t + 4 < t1) fetch(t + 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            auto dw1_step = [&](const float4 a4, const int q) {
                float4 b4[NI];
#pragma unroll
                for (int y = 0; y < NI; ++y) b4[y] = *reinterpret_cast<const float4*>(bufB + y * TILE + j * LD + 16 * h + 4 * q);
#pragma unroll
                for (int y = 0; y < NI; ++y) {
                    acc[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[y].x, acc[y], 0, 0, 0);
                    acc[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[y].y, acc[y], 0, 0, 0);
                    acc[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[y].z, acc[y], 0, 0, 0);
                    acc[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[y].w, acc[y], 0, 0, 0);
                }
            };
            if constexpr (TR) {                                        // static register indices: a dynamic q would put ca[] in scratch
#pragma unroll
                for (int q = 0; q < 4; ++q) dw1_step(ca[q], q);
            } else {
#pragma unroll 1
                for (int q = 0; q < 4; ++q) dw1_step(*reinterpret_cast<const float4*>(bufA + j * LD + 16 * h + 4 * q), q);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int y = 0; y < NI; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) bufA[(y * 16 + r) * 64 + lane] = acc[y][r];
        __syncthreads();
        constexpr int RQ = 16 * NI / 4;                                // registers per wave of the final sum
#pragma unroll
        for (int rr = 0; rr < RQ; ++rr) {
            const int r = v * RQ + rr;                                 // = y * 16 + reg
            sW1[((size_t)(kbk * NI + r / 16) * 16 + (r % 16)) * 64 + lane] = wg_sum(r);
        }
    }
}

template <int F, int HID>
static int32_t launch_small(ppo_policy_s* p, BwdSmallArgs& a, int tr_tail_wg = 0) {
    constexpr int NT = HID / 32, NI = (F + 31) / 32, WT = (1 + NI > 4) ? 1 + NI : 4;
    const size_t z2 = PPO_BWD_Z2ROW_AT(HID) ? (size_t)32 * (HID + 4) : (size_t)HID * SB_LD;
    const size_t lds_data = sizeof(float) * (z2 + (size_t)2 * HID * SB_LD + 32 * 4 + (size_t)HID * 4);
    const size_t lds_deep = sizeof(float) * (2 * z2 + (size_t)HID * SB_LD + 32 * 4 + (size_t)HID * 4);
    const size_t lds_w = sizeof(float) * (size_t)4 * WT * 32 * SB_LD;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_policy_bwd_data<F, HID>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_data));
        HIP_TRY(hipFuncSetAttribute((const void*)k_policy_bwd_data_deep<F, HID>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_deep));
        HIP_TRY(hipFuncSetAttribute((const void*)k_policy_wgrad<F, HID, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w));
        HIP_TRY(hipFuncSetAttribute((const void*)k_policy_wgrad<F, HID, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w));
        attr_set = true;
    }
    static const int dbg_cap = [] { const char* v = getenv("PPO_DEBUG_BWD_DATA_WGS"); return v ? atoi(v) : 256; }();
    const int nwg = (int)(a.B < dbg_cap ? a.B : dbg_cap);
    const int blocks = (a.L - 1) * (NT / 2) * (NT / 2) + NT;
    // K-slices: two workgroups per CU over the block list, all resident at once (a partial second round costs a whole one)
    int ks = 512 / blocks;
    if (ks < 1) ks = 1;
    // XCD-aware block order (see k_policy_wgrad; needs a multiple of 8 slices): measured SLOWER and therefore off --
    // gpurun_out/r3h, alternating on one box: wgrad 42.2 -> 47.4 us at 512 tiles, 27.6 -> 28.8 at 256, 410 -> 562 us for the
    // three-layer policy at 4096 tiles (ks = 8 / 16 / 24 alike).  A/B knobs: PPO_WGRAD_XCD=1, PPO_WGRAD_KS=n
    static const int xcd_on = [] { const char* v = getenv("PPO_WGRAD_XCD"); return v ? atoi(v) : 0; }();
    static const int ks_force = [] { const char* v = getenv("PPO_WGRAD_KS"); return v ? atoi(v) : 0; }();
    if (ks_force > 0) ks = ks_force;
    a.xcd_map = 0;
    if (xcd_on && ks >= 8 && a.B >= 8) { ks = (ks / 8) * 8; if ((int64_t)ks > a.B) ks = (int)(a.B / 8) * 8; a.xcd_map = 1; }
    if ((int64_t)ks > a.B) ks = (int)a.B;
    a.ksplit = ks;
    p->nwg_bwd = ks;                 // slabs holding weight-gradient partials (one per K-slice)
    p->nwg_small = nwg;              // slabs holding the small-gradient tails
    if (tr_tail_wg > 0) {            // the dZ / H1 tiles come from k_policy_train_tile, in operand layout: weight gradients only
        p->nwg_small = tr_tail_wg;
        ProfScope ps("k_policy_wgrad");
        hipLaunchKernelGGL((k_policy_wgrad<F, HID, true>), dim3(blocks * ks), dim3(256), lds_w, ppo_stream(), a);
        HIP_TRY(hipGetLastError());
        return PPO_OK;
    }
    {
        ProfScope ps("k_policy_bwd_data");
        if (a.L == 2) hipLaunchKernelGGL((k_policy_bwd_data<F, HID>), dim3(nwg), dim3(HID * 2), lds_data, ppo_stream(), a);
        else hipLaunchKernelGGL((k_policy_bwd_data_deep<F, HID>), dim3(nwg), dim3(HID * 2), lds_deep, ppo_stream(), a);
    }
    {
        ProfScope ps("k_policy_wgrad");
        hipLaunchKernelGGL((k_policy_wgrad<F, HID, false>), dim3(blocks * ks), dim3(256), lds_w, ppo_stream(), a);
    }
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

static int32_t bwd_small_impl(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B, int tr_tail_wg);

// PPO_ERR_UNSUPPORTED (no error text): shape or size not covered -> the caller runs the fused kernel
int32_t launch_policy_bwd_small(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B) {
    return bwd_small_impl(p, ro, idx_dev, B, 0);
}
// weight gradients from the operand-layout tiles k_policy_train_tile left in act1 / dz2f / dz1f; nwg_tail = its workgroups
// (the slabs holding the small-gradient tails)
int32_t launch_policy_wgrad_tr(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B, int nwg_tail) {
    return bwd_small_impl(p, ro, idx_dev, B, nwg_tail);
}

static int32_t bwd_small_impl(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B, int tr_tail_wg) {
    if (p->dtype != PPO_DTYPE_F32 || !p->dz1f.p || (p->L >= 2 && !p->dz2f.p) || (p->L > 2 && !p->dzm.p)) return PPO_ERR_UNSUPPORTED;
    BwdSmallArgs a;
    a.tps = ro->H / 32;
    a.states = ro->compact ? p->xs.p : ro->states.p; a.x_by_tile = ro->compact ? 1 : 0;
    a.idx = idx_dev; a.B = B * a.tps;
    a.act1 = (const float4*)p->act1.p; a.act2 = (const float4*)p->act2.p; a.dY = (const float4*)p->dY.p;
    a.w2tp = (const float4*)p->w2tp.p; a.w3p = (const float4*)p->w3p.p;
    a.dz2f = (float4*)p->dz2f.p; a.dz1f = (float4*)p->dz1f.p;
    a.slabs = p->slabs.p; a.slab_stride = slab_floats(p->F, p->HID, p->L);
    a.ksplit = 1; a.xcd_map = 0;
    a.L = p->L;
    const size_t lstride = (size_t)p->cap_tiles * (p->HID / 32) * 256;             // float4 per saved layer
    for (int l = 0; l < 4; ++l) { a.actl[l] = nullptr; a.dzl[l] = nullptr; }
    for (int l = 0; l < p->L; ++l) {
        const bool first = l == 0, last = l == p->L - 1;
        a.actl[l] = first ? (const float4*)p->act1.p : (last ? (const float4*)p->act2.p : (const float4*)p->actm.p + (size_t)(l - 1) * lstride);
        a.dzl[l] = first ? (float4*)p->dz1f.p : (last ? (float4*)p->dz2f.p : (float4*)p->dzm.p + (size_t)(l - 1) * lstride);
    }
    if (p->F == 72 && p->HID == 256) return launch_small<72, 256>(p, a, tr_tail_wg);
    if (p->F == 72 && p->HID == 128) return launch_small<72, 128>(p, a, tr_tail_wg);
    if (tr_tail_wg) return PPO_ERR_UNSUPPORTED;
    if (p->F == 216 && p->HID == 256) return launch_small<216, 256>(p, a);
    if (p->F == 216 && p->HID == 128) return launch_small<216, 128>(p, a);
    return PPO_ERR_UNSUPPORTED;
}
