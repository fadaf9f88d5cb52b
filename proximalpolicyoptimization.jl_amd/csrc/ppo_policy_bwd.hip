// ppo_policy_bwd.hip -- K11: hand-written backward of softmax+MLP (replaces Zygote's
// Flux.gradient(weights) do ... end, src/train.jl:65-79).  Math: SURVEY.md Appendix A.
//
// gfx950 mapping.  A workgroup of HID/32 waves (8 waves = two per SIMD for HID=256) walks 32-row
// tiles (one state each) and keeps its share of EVERY weight gradient resident in MFMA
// accumulators for the whole launch:
//   wave w owns output-feature tile w:
//     dW2[f-tile w, all k]  (NT 32x32 accumulators)   dW1[k-tile w, all i]  (NI accumulators)
// Per tile:
//   A  activations saved by the forward kernel in accumulator-fragment order come back with
//      coalesced 1 KiB wave loads and are TRANSPOSED THROUGH LDS ([feature][33] fp32, conflict-free
//      for both the row-major B reads and the feature-major A reads): dZ2 = (W3^T dY) . lrelu'(H2)
//   B  dH1^T = W2^T * dZ2^T  (A operand: pre-packed W2^T fragments streamed from L2, B operand:
//      dZ2 from LDS); dZ1 = dH1 . lrelu'(H1) -> LDS.     The tiny grads (dW3, db*) are VALU sums.
//   C  dW2 += dZ2 * H1^T     (contraction over the 32 rows: both operands read transposed from LDS)
//   D  dW1 += dZ1 * X^T
// At the end every workgroup writes one gradient slab; k_grad_reduce sums the slabs in a fixed
// order (bitwise reproducible, no float atomics).
//
// MFMA-bound: 2*32*(2*HID*HID + HID*F) flop per tile.
#include "ppo_internal.h"
#include "ppo_device.h"

struct BwdArgs {
    const int8_t* states; const int32_t* idx; int64_t B;
    const float4* act1; const float4* act2; const float4* dY;
    const float4* w2tp; const float4* w3p;
    float* slabs; size_t slab_stride;
};

template <int F, int HID>
__global__ __launch_bounds__(HID * 2, (HID >= 256 ? 2 : 1)) void k_policy_bwd(BwdArgs a) {
    constexpr int NT = HID / 32;                // feature tiles == waves per workgroup (wave w owns tile w)
    constexpr int NTHR = NT * 64;
    constexpr int FP = ((F + 31) / 32) * 32;
    constexpr int NI = FP / 32;
    constexpr int LD = 33;                      // padded leading dimension (rows) of the LDS tiles
    constexpr int XPT = F / 8;                  // state bytes staged per thread (threads 0..255)
    constexpr int PF = (HID >= 256) ? 4 : 8;    // W2^T fragment groups in flight per wave (register budget)
    constexpr int S4 = HID / 8;                 // fragment groups of one W2^T tile
    static_assert(S4 % PF == 0 && NTHR >= 256 && NTHR >= HID, "shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sZ2 = smem;                          // [HID][33]  dZ2^T
    float* sH1 = sZ2 + HID * LD;                // [HID][33]  H1^T
    float* sH2 = sH1 + HID * LD;                // [HID][33]  H2^T
    float* sZ1 = sH2 + HID * LD;                // [HID][33]  dZ1^T
    float* sX = sZ1 + HID * LD;                 // [FP][33]   X^T (float)
    float* sDY = sX + FP * LD;                  // [32][4]

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int j = lane & 31, h = lane >> 5;

    f32x16 accW2[NT];
    f32x16 accW1[NI];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW2[kt][r] = 0.0f;
#pragma unroll
    for (int it = 0; it < NI; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW1[it][r] = 0.0f;
    float db1 = 0.f, db2 = 0.f, db3 = 0.f, dw3[4] = {0.f, 0.f, 0.f, 0.f};

    for (int i = tid; i < FP * LD; i += NTHR) sX[i] = 0.0f;     // rows i >= F stay zero (padding of dW1)
    __syncthreads();

    const float4* w3_base = a.w3p + (size_t)(h * NT + w) * 16;
    const float4* w2t = a.w2tp + (size_t)w * S4 * 64 + lane;    // this wave's W2^T tile (k-tile w)

    for (int64_t tile = blockIdx.x; tile < a.B; tile += gridDim.x) {
        // ================= phase A: stage the tile (transposes through LDS)
        const float4 dy = a.dY[(size_t)tile * 32 + j];
        // keep the 16 W3 fragments out of the persistent register set: re-read them (L1-resident) per tile
        const float4* w3 = w3_base;
        asm volatile("" : "+v"(w3));
        if (w == 0 && h == 0) *reinterpret_cast<float4*>(&sDY[j * 4]) = dy;
        {
            const float4* s2 = a.act2 + ((size_t)tile * NT + w) * 4 * 64;
            const float4* s1 = a.act1 + ((size_t)tile * NT + w) * 4 * 64;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v2 = s2[q * 64 + lane];
                const float4 v1 = s1[q * 64 + lane];
                const float h2v[4] = {v2.x, v2.y, v2.z, v2.w};
                const float h1v[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q + e;
                    const int f = dfeat(w, r, h);
                    const float4 ww = w3[r];
                    const float dh = ww.x * dy.x + ww.y * dy.y + ww.z * dy.z + ww.w * dy.w;
                    sZ2[f * LD + j] = dh * (h2v[e] > 0.0f ? 1.0f : 0.01f);
                    sH2[f * LD + j] = h2v[e];
                    sH1[f * LD + j] = h1v[e];
                }
            }
        }
        if (tid < 256) {
            const int row = tid & 31, part = tid >> 5;            // 8 parts x F/8 features
            const int8_t* xr = a.states + (size_t)a.idx[tile] * 32 * F + (size_t)row * F + part * XPT;
#pragma unroll
            for (int i = 0; i < XPT; ++i) sX[(part * XPT + i) * LD + row] = (float)xr[i];
        }
        // start the W2^T stream before the barrier so the first groups land while the tile is staged
        float4 ring[PF];
#pragma unroll
        for (int g = 0; g < PF; ++g) ring[g] = w2t[(size_t)g * 64];
        __syncthreads();

        // ================= phase B: small VALU grads, dH1 = W2^T dZ2 (MFMA), dZ1 -> LDS
        if (tid < HID) {
            float s2 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) {
                const float z = sZ2[tid * LD + r];
                const float hv = sH2[tid * LD + r];
                const float4 y = *reinterpret_cast<const float4*>(&sDY[r * 4]);
                s2 += z; d0 += y.x * hv; d1 += y.y * hv; d2 += y.z * hv; d3 += y.w * hv;
            }
            db2 += s2; dw3[0] += d0; dw3[1] += d1; dw3[2] += d2; dw3[3] += d3;
        }
        if (tid < 4) {
            float s = 0.f;
            for (int r = 0; r < 32; ++r) s += sDY[r * 4 + tid];
            db3 += s;
        }
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll 1
            for (int s0 = 0; s0 < S4; s0 += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int s4 = s0 + u;
                    const float4 ww = ring[u];
                    ring[u] = w2t[(size_t)(s4 + PF) * 64];          // tail padding covers the over-read
                    float b[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[e] = sZ2[(2 * (4 * s4 + e) + h) * LD + j];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ww.x, b[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ww.y, b[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ww.z, b[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ww.w, b[3], acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = dfeat(w, r, h);
                const float hv = sH1[k * LD + j];
                sZ1[k * LD + j] = acc[r] * (hv > 0.0f ? 1.0f : 0.01f);
            }
        }
        __syncthreads();

        // ================= phase C: dW2[f,k] += sum_rows dZ2[f,row] * H1[k,row]   (wave w: f-tile w)
        if (tid < HID) {
            float s1 = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) s1 += sZ1[tid * LD + r];
            db1 += s1;
        }
        {
            const float* pa = sZ2 + (32 * w + j) * LD + h;
            const float* pb = sH1 + j * LD + h;
#pragma unroll 2
            for (int s = 0; s < 16; ++s) {
                const float av = pa[2 * s];
                float bv[NT];
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) bv[kt] = pb[32 * kt * LD + 2 * s];
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
                    accW2[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[kt], accW2[kt], 0, 0, 0);
            }
        }
        // ================= phase D: dW1[k,i] += sum_rows dZ1[k,row] * X[i,row]     (wave w: k-tile w)
        {
            const float* pa = sZ1 + (32 * w + j) * LD + h;
            const float* pb = sX + j * LD + h;
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {
                const float av = pa[2 * s];
                float bv[NI];
#pragma unroll
                for (int it = 0; it < NI; ++it) bv[it] = pb[32 * it * LD + 2 * s];
#pragma unroll
                for (int it = 0; it < NI; ++it)
                    accW1[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[it], accW1[it], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ================= write the slab (fragment order; k_grad_reduce maps it to Flux order)
    float* slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
    float* sW2 = slab;                                   // [(ft*NT+kt)*16 + r][64]
    float* sW1 = sW2 + (size_t)HID * HID;                // [(ft*NI+it)*16 + r][64]
    float* sb1 = sW1 + (size_t)HID * FP;
    float* sb2 = sb1 + HID;
    float* sw3 = sb2 + HID;                              // [HID][4]
    float* sb3 = sw3 + HID * 4;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) sW2[((size_t)(w * NT + kt) * 16 + r) * 64 + lane] = accW2[kt][r];
#pragma unroll
    for (int it = 0; it < NI; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) sW1[((size_t)(w * NI + it) * 16 + r) * 64 + lane] = accW1[it][r];
    if (tid < HID) {
        sb1[tid] = db1; sb2[tid] = db2;
        *reinterpret_cast<float4*>(&sw3[tid * 4]) = make_float4(dw3[0], dw3[1], dw3[2], dw3[3]);
    }
    if (tid < 4) sb3[tid] = db3;
}

template <int F, int HID>
static size_t bwd_lds_bytes() {
    constexpr int FP = ((F + 31) / 32) * 32;
    return sizeof(float) * ((size_t)4 * HID * 33 + (size_t)FP * 33 + 32 * 4);
}

int32_t launch_policy_bwd(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B) {
    BwdArgs a;
    a.states = ro->states.p; a.idx = idx_dev; a.B = B;
    a.act1 = (const float4*)p->act1.p; a.act2 = (const float4*)p->act2.p; a.dY = (const float4*)p->dY.p;
    a.w2tp = (const float4*)p->w2tp.p; a.w3p = (const float4*)p->w3p.p;
    a.slabs = p->slabs.p; a.slab_stride = slab_floats(p->F, p->HID);
    const int nwg = (int)(B < 256 ? B : 256);
    p->nwg_bwd = nwg;
    ProfScope ps("k_policy_bwd");
#define LAUNCH(FF, HH)                                                                                        \
    do {                                                                                                      \
        const size_t lds = bwd_lds_bytes<FF, HH>();                                                           \
        static bool attr_set = false;                                                                         \
        if (!attr_set) {                                                                                      \
            HIP_TRY(hipFuncSetAttribute((const void*)k_policy_bwd<FF, HH>,                                    \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
            attr_set = true;                                                                                  \
        }                                                                                                     \
        hipLaunchKernelGGL((k_policy_bwd<FF, HH>), dim3(nwg), dim3(HH * 2), lds, ppo_stream(), a);               \
    } while (0)
    if (p->F == 72 && p->HID == 256) LAUNCH(72, 256);
    else if (p->F == 72 && p->HID == 128) LAUNCH(72, 128);
    else if (p->F == 216 && p->HID == 128) LAUNCH(216, 128);
    else { ppo_set_error("unsupported policy shape (F,HID) for the gfx950 kernels"); return PPO_ERR_UNSUPPORTED; }
#undef LAUNCH
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}
