// ppo_optim.hip -- gradient slab reduction, K12 Adam (Flux legacy Adam + Flux.update!,
// src/train.jl:81; semantics SURVEY.md Appendix A) and the re-packing of the updated parameters
// into the MFMA A-operand fragment orders used by the forward/backward kernels.
#include "ppo_internal.h"
#include "ppo_device.h"

// flat Flux-order parameter vector of Policy(F, HID, NL, 4) (test/policy.jl:9-19): W1, b1, then the NL - 1 hidden->hidden
// layers (W, b) back to back -- layer l at offW2 + l * (HID*HID + HID) -- then W3, b3
struct ParamLayout {
    int F, HID, NT, FP, NI, NL2;                // NL2 = hidden->hidden layers (num_hidden_layers - 1)
    int64_t offW1, offb1, offW2, offW3, offb3, np;
};

static ParamLayout layout_of(const ppo_policy_s* p) {
    ParamLayout L;
    L.F = p->F; L.HID = p->HID; L.NT = p->HID / 32; L.FP = ((p->F + 31) / 32) * 32; L.NI = L.FP / 32; L.NL2 = p->L - 1;
    L.offW1 = 0; L.offb1 = (int64_t)p->HID * p->F; L.offW2 = L.offb1 + p->HID;
    L.offW3 = L.offW2 + (int64_t)L.NL2 * ((int64_t)p->HID * p->HID + p->HID);
    L.offb3 = L.offW3 + (int64_t)PPO_OUT * p->HID; L.np = L.offb3 + PPO_OUT;
    return L;
}

// ---------------------------------------------------------------- slab reduce
// One thread per slab element (coalesced reads across slabs); fixed summation order.
__device__ void loss_reduce_block(const double* __restrict__ terms, int64_t B, double inv_Bg, double entropy_weight,
                                  float* __restrict__ grad_tail);

struct PackPtrs {
    float* w1p; float* w2p; float* w2tp; float* b1p; float* b2p; float* w3p; float* b3;
    // bf16 compute mode (null in fp32 mode): A-operand fragments of v_mfma_f32_32x32x16_bf16 (ppo_policy_bf16.hip)
    uint16_t* w1b; uint16_t* w2b; uint16_t* w2tb; uint16_t* w3c; uint16_t* w3tb;
    // split-fp32 backward (null when the policy has none): W2 as three bf16 pieces (ppo_policy_bwd_x6.hip)
    uint16_t* w2x; uint16_t* w1x; uint16_t* w2fx;
};

// Adam fused into the slab reduction (single-rank training: no all-reduce between them): the thread that holds an element's
// gradient sum applies Flux's legacy Adam to that parameter and re-packs it -- one launch and one pass over the gradient less
struct AdamFuse { float* params; float* m; float* v; double eta, beta1, beta2, eps, bp1, bp2; float* hist2; int on; };
__device__ __forceinline__ void pack_one(const ParamLayout& L, const PackPtrs& P, int64_t i, float x);
__device__ __forceinline__ void adam_one(const AdamFuse& A, const ParamLayout& L, const PackPtrs& P, int64_t i, float g);

// Block = 64 consecutive slab elements x 4 slab groups (wave g sums slabs g, g+4, g+8, ... with 8 loads in flight);
// the four partial sums meet in LDS and are added in a fixed order.  4x the waves of a one-thread-per-element
// layout: the 87 MB slab walk needs the memory-level parallelism (341 blocks of one wave per SIMD did 3.3 TB/s).
// nwg_w slabs carry weight-gradient partials, nwg_s slabs the small-gradient tails (equal for the fused backward)
__global__ __launch_bounds__(256) void k_grad_reduce(const float* __restrict__ slabs, size_t slab_stride, int nwg_w, int nwg_s, ParamLayout L,
                                                     float* __restrict__ grad, const double* __restrict__ terms, int64_t B,
                                                     double inv_Bg, double entropy_weight, AdamFuse A, PackPtrs P) {
    if (blockIdx.x == gridDim.x - 1) {          // the extra last block reduces the per-sample loss terms
        loss_reduce_block(terms, B, inv_Bg, entropy_weight, grad + L.np);
        if (A.on && A.hist2 && threadIdx.x == 0) { A.hist2[0] = grad[L.np]; A.hist2[1] = grad[L.np + 1]; }   // per-batch loss history (k_adam's job otherwise)
        return;
    }
    __shared__ float part[4][64];
    const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const size_t e = (size_t)blockIdx.x * 64 + el;
    const size_t nW2l = (size_t)L.HID * L.HID, nW2 = (size_t)L.NL2 * nW2l, nW1 = (size_t)L.HID * L.FP;
    const size_t total = nW2 + nW1 + (size_t)L.HID * (1 + L.NL2) + (size_t)L.HID * 4 + 4;
    // fixed summation order: 8 interleaved partial sums per slab group, a fixed tree, then the 4 groups in order
    float ps[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nwg = (e < nW2 + nW1) ? nwg_w : nwg_s;
    // padding columns of the dW1 block (inputs F .. FP-1: a quarter of its last 32-column tile for F = 72) map to no parameter:
    // their slab values are never read (the split-fp32 backward does not write them either, except the ones column it keeps db1 in)
    bool dead = false;
    if (e >= nW2 && e < nW2 + nW1) {
        const size_t e1 = e - nW2;
        const int it = (int)((e1 >> 10) % L.NI);
        dead = 32 * it + (int)(e1 & 31) >= L.F;
    }
    if (e < total && !dead) {
        int g = grp;
        for (; g + 28 < nwg; g += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) ps[u] += slabs[(size_t)(g + 4 * u) * slab_stride + e];
        }
        for (int u = 0; g < nwg; g += 4, ++u) ps[u] += slabs[(size_t)g * slab_stride + e];
    }
    part[grp][el] = ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]));
    __syncthreads();
    if (grp != 0 || e >= total) return;
    const float s = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
    int64_t canon = -1;
    if (e < nW2) {
        const int lay = (int)(e / nW2l);
        const size_t el2 = e - (size_t)lay * nW2l;
        const int lane = (int)(el2 & 63), r = (int)((el2 >> 6) & 15);
        const int tile = (int)(el2 >> 10), kt = tile % L.NT, ft = tile / L.NT;
        const int f = dfeat(ft, r, lane >> 5), k = 32 * kt + (lane & 31);
        canon = L.offW2 + (int64_t)lay * ((int64_t)nW2l + L.HID) + f + (int64_t)L.HID * k;
    } else if (e < nW2 + nW1) {
        const size_t e1 = e - nW2;
        const int lane = (int)(e1 & 63), r = (int)((e1 >> 6) & 15);
        const int tile = (int)(e1 >> 10), it = tile % L.NI, ft = tile / L.NI;
        const int ko = dfeat(ft, r, lane >> 5), i = 32 * it + (lane & 31);
        if (i < L.F) canon = L.offW1 + ko + (int64_t)L.HID * i;
    } else {
        size_t e2 = e - nW2 - nW1;
        if (e2 < (size_t)L.HID) canon = L.offb1 + (int64_t)e2;
        else if ((e2 -= L.HID) < (size_t)L.HID * L.NL2)         // bias of hidden->hidden layer e2 / HID, behind its weights
            canon = L.offW2 + (int64_t)(e2 / L.HID) * ((int64_t)nW2l + L.HID) + (int64_t)nW2l + (int64_t)(e2 % L.HID);
        else if ((e2 -= (size_t)L.HID * L.NL2) < (size_t)L.HID * 4) canon = L.offW3 + (int64_t)(e2 & 3) + 4 * (int64_t)(e2 >> 2);
        else canon = L.offb3 + (int64_t)(e2 - (size_t)L.HID * 4);
    }
    if (canon >= 0) {
        grad[canon] = s;
        if (A.on) adam_one(A, L, P, canon, s);
    }
}

// loss scalars: grad[np] = -(sum min)/B_global, grad[np+1] = entropy_weight * -(sum H)/B_global.
// One block of 256 threads, fixed order: 8 interleaved fp64 partial sums per thread (loads in flight), a fixed tree
// per thread, then the LDS tree.
__device__ void loss_reduce_block(const double* __restrict__ terms, int64_t B, double inv_Bg, double entropy_weight,
                                  float* __restrict__ grad_tail) {
    __shared__ double s0[256], s1[256];
    const double2* t2 = reinterpret_cast<const double2*>(terms);
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t i = threadIdx.x;
    for (; i + 7 * 256 < B; i += 8 * 256) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { const double2 v = t2[i + u * 256]; a[u] += v.x; b[u] += v.y; }
    }
    for (int u = 0; i < B; i += 256, ++u) { const double2 v = t2[i]; a[u] += v.x; b[u] += v.y; }
    s0[threadIdx.x] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    s1[threadIdx.x] = ((b[0] + b[1]) + (b[2] + b[3])) + ((b[4] + b[5]) + (b[6] + b[7]));
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) { s0[threadIdx.x] += s0[threadIdx.x + off]; s1[threadIdx.x] += s1[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        grad_tail[0] = (float)(-(s0[0] * inv_Bg));
        grad_tail[1] = (float)(entropy_weight * (-(s1[0] * inv_Bg)));
    }
}

// ---------------------------------------------------------------- Adam + pack

__device__ __forceinline__ uint16_t to_bf16(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }   // RNE
// x = h + m + l: each piece the RNE bf16 of what the previous ones left (the differences are exact in fp32)
__device__ __forceinline__ void split3_bf16(float x, uint16_t& ph, uint16_t& pm, uint16_t& pl) {
    ph = to_bf16(x);
    const float r1 = x - __uint_as_float((uint32_t)ph << 16);
    pm = to_bf16(r1);
    const float r2 = r1 - __uint_as_float((uint32_t)pm << 16);
    pl = to_bf16(r2);
}
// k-slot of contraction index kk (0..31) inside a 32-wide tile when the other operand is a packed accumulator tile:
// k-step s = kk>>4, lane half hh and element jj such that 16s + 8(jj>>2) + 4hh + (jj&3) == kk
__device__ __forceinline__ void acc_kslot(int kk, int& s, int& hh, int& jj) {
    const int q = kk & 15;
    s = kk >> 4; hh = (q >> 2) & 1; jj = 4 * (q >> 3) + (q & 3);
}

__device__ __forceinline__ void pack_one(const ParamLayout& L, const PackPtrs& P, int64_t i, float x) {
    const int HID = L.HID, F = L.F, NT = L.NT;
    if (i < L.offb1) {                                   // W1[io][k]  (layer-1 A operand)
        const int io = (int)(i % HID), k = (int)(i / HID);
        const int hh = k / (F / 2), s = k % (F / 2);
        P.w1p[((size_t)((io >> 5) * (F / 8) + (s >> 2)) * 64 + (io & 31) + 32 * hh) * 4 + (s & 3)] = x;
        if (P.w1x) {                                     // split-fp32 train forward: natural k order, 5 k-steps, three pieces
            uint16_t ph, pm, pl;
            split3_bf16(x, ph, pm, pl);
            uint16_t* base = P.w1x + ((size_t)(io >> 5) * 5 + (k >> 4)) * 3 * 512 + ((size_t)(io & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7);
            base[0] = pl; base[512] = pm; base[1024] = ph;
        }
        if (P.w1b) {                                     // natural k order: k = 16*step + 8*half + element
            const int KS1 = (F + 15) / 16;
            P.w1b[((size_t)((io >> 5) * KS1 + (k >> 4)) * 64 + (io & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7)] = to_bf16(x);
        }
    } else if (i < L.offW2) {
        const int f = (int)(i - L.offb1), kk = f & 31, hh = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3);
        P.b1p[((f >> 5) * 2 + hh) * 16 + r] = x;
    } else if (i < L.offW3) {                            // hidden->hidden layer `lay`: W[f][k], then its bias
        const int64_t per = (int64_t)HID * HID + HID;
        const int lay = (int)((i - L.offW2) / per);
        const int64_t e = (i - L.offW2) - (int64_t)lay * per;
        if (e >= (int64_t)HID * HID) {                   // bias: accumulator-init order, like b1
            const int f = (int)(e - (int64_t)HID * HID), kk = f & 31, hh = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3);
            P.b2p[(size_t)lay * HID + ((f >> 5) * 2 + hh) * 16 + r] = x;
            return;
        }
        const size_t lo = (size_t)lay * HID * HID;       // the layers' fragment streams sit back to back
        const int f = (int)(e % HID), k = (int)(e / HID);
        {   // forward A operand: out f, contraction in the previous layer's accumulator-register order
            const int kk = k & 31, hh = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3);
            const int s = 16 * (k >> 5) + r;
            P.w2p[lo + ((size_t)((f >> 5) * (HID / 8) + (s >> 2)) * 64 + (f & 31) + 32 * hh) * 4 + (s & 3)] = x;
        }
        {   // backward A operand (W^T): out k, contraction slot (group g, component e, lane half hh) of feature f
            const bool zrow = PPO_BWD_Z2ROW_AT(HID);
            const int g = f >> 3, hh = zrow ? (f >> 2) & 1 : f & 1, e = zrow ? f & 3 : (f >> 1) & 3;   // f = 8g + 4hh + e  |  8g + 2e + hh
            P.w2tp[lo + ((size_t)((k >> 5) * (HID / 8) + g) * 64 + (k & 31) + 32 * hh) * 4 + e] = x;
        }
        if (P.w2x && lay == 0) {
            uint16_t ph, pm, pl;
            split3_bf16(x, ph, pm, pl);
            const int KS = HID / 16;
            int s, hh, jj;
            {   // backward: dH1[row, k] = sum_f dZ2[row, f] W[f][k]: B operand, column k, contraction f in the register order
                // of the packed dZ2 accumulator tile; [in-feature tile][k-step][piece lo, mid, hi][64 lanes][8]
                acc_kslot(f & 31, s, hh, jj);
                uint16_t* base = P.w2x + ((size_t)(k >> 5) * KS + 2 * (f >> 5) + s) * 3 * 512 + ((size_t)(k & 31) + 32 * hh) * 8 + jj;
                base[0] = pl; base[512] = pm; base[1024] = ph;
            }
            {   // train forward: H2^T[f, row] = sum_k W[f][k] H1[row, k]: A operand, row f, contraction k in the register order
                // of the packed H1 accumulator tile
                acc_kslot(k & 31, s, hh, jj);
                uint16_t* base = P.w2fx + ((size_t)(f >> 5) * KS + 2 * (k >> 5) + s) * 3 * 512 + ((size_t)(f & 31) + 32 * hh) * 8 + jj;
                base[0] = pl; base[512] = pm; base[1024] = ph;
            }
        }
        if (P.w2b) {
            const uint16_t xb = to_bf16(x);
            int s, hh, jj;
            acc_kslot(k & 31, s, hh, jj);                // forward: out f, contraction k against packed H1 tiles
            P.w2b[((size_t)((f >> 5) * (HID / 16) + 2 * (k >> 5) + s) * 64 + (f & 31) + 32 * hh) * 8 + jj] = xb;
            // backward (W2^T): out k, contraction f in natural order (k-step f>>4, lane half (f>>3)&1, element f&7): the B
            // operand is read straight from the row-major dZ2 image
            P.w2tb[((size_t)((k >> 5) * (HID / 16) + (f >> 4)) * 64 + (k & 31) + 32 * ((f >> 3) & 1)) * 8 + (f & 7)] = xb;
        }
    } else if (i < L.offb3) {                            // W3[oo][k]
        const int64_t e = i - L.offW3;
        const int oo = (int)(e & 3), k = (int)(e >> 2);
        const int kk = k & 31, hh = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3);
        P.w3p[((size_t)(hh * NT + (k >> 5)) * 16 + r) * 4 + oo] = x;
        if (P.w3c) {
            const uint16_t xb = to_bf16(x);
            int s, h2, jj;
            acc_kslot(k & 31, s, h2, jj);                // forward: rows 0..3 of the layer-3 A operand, [step][half][o][8]
            P.w3c[((size_t)((2 * (k >> 5) + s) * 2 + h2) * 4 + oo) * 8 + jj] = xb;
            P.w3tb[(size_t)k * 4 + oo] = xb;             // backward: W3^T rows [HID][4]
        }
    } else {
        P.b3[i - L.offb3] = x;
    }
}

// Flux legacy Adam on one parameter (the arithmetic of k_adam), then its packed copies
__device__ __forceinline__ void adam_one(const AdamFuse& A, const ParamLayout& L, const PackPtrs& P, int64_t i, float g) {
    const double gd = (double)g;
    const float mn = (float)(A.beta1 * (double)A.m[i] + (1.0 - A.beta1) * gd);
    const float vn = (float)(A.beta2 * (double)A.v[i] + ((1.0 - A.beta2) * gd) * gd);
    A.m[i] = mn; A.v[i] = vn;
    const double delta = (double)mn / (1.0 - A.bp1) / (sqrt((double)vn / (1.0 - A.bp2)) + A.eps) * A.eta;
    const float x = A.params[i] - (float)delta;
    A.params[i] = x;
    pack_one(L, P, i, x);
}

__global__ void k_pack_params(const float* __restrict__ params, ParamLayout L, PackPtrs P) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.np) return;
    pack_one(L, P, i, params[i]);
}

// Flux legacy Adam: element arithmetic in Float64 (Float64 hyper-parameters against Float32
// arrays), Float32 stores; bias-correction powers tracked in Float64 on the host.
__global__ void k_adam(float* __restrict__ params, const float* __restrict__ grad, float* __restrict__ m,
                       float* __restrict__ v, ParamLayout L, PackPtrs P, double eta, double beta1, double beta2,
                       double eps, double bp1, double bp2, float* __restrict__ hist2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (hist2 && i < 2) hist2[i] = grad[L.np + i];       // per-batch loss history (after any all-reduce)
    if (i >= L.np) return;
    const double gd = (double)grad[i];
    const float mn = (float)(beta1 * (double)m[i] + (1.0 - beta1) * gd);
    const float vn = (float)(beta2 * (double)v[i] + ((1.0 - beta2) * gd) * gd);
    m[i] = mn; v[i] = vn;
    const double delta = (double)mn / (1.0 - bp1) / (sqrt((double)vn / (1.0 - bp2)) + eps) * eta;
    const float x = params[i] - (float)delta;
    params[i] = x;
    pack_one(L, P, i, x);
}

// dataset order -> minibatch order: out[i] = index[perm_epoch(i)]
__global__ void k_feistel_index(const int32_t* __restrict__ index, int64_t len, uint64_t seed, uint32_t epoch,
                                int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    out[i] = index[feistel_perm(i, len, seed, epoch)];
}

static PackPtrs packs_of(ppo_policy_s* p) {
    PackPtrs P;
    P.w1p = p->w1p.p; P.w2p = p->w2p.p; P.w2tp = p->w2tp.p; P.b1p = p->b1p.p; P.b2p = p->b2p.p; P.w3p = p->w3p.p;
    P.b3 = p->b3.p;
    const bool b = (p->dtype == PPO_DTYPE_BF16);
    P.w1b = b ? p->w1b.p : nullptr; P.w2b = b ? p->w2b.p : nullptr; P.w2tb = b ? p->w2tb.p : nullptr;
    P.w3c = b ? p->w3c.p : nullptr; P.w3tb = b ? p->w3tb.p : nullptr;
    P.w2x = p->w2x.p; P.w1x = p->w1x.p; P.w2fx = p->w2fx.p;
    return P;
}

int32_t launch_pack_params(ppo_policy_s* p) {
    ParamLayout L = layout_of(p);
    hipLaunchKernelGGL(k_pack_params, dim3((unsigned)((L.np + 255) / 256)), dim3(256), 0, ppo_stream(), p->params.p, L,
                       packs_of(p));
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

// slab reduction + (one extra block) loss-term reduction in a single launch
int32_t launch_grad_reduce(ppo_policy_s* p, int64_t B, int64_t B_global, double entropy_weight, ppo_adam_s* fuse, float* hist2) {
    ParamLayout L = layout_of(p);
    const size_t total = (size_t)L.NL2 * L.HID * L.HID + (size_t)L.HID * L.FP + (size_t)L.HID * (1 + L.NL2) + (size_t)L.HID * 4 + 4;
    AdamFuse A = {};
    if (fuse) {
        A.params = p->params.p; A.m = fuse->m.p; A.v = fuse->v.p; A.eta = fuse->eta; A.beta1 = fuse->beta1; A.beta2 = fuse->beta2;
        A.eps = fuse->eps; A.bp1 = fuse->beta_pow[0]; A.bp2 = fuse->beta_pow[1]; A.hist2 = hist2; A.on = 1;
    }
    ProfScope ps(fuse ? "k_reduce_adam" : "k_grad_reduce");
    hipLaunchKernelGGL(k_grad_reduce, dim3((unsigned)((total + 63) / 64) + 1), dim3(256), 0, ppo_stream(), p->slabs.p,
                       slab_floats(p->F, p->HID, p->L), p->nwg_bwd, p->nwg_small ? p->nwg_small : p->nwg_bwd, L, p->grad.p, p->loss_terms.p, B, 1.0 / (double)B_global,
                       entropy_weight, A, packs_of(p));
    HIP_TRY(hipGetLastError());
    if (fuse) { fuse->beta_pow[0] *= fuse->beta1; fuse->beta_pow[1] *= fuse->beta2; }
    return PPO_OK;
}

int32_t launch_adam(ppo_adam_s* o, float* hist2) {
    ppo_policy_s* p = o->pol;
    ParamLayout L = layout_of(p);
    ProfScope ps("k_adam");
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((L.np + 255) / 256)), dim3(256), 0, ppo_stream(), p->params.p, p->grad.p,
                       o->m.p, o->v.p, L, packs_of(p), o->eta, o->beta1, o->beta2, o->eps, o->beta_pow[0],
                       o->beta_pow[1], hist2);
    HIP_TRY(hipGetLastError());
    o->beta_pow[0] *= o->beta1;
    o->beta_pow[1] *= o->beta2;
    return PPO_OK;
}

int32_t launch_feistel_index(const int32_t* index_dev, int64_t len, uint64_t seed, uint32_t epoch, int32_t* out_dev) {
    if (len <= 0) return PPO_OK;
    hipLaunchKernelGGL(k_feistel_index, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, ppo_stream(), index_dev, len,
                       seed, epoch, out_dev);
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

// batch_advantage plugin, PPO_ADV_RETURNS_NORMALISED (src/ProximalPolicyOptimization.jl:29 declares the plugin, the
// reference ships no implementation): adv = (R - mean(R)) / (std(R) + 1e-8) over the minibatch, population std,
// statistics in fp64.  One workgroup: two fixed-order block reductions, then the normalised values are scattered
// into a scratch column at their transition ids so the forward kernel reads them exactly like the returns column.
__global__ __launch_bounds__(1024) void k_adv_normalise(const float* __restrict__ returns, const int32_t* __restrict__ idx,
                                                        int64_t B, float* __restrict__ adv_col) {
    __shared__ double red[1024];
    __shared__ double stat[2];
    const int t = threadIdx.x;
    double s = 0.0;
    for (int64_t i = t; i < B; i += 1024) s += (double)returns[idx[i]];
    red[t] = s;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) { if (t < off) red[t] += red[t + off]; __syncthreads(); }
    if (t == 0) stat[0] = red[0] / (double)B;
    __syncthreads();
    const double mean = stat[0];
    s = 0.0;
    for (int64_t i = t; i < B; i += 1024) { const double d = (double)returns[idx[i]] - mean; s += d * d; }
    __syncthreads();
    red[t] = s;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) { if (t < off) red[t] += red[t + off]; __syncthreads(); }
    if (t == 0) stat[1] = 1.0 / (sqrt(red[0] / (double)B) + 1e-8);
    __syncthreads();
    const double inv = stat[1];
    for (int64_t i = t; i < B; i += 1024) { const int32_t k = idx[i]; adv_col[k] = (float)(((double)returns[k] - mean) * inv); }
}

int32_t launch_adv_normalise(const float* returns, const int32_t* idx_dev, int64_t B, float* adv_col) {
    hipLaunchKernelGGL(k_adv_normalise, dim3(1), dim3(1024), 0, ppo_stream(), returns, idx_dev, B, adv_col);
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}
