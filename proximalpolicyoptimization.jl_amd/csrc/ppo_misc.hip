// ppo_misc.hip -- standalone parity entry points: Philox4x32-10 and the forward-only loss
// (ppo_loss_with_entropy, src/train.jl:21-26,35-46) on caller-supplied probabilities.
#include "ppo_internal.h"
#include "ppo_device.h"

__global__ void k_philox(const uint32_t* __restrict__ ctr, uint32_t k0, uint32_t k1, int64_t n, uint32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[4];
    philox4x32_10(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], k0, k1, w);
    out[4 * i] = w[0]; out[4 * i + 1] = w[1]; out[4 * i + 2] = w[2]; out[4 * i + 3] = w[3];
}

// one wave per sample b: terms[b] = (min(gain, clip), H_b)
__global__ void k_loss_terms(const float* __restrict__ probs, const int64_t* __restrict__ lin1,
                             const float* __restrict__ p_old, const float* __restrict__ adv, int64_t B, int64_t A,
                             double eps, double* __restrict__ terms) {
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const float sA = 1e-8f / (float)A;
    float h = 0.0f;
    for (int64_t a = lane; a < A; a += 64) { const float sp = probs[b * A + a] + sA; h += sp * logf(sp); }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) h += __shfl_xor(h, off);
    if (lane == 0) {
        const float ps = probs[lin1[b] - 1];
        const float gain = ps / p_old[b] * adv[b];
        const double clip = adv[b] >= 0.0f ? (1.0 + eps) * (double)adv[b] : (1.0 - eps) * (double)adv[b];
        terms[2 * b] = (double)gain < clip ? (double)gain : clip;
        terms[2 * b + 1] = (double)(-h);
    }
}

__global__ void k_sum2(const double* __restrict__ terms, int64_t B, double* __restrict__ out2) {
    __shared__ double s0[256], s1[256];
    double a = 0.0, b = 0.0;
    for (int64_t i = threadIdx.x; i < B; i += 256) { a += terms[2 * i]; b += terms[2 * i + 1]; }
    s0[threadIdx.x] = a; s1[threadIdx.x] = b;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) { s0[threadIdx.x] += s0[threadIdx.x + off]; s1[threadIdx.x] += s1[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out2[0] = s0[0]; out2[1] = s1[0]; }
}

extern "C" int32_t ppo_philox4x32_10(const uint32_t* ctr4, const uint32_t* key2, int64_t n, uint32_t* out4) {
    int32_t nd = 0;
    PPO_TRY(ppo_device_count(&nd));
    PPO_TRY(ppo_device_synchronize());
    ARG_CHECK(ctr4 && key2 && out4 && n >= 0, "philox: bad argument");
    if (n == 0) return PPO_OK;
    DevBuf<uint32_t> c, o;
    PPO_TRY(c.alloc((size_t)4 * n)); PPO_TRY(o.alloc((size_t)4 * n));
    HIP_TRY(hipMemcpyAsync(c.p, ctr4, (size_t)16 * n, hipMemcpyHostToDevice, ppo_stream()));
    hipLaunchKernelGGL(k_philox, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ppo_stream(), c.p, key2[0], key2[1], n, o.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out4, o.p, (size_t)16 * n, hipMemcpyDeviceToHost, ppo_stream()));
    HIP_TRY(hipStreamSynchronize(ppo_stream()));
    return PPO_OK;
}

extern "C" int32_t ppo_loss_with_entropy(const float* probs, const int64_t* lin_idx1, const float* p_old,
                                         const float* adv, int64_t B, int64_t A, double epsilon, double* ppoloss,
                                         double* entropyloss) {
    PPO_TRY(ppo_device_synchronize());
    ARG_CHECK(probs && lin_idx1 && p_old && adv && B >= 1 && A >= 1, "ppo_loss_with_entropy: bad argument");
    for (int64_t b = 0; b < B; ++b) ARG_CHECK(lin_idx1[b] >= 1 && lin_idx1[b] <= A * B, "linear action index out of range");
    DevBuf<float> p, po, ad; DevBuf<int64_t> li; DevBuf<double> terms, out;
    PPO_TRY(p.alloc((size_t)A * B)); PPO_TRY(po.alloc(B)); PPO_TRY(ad.alloc(B)); PPO_TRY(li.alloc(B));
    PPO_TRY(terms.alloc((size_t)2 * B)); PPO_TRY(out.alloc(2));
    hipStream_t st = ppo_stream();
    HIP_TRY(hipMemcpyAsync(p.p, probs, sizeof(float) * A * B, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(po.p, p_old, sizeof(float) * B, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ad.p, adv, sizeof(float) * B, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(li.p, lin_idx1, sizeof(int64_t) * B, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_loss_terms, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, p.p, li.p, po.p, ad.p, B, A, epsilon, terms.p);
    hipLaunchKernelGGL(k_sum2, dim3(1), dim3(256), 0, st, terms.p, B, out.p);
    HIP_TRY(hipGetLastError());
    double h[2];
    HIP_TRY(hipMemcpyAsync(h, out.p, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (ppoloss) *ppoloss = -(h[0] / (double)B);
    if (entropyloss) *entropyloss = -(h[1] / (double)B);
    return PPO_OK;
}
