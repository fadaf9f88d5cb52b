// ppo_device.h -- device-side helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

// Philox4x32-10 (Salmon et al., SC'11).  Counter-based: results depend only on
// (seed; global env id, tick, stream) -- never on launch geometry or GPU count.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01_from_u32(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

// exp(x) for x <= 0 as an explicit fmaf sequence (Cephes expf polynomial); every operation is a
// single IEEE fp32 op so the CPU oracle reproduces it bit for bit.
__device__ __forceinline__ float exp_dev(float x) {
    if (!(x >= -87.0f)) return 0.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float z = r * r;
    float y = fmaf(p, z, r);
    y = y + 1.0f;
    int e = (int)n + 127;
    return y * __uint_as_float((uint32_t)e << 23);
}

// leakyrelu(x) = x > 0 ? x : 0.01x == max(x, 0.01x) for every finite x (slope < 1), bit for bit including +-0.
// v_mul + v_max instead of v_mul + v_cmp + v_cndmask: on gfx950 every vector instruction beside an fp32 MFMA costs
// MFMA time.  Inline asm because fmaxf() lowers to two v_max (an extra canonicalising one).
__device__ __forceinline__ float lrelu(float x) {
    const float y = 0.01f * x;
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

// the same on a whole accumulator tile: the 16 products as 8 v_pk_mul_f32 (IEEE multiplies, two per instruction), one
// v_max each -- 24 vector instructions instead of 32 (or 48 with fmaxf's canonicalising second v_max)
typedef float pk2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lrelu16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const pk2f x = {a[r], a[r + 1]};
        const pk2f y = x * pk2f{0.01f, 0.01f};
        float r0, r1;
        asm("v_max_f32 %0, %1, %2" : "=v"(r0) : "v"(x.x), "v"(y.x));
        asm("v_max_f32 %0, %1, %2" : "=v"(r1) : "v"(x.y), "v"(y.y));
        a[r] = r0; a[r + 1] = r1;
    }
}

// feature held by accumulator register r of 32x32 tile `tile` in lane-half hh (gfx950 C/D map)
__device__ __host__ __forceinline__ int dfeat(int tile, int r, int hh) {
    return 32 * tile + (r & 3) + 8 * (r >> 2) + 4 * hh;
}

__device__ __forceinline__ uint32_t feistel_round_fn(uint32_t x, uint32_t k) {
    x = x * 0x9E3779B1u + k;
    x ^= x >> 15; x *= 0x85EBCA77u;
    x ^= x >> 13; x *= 0xC2B2AE3Du;
    x ^= x >> 16;
    return x;
}

// bijection on [0,n): stands in for randperm (src/train.jl:93) when no permutation is supplied
__device__ __forceinline__ int64_t feistel_perm(int64_t i, int64_t n, uint64_t seed, uint32_t epoch) {
    if (n <= 1) return 0;
    int bits = 2;
    while (((int64_t)1 << bits) < n) bits += 2;
    int hb = bits / 2;
    uint32_t hmask = (uint32_t)(((uint64_t)1 << hb) - 1);
    uint64_t x = (uint64_t)i;
    do {
        uint32_t L = (uint32_t)(x >> hb) & hmask, R = (uint32_t)x & hmask;
#pragma unroll
        for (uint32_t r = 0; r < 6; ++r) {
            uint32_t k = (uint32_t)(seed >> ((r & 1) ? 32 : 0)) ^ (epoch * 0x9E3779B9u) ^ (r * 0x7F4A7C15u);
            uint32_t nL = R;
            uint32_t nR = L ^ (feistel_round_fn(R, k) & hmask);
            L = nL; R = nR;
        }
        x = ((uint64_t)L << hb) | R;
    } while ((int64_t)x >= n);
    return (int64_t)x;
}

// synthetic env template: vertex id in [0,4Q) or -1 (missing)
__device__ __host__ __forceinline__ int env_template(int Q, int h, int t) {
    int V = 4 * Q, q = h >> 2, ed = h & 3;
    if (t < 4) return 4 * q + ((ed + t) & 3);
    int c = (h * 5 + t * 7 + 3) % (V + 6);
    return c >= V ? -1 : c;
}
