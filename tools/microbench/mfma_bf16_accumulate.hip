// How does v_mfma_f32_32x32x16_bf16 add its 16 products to the fp32 accumulator?  (round 3: the split-fp32 kernels
// accumulate piece products 2^-8 .. 2^-16 of the accumulator's size.)
//   test 1: C = 1.0, 16 products of 1.5 * 2^-24 (0.75 ulp of C each; exact sum = 3 ulp): exact-sum-then-round gives 1 + 3 ulp,
//           per-product truncation at the accumulator's ulp gives 1.0, per-product rounding gives 1 + 16 ulp
//   test 2: same with C = 1.0 and products 0.25 ulp each (sum = 4 * ... = 4 ulp * 1 = 16 * 0.25 = 4 ulp)
//   test 3: random chain of 16 MFMAs (k = 256) against float64: mean signed error and rms in ulps of the result, with the
//           products ~2^-10 of the accumulator (a "mid-piece" MFMA) -- a biased mean says truncation, not rounding
// build: hipcc --offload-arch=gfx950 -O2 mfma_bf16_accumulate.hip -o mfma_bf16_accumulate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_one(const uint16_t* a, const uint16_t* b, const float* c, float* d, int chain) {
    const int lane = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c[r * 64 + lane];
    for (int s = 0; s < chain; ++s) {
        bf16x8 av, bv;
        for (int e = 0; e < 8; ++e) {
            av[e] = __builtin_bit_cast(__bf16, a[(s * 64 + lane) * 8 + e]);
            bv[e] = __builtin_bit_cast(__bf16, b[(s * 64 + lane) * 8 + e]);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) d[r * 64 + lane] = acc[r];
}
static uint16_t bf(float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)(u >> 16); }   // exact inputs only
static float fb(uint16_t h) { uint32_t u = (uint32_t)h << 16; float x; memcpy(&x, &u, 4); return x; }

int main() {
    const int CH = 16;
    std::vector<uint16_t> a(CH * 64 * 8), b(CH * 64 * 8);
    std::vector<float> c(1024), d(1024);
    uint16_t *da, *db; float *dc, *dd;
    hipMalloc(&da, a.size() * 2); hipMalloc(&db, b.size() * 2); hipMalloc(&dc, 4096); hipMalloc(&dd, 4096);
    auto run = [&](int chain) {
        hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dc, c.data(), 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, da, db, dc, dd, chain);
        hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost);
    };
    for (float prod_ulps : {0.75f, 0.25f, 0.5f, 1.25f}) {
        for (auto& x : a) x = bf(1.0f);
        for (auto& x : b) x = bf(0.0f);
        // A all ones; B[k][n] = prod_ulps * 2^-23 (exact in bf16 for these values)
        for (size_t i = 0; i < 64 * 8; ++i) b[i] = bf(prod_ulps * ldexpf(1.0f, -23));
        for (auto& x : c) x = 1.0f;
        run(1);
        printf("C = 1, 16 products of %.2f ulp: D - 1 = %.2f ulp (exact sum %.2f)\n", prod_ulps, (d[0] - 1.0f) * ldexpf(1.0f, 23), 16 * prod_ulps);
    }
    {   // one big product + 15 small ones: is the alignment to the largest ADDEND (product) or to C?
        for (auto& x : a) x = bf(1.0f);
        for (size_t i = 0; i < 64 * 8; ++i) b[i] = bf(0.75f * ldexpf(1.0f, -23));
        for (auto& x : c) x = 0.0f;
        // lane layout: B[k = 8h + e][n = lane & 31]; make k = 0 the big one for every column
        for (int lane = 0; lane < 32; ++lane) b[lane * 8 + 0] = bf(1.0f);
        run(1);
        printf("C = 0, one product 1.0 + 15 of 0.75 ulp(1): D - 1 = %.2f ulp (exact 11.25)\n", (d[0] - 1.0f) * ldexpf(1.0f, 23));
    }
    // random chains
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (float scale : {1.0f, 1.0f / 256, 1.0f / 65536}) {
        for (auto& x : a) x = bf(fb(bf(nd(rng))));
        for (auto& x : b) x = bf(fb(bf(nd(rng) * scale)));
        for (auto& x : c) x = 64.0f * nd(rng);
        run(CH);
        double sum_e = 0, sum_e2 = 0; int n = 0;
        for (int r = 0; r < 16; ++r) for (int lane = 0; lane < 64; ++lane) {
            const int col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            double ref = c[r * 64 + lane];
            for (int s = 0; s < CH; ++s) for (int k = 0; k < 16; ++k) {
                const int h = k >> 3, e = k & 7;
                ref += (double)fb(a[(s * 64 + row + 32 * h) * 8 + e]) * (double)fb(b[(s * 64 + col + 32 * h) * 8 + e]);
            }
            const double ulp = ldexp(1.0, ilogb(fabs(ref)) - 23);
            const double err = ((double)d[r * 64 + lane] - ref) / ulp;
            sum_e += err * (ref >= 0 ? 1 : -1); sum_e2 += err * err; ++n;      // signed towards larger magnitude
        }
        printf("chain of %d MFMAs, |C| ~ 64, products ~ %g: mean error %+.3f ulp (sign: away from zero), rms %.3f ulp\n", CH, scale, sum_e / n, sqrt(sum_e2 / n));
    }
    return 0;
}
