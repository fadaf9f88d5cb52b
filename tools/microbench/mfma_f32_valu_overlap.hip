// Micro-benchmark: does VALU work hide under v_mfma_f32_32x32x2_f32 on gfx950?
// One wave per SIMD (or two with -DWPS=2), a dependent MFMA chain, NV independent v_fma_f32 per MFMA interleaved
// in the instruction stream.  If the f32 MFMA had its own datapath, time would stay flat until the VALU work fills
// the 64-cycle issue interval; if it runs on the packed-f32 vector ALU, time grows by ~4 cycles per VALU op.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu tools/microbench/mfma_f32_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef WPS
#define WPS 1
#endif
template <int NV>
__global__ __launch_bounds__(256 * WPS) void k(float* out, int iters) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i % 16]) : "v"(b), "v"(a));
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV>
static void run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NV>, dim3(256), dim3(256 * WPS), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NV>, dim3(256), dim3(256 * WPS), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)iters * 8 * WPS;          // per SIMD
    printf("WPS=%d NV=%2d  %.3f ms  %.1f ns per MFMA per SIMD  (%.1f TFLOP/s chip)\n", WPS, NV, ms, ms * 1e6 / mfma,
           mfma * 1024 * 4096.0 / (ms * 1e-3) / 1e12);
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;
    run<0>(d, iters); run<2>(d, iters); run<4>(d, iters); run<8>(d, iters); run<12>(d, iters); run<16>(d, iters); run<24>(d, iters);
    return 0;
}
