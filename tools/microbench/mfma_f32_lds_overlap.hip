// Micro-benchmark: what does an LDS read cost between two v_mfma_f32_32x32x2_f32 on gfx950?
// Same shape as mfma_f32_valu_overlap.hip, with NL independent ds_read_b32 (KIND=0) or ds_read_b128 (KIND=1) per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef KIND
#define KIND 0
#endif
template <int NL>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i;
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
    float s = 0.f;
    const unsigned base = (threadIdx.x & 63) * (KIND ? 16 : 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                if (KIND) { float4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(i * 1024)); s += v.x; }
                else { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(i * 256)); s += v; }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NL>
static void run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NL>, dim3(256), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NL>, dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s NL=%2d  %.3f ms  %.1f ns per MFMA per SIMD\n", KIND ? "ds_read_b128" : "ds_read_b32 ", NL, ms, ms * 1e6 / ((double)iters * 8));
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 256 * 4);
    const int iters = 20000;
    run<0>(d, iters); run<1>(d, iters); run<2>(d, iters); run<4>(d, iters); run<8>(d, iters);
    return 0;
}
