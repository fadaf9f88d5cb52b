// ppo_returns.hip -- K6: discounted-return / GAE reverse scans (src/collect_rollouts.jl:26-42).
//
// Hot layout is time-major [T,N]: lanes = envs (coalesced rows), the scan runs along T.
// The recurrence is strictly sequential per column and the reference's running value is a
// Float64 (SURVEY.md hard part 3), so the arithmetic is NOT re-associated: a workgroup owns 64
// columns, its 4 waves each load a 32-row time chunk into registers in parallel (the HBM-bound
// part), then the chunks are chained latest-first through a 64-entry fp64 carry staged in LDS.
// Results are bit-identical to the sequential reference loop.
//
// Roofline: HBM-bound, 9 B per transition (r f32 + done u8 in, return f32 out).
#include "ppo_internal.h"
#include "ppo_device.h"
#include <cstdlib>

#define RT_CH 32          // rows per wave per pass
#define RT_WAVES 4
#define RT_COLS 64

template <int F32MODE>
__global__ __launch_bounds__(256) void k_returns_tn(const float* __restrict__ r, const uint8_t* __restrict__ done,
                                                    float* __restrict__ out, int64_t T, int64_t N, double discount) {
    __shared__ double sCarry[RT_COLS];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * RT_COLS + lane;
    const bool col_ok = n < N;
    const float gf = (float)discount;
    const int64_t npass = (T + RT_CH * RT_WAVES - 1) / (RT_CH * RT_WAVES);
    if (w == RT_WAVES - 1) sCarry[lane] = 0.0;
    __syncthreads();
    for (int64_t p = 0; p < npass; ++p) {
        const int64_t base = T - (int64_t)RT_CH * RT_WAVES * (p + 1) + (int64_t)RT_CH * w;   // may be < 0
        float rr[RT_CH];
        uint8_t dd[RT_CH];
#pragma unroll
        for (int i = 0; i < RT_CH; ++i) {
            const int64_t t = base + i;
            const bool ok = col_ok && t >= 0;
            rr[i] = ok ? r[t * N + n] : 0.0f;
            dd[i] = ok ? done[t * N + n] : (uint8_t)0;
        }
        // chain the four chunks latest-first; each wave runs the exact sequential recurrence
        for (int ww = RT_WAVES - 1; ww >= 0; --ww) {
            if (w == ww) {
                if (F32MODE) {
                    float v = (float)sCarry[lane];
#pragma unroll
                    for (int i = RT_CH - 1; i >= 0; --i) {
                        const int64_t t = base + i;
                        if (t >= 0) {
                            if (dd[i]) v = 0.0f;
                            float gv = gf * v;
                            v = rr[i] + gv;
                            if (col_ok) out[t * N + n] = v;
                        }
                    }
                    sCarry[lane] = (double)v;
                } else {
                    double v = sCarry[lane];
#pragma unroll
                    for (int i = RT_CH - 1; i >= 0; --i) {
                        const int64_t t = base + i;
                        if (t >= 0) {
                            if (dd[i]) v = 0.0;
                            double gv = discount * v;
                            v = (double)rr[i] + gv;
                            if (col_ok) out[t * N + n] = (float)v;
                        }
                    }
                    sCarry[lane] = v;
                }
            }
            __syncthreads();
        }
    }
}


// Wide variant for large N (N % 4 == 0), LDS-staged.  A workgroup owns 256 adjacent columns.  Per pass of
// RW_ROWS time rows: (1) every wave loads its share of the rows with full 1 KiB float4 row accesses (256 B for
// the done flags) and parks them in an LDS tile [rows][COLS]; (2) after a barrier wave w scans columns
// [64w, 64w+64) of the tile -- one column per lane, the exact sequential fp64 recurrence, carry kept in a
// register across passes, NO cross-wave dependency -- writing the returns back into the tile; (3) after a second
// barrier the tile leaves with wide stores.  HBM sees only wide, fully used accesses; the scan reads the tile
// row by row (conflict-free).  Two tiles are ping-ponged so the loads of pass p+1 are in flight during pass p.
// ROWS = time rows per pass, NBUF = LDS tiles (2: ping-pong over the passes)
// COLS = columns per workgroup (256: 4 waves, one row per wide wave access; 128: 2 waves, two rows per access -- twice
// the workgroups, used when 256-column workgroups would leave the chip with one small workgroup per CU)
template <int F32MODE, int COLS, int RW_ROWS, int NBUF>
__global__ __launch_bounds__(COLS) void k_returns_tn_x4(const float* __restrict__ r, const uint8_t* __restrict__ done,
                                                        float* __restrict__ out, int64_t T, int64_t N, double discount) {
    constexpr int W = COLS / 64;                                // waves; wave w scans columns [64w, 64w+64)
    constexpr int RS = 256 / COLS;                              // rows covered by one wide wave access (64 lanes x 4 columns)
    constexpr int RPW = RW_ROWS / (W * RS);                     // wide accesses per wave per pass
    __shared__ __attribute__((aligned(16))) float sR[NBUF][RW_ROWS][COLS];
    __shared__ __attribute__((aligned(16))) uint8_t sD[NBUF][RW_ROWS][COLS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * COLS;              // first column of the workgroup
    const int lr = lane / (COLS / 4), lc = (lane % (COLS / 4)) * 4;   // row within the access, first of this lane's 4 columns
    const int64_t cl = c0 + lc;
    const bool wide_ok = cl < N;                                // N % 4 == 0
    const float gf = (float)discount;
    const int64_t npass = (T + RW_ROWS - 1) / RW_ROWS;
    double v = 0.0;                                             // running value of this lane's scan column (fp64 or exact fp32 value)

    float4 rv[RPW];
    uint32_t dv[RPW];
    auto tile_row = [&](int i) { return (w * RPW + i) * RS + lr; };
    auto load_regs = [&](int64_t p) {                          // rows [T - RW_ROWS*(p+1), +RW_ROWS)
        const int64_t base = T - (int64_t)RW_ROWS * (p + 1);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int64_t t = base + tile_row(i);
            rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            dv[i] = 0u;
            if (wide_ok && t >= 0) {
                rv[i] = *reinterpret_cast<const float4*>(r + t * N + cl);
                dv[i] = *reinterpret_cast<const uint32_t*>(done + t * N + cl);
            }
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            *reinterpret_cast<float4*>(&sR[buf][tile_row(i)][lc]) = rv[i];
            *reinterpret_cast<uint32_t*>(&sD[buf][tile_row(i)][lc]) = dv[i];
        }
    };

    load_regs(0);
    park(0);
    __syncthreads();
    for (int64_t p = 0; p < npass; ++p) {
        const int buf = (int)(p & (NBUF - 1));
        const int64_t base = T - (int64_t)RW_ROWS * (p + 1);
        if (NBUF > 1 && p + 1 < npass) load_regs(p + 1);       // next pass is in flight while this one is scanned
        // ---- scan: latest row first, exact sequential recurrence, one column per lane
#pragma unroll 8
        for (int row = RW_ROWS - 1; row >= 0; --row) {
            if (base + row < 0) break;
            const float x = sR[buf][row][64 * w + lane];
            const bool dn = sD[buf][row][64 * w + lane] != 0;
            if (F32MODE) {
                float vf = dn ? 0.0f : (float)v;
                const float gv = gf * vf;
                vf = x + gv;
                v = (double)vf;
                sR[buf][row][64 * w + lane] = vf;
            } else {
                double vd = dn ? 0.0 : v;
                const double gv = discount * vd;
                vd = (double)x + gv;
                v = vd;
                sR[buf][row][64 * w + lane] = (float)vd;
            }
        }
        __syncthreads();
        // ---- wide stores of the finished tile, then park the next pass in the other tile
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int row = tile_row(i);
            const int64_t t = base + row;
            if (wide_ok && t >= 0) *reinterpret_cast<float4*>(out + t * N + cl) = *reinterpret_cast<const float4*>(&sR[buf][row][lc]);
        }
        if (NBUF > 1 && p + 1 < npass) park(buf ^ 1);
        __syncthreads();
    }
}

// Flat concatenated-episodes layout (the reference's own): each episode segment is scanned by
// the thread that owns its last element, so the arithmetic is again the sequential recurrence.
template <int F32MODE>
__global__ void k_returns_flat(const float* __restrict__ r, const uint8_t* __restrict__ term, float* __restrict__ out,
                               int64_t n, double discount) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!(term[i] || i == n - 1)) return;
    const float gf = (float)discount;
    int64_t j = i;
    if (F32MODE) {
        float v = 0.0f;
        do { float gv = gf * v; v = r[j] + gv; out[j] = v; --j; } while (j >= 0 && !term[j]);
    } else {
        double v = 0.0;
        do { double gv = discount * v; v = (double)r[j] + gv; out[j] = (float)v; --j; } while (j >= 0 && !term[j]);
    }
}

// GAE(gamma, lambda) extension: the `batch_advantage` plugin the reference declares (src/ProximalPolicyOptimization.jl:29)
// and never implements; lambda = 1, V = 0 is exactly compute_returns (src/collect_rollouts.jl:26-42).  Recurrence per
// column, latest row first, all in fp64 like the reference's running value:
//     nd = !done[t];  delta = (r[t] + gamma * (V[t+1] * nd)) - V[t];  A = delta + ((gamma * lambda) * nd) * A
//     adv[t] = float(A);  ret[t] = float(A + V[t])
// HBM-bound: 17 B per transition (r f32 + done u8 + V f32 in, adv f32 + ret f32 out; V[t+1] is the V[t] of the row
// scanned just before, carried in a register, so every value is read once).

// small-N form: one lane per column (scalar, coalesced rows)
__global__ void k_gae_tn(const float* __restrict__ r, const uint8_t* __restrict__ done, const float* __restrict__ val,
                         float* __restrict__ adv, float* __restrict__ ret, int64_t T, int64_t N, double gamma,
                         double lambda) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double a = 0.0;
    const double gl = gamma * lambda;
    double vn = (double)val[T * N + n];
    for (int64_t t = T - 1; t >= 0; --t) {
        const double nd = done[t * N + n] ? 0.0 : 1.0;
        const double v = (double)val[t * N + n];
        const double vnext = vn * nd;
        const double gvn = gamma * vnext;
        const double delta = ((double)r[t * N + n] + gvn) - v;
        const double carry = (gl * nd) * a;
        a = delta + carry;
        adv[t * N + n] = (float)a;
        ret[t * N + n] = (float)(a + v);
        vn = v;
    }
}

// Wide form for large N (N % 4 == 0), the LDS-staged structure of k_returns_tn_x4: a workgroup owns COLS adjacent
// columns; per pass of RW_ROWS time rows every wave loads its share of the three input tiles with full-width accesses
// (16 B per lane for r and V, 4 B for the done flags) into LDS; after a barrier wave w scans columns [64w, 64w+64)
// -- one column per lane, the exact sequential fp64 recurrence, A and V[t+1] carried in registers across passes, no
// cross-wave dependency -- writing adv over the r tile and ret over the V tile; after a second barrier both tiles
// leave with wide stores.  Two tile sets are ping-ponged so the loads of pass p+1 are in flight during the scan of p.
template <int COLS, int RW_ROWS>
__global__ __launch_bounds__(COLS) void k_gae_tn_x4(const float* __restrict__ r, const uint8_t* __restrict__ done,
                                                    const float* __restrict__ val, float* __restrict__ adv,
                                                    float* __restrict__ ret, int64_t T, int64_t N, double gamma,
                                                    double lambda) {
    constexpr int W = COLS / 64, RS = 256 / COLS, RPW = RW_ROWS / (W * RS);
    static_assert(RPW >= 1 && RPW * W * RS == RW_ROWS, "tile shape");
    __shared__ __attribute__((aligned(16))) float sR[2][RW_ROWS][COLS];
    __shared__ __attribute__((aligned(16))) float sV[2][RW_ROWS][COLS];
    __shared__ __attribute__((aligned(16))) uint8_t sD[2][RW_ROWS][COLS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * COLS;
    const int lr = lane / (COLS / 4), lc = (lane % (COLS / 4)) * 4;
    const int64_t cl = c0 + lc;
    const bool wide_ok = cl < N;                                // N % 4 == 0
    const int64_t sc = c0 + 64 * w + lane;                      // this lane's scan column
    const int64_t npass = (T + RW_ROWS - 1) / RW_ROWS;
    const double gl = gamma * lambda;
    double a = 0.0;
    double vn = (sc < N) ? (double)val[T * N + sc] : 0.0;       // V[T]: bootstrap value behind the last row

    float4 rv[RPW], vv[RPW];
    uint32_t dv[RPW];
    auto tile_row = [&](int i) { return (w * RPW + i) * RS + lr; };
    auto load_regs = [&](int64_t p) {
        const int64_t base = T - (int64_t)RW_ROWS * (p + 1);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int64_t t = base + tile_row(i);
            rv[i] = make_float4(0.f, 0.f, 0.f, 0.f); vv[i] = rv[i]; dv[i] = 0u;
            if (wide_ok && t >= 0) {
                rv[i] = *reinterpret_cast<const float4*>(r + t * N + cl);
                vv[i] = *reinterpret_cast<const float4*>(val + t * N + cl);
                dv[i] = *reinterpret_cast<const uint32_t*>(done + t * N + cl);
            }
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            *reinterpret_cast<float4*>(&sR[buf][tile_row(i)][lc]) = rv[i];
            *reinterpret_cast<float4*>(&sV[buf][tile_row(i)][lc]) = vv[i];
            *reinterpret_cast<uint32_t*>(&sD[buf][tile_row(i)][lc]) = dv[i];
        }
    };
    load_regs(0);
    park(0);
    __syncthreads();
    for (int64_t p = 0; p < npass; ++p) {
        const int buf = (int)(p & 1);
        const int64_t base = T - (int64_t)RW_ROWS * (p + 1);
        if (p + 1 < npass) load_regs(p + 1);
#pragma unroll 8
        for (int row = RW_ROWS - 1; row >= 0; --row) {
            if (base + row < 0) break;
            const double x = (double)sR[buf][row][64 * w + lane];
            const double v = (double)sV[buf][row][64 * w + lane];
            const double nd = sD[buf][row][64 * w + lane] ? 0.0 : 1.0;
            const double vnext = vn * nd;
            const double gvn = gamma * vnext;
            const double delta = (x + gvn) - v;
            const double carry = (gl * nd) * a;
            a = delta + carry;
            sR[buf][row][64 * w + lane] = (float)a;
            sV[buf][row][64 * w + lane] = (float)(a + v);
            vn = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int row = tile_row(i);
            const int64_t t = base + row;
            if (wide_ok && t >= 0) {
                *reinterpret_cast<float4*>(adv + t * N + cl) = *reinterpret_cast<const float4*>(&sR[buf][row][lc]);
                *reinterpret_cast<float4*>(ret + t * N + cl) = *reinterpret_cast<const float4*>(&sV[buf][row][lc]);
            }
        }
        if (p + 1 < npass) park(buf ^ 1);
        __syncthreads();
    }
}

int32_t launch_returns_tn(const float* r, const uint8_t* done, float* out, int64_t T, int64_t N, double discount,
                          int f32mode) {
    if (T <= 0 || N <= 0) return PPO_OK;
    ProfScope ps("k_returns_tn");
    if (N % 4 == 0 && N >= 16384) {                  // wide columns: 1 KiB per wave access
        // Tile shape by column count, from an A/B of the shapes on one box (PPO_RETURNS_VARIANT, tools/returns_ab.py,
        // gpurun_out/r3d; us at 16384 / 65536 / 262144 columns x 128 rows):
        //   128 cols x 32 rows (round 2)  13.8 / 19.1 / 61.3      64 x 16   14.9 / 17.8 / 57.6
        //   128 x 16                      14.9 / 18.7 / 50.8     256 x 16   15.9 / 17.7 / 52.5
        //   128 x 8                       16.4 / 19.6 / 51.2      64 x 8    15.5 / 18.3 / 55.8     256 x 32  16.2 / 20.2 / 61.6
        // 16-row passes beat 32-row ones once the launch is more than one round of workgroups (20 KB of LDS: eight
        // workgroups per CU, finer interleaving of load / scan / store phases); measured and dropped: a single 128-row
        // pass (22.4 us at 65536), two passes of load lookahead instead of one (+1 us at every size: the second register
        // set costs more in the scan loop than the deeper queue gains -- the workgroups of a CU already overlap each other).
        static const int variant = [] { const char* v = getenv("PPO_RETURNS_VARIANT"); return v ? atoi(v) : 0; }();
#define RV(C, R) { dim3 g((unsigned)((N + (C) - 1) / (C)));                                                                  \
            if (f32mode) hipLaunchKernelGGL((k_returns_tn_x4<1, C, R, 2>), g, dim3(C), 0, ppo_stream(), r, done, out, T, N, discount); \
            else hipLaunchKernelGGL((k_returns_tn_x4<0, C, R, 2>), g, dim3(C), 0, ppo_stream(), r, done, out, T, N, discount); \
            HIP_TRY(hipGetLastError()); return PPO_OK; }
        if (variant == 12832) RV(128, 32)
        if (variant == 12816) RV(128, 16)
        if (variant == 6416) RV(64, 16)
        if (variant == 25616) RV(256, 16)
        if (N >= 131072) RV(128, 16)
        if (N >= 32768) RV(64, 16)
        RV(128, 32)
#undef RV
    }
    dim3 grid((unsigned)((N + RT_COLS - 1) / RT_COLS));
    if (f32mode) hipLaunchKernelGGL(k_returns_tn<1>, grid, dim3(256), 0, ppo_stream(), r, done, out, T, N, discount);
    else hipLaunchKernelGGL(k_returns_tn<0>, grid, dim3(256), 0, ppo_stream(), r, done, out, T, N, discount);
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

int32_t launch_returns_flat(const float* r, const uint8_t* term, float* out, int64_t n, double discount, int f32mode) {
    if (n <= 0) return PPO_OK;
    dim3 grid((unsigned)((n + 255) / 256));
    if (f32mode) hipLaunchKernelGGL(k_returns_flat<1>, grid, dim3(256), 0, ppo_stream(), r, term, out, n, discount);
    else hipLaunchKernelGGL(k_returns_flat<0>, grid, dim3(256), 0, ppo_stream(), r, term, out, n, discount);
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}

int32_t launch_gae_tn(const float* r, const uint8_t* done, const float* values, float* adv, float* ret, int64_t T,
                      int64_t N, double gamma, double lambda) {
    if (T <= 0 || N <= 0) return PPO_OK;
    ProfScope ps("k_gae_tn");
    if (N % 4 == 0 && N >= 16384) {                  // wide columns, LDS-staged (73.7 KB of LDS: two workgroups per CU)
        const int cols = (N / 256 >= 1024) ? 256 : 128;
        dim3 gridw((unsigned)((N + cols - 1) / cols));
        if (cols == 256) hipLaunchKernelGGL((k_gae_tn_x4<256, 16>), gridw, dim3(256), 0, ppo_stream(), r, done, values, adv, ret, T, N, gamma, lambda);
        else hipLaunchKernelGGL((k_gae_tn_x4<128, 16>), gridw, dim3(128), 0, ppo_stream(), r, done, values, adv, ret, T, N, gamma, lambda);
        HIP_TRY(hipGetLastError());
        return PPO_OK;
    }
    dim3 grid((unsigned)((N + 63) / 64));
    hipLaunchKernelGGL(k_gae_tn, grid, dim3(64), 0, ppo_stream(), r, done, values, adv, ret, T, N, gamma, lambda);
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}
