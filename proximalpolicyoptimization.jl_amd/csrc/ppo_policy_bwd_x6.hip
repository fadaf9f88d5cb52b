// ppo_policy_bwd_x6.hip -- K11 (src/train.jl:65-79): the fused fp32 backward with its three big products
//   dH1 = dZ2 W2      dW2 += dZ2^T H1      dW1 += dZ1^T X
// on the bf16 matrix pipe as SPLIT-fp32 products ("bf16x6"); same inputs, same gradient slab as ppo_policy_bwd.hip.
//
// Why.  On gfx950 v_mfma_f32_32x32x2_f32 retires 64 flop/clk/SIMD and blocks the vector ALU while it runs;
// v_mfma_f32_32x32x16_bf16 retires 1024 flop/clk/SIMD and leaves the vector ALU free for 24 of its 32 clocks.  An fp32
// number is the exact sum of three bfloat16 numbers (a = a_h + a_m + a_l: each an RNE rounding of what the previous ones
// left, |a_m| <= 2^-8 |a|, |a_l| <= 2^-17 |a|), a product of two bf16 numbers is exact in fp32, and the MFMA accumulates in
// fp32.  So
//     a b  =  a_h b_h + (a_h b_m + a_m b_h) + (a_h b_l + a_m b_m + a_l b_h)  +  O(2^-24 |a b|)
// six bf16 MFMAs of k = 16 (192 clocks) stand for the eight fp32 MFMAs of k = 2 (512 clocks) that cover the same 16
// contraction indices, with a truncation error of the order of ONE fp32 rounding -- the fp32 fma chain they replace
// rounds once per index, this form once per MFMA.  Where one operand is exact in bf16 (the int8 state rows X) three
// MFMAs do.  The gradient tests hold this kernel to the same 2e-5 max|g| bar against the float64 restatement of the tests as the
// pure-fp32 kernel.
//
// Data flow per 32-row tile (wave w owns feature tile w of every product):
//   A  H1 fragments (lane = row) -> three bf16 pieces -> row-major [row][32 features] images per feature tile
//      (8-byte-chunk swizzle: ds_write_b64 fill and ds_read_b64_tr_b16 transposed reads both conflict-free);
//      dZ2 = (dY W3) . lrelu'(H2), lane = row -> three pieces, which ARE A-operand fragments of the next product
//      (k order = accumulator-register order; the W2 pieces are packed to match) -> lane-linear fragment images;
//      H2^T -> fp32 tile (dW3 sums, and phase C); X -> bf16 [input][k-slot] image with a ONES column at input 72
//      (db1 = the dW1 column of that input), zeros up to 96.
//   B  dH1[row, k] = sum_f dZ2[row, f] W2[f, k]: A = the dZ2 fragments of all feature tiles (LDS), B = this wave's
//      W2 piece stream from L2 (three passes: W_l x Z_h | W_m x Z_m, Z_h | W_h x Z_l, Z_m, Z_h -- one 4-register
//      operand ring instead of three).  The result has lane = feature, registers = rows: dZ1 = dH1 . lrelu'(H1) (signs
//      by transposed reads of the H1 image) is, split in three, already the A operand of
//   D  dW1[w,:] += dZ1^T X : 3 input tiles x 2 k-steps x 3 pieces = 18 MFMAs
//   C  dW2[w,:] += dZ2^T H1: A = this wave's own dZ2 fragments, each piece transposed exactly by an identity MFMA
//      (lane = row -> lane = feature) and re-packed; B = the H1 piece images read transposed:
//      8 k-tiles x 2 k-steps x 6 = 96 MFMAs.
// The next tile's layer-1 / layer-2 fragments arrive by LDS-DMA during phase C into regions only this wave touches
// and that are dead by then (its dZ2 fragment images, its H2^T rows).
#include "ppo_internal.h"
#include "ppo_device.h"
#include <cstdlib>
#include <type_traits>

#include "ppo_x6.h"
typedef float pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2 pk_fma(pk2 a, pk2 b, pk2 c) { return __builtin_elementwise_fma(a, b, c); }

struct BwdXArgs {
    unsigned long long* stamps;   // diagnostic build only (-DPPO_X6_STAMP): [nwg][2 waves][12 phases]
    const int8_t* states; const int32_t* idx; int64_t B;   // B = number of 32-row tiles (states * tps)
    int tps; int x_by_tile;
    const float4* act1; const float4* act2; const float4* dY;
    const uint4* w2x; const float4* w3p;
    float* slabs; size_t slab_stride;
};

template <int F, int HID>
struct XCfg {
    static_assert(F == 72, "state rows of 72 features (+ the ones column, zero padded to 96)");
    static constexpr int NT = HID / 32, KS = HID / 16;       // feature tiles; k-steps of the dH1 contraction
    static constexpr int LD = 36;
    static constexpr int WG_PER_CU = HID == 128 ? 2 : 1;
    static constexpr int XROW = 80;                          // bytes per input row of the X image (32 k-slots of bf16 + pad: 16-byte reads conflict-free)
    static constexpr int NIX = 3;                            // input tiles: 72 features, ones at 72, zeros to 96
    static constexpr size_t oZ2 = 0, szZ2 = (size_t)NT * 6 * 1024;                 // dZ2 fragments [feature tile][k-step 2][piece 3][64 lanes][16 B]
    static constexpr size_t oH2 = oZ2 + szZ2, szH2 = sizeof(float) * HID * LD;      // H2^T fp32 [feature][36]
    static constexpr size_t oH1 = oH2 + szH2, szH1 = (size_t)3 * NT * 2048;        // H1 pieces [piece][feature tile][32 rows][64 B]
    static constexpr size_t oX = oH1 + szH1, szX = (size_t)32 * NIX * XROW;
    static constexpr size_t oDY = oX + szX, oW3 = oDY + 512, oID = oW3 + (size_t)HID * 16, total = oID + 2048;
    static_assert(total <= 160 * 1024 / WG_PER_CU, "LDS budget");
};

// per-phase opaque lane id: everything lane-derived (row, half, LDS offsets) is recomputed from it where it is used, a few
// VALU ops, instead of living in ~30 loop-invariant registers that hipcc hoists out of the tile loop and spills
#define X6_LANE() unsigned ln = (unsigned)lane; asm volatile("" : "+v"(ln)); const int j = (int)(ln & 31u), h = (int)(ln >> 5); (void)j; (void)h

// W2 piece fragments in flight ahead of the dH1 chain (3 per k-step).  HID = 256 runs at the 256-register budget: with
// the chain's two accumulators a ring of 6 spills one accumulator tile inside the chain loop, 3 fits
#ifndef PPO_X6_DMA_SPREAD
// next-tile LDS-DMA loads spread through phase C's MFMA loop (bit 0: layer 2, bit 1: layer 1) or issued together in front of
// it (0).  Off: the loads have to land before the barrier that ends the tile (see the loop), and the pieces issued last
// would make that barrier wait for them.
#define PPO_X6_DMA_SPREAD 0
#endif
// A/B knob (make -C csrc xprio): wave priority raised while a wave runs its MFMA loops (bit 0: the dH1 chain, bit 1: dW2), so that
// the SIMD partner's vector phases (splits, small gradients) do not take issue slots from it.  Not measured yet: off.
#ifndef PPO_X6_PRIO
#define PPO_X6_PRIO 0
#endif
#ifndef PPO_X6_ZPIPE
#define PPO_X6_ZPIPE 0
#endif
// A/B knob (make -C csrc xnodangle): the last ring round of the dH1 chain does not issue the reloads that run past the stream.
// Not measured yet: off.
#ifndef PPO_X6_NODANGLE
#define PPO_X6_NODANGLE 0
#endif
#ifndef PPO_X6_RING
#define PPO_X6_RING 6
#endif
// W2 pieces in flight per wave in the dH1 chain.  The chain waits on the L2 (one k-step of prefetch distance is ~200-400 clocks
// of MFMA work, an L2 hit is longer), so the ring is as deep as the registers allow: at HID = 256 that is 4 (256 VGPRs, no
// scratch; 6 spills 72 B and measured slower, 3 -> 4: 0.2484 -> 0.2440 ms, gpurun_out/rd4, tools/r3_rd4_ab.sh)
#ifndef PPO_X6_RING_256
#define PPO_X6_RING_256 4
#endif

template <int F, int HID>
__global__ __launch_bounds__(HID * 2, 2) void k_policy_bwd_x6(BwdXArgs a) {
    using C = XCfg<F, HID>;
    constexpr int NT = C::NT, KS = C::KS, NTHR = NT * 64, LD = C::LD, NIX = C::NIX, XROW = C::XROW;
    constexpr int XQW = 32 * F / 8, XPD = (XQW + NTHR - 1) / NTHR;
    constexpr int RD = HID >= 256 ? PPO_X6_RING_256 : PPO_X6_RING;
    static_assert(NTHR >= 256 && NTHR >= HID, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* const fragZ2 = smem_c + C::oZ2;
    float* const sH2 = reinterpret_cast<float*>(smem_c + C::oH2);
    char* const imgH1 = smem_c + C::oH1;
    char* const imgX = smem_c + C::oX;
    float* const sDY = reinterpret_cast<float*>(smem_c + C::oDY);
    float* const sW3 = reinterpret_cast<float*>(smem_c + C::oW3);
    uint4* const sID = reinterpret_cast<uint4*>(smem_c + C::oID);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    f32x16 accW2[NT], accW1[NIX];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW2[kt][r] = 0.0f;
#pragma unroll
    for (int it = 0; it < NIX; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW1[it][r] = 0.0f;
    float db2 = 0.f, db3 = 0.f, dw3[4] = {0.f, 0.f, 0.f, 0.f};

#ifdef PPO_X6_ZERO_LDS
    for (int i = tid; i < (int)(C::total / 4); i += NTHR) reinterpret_cast<uint32_t*>(smem_c)[i] = 0u;
    __syncthreads();
#endif
    if (tid < HID) {                                            // w3p is [h][tile][r][4]: un-permute to [f][4]
        const int kk = tid & 31, hh = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3);
        *reinterpret_cast<float4*>(&sW3[tid * 4]) = a.w3p[(size_t)(hh * NT + (tid >> 5)) * 16 + r];
    }
    // identity B operands of the transposing MFMAs: k-slot (step s, lane half hB, element e) carries feature
    // 16s + 8(e>>2) + 4hB + (e&3) of a packed accumulator tile, so lane (n, hB) holds a single 1.0 -- in step n>>4,
    // element 4((n>>3)&1) + (n&3), and only if hB == (n>>2)&1
    if (tid < 64) {
        const int n = tid & 31, hB = tid >> 5, e = 4 * ((n >> 3) & 1) + (n & 3);
#pragma unroll
        for (int s1 = 0; s1 < 2; ++s1) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (hB == ((n >> 2) & 1) && s1 == (n >> 4)) {
                const uint32_t one = (e & 1) ? 0x3F800000u : 0x00003F80u;
                if ((e >> 1) == 0) v.x = one; else if ((e >> 1) == 1) v.y = one; else if ((e >> 1) == 2) v.z = one; else v.w = one;
            }
            sID[s1 * 64 + tid] = v;
        }
    }
    // X image: zero, ones in the 32 k-slots of input 72 (never written again: the staging below touches inputs < 72)
    for (int i = tid; i < 32 * NIX * XROW / 4; i += NTHR) {
        const int row = i / (XROW / 4), c = i % (XROW / 4);
        reinterpret_cast<uint32_t*>(imgX)[i] = (row == F && c < 16) ? 0x3F803F80u : 0u;
    }
    __syncthreads();

    // this wave's W2 piece stream: [k-step][piece: lo, mid, hi][64 lanes][8 bf16], 3 KS KiB back to back
    const char* const wx = reinterpret_cast<const char*>(a.w2x + (size_t)w * 3 * KS * 64);
    float* const h2slice = sH2 + (size_t)(32 * w) * LD;
    char* const z2own = fragZ2 + (size_t)w * 6 * 1024;              // this wave's dZ2 fragments; DMA landing zone of the next tile's layer-1 fragments
    const unsigned h2slice_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)h2slice);
    const unsigned z2own_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)z2own);
    // LDS-DMA of a tile's fragments (4 x 1 KiB, lane-linear) into a region that only this wave touches; issued from inline
    // asm (see ppo_policy_bwd.hip for why hipcc must not know) and waited for with an explicit vmcnt(0) in front of the barrier that
    // ends the tile
    auto dma_frag = [&](const float4* base, int64_t t, unsigned dst_lds, unsigned ln) {
        const float4* src = base + ((size_t)t * NT + w) * 4 * 64 + ln;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned keep;
            const float4* gsrc = src + q * 64;
            const unsigned dst = dst_lds + (unsigned)q * 1024u;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
        }
    };
    // one of the four 1 KiB pieces of such a load (phase C issues them one per fragment set, between the MFMAs: eight loads
    // issued back to back hold the wave at the issue stage for ~1200 clocks)
    auto dma_one = [&](const float4* base, int64_t t, unsigned dst_lds, int q, unsigned ln) {
        unsigned keep;
        const float4* gsrc = base + ((size_t)t * NT + w) * 4 * 64 + ln + q * 64;
        const unsigned dst = dst_lds + (unsigned)q * 1024u;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    };
    (void)dma_one;
    float4 dy;
    uint2 xd[XPD];
    auto issue_tile_loads = [&](int64_t t, int sidx, unsigned ln) {
        dy = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.dY + (size_t)t * 32) + (ln & 31u) * 16u);
        const char* xs = reinterpret_cast<const char*>(a.states + (a.x_by_tile ? (size_t)t : ((size_t)sidx * a.tps + (size_t)(t % a.tps))) * 32 * F);
#pragma unroll
        for (int i = 0; i < XPD; ++i) {
            const unsigned u = (unsigned)(64 * w) + ln + (unsigned)i * NTHR;
            const unsigned d = (u & 31u) * (unsigned)(F / 8) + (u >> 5);      // lane -> row, 32-lane group -> one 8-feature unit
            xd[i] = u < (unsigned)XQW ? *reinterpret_cast<const uint2*>(xs + d * 8u) : make_uint2(0u, 0u);
        }
    };
    if ((int64_t)blockIdx.x < a.B) {
        dma_frag(a.act2, blockIdx.x, h2slice_lds, (unsigned)lane);
        dma_frag(a.act1, blockIdx.x, z2own_lds, (unsigned)lane);
        issue_tile_loads(blockIdx.x, a.x_by_tile ? 0 : __builtin_amdgcn_readfirstlane(a.idx[blockIdx.x / a.tps]), (unsigned)lane);
    }
    // LDS-DMA data is ordered for a later ds_read only by the issuing wave's vmcnt wait FOLLOWED BY A BARRIER (cdna_hip_programming.md,
    // "read a staged buffer one phase after the wait that retires it"): a read right behind the wait can still see the old LDS
    // content for part of the wave.  So the wait sits in front of the barrier that ends a tile (and this one, for the first tile),
    // the reads of the landing zones behind it.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#ifdef PPO_X6_STAMP
    unsigned long long st_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = clock64();
#define XSTAMP(i) do { unsigned long long _n = clock64(); st_sum[i] += _n - st_t; st_t = _n; } while (0)
#else
#define XSTAMP(i) do {} while (0)
#endif
    for (int64_t tile = blockIdx.x; tile < a.B; tile += gridDim.x) {
        // ================= phase A: stage the tile
        {   // H1 (lane = row j, register 4q+e <-> feature e + 8q + 4h of tile w) -> three bf16 pieces -> images
            X6_LANE();
            // this lane's 8-byte chunks in the H1 piece images of feature tile w: row j, chunk (2q + h) ^ ((j >> 2) & 7)
            const unsigned hw = (unsigned)(w * 2048 + 64 * j), cx = (unsigned)(j >> 2) & 7u;
            float4 v1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v1[q] = *reinterpret_cast<const float4*>(z2own + q * 1024 + ln * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float hv[4] = {v1[q].x, v1[q].y, v1[q].z, v1[q].w};
                uint2 ph, pm, pl;
                x_split4(hv, ph, pm, pl);
                const unsigned off = hw + 8u * (((unsigned)(2 * q + h)) ^ cx);
                *reinterpret_cast<uint2*>(imgH1 + off) = ph;
                *reinterpret_cast<uint2*>(imgH1 + NT * 2048 + off) = pm;
                *reinterpret_cast<uint2*>(imgH1 + 2 * NT * 2048 + off) = pl;
            }
        }
        __builtin_amdgcn_sched_barrier(0);                          // (keeps the two halves of the phase from sharing registers)
        XSTAMP(0);
        {
            X6_LANE();
            float* const h2b = sH2 + (32 * w + 4 * h) * LD + j;
            const float w3a0 = sW3[(32 * w + j) * 4 + h], w3a1 = sW3[(32 * w + j) * 4 + 2 + h];
            float4 v2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v2[q] = *reinterpret_cast<const float4*>(h2slice + q * 256 + ln * 4);
            if (w == 0 && h == 0) *reinterpret_cast<float4*>(&sDY[j * 4]) = dy;
            // dZ2 (lane = row): dH2^T[f, row] = sum_o W3[o, f] dY[row, o] as two fp32 MFMAs (k = the 4 outputs)
            f32x16 dh2;
#pragma unroll
            for (int r = 0; r < 16; ++r) dh2[r] = 0.0f;
            dh2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w3a0, h ? dy.y : dy.x, dh2, 0, 0, 0);
            dh2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w3a1, h ? dy.w : dy.z, dh2, 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the raw fragments are in registers: the slice may be overwritten
            // registers 8s .. 8s+7 packed = the A-operand fragment of k-step (w, s) of dH1 = dZ2 W2 (k order: the
            // accumulator-register order, acc_kslot in ppo_optim.hip)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint2 zh[2], zm[2], zl[2];
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int q = 2 * s + qq;
                    const float h2v[4] = {v2[q].x, v2[q].y, v2[q].z, v2[q].w};
                    float z[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        z[e] = dh2[4 * q + e] * (h2v[e] > 0.0f ? 1.0f : 0.01f);
                        h2b[(e + 8 * q) * LD] = h2v[e];
                    }
                    x_split4(z, zh[qq], zm[qq], zl[qq]);
                }
                *reinterpret_cast<uint4*>(z2own + (s * 3 + 0) * 1024 + ln * 16) = make_uint4(zh[0].x, zh[0].y, zh[1].x, zh[1].y);
                *reinterpret_cast<uint4*>(z2own + (s * 3 + 1) * 1024 + ln * 16) = make_uint4(zm[0].x, zm[0].y, zm[1].x, zm[1].y);
                *reinterpret_cast<uint4*>(z2own + (s * 3 + 2) * 1024 + ln * 16) = make_uint4(zl[0].x, zl[0].y, zl[1].x, zl[1].y);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        XSTAMP(1);
        // state rows -> bf16 X image [input][k-slot]; tile row R sits in k-slot R with bits 2 and 3 swapped (the row order
        // of accumulator registers: phase D's A operands are the dH1 accumulators)
        {
        X6_LANE();
#pragma unroll
        for (int i = 0; i < XPD; ++i) {
            const int d = 64 * w + (int)ln + i * NTHR;
            if (d < XQW) {
                const int row = d & 31, c = d >> 5;
                const int slot = (row & 19) | ((row & 4) << 1) | ((row & 8) >> 1);
                const uint32_t xw[2] = {xd[i].x, xd[i].y};
                char* const xp = imgX + (8 * c) * XROW + 2 * slot;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xv = (float)(int)(int8_t)(xw[e >> 2] >> (8 * (e & 3)));
                    *reinterpret_cast<uint16_t*>(xp + e * XROW) = (uint16_t)(__float_as_uint(xv) >> 16);     // |x| <= 128: exact in bf16
                }
            }
        }
        }
        XSTAMP(2);
        __syncthreads();
        XSTAMP(3);
        // ================= phase B: small VALU grads, dH1 = dZ2 W2, dZ1, dW1
        auto small_grads = [&]() {
            X6_LANE();
            const float* gh = sH2 + (32 * w + j) * LD + 16 * h;
            const float* gy = sDY + 64 * h;
            pk2 d01 = {0.f, 0.f}, d23 = {0.f, 0.f};
#pragma unroll 1
            for (int rc = 0; rc < 16; rc += 4) {
                const float4 h4 = *reinterpret_cast<const float4*>(gh + rc);
                const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
                float4 y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = *reinterpret_cast<const float4*>(gy + (rc + i) * 4);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const pk2 hh2 = {hv[i], hv[i]};
                    d01 = pk_fma(pk2{y[i].x, y[i].y}, hh2, d01); d23 = pk_fma(pk2{y[i].z, y[i].w}, hh2, d23);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            dw3[0] += d01.x; dw3[1] += d01.y; dw3[2] += d23.x; dw3[3] += d23.y;
            if (w == 0 && ln < 4u) {
                float s = 0.f;
                for (int r = 0; r < 32; ++r) s += sDY[r * 4 + ln];
                db3 += s;
            }
        };
        uint4 ring[RD];
#pragma unroll
        for (int g = 0; g < RD; ++g) { unsigned lo = (unsigned)lane * 16u; asm volatile("" : "+v"(lo)); ring[g] = *reinterpret_cast<const uint4*>(wx + (lo + (unsigned)g * 1024u)); }
        const bool grads_first = (2 * w < NT);
        const int64_t ntile = (tile + gridDim.x < a.B) ? tile + gridDim.x : tile;
        if (grads_first) small_grads();
        XSTAMP(4);
        {
#if PPO_X6_PRIO & 1
            __builtin_amdgcn_s_setprio(2);
#endif
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            // one pass over the k-steps; per k-step the three W2 pieces (lo, mid, hi: 1 + 2 + 3 MFMAs against the dZ2 pieces).
            // The stream pointer is a scalar that advances 1 KiB per piece, the fragment pointer advances per ring round:
            // nothing here is a per-step address the compiler could hoist out of the tile loop and spill
            // The leading terms (h h) and the small ones (h m + m h, 2^-8; h l + m m + l h, 2^-16) in separate accumulators,
            // added at the end: inside one MFMA the 16 products and the accumulator are aligned to the largest of them before
            // they are added (tools/microbench/mfma_bf16_accumulate.hip), so small terms fed into the leading running sum
            // would lose their low bits at every one of the 6 KS steps; summed among themselves they keep them (the 2^-16
            // terms then lose bits 2^-32 of the result).
            f32x16 accs;
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[r] = 0.0f;
            const char* wn = wx + (size_t)RD * 1024;
            // ring entry i of the stream (i = 3 k + piece) lives in slot i % RD; a round is the fewest k-steps after which the
            // slots repeat (RD = 3: 1, RD = 6: 2, RD = 4: 4), so every slot index below is a compile-time constant
            constexpr int RU = (RD % 3 == 0) ? RD / 3 : RD;
            static_assert(KS % RU == 0 && (3 * RU) % RD == 0, "ring rounds");
            unsigned zo = (unsigned)lane * 16u;
            asm volatile("" : "+v"(zo));
            const char* zp = fragZ2 + zo;
            const unsigned lo16 = zo;
#if PPO_X6_ZPIPE
            // A/B knob: the dZ2 pieces of the next k-step are read in front of the current one's six MFMAs (12 more registers: only
            // where the budget allows, i.e. HID = 128); the last read runs one k-step past the fragments (inside the LDS block)
            constexpr bool ZP = HID <= 128;
#else
            constexpr bool ZP = false;
#endif
            uint4 zc[3] = {}, zn[3] = {};
            if (ZP) {
                zc[0] = *reinterpret_cast<const uint4*>(zp); zc[1] = *reinterpret_cast<const uint4*>(zp + 1024);
                zc[2] = *reinterpret_cast<const uint4*>(zp + 2048);
            }
            // one ring round (RU k-steps).  LAST: the round whose reloads would run past the wave's stream -- with PPO_X6_NODANGLE those
            // loads are not issued (their registers are reused right behind the loop, and overwriting a register with a load in
            // flight costs a vmcnt wait: one exposed L2 latency per tile)
            auto chain_round = [&](auto last_c) {
                constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int s0 = (3 * u + 0) % RD, s1 = (3 * u + 1) % RD, s2 = (3 * u + 2) % RD;
                    // stream entry reloaded into slot s<pc>: 3 (k0 + u) + pc + RD, k0 = KS - RU in the last round
                    const bool ld0 = !LAST || 3 * (KS - RU + u) + 0 + RD < 3 * KS;
                    const bool ld1 = !LAST || 3 * (KS - RU + u) + 1 + RD < 3 * KS;
                    const bool ld2 = !LAST || 3 * (KS - RU + u) + 2 + RD < 3 * KS;
                    uint4 z_h, z_m, z_l;
                    if (ZP) {
                        zn[0] = *reinterpret_cast<const uint4*>(zp + ((u + 1) * 3 + 0) * 1024);
                        zn[1] = *reinterpret_cast<const uint4*>(zp + ((u + 1) * 3 + 1) * 1024);
                        zn[2] = *reinterpret_cast<const uint4*>(zp + ((u + 1) * 3 + 2) * 1024);
                        __builtin_amdgcn_sched_barrier(0);
                        z_h = zc[0]; z_m = zc[1]; z_l = zc[2];
                    } else {
                        z_h = *reinterpret_cast<const uint4*>(zp + (u * 3 + 0) * 1024);
                        z_m = *reinterpret_cast<const uint4*>(zp + (u * 3 + 1) * 1024);
                        z_l = *reinterpret_cast<const uint4*>(zp + (u * 3 + 2) * 1024);
                    }
                    accs = x_mfma(z_h, ring[s0], accs);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ld0) ring[s0] = *reinterpret_cast<const uint4*>(wn + lo16);   // (without PPO_X6_NODANGLE the last round reads RD KiB ahead: padding / next wave's stream)
                    __builtin_amdgcn_sched_barrier(0);
                    accs = x_mfma(z_m, ring[s1], accs);
                    accs = x_mfma(z_h, ring[s1], accs);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ld1) ring[s1] = *reinterpret_cast<const uint4*>(wn + 1024 + lo16);
                    __builtin_amdgcn_sched_barrier(0);
                    accs = x_mfma(z_l, ring[s2], accs);
                    accs = x_mfma(z_m, ring[s2], accs);
                    acc = x_mfma(z_h, ring[s2], acc);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ld2) ring[s2] = *reinterpret_cast<const uint4*>(wn + 2048 + lo16);
                    wn += 3 * 1024;
                    __builtin_amdgcn_sched_barrier(0);
                    if (ZP) { zc[0] = zn[0]; zc[1] = zn[1]; zc[2] = zn[2]; }
                }
                zp += RU * 3 * 1024;
            };
#if PPO_X6_NODANGLE
#pragma unroll 1
            for (int k0 = 0; k0 < KS - RU; k0 += RU) chain_round(std::false_type{});
            chain_round(std::true_type{});
#else
#pragma unroll 1
            for (int k0 = 0; k0 < KS; k0 += RU) chain_round(std::false_type{});
#endif
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = acc[r] + accs[r];
#if PPO_X6_PRIO & 1
            __builtin_amdgcn_s_setprio(0);
#endif
            XSTAMP(5);
            // acc: dH1, lane = feature 32w + j, register r <-> tile row (r&3) + 8(r>>2) + 4h.  dZ1 = dH1 . lrelu'(H1): the sign
            // of H1 from the first piece of its image, read transposed (block rows 8g + 4h .. +3 = registers 4g .. 4g+3)
            X6_LANE();
            const unsigned q4 = (ln & 15u) >> 2, p4 = ln & 3u, g1 = (ln >> 4) & 1u;
            typedef __attribute__((address_space(3))) xs16x4 lds_s16x4;
            uint2 zh[4], zm[4], zl[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned r0 = 8u * g + 4u * h;
                const char* hp = imgH1 + w * 2048 + 64u * (r0 + q4) + 8u * ((4u * g1 + p4) ^ ((r0 >> 2) & 7u));
                const uint2 sg = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hp)));
                const float z1[4] = {acc[4 * g + 0] * ((int16_t)(sg.x & 0xFFFFu) > 0 ? 1.0f : 0.01f),
                                     acc[4 * g + 1] * ((int32_t)sg.x >= 0x10000 ? 1.0f : 0.01f),
                                     acc[4 * g + 2] * ((int16_t)(sg.y & 0xFFFFu) > 0 ? 1.0f : 0.01f),
                                     acc[4 * g + 3] * ((int32_t)sg.y >= 0x10000 ? 1.0f : 0.01f)};
                x_split4(z1, zh[g], zm[g], zl[g]);
            }
            // ================= phase D: dW1[k,i] += sum_rows dZ1[k,row] * X[i,row]   (wave w: k-tile w; input 72 = ones: db1)
            const char* const xb = imgX + j * XROW + 16 * h;           // k-slots 16s + 8h .. +7 of input 32 it + j
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 bx[NIX];
#pragma unroll
                for (int it = 0; it < NIX; ++it) bx[it] = *reinterpret_cast<const uint4*>(xb + 32 * it * XROW + 32 * s);
                const uint4 a_l = make_uint4(zl[2 * s].x, zl[2 * s].y, zl[2 * s + 1].x, zl[2 * s + 1].y);
                const uint4 a_m = make_uint4(zm[2 * s].x, zm[2 * s].y, zm[2 * s + 1].x, zm[2 * s + 1].y);
                const uint4 a_h = make_uint4(zh[2 * s].x, zh[2 * s].y, zh[2 * s + 1].x, zh[2 * s + 1].y);
#pragma unroll
                for (int it = 0; it < NIX; ++it) {
                    accW1[it] = x_mfma(a_l, bx[it], accW1[it]);
                    accW1[it] = x_mfma(a_m, bx[it], accW1[it]);
                    accW1[it] = x_mfma(a_h, bx[it], accW1[it]);
                }
            }
        }
        XSTAMP(6);
        if (!grads_first) small_grads();
        XSTAMP(7);
        // every wave is through with the dZ2 fragments (A operands of the chain) and with its own H2^T rows' dW3 sums
        __syncthreads();
        XSTAMP(8);
        // ================= phase C: dW2[f,k] += sum_rows dZ2[f,row] * H1[k,row]   (wave w: f-tile w)
        {
            // A operand: dZ2^T of this wave's features with lane = feature.  The pieces sit in LDS as this wave's own fragments
            // (lane = row, written in phase A): an IDENTITY MFMA transposes each piece exactly (D = P I: lane (feature, h) gets
            // rows (r&3) + 8(r>>2) + 4h of the packed tile in register r, every value one bf16 number), v_perm packs them
            // again.  6 MFMAs + 24 packs instead of recomputing dZ2 in the other orientation and splitting it a second time.
            X6_LANE();
            uint4 ah[2], am[2], al[2];
            {
                const uint4 id0 = sID[ln], id1 = sID[64 + ln];
                float s2 = 0.f;
#pragma unroll
                for (int p = 2; p >= 0; --p) {                          // smallest piece first (db2 = the sum over the 32 rows)
                    const uint4 f0 = *reinterpret_cast<const uint4*>(z2own + (0 * 3 + p) * 1024 + ln * 16);
                    const uint4 f1 = *reinterpret_cast<const uint4*>(z2own + (1 * 3 + p) * 1024 + ln * 16);
                    f32x16 t;
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[r] = 0.0f;
                    t = x_mfma(f0, id0, t);
                    t = x_mfma(f1, id1, t);
                    float sp = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sp += t[r];
                    s2 += sp;
                    uint4 pk[2];
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        pk[s] = make_uint4(x_perm(t[8 * s + 0], t[8 * s + 1]), x_perm(t[8 * s + 2], t[8 * s + 3]),
                                           x_perm(t[8 * s + 4], t[8 * s + 5]), x_perm(t[8 * s + 6], t[8 * s + 7]));
                    if (p == 0) { ah[0] = pk[0]; ah[1] = pk[1]; } else if (p == 1) { am[0] = pk[0]; am[1] = pk[1]; } else { al[0] = pk[0]; al[1] = pk[1]; }
                }
                db2 += s2;
            }
            // the next tile's inputs: fragments by LDS-DMA into the (now idle) regions, dY and the state dwords into registers
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's reads of its H2^T rows have returned
#if !(PPO_X6_DMA_SPREAD & 1)
            dma_frag(a.act2, ntile, h2slice_lds, ln);
#endif
#if !(PPO_X6_DMA_SPREAD & 2)
            dma_frag(a.act1, ntile, z2own_lds, ln);
#endif
            {
                int nidx = 0;
                if (!a.x_by_tile) {
                    const int32_t* ip = a.idx + ntile / a.tps;
                    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(nidx) : "s"(ip) : "memory");
                }
                issue_tile_loads(ntile, nidx, ln);
            }
            // B operands: H1 pieces of k-tile kt, transposed reads.  k-slot (s, h, e) <-> tile row 16s + 8(e>>2) + 4h + (e&3): the
            // two block reads of a fragment take rows r0 .. r0+3, r0 = 16s + 8u + 4h; 16-lane group G = lane >> 4 takes columns
            // 16(G&1) .., its lane 4q + p addresses row r0 + q, 8-byte chunk 4(G&1) + p
            unsigned tb[2][2];
            {
                const unsigned q4 = (ln & 15u) >> 2, p4 = ln & 3u, g1 = (ln >> 4) & 1u;
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const unsigned r0 = 16u * s + 8u * u + 4u * h;
                        tb[s][u] = 64u * (r0 + q4) + 8u * ((4u * g1 + p4) ^ ((r0 >> 2) & 7u));
                    }
                asm volatile("" : "+v"(tb[0][0]), "+v"(tb[0][1]), "+v"(tb[1][0]), "+v"(tb[1][1]));
            }
            // one fragment set (k-tile kt, k-step s) of lookahead: its six transposed reads are issued before the six MFMAs
            // of the set in front of it (pinned: left alone hipcc hoists every read of the unrolled loop and spills)
            auto load_b = [&](int g, uint4 (&b)[3]) {
                const char* i0 = imgH1 + (g >> 1) * 2048;
                const int s = g & 1;
                b[0] = x_tr_frag(i0 + tb[s][0], i0 + tb[s][1]);
                b[1] = x_tr_frag(i0 + NT * 2048 + tb[s][0], i0 + NT * 2048 + tb[s][1]);
                b[2] = x_tr_frag(i0 + 2 * NT * 2048 + tb[s][0], i0 + 2 * NT * 2048 + tb[s][1]);
            };
            XSTAMP(9);
#if PPO_X6_PRIO & 2
            __builtin_amdgcn_s_setprio(2);
#endif
            uint4 bc[3], bn[3];
            load_b(0, bc);
#pragma unroll
            for (int g = 0; g < 2 * NT; ++g) {
                const int kt = g >> 1, s = g & 1;
                if (g + 1 < 2 * NT) load_b(g + 1, bn);
                __builtin_amdgcn_sched_barrier(0);
                accW2[kt] = x_mfma(al[s], bc[0], accW2[kt]);
                accW2[kt] = x_mfma(am[s], bc[1], accW2[kt]);
                accW2[kt] = x_mfma(ah[s], bc[2], accW2[kt]);
                accW2[kt] = x_mfma(am[s], bc[0], accW2[kt]);
                accW2[kt] = x_mfma(ah[s], bc[1], accW2[kt]);
                accW2[kt] = x_mfma(ah[s], bc[0], accW2[kt]);
                __builtin_amdgcn_sched_barrier(0);
#if PPO_X6_DMA_SPREAD
                // the next tile's saved activations, one 1 KiB piece per fragment set (this wave's regions are idle since the
                // lgkmcnt(0) above): layer 2 into the H2^T rows, layer 1 into the dZ2 fragment images
                if ((PPO_X6_DMA_SPREAD & 1) && g < 4) dma_one(a.act2, ntile, h2slice_lds, g, ln);
                else if ((PPO_X6_DMA_SPREAD & 2) && g >= 4 && g < 8) dma_one(a.act1, ntile, z2own_lds, g - 4, ln);
                __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int p = 0; p < 3; ++p) bc[p] = bn[p];
            }
        }
#if PPO_X6_PRIO & 2
        __builtin_amdgcn_s_setprio(0);
#endif
        XSTAMP(10);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's LDS-DMA (and register prefetch) of the next tile has landed ...
        __syncthreads();                                            // ... and only behind this barrier are its landing zones read
        XSTAMP(11);
    }
#ifdef PPO_X6_STAMP
    if (a.stamps && lane == 0 && (w == 0 || w == NT - 1))
        for (int i = 0; i < 12; ++i) a.stamps[((size_t)blockIdx.x * 2 + (w ? 1 : 0)) * 12 + i] = st_sum[i];
#endif

    // ================= write the slab (fragment order; k_grad_reduce maps it to Flux order)
    constexpr int FP = 96, NI = FP / 32;
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
    for (int it = 0; it < NI; ++it) {
        if (32 * it + j >= F) continue;                 // padding columns (inputs 72 .. 95): k_grad_reduce never reads them
#pragma unroll
        for (int r = 0; r < 16; ++r) sW1[((size_t)(w * NI + it) * 16 + r) * 64 + lane] = accW1[it][r];
    }
    // db1[k] = dW1[k][input 72] (the ones column): column 8 of input tile 2, held by lanes 8 and 40
    if (j == F - 64) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sb1[32 * w + (r & 3) + 8 * (r >> 2) + 4 * h] = accW1[2][r];
    }
    {   // combine the two row halves (lanes l and l^32 own the same feature)
        db2 += __shfl_xor(db2, 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) dw3[i] += __shfl_xor(dw3[i], 32);
        if (h == 0) {
            const int f = 32 * w + j;
            sb2[f] = db2;
            *reinterpret_cast<float4*>(&sw3[f * 4]) = make_float4(dw3[0], dw3[1], dw3[2], dw3[3]);
        }
    }
    if (tid < 4) sb3[tid] = db3;
    // The last tile's phase C has issued one more set of LDS-DMA loads (a harmless re-load of the same tile).  They must have
    // LANDED before this wave ends: a DMA that arrives after the workgroup's LDS has been handed to the next workgroup on the
    // CU writes into THAT workgroup's LDS (seen as run-to-run differences in a handful of gradient elements at HID = 128, where
    // the workgroup's tail is short: tools/x6_repro_check2.py).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#ifdef PPO_X6_STAMP
static unsigned long long* g_x6_stamps = nullptr;
extern "C" int32_t ppo_debug_x6_stamps(unsigned long long* out) {
    if (!g_x6_stamps) return -1;
    (void)hipDeviceSynchronize();
    return hipMemcpy(out, g_x6_stamps, 512 * 24 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

int32_t launch_policy_bwd_x6(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B) {
    BwdXArgs a;
    a.stamps = nullptr;
#ifdef PPO_X6_STAMP
    { if (!g_x6_stamps) (void)hipMalloc((void**)&g_x6_stamps, 512 * 24 * 8); a.stamps = g_x6_stamps; }
#endif
    a.tps = ro->H / 32;
    a.states = ro->compact ? p->xs.p : ro->states.p; a.x_by_tile = ro->compact ? 1 : 0;
    a.idx = idx_dev; a.B = B * a.tps;
    a.act1 = (const float4*)p->act1.p; a.act2 = (const float4*)p->act2.p; a.dY = (const float4*)p->dY.p;
    if (p->L != 2 || !p->w2x.p) return PPO_ERR_UNSUPPORTED;
    a.w2x = (const uint4*)p->w2x.p; a.w3p = (const float4*)p->w3p.p;
    a.slabs = p->slabs.p; a.slab_stride = slab_floats(p->F, p->HID);
    int nwg = 0;
    ProfScope ps("k_policy_bwd");
#define LAUNCH(FF, HH)                                                                                        \
    do {                                                                                                      \
        const int64_t cap = 256 * XCfg<FF, HH>::WG_PER_CU;                                                    \
        nwg = (int)(a.B < cap ? a.B : cap);                                                                   \
        p->nwg_bwd = nwg; p->nwg_small = 0;                                                                   \
        const size_t lds = XCfg<FF, HH>::total;                                                               \
        static thread_local bool attr_set = false;                                                            \
        if (!attr_set) {                                                                                      \
            HIP_TRY(hipFuncSetAttribute((const void*)k_policy_bwd_x6<FF, HH>,                                 \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
            attr_set = true;                                                                                  \
        }                                                                                                     \
        hipLaunchKernelGGL((k_policy_bwd_x6<FF, HH>), dim3(nwg), dim3(HH * 2), lds, ppo_stream(), a);          \
    } while (0)
    if (p->F == 72 && p->HID == 256) LAUNCH(72, 256);
    else if (p->F == 72 && p->HID == 128) LAUNCH(72, 128);
    else return PPO_ERR_UNSUPPORTED;
#undef LAUNCH
    HIP_TRY(hipGetLastError());
    return PPO_OK;
}
