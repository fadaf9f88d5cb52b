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
//      coalesced 1 KiB wave loads and are TRANSPOSED THROUGH LDS ([feature][36] fp32, conflict-free
//      for the row-major B reads and for the 16-byte feature-major reads of phases C/D): dZ2 = (W3^T dY) . lrelu'(H2)
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
#include <cstdlib>

#ifndef PPO_BWD_LD
#define PPO_BWD_LD 36
#endif
#ifndef PPO_BWD_ACT_NT
#define PPO_BWD_ACT_NT 1          // read-once saved activations: non-temporal loads
#endif
#if PPO_BWD_ACT_NT
#define PPO_BWD_NT_SFX " nt"
#else
#define PPO_BWD_NT_SFX ""
#endif
#ifndef PPO_BWD_KG
#define PPO_BWD_KG 2
#endif

// two fp32 FMAs per vector instruction (v_pk_fma_f32; each element is the same IEEE fma as fmaf)
typedef float pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2 pk_fma(pk2 a, pk2 b, pk2 c) { return __builtin_elementwise_fma(a, b, c); }

struct BwdArgs {
    unsigned long long* stamps;   // diagnostic build only (-DPPO_BWD_STAMP): [nwg][2 waves][6 phases]

    const int8_t* states; const int32_t* idx; int64_t B;   // B = number of 32-row tiles (states * tps)
    int tps;                                                // tiles per state (H / 32)
    int x_by_tile;                                          // 1: `states` is the forward's row scratch in minibatch order
                                                            //    (tile t's rows at t*32*F: compact rollouts); 0: gather by idx
    const float4* act1; const float4* act2; const float4* dY;
    const float4* w2tp; const float4* w3p;
    float* slabs; size_t slab_stride;
};

// HID = 128, F = 72: a workgroup is 4 waves (one per SIMD) and nothing covers its barriers and VALU gradient sums --
// unless a SECOND workgroup shares the CU.  Registers allow it (~220 per wave); LDS did not (86.5 KB).  So in this
// shape X^T is staged as int8 (2.3 KB instead of 9.2 KB as floats; the B operands of phase D are converted on the way
// in, 8 VALU per 16-byte read), which brings the workgroup to 79.6 KB: two per CU, 512 workgroups per launch.
// How the state rows X reach phase D's B operands (F = 72 only; the 8 tail columns always go through sXt as floats):
//   0  X^T as floats [feature][36 rows]: 288 cvt pairs + 288 ds_write_b32 per wave and tile, 8 ds_read_b128 in phase D
//   1  X^T as int8   [feature][36 rows]: 288 ds_write_b8, 8 ds_read_b32 + 32 cvt in phase D
//   2  X   as int8   [row][68 B] (the rows as they come): 64 ds_write_b32, 32 ds_read_i8 + 32 cvt in phase D
// Every vector instruction beside an fp32 MFMA costs MFMA time on gfx950, so the forms differ by
// what they issue, not by what they compute.
// Measured (gpurun_out/r2i, same box, alternating): form 0 is the fastest at HID = 256 (0.3523 vs 0.3555-0.3560 ms:
// the byte / dword reads of forms 1-2 put their LDS latency in front of phase D's MFMAs, which costs more than the
// 500 vector instructions they save); at HID = 128 the three are level (0.1211-0.1221 ms) and form 0 does not fit.
#ifndef PPO_BWD_XMODE
#define PPO_BWD_XMODE 0
#endif
// which state dword a thread stages: 1 = lane -> tile row, so the 32 lanes of a transposing LDS write land in 32 banks
// (the global load is then a 72-byte-stride gather inside a 2.3 KB, cache-resident tile); 0 = linear (coalesced load,
// 4.5-way bank conflicts on the writes: the 6 % SQ_LDS_BANK_CONFLICT of round 1)
#ifndef PPO_BWD_XROWLANE
#define PPO_BWD_XROWLANE 1
#endif
// dH2 = W3^T dY of the wave's feature tile: 1 = two MFMAs (k = the 4 outputs) whose accumulator IS the lane-is-row
// fragment the phase needs; 0 = 64 FMAs + 16 LDS reads of W3 per lane (round-1 form).  Only where the 16 extra live
// registers fit: at HID = 256 the kernel sits at 256 VGPRs and the MFMA form spills 10 of them.
#ifndef PPO_BWD_DH2_MFMA
#define PPO_BWD_DH2_MFMA 1
#endif
#ifndef PPO_BWD_DH2_MFMA_MAX_HID
#define PPO_BWD_DH2_MFMA_MAX_HID 256
#endif
// workgroup barrier between phases B and C/D (round-1 form; no data flow needs it, see the loop) for hidden widths below
// this one.  Measured (gpurun_out/r2v): HID = 256 0.3430 -> 0.3420 ms without it, HID = 128 0.1165 -> 0.1178 ms (two
// workgroups share a CU there and the barrier keeps their MFMA phases apart).
#ifndef PPO_BWD_BAR2_BELOW_HID
#define PPO_BWD_BAR2_BELOW_HID 256
#endif
template <int F, int HID>
struct BwdCfg {
    static constexpr int XMODE = (F == 72) ? ((HID == 128 && PPO_BWD_XMODE == 0) ? 1 : PPO_BWD_XMODE) : 0;   // HID = 128 needs an int8 form (LDS)
    static constexpr bool XI8 = XMODE == 1, XN8 = XMODE == 2;
    static constexpr int XS = 68;                                        // XN8 row stride in bytes (17 dwords: rows 16 apart sit 16 banks apart)
    static constexpr int WG_PER_CU = (HID == 128 && F == 72) ? 2 : 1;
    static constexpr int MIN_WAVES = (HID >= 256 || WG_PER_CU == 2) ? 2 : 1;   // per SIMD (caps the register budget at 256)
};

template <int F, int HID>
__global__ __launch_bounds__(HID * 2, (BwdCfg<F, HID>::MIN_WAVES)) void k_policy_bwd(BwdArgs a) {
    constexpr bool XI8 = BwdCfg<F, HID>::XI8, XN8 = BwdCfg<F, HID>::XN8;
    constexpr int XS = BwdCfg<F, HID>::XS;
    constexpr int NT = HID / 32;                // feature tiles == waves per workgroup (wave w owns tile w)
    constexpr int NTHR = NT * 64;
    constexpr int FP = ((F + 31) / 32) * 32;
    constexpr int NI = FP / 32;                 // i-tiles in the slab layout (last one partly padding)
    constexpr int NIM = F / 32;                 // full 32-column i-tiles done on the MFMA pipe
    constexpr int FT = F % 32;                  // tail columns (8 for F=72, 24 for F=216): VALU beside the MFMAs
    static_assert(FT % 4 == 0, "F must be a multiple of 4");
    // leading dimension (rows) of the LDS tiles: 36 = 4*9 keeps every row 16-byte aligned and makes the 16 lanes of a
    // ds_read_b128 group (consecutive features, 36 dwords apart) cover the 64 banks exactly once, so the products that
    // contract over the 32 rows fetch FOUR consecutive rows per LDS instruction (row = 16*half + step: the k-slot
    // order of an MFMA contraction is free as long as A and B agree).  Beside the fp32 MFMA an LDS read costs the same
    // ~6 ns whether it returns 4 or 16 bytes per lane (tools/microbench/mfma_f32_lds_overlap.hip), and these kernels
    // issued one ds_read_b32 per MFMA.
    constexpr int LD = PPO_BWD_LD;
    static_assert(F % 8 == 0, "state rows are staged in 8-byte units");
    constexpr int XQW = 32 * F / 8;             // 8-byte units of one state (F = 72, 512 threads: ONE pass, 288 threads busy;
    constexpr int XPD = (XQW + NTHR - 1) / NTHR;  //   as dwords it was a full pass plus a 64-thread one with its own addresses)
    constexpr int PF = (HID >= 256) ? 2 : 4;    // W2^T fragment groups per register set (two sets, ping-ponged)
    constexpr int S4 = HID / 8;                 // fragment groups of one W2^T tile
    static_assert(S4 % (2 * PF) == 0 && NTHR >= 256 && NTHR >= HID, "shape");
    // dZ2 row-major [32 rows][RS]: RS = HID + 4 puts the 16 lanes of a 16-byte access (rows j .. j+15) on 64 different banks
    constexpr bool Z2R = PPO_BWD_Z2ROW_AT(HID);
    constexpr int RS = HID + 4, Z2SZ = Z2R ? 32 * RS : HID * LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sZ2 = smem;                          // Z2R: dZ2 [32][RS]; else [HID][LD] dZ2^T
    float* sH1 = sZ2 + Z2SZ;                    // [HID][LD]  H1^T
    float* sH2 = sH1 + HID * LD;                // [HID][33]  H2^T
    float* sZ1 = sH2 + HID * LD;                // [HID][33]  dZ1^T
    float* sX = sZ1 + HID * LD;                 // [NIM*32][33]  X^T (float), MFMA part  (XI8: int8 [NIM*32][LD bytes])
    int8_t* const sXb = reinterpret_cast<int8_t*>(sX);
    float* sXt = XN8 ? sX + 32 * XS / 4 : (XI8 ? sX + NIM * 32 * LD / 4 : sX + NIM * 32 * LD);   // [32][FT]  X tail columns, row-major per tile row
    float* sDY = sXt + 32 * (FT > 0 ? FT : 4);  // [32][4]
    float* sW3 = sDY + 32 * 4;                  // [HID][4]   W3[:,f] per feature (staged once)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // wave index as a provably wave-uniform scalar: everything derived from it (tile bases, global
    // pointers) lives in SGPRs instead of per-lane 64-bit VGPR pairs
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    f32x16 accW2[NT];
    f32x16 accW1[NIM];
    pk2 tl2[FT > 0 ? FT / 2 : 1];               // dW1[k = 32w+j][NIM*32 + c], rows of this lane half (pairs: v_pk_fma_f32)
#pragma unroll
    for (int c = 0; c < (FT > 0 ? FT / 2 : 1); ++c) tl2[c] = pk2{0.0f, 0.0f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW2[kt][r] = 0.0f;
#pragma unroll
    for (int it = 0; it < NIM; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW1[it][r] = 0.0f;
    float db1 = 0.f, db2 = 0.f, db3 = 0.f, dw3[4] = {0.f, 0.f, 0.f, 0.f};

    if (tid < HID) {                                            // w3p is [h][tile][r][4]: un-permute to [f][4]
        const int kk = tid & 31, hh = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3);
        *reinterpret_cast<float4*>(&sW3[tid * 4]) = a.w3p[(size_t)(hh * NT + (tid >> 5)) * 16 + r];
    }
    __syncthreads();

    constexpr bool DH2M = PPO_BWD_DH2_MFMA && HID <= PPO_BWD_DH2_MFMA_MAX_HID;

    const char* const w2t = reinterpret_cast<const char*>(a.w2tp + (size_t)w * S4 * 64);   // this wave's W2^T tile (scalar base)
    const unsigned lo16 = (unsigned)lane * 16u;
    const unsigned fb = (unsigned)(32 * w + 4 * h);
    // LDS-DMA prefetch of the NEXT tile's layer-2 fragments (4 x 1 KiB, lane-linear) into this wave's own sH2
    // slice: rows [32w, 32w+32) of sH2 are written (phase A) and read (small_grads) by wave w only, so once the
    // wave's small_grads are done the slice is free until phase A of the next tile.  No registers involved.
    float* const h2slice = sH2 + (size_t)(32 * w) * LD;
    // The DMA is issued from inline asm on purpose: if hipcc knows an LDS-DMA is pending it (a) puts vmcnt(0) in front
    // of the next LDS read of ANY address (here: the VALU gradient loops, i.e. a full HBM round trip of stall),
    // (b) degrades every counted vmcnt(N) to vmcnt(0) and (c) drains it at each __syncthreads().  The only consumer
    // is this same wave in phase A of the next tile, behind an explicit s_waitcnt vmcnt(0).  M0 (the LDS-DMA
    // destination base) is saved/restored inside the statement (cdna_hip_programming.md 5.7).
    const unsigned h2slice_lds = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) void*)h2slice);
    auto dma_next_act2 = [&](int64_t t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's reads of the slice have returned
        const float4* src = a.act2 + ((size_t)t * NT + w) * 4 * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned keep;
            const float4* gsrc = src + q * 64;
            const unsigned dst = h2slice_lds + (unsigned)q * 1024u;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" PPO_BWD_NT_SFX "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
        }
    };
    if ((int64_t)blockIdx.x < a.B) dma_next_act2(blockIdx.x);

    float4 v1[4], dy;
    uint2 xd[XPD];
    auto issue_tile_loads = [&](int64_t t, int sidx) {      // sidx: transition id of tile t's state (wave-uniform)
        // scalar (SGPR) base + 32-bit per-lane byte offset -> saddr-form loads, no per-lane 64-bit pointers
        const char* s1 = reinterpret_cast<const char*>(a.act1 + ((size_t)t * NT + w) * 4 * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#if PPO_BWD_ACT_NT
            typedef float f32x4l __attribute__((ext_vector_type(4)));
            const f32x4l t4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4l*>(s1 + (lo16 + (unsigned)q * 1024u)));
            v1[q] = make_float4(t4.x, t4.y, t4.z, t4.w);
#else
            v1[q] = *reinterpret_cast<const float4*>(s1 + (lo16 + (unsigned)q * 1024u));
#endif
        }
        dy = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.dY + (size_t)t * 32) + (unsigned)j * 16u);
        const char* xs = reinterpret_cast<const char*>(a.states + (a.x_by_tile ? (size_t)t : ((size_t)sidx * a.tps + (size_t)(t % a.tps))) * 32 * F);
#pragma unroll
        for (int i = 0; i < XPD; ++i) {
            const unsigned u = (unsigned)tid + (unsigned)i * NTHR;
#if PPO_BWD_XROWLANE
            const unsigned d = (u & 31u) * (unsigned)(F / 8) + (u >> 5);      // lane -> row, 32-lane group -> one 8-feature unit
#else
            const unsigned d = u;
#endif
            xd[i] = u < (unsigned)XQW ? *reinterpret_cast<const uint2*>(xs + d * 8u) : make_uint2(0u, 0u);
        }
    };
    if ((int64_t)blockIdx.x < a.B)
        issue_tile_loads(blockIdx.x, a.x_by_tile ? 0 : __builtin_amdgcn_readfirstlane(a.idx[blockIdx.x / a.tps]));
    // LDS-DMA data is ordered for a later ds_read only by the issuing wave's vmcnt wait FOLLOWED BY A BARRIER (see
    // ppo_policy_bwd_x6.hip): the wait sits in front of the barrier that ends a tile (and this one, for the first tile)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef PPO_BWD_STAMP
    unsigned long long st_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = clock64();
#define STAMP(i) do { unsigned long long _n = clock64(); st_sum[i] += _n - st_t; st_t = _n; } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
    for (int64_t tile = blockIdx.x; tile < a.B; tile += gridDim.x) {
        // per-lane LDS bases: element (feature 32w+4h+fo, row j) of each transposed tile.  `lb` is made opaque
        // once per tile: everything derived from it is recomputed here (a few VALU ops) instead of being
        // hoisted out of the tile loop as ~50 loop-invariant address registers that then spill to scratch
        unsigned lb = fb * LD + j;
        asm volatile("" : "+v"(lb));
        float* const z2b = sZ2 + lb;
        float* const h1b = sH1 + lb;
        float* const h2b = sH2 + lb;
        float* const z1b = sZ1 + lb;
        const float* const w3b = sW3 + fb * 4;
        // ================= phase A: stage the tile (transposes through LDS)
        // Nothing is fetched here any more: the layer-1 fragments, dY and the state dwords were issued into
        // registers at the top of phase D of the previous tile (v1/dy/xd), the layer-2 fragments came by LDS-DMA.
        float4 v2[4];
        {
            // wait for everything this wave has in flight (its DMA is the oldest)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (DH2M) {
                // the layer-1 fragments go to their transposed places FIRST: their 16 registers are free again before the
                // dH2 accumulator and the layer-2 fragments come alive (the kernel sits at the 256-register budget)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float h1v[4] = {v1[q].x, v1[q].y, v1[q].z, v1[q].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) h1b[(e + 8 * q) * LD] = h1v[e];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // read the raw 4 KiB of layer-2 fragments out of the slice, and only then overwrite the slice with the
            // transposed tile below
#pragma unroll
            for (int q = 0; q < 4; ++q) v2[q] = *reinterpret_cast<const float4*>(h2slice + q * 256 + lane * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#ifdef PPO_BWD_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(6);
#endif
        if (w == 0 && h == 0) *reinterpret_cast<float4*>(&sDY[j * 4]) = dy;
        // feature of register r: 32w + 4h + (r&3) + 8(r>>2)  ->  one base per array + compile-time offsets
        // (the offsets fold into the ds_* immediate field; no per-element address registers)
        f32x16 dh2;
        if constexpr (DH2M) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dh2[r] = 0.0f;
            // A operand of dH2^T[f, row] = sum_o W3[o, f] dY[row, o] for feature f = 32w + j: k-step s, lane half h -> output
            // o = 2s + h (two LDS dwords per tile: not worth two registers across the whole tile loop)
            const float w3a0 = sW3[(32 * w + j) * 4 + h], w3a1 = sW3[(32 * w + j) * 4 + 2 + h];
            dh2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w3a0, h ? dy.y : dy.x, dh2, 0, 0, 0);     // register r = 4q + e <-> feature e + 8q (+ 4h)
            dh2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w3a1, h ? dy.w : dy.z, dh2, 0, 0, 0);
        }
        float* const z2r = sZ2 + j * RS + fb;                      // Z2R: this lane's row, features 32w + 4h + 8q .. +3
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float h2v[4] = {v2[q].x, v2[q].y, v2[q].z, v2[q].w};
            const float h1v[4] = {v1[q].x, v1[q].y, v1[q].z, v1[q].w};
            float z[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int fo = e + 8 * q;                          // feature offset inside the tile
                float dh;
                if constexpr (DH2M) dh = dh2[4 * q + e];
                else {
                    const float4 ww = *reinterpret_cast<const float4*>(w3b + fo * 4);
                    dh = fmaf(ww.w, dy.w, fmaf(ww.z, dy.z, fmaf(ww.y, dy.y, ww.x * dy.x)));
                }
                z[e] = dh * (h2v[e] > 0.0f ? 1.0f : 0.01f);
                if constexpr (!Z2R) z2b[fo * LD] = z[e];
                h2b[fo * LD] = h2v[e];
                if constexpr (!DH2M) h1b[fo * LD] = h1v[e];
            }
            if constexpr (Z2R) *reinterpret_cast<float4*>(z2r + 8 * q) = make_float4(z[0], z[1], z[2], z[3]);
        }
#pragma unroll
        for (int i = 0; i < XPD; ++i) {
            const int d = tid + i * NTHR;                         // unit row*(F/8) + c : features 8c..8c+7 of a row
            if (d < XQW) {
#if PPO_BWD_XROWLANE
                const int row = d & 31, c = d >> 5;               // the 32 lanes of an LDS write hold 32 rows of ONE feature:
#else                                                             // banks (LD*f + row) mod 64 are all different
                const int row = d / (F / 8), c = d % (F / 8);     // consecutive lanes = consecutive c: bank conflicts on the writes
#endif
                const uint32_t xw[2] = {xd[i].x, xd[i].y};
                if (8 * c < NIM * 32) {
                    if (XN8) { *reinterpret_cast<uint2*>(sXb + row * XS + 8 * c) = xd[i]; continue; }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (XI8) sXb[(8 * c + e) * LD + row] = (int8_t)(xw[e >> 2] >> (8 * (e & 3)));
                        else sX[(8 * c + e) * LD + row] = (float)(int)(int8_t)(xw[e >> 2] >> (8 * (e & 3)));
                    }
                } else {                                          // tail columns, row-major per tile row: two 16-byte writes
                    float xv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) xv[e] = (float)(int)(int8_t)(xw[e >> 2] >> (8 * (e & 3)));
                    float* xt = sXt + row * FT + (8 * c - NIM * 32);
                    *reinterpret_cast<float4*>(xt) = make_float4(xv[0], xv[1], xv[2], xv[3]);
                    *reinterpret_cast<float4*>(xt + 4) = make_float4(xv[4], xv[5], xv[6], xv[7]);
                }
            }
        }
        STAMP(0);
        __syncthreads();
        STAMP(1);
        // ================= phase B: small VALU grads, dH1 = W2^T dZ2 (MFMA), dZ1 -> LDS
        // small VALU grads, spread over all waves: lane (fl, hh) of wave w owns feature 32w+fl and rows
        // [16hh, 16hh+16); the two halves are added once at the end of the kernel.  The two waves that share
        // a SIMD (w and w + NT/2) run them at opposite ends of the phase, so one wave's VALU/LDS work sits
        // beside the other's MFMAs instead of both idling the matrix pipe together.
        auto small_grads = [&]() {
            const float* gz = sZ2 + (32 * w + j) * LD + 16 * h;     // (Z2R: db2 is summed from phase C's A operands instead)
            const float* gh = sH2 + (32 * w + j) * LD + 16 * h;
            const float* gy = sDY + 64 * h;
            float s2 = 0.f;
            pk2 d01 = {0.f, 0.f}, d23 = {0.f, 0.f};                // v_pk_fma_f32: two of the four dW3 sums per instruction
#pragma unroll 1
            for (int rc = 0; rc < 16; rc += 4) {                   // one 16-byte read per operand covers 4 rows
                float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (!Z2R) z4 = *reinterpret_cast<const float4*>(gz + rc);
                const float4 h4 = *reinterpret_cast<const float4*>(gh + rc);
                const float z[4] = {z4.x, z4.y, z4.z, z4.w}, hv[4] = {h4.x, h4.y, h4.z, h4.w};
                float4 y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = *reinterpret_cast<const float4*>(gy + (rc + i) * 4);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (!Z2R) s2 += z[i];
                    const pk2 hh2 = {hv[i], hv[i]};
                    d01 = pk_fma(pk2{y[i].x, y[i].y}, hh2, d01); d23 = pk_fma(pk2{y[i].z, y[i].w}, hh2, d23);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!Z2R) db2 += s2;
            dw3[0] += d01.x; dw3[1] += d01.y; dw3[2] += d23.x; dw3[3] += d23.y;
            if (tid < 4) {
                float s = 0.f;
                for (int r = 0; r < 32; ++r) s += sDY[r * 4 + tid];
                db3 += s;
            }
        };
        // W2^T stream (L2-resident): two register sets of PF fragment groups, ping-ponged.  The loads of one set are
        // issued BEFORE the MFMAs that consume the other (sched_barrier pins that order: left alone, hipcc sinks the
        // loads below the MFMAs and waits vmcnt(0) right behind them, exposing an L2 round trip per 16 MFMAs).
        float4 ringA[PF], ringB[PF];
#pragma unroll
        for (int g = 0; g < PF; ++g) ringA[g] = *reinterpret_cast<const float4*>(w2t + (lo16 + (unsigned)g * 1024u));
        const bool grads_first = (2 * w < NT);                    // wave-uniform (w is an SGPR)
        const int64_t ntile = (tile + gridDim.x < a.B) ? tile + gridDim.x : tile;    // harmless re-load on the last tile
        if (grads_first) small_grads();
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            // B operand: Z2R: dZ2[row j][f = 8g + 4h + e], one 16-byte read per fragment group; else dZ2^T[f = 8g + 2e + h][row j]
            const float* bz = Z2R ? sZ2 + j * RS + 4 * h : sZ2 + h * LD + j;
            const char* wn = w2t + (size_t)PF * 1024;                // scalar pointer to the next group set
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
#pragma unroll 1
            for (int s0 = 0; s0 < S4; s0 += 2 * PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) ringB[u] = *reinterpret_cast<const float4*>(wn + (lo16 + (unsigned)u * 1024u));
                wn += (size_t)PF * 1024;
                __builtin_amdgcn_sched_barrier(0);
                mfma_set(ringA);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < PF; ++u) ringA[u] = *reinterpret_cast<const float4*>(wn + (lo16 + (unsigned)u * 1024u));   // tail padding covers the over-read
                wn += (size_t)PF * 1024;
                __builtin_amdgcn_sched_barrier(0);
                mfma_set(ringB);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int fo = (r & 3) + 8 * (r >> 2);
                const float hv = h1b[fo * LD];
                z1b[fo * LD] = acc[r] * (hv > 0.0f ? 1.0f : 0.01f);
            }
        }
        if (!grads_first) small_grads();
        STAMP(2);
        // No workgroup barrier between phases B and C/D: phase C reads this wave's own dZ2 rows and everybody's H1^T
        // (complete since the barrier behind phase A), phase D this wave's own dZ1 rows (written by itself above) and
        // X^T (phase A), and nothing in C/D writes LDS.  A wave that finishes its dH1 chain early goes straight on, so
        // the two waves of a SIMD keep the skew grads_first gives them instead of being re-aligned here.
        if constexpr (HID < PPO_BWD_BAR2_BELOW_HID) __syncthreads();
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's dZ1 writes have landed
        STAMP(3);
        // LDS-DMA of the next tile's layer-2 fragments into this wave's (now idle) sH2 slice.  Issued here, after
        // the barrier, by every wave: while a DMA is pending hipcc turns every counted vmcnt(N) into vmcnt(0), which
        // must not happen inside the weight-stream loop of phase B; phases C/D have no counted waits.
        dma_next_act2(ntile);

        // ================= phase C: dW2[f,k] += sum_rows dZ2[f,row] * H1[k,row]   (wave w: f-tile w)
        // db1 and the dW1 tail columns (i >= NIM*32) on the VALU: lane (j, h) owns k = 32w+j and tile rows
        // [16h, 16h+16).  Like small_grads, the two waves of a SIMD run this at opposite ends of the C/D phases.
        auto tail_grads = [&]() {
            const float* g1 = sZ1 + (32 * w + j) * LD + 16 * h;
            const float* xt = sXt + 16 * h * FT;
            float s1 = 0.f;
            // the LDS reads of 4 rows are issued together, then consumed (left alone hipcc emits
            // read / lgkmcnt(0) / use for every single read: ~40 serialized LDS round trips per tile)
#pragma unroll 1
            for (int rc = 0; rc < 16; rc += 4) {
                const float4 z4 = *reinterpret_cast<const float4*>(g1 + rc);
                const float z[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
                for (int ib = 0; ib < 4; ib += 2) {            // two rows of X tail columns in flight (register budget)
                    float4 xv[2][FT / 4 > 0 ? FT / 4 : 1];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
#pragma unroll
                        for (int c4 = 0; c4 < FT / 4; ++c4) xv[i][c4] = *reinterpret_cast<const float4*>(xt + (rc + ib + i) * FT + 4 * c4);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        s1 += z[ib + i];
                        const pk2 zz = {z[ib + i], z[ib + i]};
#pragma unroll
                        for (int c4 = 0; c4 < FT / 4; ++c4) {
                            tl2[2 * c4 + 0] = pk_fma(zz, pk2{xv[i][c4].x, xv[i][c4].y}, tl2[2 * c4 + 0]);
                            tl2[2 * c4 + 1] = pk_fma(zz, pk2{xv[i][c4].z, xv[i][c4].w}, tl2[2 * c4 + 1]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            db1 += s1;
        };
        if (grads_first) tail_grads();
        {
            constexpr int KG = PPO_BWD_KG;                           // k-tiles whose B operands are in flight together
            // rows 16h .. 16h+15 of this lane's feature (Z2R: four 4-byte reads per step, and db2 is their sum)
            const float* pa = Z2R ? sZ2 + (16 * h) * RS + 32 * w + j : sZ2 + (32 * w + j) * LD + 16 * h;
            const float* pb = sH1 + j * LD + 16 * h;
            float s2 = 0.f;
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {                            // 4 rows per step
                float4 a4;
                if constexpr (Z2R) {
                    a4 = make_float4(pa[(4 * q) * RS], pa[(4 * q + 1) * RS], pa[(4 * q + 2) * RS], pa[(4 * q + 3) * RS]);
                    s2 += a4.x; s2 += a4.y; s2 += a4.z; s2 += a4.w;
                } else a4 = *reinterpret_cast<const float4*>(pa + 4 * q);
#pragma unroll
                for (int kh = 0; kh < NT; kh += KG) {
                    float4 b4[KG];
#pragma unroll
                    for (int u = 0; u < KG; ++u) b4[u] = *reinterpret_cast<const float4*>(pb + 32 * (kh + u) * LD + 4 * q);
#pragma unroll
                    for (int u = 0; u < KG; ++u) {
                        accW2[kh + u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[u].x, accW2[kh + u], 0, 0, 0);
                        accW2[kh + u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[u].y, accW2[kh + u], 0, 0, 0);
                        accW2[kh + u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[u].z, accW2[kh + u], 0, 0, 0);
                        accW2[kh + u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[u].w, accW2[kh + u], 0, 0, 0);
                    }
                }
            }
            if constexpr (Z2R) db2 += s2;
        }
        STAMP(7);
        // ================= phase D: dW1[k,i] += sum_rows dZ1[k,row] * X[i,row]     (wave w: k-tile w)
        // register pressure is lowest here: the NEXT tile's layer-1 fragments, dY and state dwords are issued
        // now (22 registers) and land under this phase's MFMAs and the barrier; consumed in phase A.
        // Unconditional (a harmless re-load on the last tile): no phi, so the staging registers live D..A only.
        {
            // next tile's transition id through the scalar cache (s_load + lgkmcnt): a vector load here would be
            // waited for with vmcnt(0), which also drains the pending LDS-DMA (a full HBM round trip)
            int nidx = 0;
            if (!a.x_by_tile) {                                  // wave-uniform
                const int32_t* ip = a.idx + ntile / a.tps;
                asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(nidx) : "s"(ip) : "memory");
            }
            issue_tile_loads(ntile, nidx);
        }
        STAMP(8);
        {
            const float* pa = sZ1 + (32 * w + j) * LD + 16 * h;
            const float* pb = sX + j * LD + 16 * h;
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                const float4 a4 = *reinterpret_cast<const float4*>(pa + 4 * q);
#pragma unroll
                for (int it = 0; it < NIM; ++it) {
                    float4 b4;
                    if (XN8) {                                   // feature 32 it + j of rows 16h + 4q .. +3: four sign-extending byte reads
                        const int8_t* xb = sXb + (16 * h + 4 * q) * XS + 32 * it + j;
                        b4 = make_float4((float)(int)xb[0], (float)(int)xb[XS], (float)(int)xb[2 * XS], (float)(int)xb[3 * XS]);
                    } else if (XI8) {                            // four rows of feature 32 it + j as one dword of int8
                        const uint32_t d = *reinterpret_cast<const uint32_t*>(sXb + (32 * it + j) * LD + 16 * h + 4 * q);
                        b4 = make_float4((float)(int)(int8_t)(d), (float)(int)(int8_t)(d >> 8), (float)(int)(int8_t)(d >> 16), (float)(int)(int8_t)(d >> 24));
                    } else b4 = *reinterpret_cast<const float4*>(pb + 32 * it * LD + 4 * q);
                    accW1[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, accW1[it], 0, 0, 0);
                    accW1[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, accW1[it], 0, 0, 0);
                    accW1[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, accW1[it], 0, 0, 0);
                    accW1[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, accW1[it], 0, 0, 0);
                }
            }
        }
        STAMP(9);
        if (!grads_first) tail_grads();
        STAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the next tile's LDS-DMA has landed before the barrier its reads sit behind
        __syncthreads();
        STAMP(5);
    }
#ifdef PPO_BWD_STAMP
    if (a.stamps && lane == 0 && (w == 0 || w == NT - 1))
        for (int i = 0; i < 10; ++i) a.stamps[((size_t)blockIdx.x * 2 + (w ? 1 : 0)) * 10 + i] = st_sum[i];
#endif

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
    for (int it = 0; it < NIM; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) sW1[((size_t)(w * NI + it) * 16 + r) * 64 + lane] = accW1[it][r];
    if (FT > 0) {   // tail columns: combine the two row halves, then place them at their fragment-order positions of i-tile NIM
        float tl[FT > 0 ? FT : 1];
#pragma unroll
        for (int c = 0; c < FT; ++c) tl[c] = (c & 1) ? tl2[c >> 1].y : tl2[c >> 1].x;
#pragma unroll
        for (int c = 0; c < FT; ++c) tl[c] += __shfl_xor(tl[c], 32);
        if (h == 0) {
            const int r = (j & 3) + 4 * (j >> 3), hh = (j >> 2) & 1;      // accumulator register / lane half that holds row k = 32w+j
#pragma unroll
            for (int c = 0; c < FT; ++c) sW1[((size_t)(w * NI + NIM) * 16 + r) * 64 + c + 32 * hh] = tl[c];
        }
    }
    {   // combine the two row halves (lanes l and l^32 own the same feature)
        db1 += __shfl_xor(db1, 32); db2 += __shfl_xor(db2, 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) dw3[i] += __shfl_xor(dw3[i], 32);
        if (h == 0) {
            const int f = 32 * w + j;
            sb1[f] = db1; sb2[f] = db2;
            *reinterpret_cast<float4*>(&sw3[f * 4]) = make_float4(dw3[0], dw3[1], dw3[2], dw3[3]);
        }
    }
    if (tid < 4) sb3[tid] = db3;
    // the LDS-DMA issued for the (re-loaded) last tile must have landed before the workgroup's LDS is released to the next
    // workgroup on this CU (see ppo_policy_bwd_x6.hip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#ifdef PPO_BWD_STAMP
unsigned long long* g_bwd_stamps = nullptr;
extern "C" int32_t ppo_debug_bwd_stamps(unsigned long long* out) {
    if (!g_bwd_stamps) return -1;
    (void)hipDeviceSynchronize();
    return hipMemcpy(out, g_bwd_stamps, 256 * 20 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

template <int F, int HID>
static size_t bwd_lds_bytes() {
    const size_t xt = (size_t)(F / 32) * 32 * PPO_BWD_LD;      // X^T elements: floats, or bytes in the int8 forms
    const size_t xfl = BwdCfg<F, HID>::XN8 ? (size_t)32 * BwdCfg<F, HID>::XS / 4 : (BwdCfg<F, HID>::XI8 ? xt / 4 : xt);
    const size_t z2 = PPO_BWD_Z2ROW_AT(HID) ? (size_t)32 * (HID + 4) : (size_t)HID * PPO_BWD_LD;
    return sizeof(float) * (z2 + (size_t)3 * HID * PPO_BWD_LD + xfl + (size_t)32 * ((F % 32) ? (F % 32) : 4) + 32 * 4 + (size_t)HID * 4);
}

int32_t launch_policy_bwd(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B) {
    if (p->dtype == PPO_DTYPE_BF16) return launch_policy_bwd_bf16(p, ro, idx_dev, B);
    if (ppo_bwd_split_enabled()) {
        const int32_t xs = launch_policy_bwd_x6(p, ro, idx_dev, B);
        if (xs != PPO_ERR_UNSUPPORTED) return xs;
    }
    BwdArgs a;
    a.tps = ro->H / 32;
    a.states = ro->compact ? p->xs.p : ro->states.p; a.x_by_tile = ro->compact ? 1 : 0;
    a.idx = idx_dev; a.B = B * a.tps;
    a.act1 = (const float4*)p->act1.p; a.act2 = (const float4*)p->act2.p; a.dY = (const float4*)p->dY.p;
    a.w2tp = (const float4*)p->w2tp.p; a.w3p = (const float4*)p->w3p.p;
    a.slabs = p->slabs.p; a.slab_stride = slab_floats(p->F, p->HID);
    a.stamps = nullptr;
#ifdef PPO_BWD_STAMP
    { static unsigned long long* dbg = nullptr; if (!dbg) (void)hipMalloc((void**)&dbg, 256 * 20 * 8); a.stamps = dbg; g_bwd_stamps = dbg; }
#endif
    int nwg = 0;
    ProfScope ps("k_policy_bwd");
#define LAUNCH(FF, HH)                                                                                        \
    do {                                                                                                      \
        const int64_t cap = 256 * BwdCfg<FF, HH>::WG_PER_CU;                                                  \
        nwg = (int)(a.B < cap ? a.B : cap);                                                                   \
        p->nwg_bwd = nwg; p->nwg_small = 0;                                                                                     \
        const size_t lds = bwd_lds_bytes<FF, HH>();                                                           \
        static thread_local bool attr_set = false;                                                                         \
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
