// ppo_x6.h -- split-fp32 ("bf16x6") helpers shared by ppo_policy_bwd_x6.hip and ppo_policy_fwd_x6.hip: an fp32 number as the
// exact sum of three bfloat16 pieces, piece products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
#pragma once
#include "ppo_internal.h"
#include "ppo_device.h"

typedef __bf16 xbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 xbf16x2 __attribute__((ext_vector_type(2)));
typedef float xf32x2 __attribute__((ext_vector_type(2)));
typedef short xs16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t x_pack(float lo, float hi) {             // v_cvt_pk_bf16_f32 (RNE)
    const xf32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, xbf16x2));
}
__device__ __forceinline__ f32x16 x_mfma(const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8, a), __builtin_bit_cast(xbf16x8, b), c, 0, 0, 0);
}
// four fp32 values -> three packed bf16 pieces (two dwords each): a = h + m + l exactly up to 2^-26 |a|
__device__ __forceinline__ void x_split4(const float (&a)[4], uint2& ph, uint2& pm, uint2& pl) {
    uint32_t P[2], M[2], L[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float a0 = a[2 * u], a1 = a[2 * u + 1];
        P[u] = x_pack(a0, a1);
        const float r0 = a0 - __uint_as_float(P[u] << 16), r1 = a1 - __uint_as_float(P[u] & 0xFFFF0000u);      // exact
        M[u] = x_pack(r0, r1);
        const float s0 = r0 - __uint_as_float(M[u] << 16), s1 = r1 - __uint_as_float(M[u] & 0xFFFF0000u);      // exact
        L[u] = x_pack(s0, s1);
    }
    ph = make_uint2(P[0], P[1]); pm = make_uint2(M[0], M[1]); pl = make_uint2(L[0], L[1]);
}
// two fp32 registers that hold exact bf16 values -> one packed dword {lo16 = a, hi16 = b}
__device__ __forceinline__ uint32_t x_perm(float a, float b) {
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// two transposed 4x16 block reads -> one 32x32x16 operand fragment
__device__ __forceinline__ uint4 x_tr_frag(const char* p0, const char* p1) {
    typedef __attribute__((address_space(3))) xs16x4 lds_s16x4;
    const xs16x4 u0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
    const xs16x4 u1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
    const uint2 a = __builtin_bit_cast(uint2, u0), b = __builtin_bit_cast(uint2, u1);
    return make_uint4(a.x, a.y, b.x, b.y);
}

