// modarith.cuh -- 64-bit modular arithmetic primitives for gfx950 (device side).
//
// Replaces the reference's uint128.h device layer (mul64 / sub128 PTX, uint128.h:343-373) and
// singleBarrett (ntt_60bit.cuh:44-61).  gfx950 has a native 32x32+64 -> 64 multiply-add
// (v_mad_u64_u32); 64x64 -> 128 products are four of those, written out here so the partial
// products each variant needs are explicit.
#pragma once
#include <hip/hip_runtime.h>

namespace mi355ntt {

using u64 = unsigned long long;
using u128 = unsigned __int128;
using u32 = unsigned int;

__device__ __forceinline__ u32 lo32(u64 x) { return (u32)x; }
__device__ __forceinline__ u32 hi32(u64 x) { return (u32)(x >> 32); }

// 32x32 + 64 -> 64 (v_mad_u64_u32)
__device__ __forceinline__ u64 mad32(u32 a, u32 b, u64 c) { return (u64)a * b + c; }

// high 64 bits of a 64x64 product, exact
__device__ __forceinline__ u64 mul_hi(u64 a, u64 b)
{
    u32 a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    u64 p00 = mad32(a0, b0, 0);
    u64 p01 = mad32(a0, b1, hi32(p00));          // < 2^64: (2^32-1)^2 + 2^32-1
    u64 p10 = mad32(a1, b0, lo32(p01));
    return mad32(a1, b1, (u64)hi32(p01) + hi32(p10));
}

// low 64 bits of a 64x64 product
__device__ __forceinline__ u64 mul_lo(u64 a, u64 b) { return a * b; }

// full 64x64 -> 128
__device__ __forceinline__ void mul_wide(u64 a, u64 b, u64& lo, u64& hi)
{
    u32 a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    u64 p00 = mad32(a0, b0, 0);
    u64 p01 = mad32(a0, b1, hi32(p00));
    u64 p10 = mad32(a1, b0, lo32(p01));
    hi = mad32(a1, b1, (u64)hi32(p01) + hi32(p10));
    lo = ((u64)lo32(p10) << 32) | lo32(p00);
}

// Algorithm 7 (singleBarrett, ntt_60bit.cuh:44-61) on a 128-bit value {hi:lo} < 2^(2k):
//   x1 = (a >> (k-2)).low ; s = ((x1*mu) >> (k+2)).low ; r = (a - s*q).low ; r -= q if r >= q
// Only the low 64 bits of a - s*q are ever inspected, so only lo(s*q) is formed.
__device__ __forceinline__ u64 barrett_reduce(u64 lo, u64 hi, u64 q, u64 mu, u32 k)
{
    u64 x1 = (lo >> (k - 2)) | (hi << (66 - k));       // 3 <= k <= 62
    u64 ml, mh;
    mul_wide(x1, mu, ml, mh);
    u64 s = (k == 62) ? mh : ((ml >> (k + 2)) | (mh << (62 - k)));
    u64 r = lo - s * q;
    return r >= q ? r - q : r;
}

__device__ __forceinline__ u64 barrett_mul(u64 a, u64 b, u64 q, u64 mu, u32 k)
{
    u64 lo, hi;
    mul_wide(a, b, lo, hi);
    return barrett_reduce(lo, hi, q, mu, k);
}

// canonical add / sub / halve, as the reference butterflies write them (ntt_60bit.cuh:102-110,166)
__device__ __forceinline__ u64 add_mod(u64 a, u64 b, u64 q)
{
    u64 t = a + b;
    return t >= q ? t - q : t;
}
__device__ __forceinline__ u64 sub_mod(u64 a, u64 b, u64 q) { return a < b ? a + q - b : a - b; }
__device__ __forceinline__ u64 half_mod(u64 x, u64 q2) { return (x >> 1) + ((x & 1) ? q2 : 0); }

// Shoup multiplication by a constant w < q with companion wp = floor(w * 2^64 / q).
// y may be ANY 64-bit value; result is congruent to y*w and lies in [0, 2q).
__device__ __forceinline__ u64 shoup_mul_lazy(u64 y, u64 w, u64 wp, u64 q)
{
    u64 h = mul_hi(y, wp);
    return y * w - h * q;
}

// x in [0, 2q) -> [0, q)
__device__ __forceinline__ u64 csub(u64 x, u64 q) { return x >= q ? x - q : x; }

}  // namespace mi355ntt
