// kernels_fast_impl.cuh -- kernel templates of the single-pass path, instantiated once per ring degree in
// kernels_fast_n<LOGN>.hip so the five sizes compile in parallel.
#pragma once
#include "kernels.hpp"
#include "modarith.cuh"
#include "ntt_core.cuh"

namespace mi355ntt {

// coalesced layout B0: register r of thread t holds coefficient (r << B0) | t; lane offset in a VGPR,
// the r * (n/32) * 8 byte displacement in the buffer instruction's scalar offset
template <int LOGN>
__device__ __forceinline__ void load_coalesced(u64 (&v)[32], const u64* __restrict__ poly, unsigned t)
{
    const BufRsrc rs = make_rsrc(poly, Geo<LOGN>::N * 8u);
#pragma unroll
    for (int r = 0; r < 32; r++) v[r] = buf_load_u64(rs, t * 8u, ((unsigned)r << Geo<LOGN>::B0) * 8u);
}

template <int LOGN>
__device__ __forceinline__ void store_coalesced(const u64 (&v)[32], u64* __restrict__ poly, unsigned t)
{
    const BufRsrc rs = make_rsrc(poly, Geo<LOGN>::N * 8u);
#pragma unroll
    for (int r = 0; r < 32; r++) buf_store_u64(rs, t * 8u, ((unsigned)r << Geo<LOGN>::B0) * 8u, v[r]);
}

// ---- forward: natural -> bit-reversed, canonical ------------------------------------------------
template <int LOGN, int HL>
__global__ void __launch_bounds__(Geo<LOGN>::T, 4)
k_forward(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division,
          unsigned prime_base)
{
    using G = Geo<LOGN>;
    __shared__ __attribute__((aligned(16))) u64 lds[G::LDS_WORDS];
    const unsigned y = blockIdx.x;
    const unsigned idx = prime_base + y % division;
    const PrimeDev p = primes[idx];
    const TwPair* twp = tw + (size_t)idx * G::N;
    u64* poly = a + (size_t)y * G::N;
    const unsigned t = threadIdx.x;
    u64 v[32];
    load_coalesced<LOGN>(v, poly, t);
    forward_core<LOGN, HL>(v, twp, t, p, lds);
#pragma unroll
    for (int r = 0; r < 32; r++) v[r] = canon_2q(reduce_2q(v[r], p), p.q);
    exchange<LOGN, 0, G::B0>(v, lds, t);
    store_coalesced<LOGN>(v, poly, t);
}

// ---- inverse: bit-reversed -> natural, scaled by n^-1, canonical --------------------------------
template <int LOGN, int HL>
__global__ void __launch_bounds__(Geo<LOGN>::T, 4)
k_inverse(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division,
          unsigned prime_base)
{
    using G = Geo<LOGN>;
    __shared__ __attribute__((aligned(16))) u64 lds[G::LDS_WORDS];
    const unsigned y = blockIdx.x;
    const unsigned idx = prime_base + y % division;
    const PrimeDev p = primes[idx];
    const TwPair* twp = tw + (size_t)idx * G::N;
    u64* poly = a + (size_t)y * G::N;
    const unsigned t = threadIdx.x;
    u64 v[32];
    load_coalesced<LOGN>(v, poly, t);
    exchange<LOGN, G::B0, 0>(v, lds, t);
    inverse_core<LOGN, HL>(v, twp, t, p, lds);
#pragma unroll
    for (int r = 0; r < 32; r++) v[r] = canon_after_inverse<HL>(v[r], p);
    store_coalesced<LOGN>(v, poly, t);
}

// ---- fused: a = INTT( NTT(a) (.) bhat ) ---------------------------------------------------------
template <int LOGN, int HL>
__global__ void __launch_bounds__(Geo<LOGN>::T, 4)
k_polymul(u64* __restrict__ a, const u64* __restrict__ bhat, const TwPair* __restrict__ twf, const TwPair* __restrict__ twi,
          const PrimeDev* __restrict__ primes, unsigned division)
{
    using G = Geo<LOGN>;
    __shared__ __attribute__((aligned(16))) u64 lds[G::LDS_WORDS];
    const unsigned y = blockIdx.x;
    const unsigned idx = y % division;
    const PrimeDev p = primes[idx];
    u64* poly = a + (size_t)y * G::N;
    const u64* bp = bhat + (size_t)y * G::N;
    const unsigned t = threadIdx.x;
    u64 v[32];
    load_coalesced<LOGN>(v, poly, t);
    forward_core<LOGN, HL>(v, twf + (size_t)idx * G::N, t, p, lds);
    // layout 0: this thread holds NTT values 32t .. 32t+31; the inverse starts from the same layout
    const BufRsrc brs = make_rsrc(bp, G::N * 8u);
#pragma unroll
    for (int r = 0; r < 32; r += 2) {
        const TwPair bb = buf_load_tw(brs, t * 256u, (unsigned)r * 8u);      // two consecutive words of bhat
        const u64 x0 = canon_2q(reduce_2q(v[r], p), p.q);
        const u64 x1 = canon_2q(reduce_2q(v[r + 1], p), p.q);
        v[r] = barrett_mul(x0, bb.w, p.q, p.mu, p.k);          // poly_arithmetic.cuh:36-66, Algorithm 7
        v[r + 1] = barrett_mul(x1, bb.wp, p.q, p.mu, p.k);
        if ((r & 6) == 6) __builtin_amdgcn_sched_barrier(0);
    }
    inverse_core<LOGN, HL>(v, twi + (size_t)idx * G::N, t, p, lds);
#pragma unroll
    for (int r = 0; r < 32; r++) v[r] = canon_after_inverse<HL>(v[r], p);
    store_coalesced<LOGN>(v, poly, t);
}

template <int LOGN>
hipError_t launch_fwd(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division, unsigned base,
                      hipStream_t s)
{
    dim3 g(num), b(Geo<LOGN>::T);
    if (hl >= 6) k_forward<LOGN, 6><<<g, b, 0, s>>>(d_a, tw, pr, division, base);
    else if (hl >= 4) k_forward<LOGN, 4><<<g, b, 0, s>>>(d_a, tw, pr, division, base);
    else k_forward<LOGN, 2><<<g, b, 0, s>>>(d_a, tw, pr, division, base);
    return hipGetLastError();
}

template <int LOGN>
hipError_t launch_inv(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division, unsigned base,
                      hipStream_t s)
{
    dim3 g(num), b(Geo<LOGN>::T);
    if (hl >= 6) k_inverse<LOGN, 6><<<g, b, 0, s>>>(d_a, tw, pr, division, base);
    else if (hl >= 4) k_inverse<LOGN, 4><<<g, b, 0, s>>>(d_a, tw, pr, division, base);
    else k_inverse<LOGN, 2><<<g, b, 0, s>>>(d_a, tw, pr, division, base);
    return hipGetLastError();
}

template <int LOGN>
hipError_t launch_mul(int hl, u64* d_a, const u64* d_b, const TwPair* twf, const TwPair* twi, const PrimeDev* pr, unsigned num,
                      unsigned division, hipStream_t s)
{
    dim3 g(num), b(Geo<LOGN>::T);
    if (hl >= 6) k_polymul<LOGN, 6><<<g, b, 0, s>>>(d_a, d_b, twf, twi, pr, division);
    else if (hl >= 4) k_polymul<LOGN, 4><<<g, b, 0, s>>>(d_a, d_b, twf, twi, pr, division);
    else k_polymul<LOGN, 2><<<g, b, 0, s>>>(d_a, d_b, twf, twi, pr, division);
    return hipGetLastError();
}


// explicit per-size entry points (defined in kernels_fast_n<LOGN>.hip)
#define MI355NTT_DECLARE_SIZE(LOGN)                                                                                          \
    hipError_t fast_fwd_##LOGN(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division,     \
                               unsigned base, hipStream_t s);                                                                \
    hipError_t fast_inv_##LOGN(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division,     \
                               unsigned base, hipStream_t s);                                                                \
    hipError_t fast_mul_##LOGN(int hl, u64* d_a, const u64* d_b, const TwPair* twf, const TwPair* twi, const PrimeDev* pr,  \
                               unsigned num, unsigned division, hipStream_t s);
MI355NTT_DECLARE_SIZE(11)
MI355NTT_DECLARE_SIZE(12)
MI355NTT_DECLARE_SIZE(13)
MI355NTT_DECLARE_SIZE(14)
MI355NTT_DECLARE_SIZE(15)

#define MI355NTT_DEFINE_SIZE(LOGN)                                                                                           \
    hipError_t fast_fwd_##LOGN(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division,     \
                               unsigned base, hipStream_t s)                                                                 \
    {                                                                                                                        \
        return launch_fwd<LOGN>(hl, d_a, tw, pr, num, division, base, s);                                                    \
    }                                                                                                                        \
    hipError_t fast_inv_##LOGN(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division,     \
                               unsigned base, hipStream_t s)                                                                 \
    {                                                                                                                        \
        return launch_inv<LOGN>(hl, d_a, tw, pr, num, division, base, s);                                                    \
    }                                                                                                                        \
    hipError_t fast_mul_##LOGN(int hl, u64* d_a, const u64* d_b, const TwPair* twf, const TwPair* twi, const PrimeDev* pr,  \
                               unsigned num, unsigned division, hipStream_t s)                                               \
    {                                                                                                                        \
        return launch_mul<LOGN>(hl, d_a, d_b, twf, twi, pr, num, division, s);                                               \
    }

}  // namespace mi355ntt
