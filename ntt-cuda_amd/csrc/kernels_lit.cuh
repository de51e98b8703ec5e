// kernels_lit.cuh -- kernel class HL_LIT: the reference's own arithmetic in the single-pass kernel shape (round 6).
//
// Contexts that hold a Barrett-inexact prime must return the reference's words, not the exact transform's (ntt_core.cuh, "kernel
// class HL_LIT"; capi.cpp).  Until round 5 their polynomials ran the stage-per-launch kernels of kernels_compat.hip (2 passes over
// memory at n = 2^15, 0.10 of the HBM roofline, plus a gather buffer for contexts that mix both kinds of primes).  These kernels run
// the register-resident rounds of the lazy classes with the literal butterflies: ONE read and ONE write of HBM per transform, in
// place, one launch for the whole call.  (Small batches: the kernel shape of kernels_lat.cuh with the same butterflies, at the end
// of this file; which shape a call takes: lit_use_latency_path, kernels_fast_impl.cuh.)
//
// A context of this class may hold Barrett-EXACT primes next to the inexact ones -- the reference's own decryption_test.cu:47-48 set:
// two exact, one not.  For those the reference's words ARE the exact transform's, so their polynomials take the lazy butterflies of
// class HL_LIT_EXACT (2: exact quotients, valid for every q < 2^62).  A persistent workgroup therefore walks its polynomials in TWO
// passes -- first those of the inexact primes with the literal rounds, then those of the exact primes with the lazy rounds -- each
// pass a loop of its own, so that the two bodies never meet in one loop (as the two arms of a branch inside one polynomial loop they
// cost 170-240 bytes of scratch per lane: the 64 registers that carry the polynomial across the join pin both allocations).  Which
// polynomials belong to which pass is a 16-bit mask of the call's primes, read once per workgroup with scalar loads.
#pragma once

namespace mi355ntt {

// bit i = prime (prime_base + i) of this call is Barrett-inexact (PrimeDev::lit); division <= 16
__device__ __forceinline__ unsigned lit_mask_of(const PrimeDev* __restrict__ primes, unsigned prime_base, unsigned division)
{
    unsigned m = 0;
    for (unsigned i = 0; i < division; i++) m |= (primes[prime_base + i].lit ? 1u : 0u) << i;      // (uniform addresses: scalar loads)
    m = __builtin_amdgcn_readfirstlane(m);
    asm volatile("" : "+s"(m));
    return m;
}

// The polynomials of one pass of one workgroup: positions first, first + stride, ... below num whose prime's bit in `mask` equals
// `want`.  All of it lives in SGPRs (wave-uniform values pinned as in the lazy kernels: a `%` by a run-time value is a VALU sequence).
struct LitWalk {
    unsigned y, ymod, stride, ystep, division, num, mask, want;
    __device__ __forceinline__ LitWalk(unsigned first, unsigned stride_, unsigned division_, unsigned num_, unsigned mask_, bool want_)
        : y(first), stride(stride_), division(division_), num(num_), mask(mask_), want(want_ ? 1u : 0u)
    {
        ymod = __builtin_amdgcn_readfirstlane(first % division_);
        ystep = __builtin_amdgcn_readfirstlane(stride_ % division_);
        asm volatile("" : "+s"(ymod), "+s"(ystep));
        seek();
    }
    __device__ __forceinline__ bool done() const { return y >= num; }
    __device__ __forceinline__ void step()
    {
        y += stride;
        ymod = (ymod + ystep >= division ? ymod + ystep - division : ymod + ystep);
    }
    __device__ __forceinline__ void seek()
    {
        while (y < num && ((mask >> ymod) & 1u) != want) step();
    }
    __device__ __forceinline__ void advance()
    {
        step();
        seek();
    }
};

// ================================================================================================
// n = 2^15 (the rounds, exchanges and row staging of k_forward15 / k_inverse15 / k_polymul15, kernels_fast_impl.cuh)
// ================================================================================================
template <int HB, bool WANT>
__device__ __forceinline__ void lit_pass_forward15(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes,
                                                   unsigned division, unsigned prime_base, unsigned num, unsigned mask, u64* lds, unsigned wave_s)
{
    constexpr int LOGN = 15;
    using G = Geo<LOGN>;
    auto fresh_t = [&]() { return (wave_s << 6) | fresh_lane_id(); };
    LitWalk w(blockIdx.x, gridDim.x, division, num, mask, WANT);
    if (w.done()) return;
    u64 v[32];
    load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)w.y * G::N, fresh_t());
    while (!w.done()) {
        const unsigned idx = prime_base + w.ymod;
        const PrimeDev p = primes[idx];
        const TwPair* twp = tw + (size_t)idx * G::N;
        const BufRsrc twr = make_rsrc(twp, G::N * 16u);
        u64* poly = a + (size_t)w.y * G::N;
        w.advance();
        __builtin_amdgcn_s_setprio(Tune::kPrioR1);
        ct_round<LOGN, HB, 10, 4, false, Tune::kPsplitR1, Tune::kPrioR1B>(v, twp, twr, 0u, p);
        __syncthreads();                                  // every wave has left its private slice (previous polynomial)
        exchange<LOGN, 10, 5>(v, lds, fresh_t());
        __builtin_amdgcn_s_setprio(Tune::kPrioR2);
        ct_round<LOGN, HB, 5, 4, false, Tune::kPsplitR2, Tune::kPrioR2B>(v, twp, twr, fresh_t(), p);
        wave_transpose_5_to_0(v, lds + wave_s * WAVE_SLICE_WORDS, fresh_lane_id());
        __builtin_amdgcn_s_setprio(Tune::kPrioR3);
        ct_round<LOGN, HB, 0, 4, false, Tune::kPsplitR3, Tune::kPrioR3B>(v, twp, twr, fresh_t(), p);
        static_for<32>([&](auto rc) { v[decltype(rc)::value] = canon_after_forward<HB, false>(v[decltype(rc)::value], p); });
        wave_store_rows(v, lds + wave_s * WAVE_SLICE_WORDS, make_rsrc(poly + wave_s * 2048u, 16384u), 0u, 0u);
        if (!w.done()) load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)w.y * G::N, fresh_t());
    }
}

template <int LOGN>      // (= 15: a template so that only the translation unit that launches it holds the kernel)
__global__ void __launch_bounds__(1024, 4)
k_forward15_lit(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division,
                unsigned prime_base, unsigned num)
{
    if (guard_says_skip(primes, prime_base)) return;      // checked raw call whose table is not the cached one
    static_assert(LOGN == 15, "n = 2^15 only");
    __shared__ __attribute__((aligned(16))) u64 lds[Geo<15>::LDS_WORDS];
    unsigned wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+s"(wave_s));
    const unsigned mask = lit_mask_of(primes, prime_base, division);
    lit_pass_forward15<HL_LIT, true>(a, tw, primes, division, prime_base, num, mask, lds, wave_s);
    __syncthreads();                                      // (the second pass starts on idle slices)
    lit_pass_forward15<HL_LIT_EXACT, false>(a, tw, primes, division, prime_base, num, mask, lds, wave_s);
}

template <int HB, bool WANT>
__device__ __forceinline__ void lit_pass_inverse15(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes,
                                                   unsigned division, unsigned prime_base, unsigned num, unsigned mask, u64* lds, unsigned wave_s)
{
    constexpr int LOGN = 15;
    using G = Geo<LOGN>;
    auto fresh_t = [&]() { return (wave_s << 6) | fresh_lane_id(); };
    u64* slice = lds + wave_s * WAVE_SLICE_WORDS;
    LitWalk w(blockIdx.x, gridDim.x, division, num, mask, WANT);
    if (w.done()) return;
    u64 v[32];
    wave_load_rows(v, slice, make_rsrc(a + (size_t)w.y * G::N + wave_s * 2048u, 16384u), 0u, 0u);
    while (!w.done()) {
        const unsigned idx = prime_base + w.ymod;
        const PrimeDev p = primes[idx];
        const TwPair* twp = tw + (size_t)idx * G::N;
        const BufRsrc twr = make_rsrc(twp, G::N * 16u);
        u64* poly = a + (size_t)w.y * G::N;
        w.advance();
        __builtin_amdgcn_s_setprio(Tune::kPrioI1);
        gs_round<LOGN, HB, 0, 0, false, Tune::kPsplitI1, Tune::kPrioI1B>(v, twp, twr, fresh_t(), p, primes[idx].twn);
        wave_transpose_0_to_5(v, slice, fresh_lane_id());
        __builtin_amdgcn_s_setprio(Tune::kPrioI2);
        gs_round<LOGN, HB, 5, 0, false, Tune::kPsplitI2, Tune::kPrioI2B>(v, twp, twr, fresh_t(), p, primes[idx].twn);
        __syncthreads();                                  // private slices are idle from here on
        exchange<LOGN, 5, 10>(v, lds, fresh_t());
        __builtin_amdgcn_s_setprio(Tune::kPrioI3);
        gs_round<LOGN, HB, 10, 0, false, Tune::kPsplitI3, Tune::kPrioI3B>(v, twp, twr, 0u, p, primes[idx].twn);
        static_for<32>([&](auto rc) { v[decltype(rc)::value] = canon_after_inverse<HB, false>(v[decltype(rc)::value], p); });
        store_coalesced<LOGN, Tune::kInv15AuxSt>(v, poly, fresh_t());
        // (the exchange ended with a barrier: every slice is free for this wave's own row staging)
        if (!w.done()) wave_load_rows(v, slice, make_rsrc(a + (size_t)w.y * G::N + wave_s * 2048u, 16384u), 0u, 0u);
    }
}

template <int LOGN>      // (= 15: a template so that only the translation unit that launches it holds the kernel)
__global__ void __launch_bounds__(1024, 4)
k_inverse15_lit(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division,
                unsigned prime_base, unsigned num)
{
    if (guard_says_skip(primes, prime_base)) return;
    static_assert(LOGN == 15, "n = 2^15 only");
    __shared__ __attribute__((aligned(16))) u64 lds[Geo<15>::LDS_WORDS];
    unsigned wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+s"(wave_s));
    const unsigned mask = lit_mask_of(primes, prime_base, division);
    lit_pass_inverse15<HL_LIT, true>(a, tw, primes, division, prime_base, num, mask, lds, wave_s);
    __syncthreads();
    lit_pass_inverse15<HL_LIT_EXACT, false>(a, tw, primes, division, prime_base, num, mask, lds, wave_s);
}

// fused a = INTT(NTT(a) (.) bhat): forwardNTT_batch -> barrett_batch -> inverseNTT_batch (bfv_encryption.cuh:268-271) on the literal
// rounds' words as they are / the exact primes' lazy values (FusedMul, ntt_core.cuh)
template <int HB, bool WANT>
__device__ __forceinline__ void lit_pass_polymul15(u64* __restrict__ a, const u64* __restrict__ bhat, const TwPair* __restrict__ twf,
                                                   const TwPair* __restrict__ twi, const PrimeDev* __restrict__ primes, unsigned division,
                                                   unsigned num, unsigned mask, const SharedB& sb, u64* lds, unsigned wave_s)
{
    constexpr int LOGN = 15;
    using G = Geo<LOGN>;
    auto fresh_t = [&]() { return (wave_s << 6) | fresh_lane_id(); };
    u64* slice = lds + wave_s * WAVE_SLICE_WORDS;
    LitWalk w(blockIdx.x, gridDim.x, division, num, mask, WANT);
    if (w.done()) return;
    u64 v[32];
    load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)w.y * G::N, fresh_t());
    while (!w.done()) {
        const unsigned idx = w.ymod;
        const PrimeDev p = primes[idx];
        const TwPair* tf = twf + (size_t)idx * G::N;
        const TwPair* ti = twi + (size_t)idx * G::N;
        const BufRsrc tfr = make_rsrc(tf, G::N * 16u), tir = make_rsrc(ti, G::N * 16u);
        u64* poly = a + (size_t)w.y * G::N;
        const BufRsrc brs = make_rsrc(bhat + (size_t)sb.index(w.y, idx, division) * G::N + wave_s * 2048u, 16384u);
        w.advance();
        __builtin_amdgcn_s_setprio(Tune::kPrioR1);
        ct_round<LOGN, HB, 10, 4, false>(v, tf, tfr, 0u, p);
        __syncthreads();
        exchange<LOGN, 10, 5>(v, lds, fresh_t());
        __builtin_amdgcn_s_setprio(Tune::kPrioR2);
        ct_round<LOGN, HB, 5, 4, false>(v, tf, tfr, fresh_t(), p);
        wave_transpose_5_to_0(v, slice, fresh_lane_id());
        __builtin_amdgcn_s_setprio(Tune::kPrioR3);
        ct_round<LOGN, HB, 0, 4, false>(v, tf, tfr, fresh_t(), p);
        {
            u64 bb[16];
            wave_load_rows_half_direct<0>(bb, slice, brs);
            static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                v[r] = FusedMul<HB, false>::mul(v[r], bb[r], p);
            });
            wave_load_rows_half_direct<1>(bb, slice, brs);
            static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                v[16 + r] = FusedMul<HB, false>::mul(v[16 + r], bb[r], p);
            });
        }
        gs_round<LOGN, HB, 0, 0, false>(v, ti, tir, fresh_t(), p, primes[idx].twn);
        wave_transpose_0_to_5(v, slice, fresh_lane_id());
        __builtin_amdgcn_s_setprio(Tune::kPrioI2);
        gs_round<LOGN, HB, 5, 0, false>(v, ti, tir, fresh_t(), p, primes[idx].twn);
        __syncthreads();
        exchange<LOGN, 5, 10>(v, lds, fresh_t());
        __builtin_amdgcn_s_setprio(Tune::kPrioI3);
        gs_round<LOGN, HB, 10, 0, false>(v, ti, tir, fresh_t(), p, primes[idx].twn);
        static_for<32>([&](auto rc) { v[decltype(rc)::value] = canon_after_inverse<HB, false>(v[decltype(rc)::value], p); });
        store_coalesced<LOGN, Tune::kInv15AuxSt>(v, poly, fresh_t());
        if (!w.done()) load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)w.y * G::N, fresh_t());
    }
}

template <int LOGN>      // (= 15: a template so that only the translation unit that launches it holds the kernel)
__global__ void __launch_bounds__(1024, 4)
k_polymul15_lit(u64* __restrict__ a, const u64* __restrict__ bhat, const TwPair* __restrict__ twf, const TwPair* __restrict__ twi,
                const PrimeDev* __restrict__ primes, unsigned division, unsigned num)
{
    const SharedB sb(division);
    static_assert(LOGN == 15, "n = 2^15 only");
    __shared__ __attribute__((aligned(16))) u64 lds[Geo<15>::LDS_WORDS];
    unsigned wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+s"(wave_s));
    const unsigned mask = lit_mask_of(primes, 0u, division);
    lit_pass_polymul15<HL_LIT, true>(a, bhat, twf, twi, primes, division, num, mask, sb, lds, wave_s);
    __syncthreads();
    lit_pass_polymul15<HL_LIT_EXACT, false>(a, bhat, twf, twi, primes, division, num, mask, sb, lds, wave_s);
}

// ================================================================================================
// n = 2^11 .. 2^14 (the shapes of k_forward / k_inverse / k_polymul, kernels_fast_impl.cuh)
// ================================================================================================
template <int LOGN, int HB, bool WANT>
__device__ __forceinline__ void lit_pass_forward(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes,
                                                 unsigned division, unsigned prime_base, unsigned num, unsigned mask, u64* lds, unsigned wave_s)
{
    using G = Geo<LOGN>;
    auto tid = [&]() { return (wave_s << 6) | fresh_lane_id(); };
    LitWalk w(blockIdx.x, gridDim.x, division, num, mask, WANT);
    if (w.done()) return;
    u64 v[32];
    load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)w.y * G::N, tid());
    while (!w.done()) {
        const unsigned idx = prime_base + w.ymod;
        const PrimeDev p = primes[idx];
        const TwPair* twp = tw + (size_t)idx * G::N;
        u64* poly = a + (size_t)w.y * G::N;
        w.advance();
        forward_core<LOGN, HB, false>(v, twp, tid, p, lds);
        static_for<32>([&](auto rc) { v[decltype(rc)::value] = canon_after_forward<HB, false>(v[decltype(rc)::value], p); });
        __syncthreads();        // every wave has read the last exchange: the image is free
        wave_store_rows(v, lds + wave_s * 1024u, make_rsrc(poly + wave_s * 2048u, 16384u), 0u, 0u);
        if (!w.done()) load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)w.y * G::N, tid());
        __syncthreads();        // the next polynomial's first exchange reuses the LDS image
    }
}

template <int LOGN>
__global__ void __launch_bounds__(Geo<LOGN>::T, 4)
k_forward_lit(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division,
              unsigned prime_base, unsigned num)
{
    if (guard_says_skip(primes, prime_base)) return;
    __shared__ __attribute__((aligned(16))) u64 lds[Geo<LOGN>::LDS_WORDS];
    unsigned wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+s"(wave_s));
    const unsigned mask = lit_mask_of(primes, prime_base, division);
    lit_pass_forward<LOGN, HL_LIT, true>(a, tw, primes, division, prime_base, num, mask, lds, wave_s);
    __syncthreads();
    lit_pass_forward<LOGN, HL_LIT_EXACT, false>(a, tw, primes, division, prime_base, num, mask, lds, wave_s);
}

template <int LOGN, int HB, bool WANT>
__device__ __forceinline__ void lit_pass_inverse(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes,
                                                 unsigned division, unsigned prime_base, unsigned num, unsigned mask, u64* lds, unsigned wave_s)
{
    using G = Geo<LOGN>;
    auto tid = [&]() { return (wave_s << 6) | fresh_lane_id(); };
    LitWalk w(blockIdx.x, gridDim.x, division, num, mask, WANT);
    if (w.done()) return;
    u64 v[32];
    wave_load_rows(v, lds + wave_s * 1024u, make_rsrc(a + (size_t)w.y * G::N + wave_s * 2048u, 16384u), 0u, 0u);
    __syncthreads();        // every wave has left its staging slice: the first exchange writes the workgroup-wide image over them
    while (!w.done()) {
        const unsigned idx = prime_base + w.ymod;
        const PrimeDev p = primes[idx];
        const TwPair* twp = tw + (size_t)idx * G::N;
        u64* poly = a + (size_t)w.y * G::N;
        w.advance();
        inverse_core<LOGN, HB, false>(v, twp, tid, p, lds, primes[idx].twn);
        static_for<32>([&](auto rc) { v[decltype(rc)::value] = canon_after_inverse<HB, false>(v[decltype(rc)::value], p); });
        store_coalesced<LOGN, Tune::kInvAuxSt>(v, poly, tid());
        __syncthreads();        // every wave has read the last exchange: the image is free for the row staging
        if (!w.done()) wave_load_rows(v, lds + wave_s * 1024u, make_rsrc(a + (size_t)w.y * G::N + wave_s * 2048u, 16384u), 0u, 0u);
        __syncthreads();
    }
}

template <int LOGN>
__global__ void __launch_bounds__(Geo<LOGN>::T, 4)
k_inverse_lit(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division,
              unsigned prime_base, unsigned num)
{
    if (guard_says_skip(primes, prime_base)) return;
    __shared__ __attribute__((aligned(16))) u64 lds[Geo<LOGN>::LDS_WORDS];
    unsigned wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+s"(wave_s));
    const unsigned mask = lit_mask_of(primes, prime_base, division);
    lit_pass_inverse<LOGN, HL_LIT, true>(a, tw, primes, division, prime_base, num, mask, lds, wave_s);
    __syncthreads();
    lit_pass_inverse<LOGN, HL_LIT_EXACT, false>(a, tw, primes, division, prime_base, num, mask, lds, wave_s);
}

// one workgroup per polynomial (as k_polymul<LOGN>): the polynomial's class is picked once, at the top -- nothing is live across the branch
template <int LOGN, int HB>
__device__ __forceinline__ void lit_polymul_one(u64* __restrict__ poly, const u64* __restrict__ bp, const TwPair* __restrict__ twf,
                                                const TwPair* __restrict__ twi, const PrimeDev* __restrict__ primes, unsigned idx, u64* lds,
                                                unsigned wave_s)
{
    using G = Geo<LOGN>;
    auto tid = [&]() { return (wave_s << 6) | fresh_lane_id(); };
    const PrimeDev p = primes[idx];
    u64 v[32];
    load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, poly, tid());
    forward_core<LOGN, HB, false>(v, twf + (size_t)idx * G::N, tid, p, lds);
    const BufRsrc brs = make_rsrc(bp, G::N * 8u);
    const unsigned boff = tid() * 256u;
#pragma unroll
    for (int r = 0; r < 32; r += 2) {
        const TwPair bb = buf_load_tw(brs, boff, (unsigned)r * 8u);      // two consecutive words of bhat
        v[r] = FusedMul<HB, false>::mul(v[r], bb.w, p);
        v[r + 1] = FusedMul<HB, false>::mul(v[r + 1], bb.wp, p);
        if ((r & 6) == 6) __builtin_amdgcn_sched_barrier(0);
    }
    inverse_core<LOGN, HB, false>(v, twi + (size_t)idx * G::N, tid, p, lds, primes[idx].twn);
#pragma unroll
    for (int r = 0; r < 32; r++) v[r] = canon_after_inverse<HB, false>(v[r], p);
    store_coalesced<LOGN, Tune::kInvAuxSt>(v, poly, tid());
}

template <int LOGN>
__global__ void __launch_bounds__(Geo<LOGN>::T, 4)
k_polymul_lit(u64* __restrict__ a, const u64* __restrict__ bhat, const TwPair* __restrict__ twf, const TwPair* __restrict__ twi,
              const PrimeDev* __restrict__ primes, unsigned division)
{
    const SharedB sb(division);
    using G = Geo<LOGN>;
    __shared__ __attribute__((aligned(16))) u64 lds[G::LDS_WORDS];
    const unsigned y = blockIdx.x;
    unsigned idx = __builtin_amdgcn_readfirstlane(y % division);
    asm volatile("" : "+s"(idx));
    unsigned wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+s"(wave_s));
    u64* poly = a + (size_t)y * G::N;
    const u64* bp = bhat + (size_t)sb.index(y, idx, division) * G::N;
    if (primes[idx].lit) lit_polymul_one<LOGN, HL_LIT>(poly, bp, twf, twi, primes, idx, lds, wave_s);
    else lit_polymul_one<LOGN, HL_LIT_EXACT>(poly, bp, twf, twi, primes, idx, lds, wave_s);
}

// ================================================================================================
// small batches (the shapes of kernels_lat.cuh: a polynomial over n/512 waves of eight coefficients per thread, two launches per
// transform, three per product).  One workgroup works on one polynomial, so the butterfly kind is picked once at the top of the
// kernel (nothing is live across the branch): the reference's own butterflies for a Barrett-inexact prime, the lazy class-2 ones
// for the exact primes of the same context.  Literal values travel between the launches as the reference's memory holds them
// between its stage launches; nothing is canonicalised and the inverse halves in every stage (no n^-1 table entries).
// ================================================================================================
template <int LOGN, int HB>
__device__ __forceinline__ void lat_lit_fwd_a(u64* __restrict__ poly, const TwPair* __restrict__ twp, const PrimeDev& p, u64* lds, unsigned g,
                                              unsigned k, unsigned lane)
{
    using L = LatGeo<LOGN>;
    const BufRsrc twr = make_rsrc(twp, L::N * 16u), prs = make_rsrc(poly, L::N * 8u);
    const unsigned voff = ((g << 6) | lane) * 8u;
    u64 v[8];
    static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; v[r] = buf_load_u64(prs, voff, ((r << (LOGN - 3)) | (k << (6 + L::GB))) * 8u); });
    lat_fwd_round<LOGN, HB, false, true, LOGN - 1, L::NST1>(v, twp, twr, p, 0u);
    if constexpr (L::NST2 > 0) {
        lat_swap_kr(v, lds, k, lane);
        lat_fwd_round<LOGN, HB, false, true, LOGN - 4, L::NST2>(v, twp, twr, p, k);
        static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; buf_store_u64(prs, voff, ((k << (LOGN - 3)) | (r << (LOGN - 6))) * 8u, v[r]); });
    } else {
        static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; buf_store_u64(prs, voff, ((r << (LOGN - 3)) | (k << (6 + L::GB))) * 8u, v[r]); });
    }
}

template <int LOGN, int HB>
__device__ __forceinline__ void lat_lit_inv_a(u64* __restrict__ poly, const TwPair* __restrict__ twp, const PrimeDev& p, const TwPair* __restrict__ twn,
                                              u64* lds, unsigned g, unsigned k, unsigned lane)
{
    using L = LatGeo<LOGN>;
    constexpr bool SCALE = (HB != HL_LIT);               // (the literal butterflies halve in every stage)
    const BufRsrc twr = make_rsrc(twp, L::N * 16u), prs = make_rsrc(poly, L::N * 8u);
    const unsigned voff = ((g << 6) | lane) * 8u;
    u64 v[8];
    if constexpr (L::NST2 > 0) {
        static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; v[r] = buf_load_u64(prs, voff, ((k << (LOGN - 3)) | (r << (LOGN - 6))) * 8u); });
        lat_inv_round<LOGN, HB, false, true, LOGN - 6, L::NST2>(v, twp, twr, p, k);
        lat_swap_kr(v, lds, k, lane);
    } else {
        static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; v[r] = buf_load_u64(prs, voff, ((r << (LOGN - 3)) | (k << (6 + L::GB))) * 8u); });
    }
    lat_inv_round<LOGN, HB, false, true, LOGN - 3, L::NST1, SCALE>(v, twp, twr, p, 0u, twn);
    static_for<8>([&](auto rc) {
        constexpr unsigned r = decltype(rc)::value;
        buf_store_u64(prs, voff, ((r << (LOGN - 3)) | (k << (6 + L::GB))) * 8u, canon_after_inverse<HB, false>(v[r], p));
    });
}

template <int LOGN>
__global__ void __launch_bounds__(LatGeo<LOGN>::WA, 1)
k_lat_fwd_a_lit(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division, unsigned prime_base)
{
    if (guard_says_skip(primes, prime_base)) return;
    using L = LatGeo<LOGN>;
    __shared__ u64 lds[L::NST2 ? 4096 : 1];
    const unsigned y = blockIdx.x >> L::GB, g = blockIdx.x & ((1u << L::GB) - 1u);
    const unsigned k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    unsigned idx = __builtin_amdgcn_readfirstlane(prime_base + y % division);
    asm volatile("" : "+s"(idx));
    const PrimeDev p = primes[idx];
    u64* poly = a + (size_t)y * L::N;
    const TwPair* twp = tw + (size_t)idx * L::N;
    if (primes[idx].lit) lat_lit_fwd_a<LOGN, HL_LIT>(poly, twp, p, lds, g, k, lane);
    else lat_lit_fwd_a<LOGN, HL_LIT_EXACT>(poly, twp, p, lds, g, k, lane);
}

template <int LOGN>
__global__ void __launch_bounds__(LatGeo<LOGN>::WA, 1)
k_lat_inv_a_lit(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division, unsigned prime_base)
{
    if (guard_says_skip(primes, prime_base)) return;
    using L = LatGeo<LOGN>;
    __shared__ u64 lds[L::NST2 ? 4096 : 1];
    const unsigned y = blockIdx.x >> L::GB, g = blockIdx.x & ((1u << L::GB) - 1u);
    const unsigned k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    unsigned idx = __builtin_amdgcn_readfirstlane(prime_base + y % division);
    asm volatile("" : "+s"(idx));
    const PrimeDev p = primes[idx];
    u64* poly = a + (size_t)y * L::N;
    const TwPair* twp = tw + (size_t)idx * L::N;
    if (primes[idx].lit) lat_lit_inv_a<LOGN, HL_LIT>(poly, twp, p, primes[idx].twn, lds, g, k, lane);
    else lat_lit_inv_a<LOGN, HL_LIT_EXACT>(poly, twp, p, primes[idx].twn, lds, g, k, lane);
}

// the "b" kernels: one wave on 512 consecutive coefficients (index bits 8 .. 0)
template <int LOGN, int HB>
__device__ __forceinline__ void lat_lit_fwd_b(const BufRsrc prs, const TwPair* __restrict__ twp, const PrimeDev& p, u64* slice, unsigned c, unsigned lane)
{
    const BufRsrc twr = make_rsrc(twp, (1u << LOGN) * 16u);
    u64 v[8];
    lat_load_l6(v, prs, c, lane);
    lat_fwd_b_rounds<LOGN, HB, false>(v, twp, twr, p, slice, c, lane);
    static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; v[r] = canon_after_forward<HB, false>(v[r], p); });
    lat_store_l0(v, prs, c, lane);
}
template <int LOGN, int HB>
__device__ __forceinline__ void lat_lit_inv_b(const BufRsrc prs, const TwPair* __restrict__ twp, const PrimeDev& p, u64* slice, unsigned c, unsigned lane)
{
    const BufRsrc twr = make_rsrc(twp, (1u << LOGN) * 16u);
    u64 v[8];
    lat_load_l0(v, prs, c, lane);
    lat_inv_b_rounds<LOGN, HB, false>(v, twp, twr, p, slice, c, lane);
    lat_store_l6(v, prs, c, lane);
}
template <int LOGN, int HB>
__device__ __forceinline__ void lat_lit_mul_b(const BufRsrc prs, const BufRsrc brs, const TwPair* __restrict__ tf, const TwPair* __restrict__ ti,
                                              const PrimeDev& p, u64* slice, unsigned c, unsigned lane)
{
    const BufRsrc tfr = make_rsrc(tf, (1u << LOGN) * 16u), tir = make_rsrc(ti, (1u << LOGN) * 16u);
    u64 v[8], bb[8];
    lat_load_l6(v, prs, c, lane);
    lat_load_l0(bb, brs, c, lane);
    lat_fwd_b_rounds<LOGN, HB, false>(v, tf, tfr, p, slice, c, lane);
    static_for<8>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        v[r] = FusedMul<HB, false>::mul(v[r], bb[r], p);
    });
    lat_inv_b_rounds<LOGN, HB, false, FusedMul<HB, false>::LAZY>(v, ti, tir, p, slice, c, lane);
    lat_store_l6(v, prs, c, lane);
}

template <int LOGN>
__global__ void __launch_bounds__(64, 1)
k_lat_fwd_b_lit(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division, unsigned prime_base)
{
    if (guard_says_skip(primes, prime_base)) return;
    __shared__ u64 slice[LAT_SLICE_WORDS];
    const unsigned y = blockIdx.x >> (LOGN - 9), c = blockIdx.x & ((1u << (LOGN - 9)) - 1u), lane = threadIdx.x;
    unsigned idx = __builtin_amdgcn_readfirstlane(prime_base + y % division);
    asm volatile("" : "+s"(idx));
    const PrimeDev p = primes[idx];
    const TwPair* twp = tw + (size_t)idx * (1u << LOGN);
    const BufRsrc prs = make_rsrc(a + (size_t)y * (1u << LOGN), (1u << LOGN) * 8u);
    if (primes[idx].lit) lat_lit_fwd_b<LOGN, HL_LIT>(prs, twp, p, slice, c, lane);
    else lat_lit_fwd_b<LOGN, HL_LIT_EXACT>(prs, twp, p, slice, c, lane);
}

template <int LOGN>
__global__ void __launch_bounds__(64, 1)
k_lat_inv_b_lit(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division, unsigned prime_base)
{
    if (guard_says_skip(primes, prime_base)) return;
    __shared__ u64 slice[LAT_SLICE_WORDS];
    const unsigned y = blockIdx.x >> (LOGN - 9), c = blockIdx.x & ((1u << (LOGN - 9)) - 1u), lane = threadIdx.x;
    unsigned idx = __builtin_amdgcn_readfirstlane(prime_base + y % division);
    asm volatile("" : "+s"(idx));
    const PrimeDev p = primes[idx];
    const TwPair* twp = tw + (size_t)idx * (1u << LOGN);
    const BufRsrc prs = make_rsrc(a + (size_t)y * (1u << LOGN), (1u << LOGN) * 8u);
    if (primes[idx].lit) lat_lit_inv_b<LOGN, HL_LIT>(prs, twp, p, slice, c, lane);
    else lat_lit_inv_b<LOGN, HL_LIT_EXACT>(prs, twp, p, slice, c, lane);
}

template <int LOGN>
__global__ void __launch_bounds__(64, 1)
k_lat_mul_b_lit(u64* __restrict__ a, const u64* __restrict__ bhat, const TwPair* __restrict__ twf, const TwPair* __restrict__ twi,
                const PrimeDev* __restrict__ primes, unsigned division)
{
    const SharedB sb(division);
    __shared__ u64 slice[LAT_SLICE_WORDS];
    const unsigned y = blockIdx.x >> (LOGN - 9), c = blockIdx.x & ((1u << (LOGN - 9)) - 1u), lane = threadIdx.x;
    unsigned idx = __builtin_amdgcn_readfirstlane(y % division);
    asm volatile("" : "+s"(idx));
    const PrimeDev p = primes[idx];
    const TwPair* tf = twf + (size_t)idx * (1u << LOGN);
    const TwPair* ti = twi + (size_t)idx * (1u << LOGN);
    const BufRsrc prs = make_rsrc(a + (size_t)y * (1u << LOGN), (1u << LOGN) * 8u);
    const BufRsrc brs = make_rsrc(bhat + (size_t)sb.index(y, idx, division) * (1u << LOGN), (1u << LOGN) * 8u);
    if (primes[idx].lit) lat_lit_mul_b<LOGN, HL_LIT>(prs, brs, tf, ti, p, slice, c, lane);
    else lat_lit_mul_b<LOGN, HL_LIT_EXACT>(prs, brs, tf, ti, p, slice, c, lane);
}

}  // namespace mi355ntt
