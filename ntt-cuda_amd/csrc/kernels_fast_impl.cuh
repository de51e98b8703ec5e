// kernels_fast_impl.cuh -- kernel templates of the single-pass path, instantiated once per ring degree in
// kernels_fast_n<LOGN>.hip so the five sizes compile in parallel.
#pragma once
#include "kernels.hpp"
#include "modarith.cuh"
#include "ntt_core.cuh"

#include <cstdlib>

namespace mi355ntt {

// coalesced layout B0: register r of thread t holds coefficient (r << B0) | t; lane offset in a VGPR,
// the r * (n/32) * 8 byte displacement in the buffer instruction's scalar offset
// PAIR16: issue order (0, 16, 1, 17, ...) -- the order in which a forward round on register bits 4..0 consumes the
// registers (its first stage pairs r with r + 16; loads return in order, so the first butterfly can start after two loads
// have landed instead of seventeen)
template <int LOGN, bool PAIR16 = false>
__device__ __forceinline__ void load_coalesced(u64 (&v)[32], const u64* __restrict__ poly, unsigned t)
{
    const BufRsrc rs = make_rsrc(poly, Geo<LOGN>::N * 8u);
    static_for<32>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int r = PAIR16 ? ((i >> 1) | ((i & 1) << 4)) : i;
        v[r] = buf_load_u64(rs, t * 8u, ((unsigned)r << Geo<LOGN>::B0) * 8u);
        if constexpr (PAIR16) __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise re-sorts the loads by register)
    });
}

// AUX: cache policy of the stores (the inverse / fused kernels' results are written through: Tune::kInvAuxSt, kInv15AuxSt)
constexpr int kStoreCoalescedOps = 32;      // vector-memory instructions store_coalesced issues per lane (counted waits behind it rely on this)
template <int LOGN, int AUX = Tune::kStreamAuxSt>
__device__ __forceinline__ void store_coalesced(const u64 (&v)[32], u64* __restrict__ poly, unsigned t)
{
    static_assert(kStoreCoalescedOps == 32, "one buffer_store_dwordx2 per register");
    const BufRsrc rs = make_rsrc(poly, Geo<LOGN>::N * 8u);
#pragma unroll
    for (int r = 0; r < 32; r++) {
        v2u32 x;
        x.x = lo32(v[r]);
        x.y = hi32(v[r]);
        __builtin_amdgcn_raw_buffer_store_b64(x, rs, t * 8u, ((unsigned)r << Geo<LOGN>::B0) * 8u, AUX);
    }
}

// Persistent workgroups: grid = min(num, resident workgroups); each workgroup walks polynomials
// y = blockIdx.x, blockIdx.x + gridDim.x, ...  The loads of the next polynomial are issued right behind the
// stores of the current one, so the write drain, the dispatch gap and the read latency overlap.

// ---- forward: natural -> bit-reversed, canonical ------------------------------------------------
template <int LOGN, int HL, bool NEAR>
__global__ void __launch_bounds__(Geo<LOGN>::T, 4)
k_forward(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division,
          unsigned prime_base, unsigned num)
{
    if (guard_says_skip(primes, prime_base)) return;      // checked raw call whose table is not the cached one
    using G = Geo<LOGN>;
    __shared__ __attribute__((aligned(16))) u64 lds[G::LDS_WORDS];
    // thread-derived values are rebuilt where they are used (wave index in an SGPR, lane index from v_mbcnt), as in the
    // n = 2^15 kernels: one VGPR kept live across the polynomial loop was a spill in several instantiations
    unsigned wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+s"(wave_s));
    auto tid = [&]() { return (wave_s << 6) | fresh_lane_id(); };
    u64 v[32];
    unsigned y = blockIdx.x;
    load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)y * G::N, tid());
    // modulus index of polynomial y, carried along in SGPRs instead of y % division per iteration (a division by a runtime
    // value is a multi-instruction VALU sequence; its reciprocal sat in a VGPR across the loop and was the forward kernel's
    // spill, reloaded behind a vmcnt(0) that also waited for the next polynomial's loads)
    unsigned ymod = __builtin_amdgcn_readfirstlane(blockIdx.x % division), ystep = __builtin_amdgcn_readfirstlane(gridDim.x % division);
    asm volatile("" : "+s"(ymod), "+s"(ystep));          // in SGPRs from here on (the quotient sequence itself runs on the VALU)
    for (; y < num; y += gridDim.x, ymod = (ymod + ystep >= division ? ymod + ystep - division : ymod + ystep)) {
        const unsigned idx = prime_base + ymod;
        const PrimeDev p = primes[idx];
        const TwPair* twp = tw + (size_t)idx * G::N;
        u64* poly = a + (size_t)y * G::N;
        forward_core<LOGN, HL, NEAR>(v, twp, tid, p, lds);
        static_for<32>([&](auto rc) { v[decltype(rc)::value] = canon_after_forward<HL, NEAR>(v[decltype(rc)::value], p); });
        // layout 0 (32 consecutive words per thread) leaves through the wave's own 8 KiB of the image with 16-byte stores
        // (as on n = 2^15) instead of a workgroup-wide layout exchange and 8-byte stores
        __syncthreads();        // every wave has read the last exchange: the image is free
        wave_store_rows(v, lds + wave_s * 1024u, make_rsrc(poly + wave_s * 2048u, 16384u), 0u, 0u);
        if (y + gridDim.x < num) load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)(y + gridDim.x) * G::N, tid());
        __syncthreads();        // the next polynomial's first exchange reuses the LDS image
    }
}

// ---- inverse: bit-reversed -> natural, scaled by n^-1, canonical --------------------------------
template <int LOGN, int HL, bool NEAR>
__global__ void __launch_bounds__(Geo<LOGN>::T, 4)
k_inverse(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division,
          unsigned prime_base, unsigned num)
{
    if (guard_says_skip(primes, prime_base)) return;      // checked raw call whose table is not the cached one
    using G = Geo<LOGN>;
    __shared__ __attribute__((aligned(16))) u64 lds[G::LDS_WORDS];
    unsigned wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (thread-derived values: see k_forward)
    asm volatile("" : "+s"(wave_s));
    auto tid = [&]() { return (wave_s << 6) | fresh_lane_id(); };
    u64 v[32];
    unsigned y = blockIdx.x;
    // rows (16-byte loads) through the wave's own 8 KiB of the image straight into layout 0: no layout exchange
    wave_load_rows(v, lds + wave_s * 1024u, make_rsrc(a + (size_t)y * G::N + wave_s * 2048u, 16384u), 0u, 0u);
    __syncthreads();        // every wave has left its staging slice: the first exchange writes the workgroup-wide image over them
                            // (later iterations are covered by the barrier at the loop's end)
    // modulus index of polynomial y, carried along in SGPRs instead of y % division per iteration (a division by a runtime
    // value is a multi-instruction VALU sequence; its reciprocal sat in a VGPR across the loop and was the forward kernel's
    // spill, reloaded behind a vmcnt(0) that also waited for the next polynomial's loads)
    unsigned ymod = __builtin_amdgcn_readfirstlane(blockIdx.x % division), ystep = __builtin_amdgcn_readfirstlane(gridDim.x % division);
    asm volatile("" : "+s"(ymod), "+s"(ystep));          // in SGPRs from here on (the quotient sequence itself runs on the VALU)
    for (; y < num; y += gridDim.x, ymod = (ymod + ystep >= division ? ymod + ystep - division : ymod + ystep)) {
        const unsigned idx = prime_base + ymod;
        const PrimeDev p = primes[idx];
        const TwPair* twp = tw + (size_t)idx * G::N;
        u64* poly = a + (size_t)y * G::N;
        inverse_core<LOGN, HL, NEAR>(v, twp, tid, p, lds, primes[idx].twn);
        static_for<32>([&](auto rc) { v[decltype(rc)::value] = canon_after_inverse<HL, NEAR>(v[decltype(rc)::value], p); });
        store_coalesced<LOGN, Tune::kInvAuxSt>(v, poly, tid());
        __syncthreads();        // every wave has read the last exchange: the image is free for the row staging
        if (y + gridDim.x < num)
            wave_load_rows(v, lds + wave_s * 1024u, make_rsrc(a + (size_t)(y + gridDim.x) * G::N + wave_s * 2048u, 16384u), 0u, 0u);
        __syncthreads();
    }
}

// Wave priorities per phase (s_setprio; priority outranks age in the SIMD's issue arbitration; values: Tune::kPrio*, tune.hpp).
// Between two workgroup-wide exchanges the SIMD arbitrates strictly oldest-first, so its four waves run almost one after the other
// and the last one finishes alone.  Lowering a wave's priority as it progresses past the exchange (phase right after the exchange
// highest, the round that feeds the next exchange lowest) lets the waves that are behind catch up: +3 % (1024 polynomials) to +7 %
// (8192) on k_forward15.
//
// Start-time stagger of the persistent workgroups: 8 phase groups, UNITS x 2048 cycles apart (Tune::kStagger*).  Every workgroup
// does the same work, so without it all CUs load and store in the same instants and HBM sees bursts instead of a steady stream.
template <int UNITS, int UNITS_MULTI = UNITS>
__device__ __forceinline__ void stagger_start(bool multi = false)
{
    if constexpr (UNITS > 0 || UNITS_MULTI > 0) {
        const unsigned ph = (blockIdx.x >> 3) & 7u;
        const unsigned n = ph * (multi ? (unsigned)UNITS_MULTI : (unsigned)UNITS);
        for (unsigned i = 0; i < n; i++) __builtin_amdgcn_s_sleep(32);
    }
}
// k_inverse15's division word: kStreamLoads set for launches that stream (batches of kInvStreamLoadsMin polynomials = 1 GiB and more):
// their 16-byte row loads carry the non-temporal hint (Tune::kInv15AuxLd).  Only the load instructions exist twice in the code
// (wave_load_rows_half, alt).
inline unsigned inv15_division_word(unsigned division, unsigned num) { return division | (num >= kInvStreamLoadsMin ? kStreamLoads : 0u); }
// INV_POS(y): position of the y-th polynomial k_inverse15 walks: from the last polynomial down (Tune::kInvDescending).  A forward transform is
// normally followed by an inverse over the same polynomials (and the other way round): walking them in opposite directions makes
// each kernel start on what the previous one wrote last, i.e. on what is still in the memory-side cache.
#define INV_POS(y) (Tune::kInvDescending ? num - 1u - (y) : (y))      // (`num`: the kernel's argument)

// ================================================================================================
// n = 2^15: one workgroup-wide exchange per transform, everything else wave-local (ntt_core.cuh).
// forward : load(layout 10) R1 | sync, exchange 10->5 | R2 | wave transpose 5->0 | R3 | canon | wave-local row store
// inverse : wave-local row load (layout 0) | R1' | wave transpose 0->5 | R2' | sync, exchange 5->10 | R3' | canon | store
// ================================================================================================
// The fused coupling stage of SPLIT = 1 reads the partner half through the wave's private LDS slice: `buffer_load_dwordx4 ... lds`
// (LDS-direct: the data never touches a VGPR; lane L's 16 bytes land at M0 + 16 L whatever its global offset -- probed on the
// hardware, tools/probe/lds_direct.hip) moves two 512-byte partner rows per instruction, lanes 0..31 the row r and lanes 32..63 the
// row r + 1 of this wave's 64 columns.  Four phases of eight rows, double-buffered in the slice's first 8 KiB: phases 0 and 1 are
// requested at the tail of the previous iteration, phase c + 2 as soon as phase c has been read out.  (Through registers, eight partner
// loads per thread at a time, the stage cost 24 us per polynomial -- four exposed HBM latencies: r03 first version.)
typedef __attribute__((address_space(3))) void* LdsPtr;
template <int PH>
__device__ __forceinline__ void split_partner_fetch(BufRsrc hrs, u64* slice, unsigned wave_s, unsigned lane)
{
    const unsigned voff = ((lane & 32u) << 8) | ((lane & 31u) << 4);           // upper half-wave: the next row (8192 B further)
    static_for<4>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(hrs, (LdsPtr)(slice + (PH & 1) * 512 + j * 128), 16, voff,
                                                 ((unsigned)((8 * PH + 2 * j) << Geo<15>::B0) + (wave_s << 6)) * 8u, 0, 0);
    });
}
// (row 8 PH + i of this wave, this lane's column)
template <int PH>
__device__ __forceinline__ void split_partner_read(u64 (&V)[8], const u64* slice, unsigned lane)
{
    static_for<8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        V[i] = slice[(PH & 1) * 512 + (i >> 1) * 128 + (i & 1) * 64 + lane];
    });
}

// SPLIT (n = 2^16 contexts: two half-size transforms per polynomial y, halves h = 0 / 1 at a + (2y + h) 2^15, tables and constants
// of "virtual prime" 2 i + h): the workgroup transforms both halves of its polynomial one after the other.  Before the rounds of
// the lower half it reads the partner coefficients V of the upper half, forms T = V w (the stage that couples the halves), stores
// U - T (canonical: the upper half's input) in place over V and keeps U + T in registers as its own input (in [0, 2q): every
// class's bound tracking admits that, see fwd_reduce_mask); the upper half is then an ordinary transform of what the same threads
// stored.  Reads 1.5 x, writes 1.5 x the polynomial in ONE launch instead of 2 x / 2 x in two (a stage kernel in front).
template <int HL, bool NEAR, int SPLIT = 0>
__global__ void __launch_bounds__(1024, 4)
k_forward15(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division,
            unsigned prime_base, unsigned num)
{
    if (guard_says_skip(primes, prime_base)) return;      // checked raw call whose table is not the cached one
    constexpr int LOGN = 15;
    using G = Geo<LOGN>;
    __shared__ __attribute__((aligned(16))) u64 lds[G::LDS_WORDS];
    // Thread-derived values are rebuilt where they are used -- the wave index lives in an SGPR, the lane index comes from
    // v_mbcnt -- instead of surviving the polynomial loop in VGPRs the kernel does not have (they were its spills, reloaded
    // from scratch in front of the exchange and of the row store).
    unsigned wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+s"(wave_s));
    auto fresh_t = [&]() { return (wave_s << 6) | fresh_lane_id(); };
    u64 v[32];
    unsigned y = blockIdx.x;
    stagger_start<Tune::kStaggerFwd, Tune::kStaggerFwdMulti>(num > gridDim.x);
    // (SPLIT: half h of polynomial y is the half-size polynomial 2 y + h)
    [[maybe_unused]] unsigned h = 0;
    auto half_of = [](unsigned yy, unsigned hh) { return SPLIT ? 2 * yy + hh : yy; };
    load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)(half_of(y, 0)) * G::N, fresh_t());
    if constexpr (SPLIT != 0) {
        const BufRsrc hrs = make_rsrc(a + (size_t)(half_of(y, 0) + 1) * G::N, G::N * 8u);
        split_partner_fetch<0>(hrs, lds + wave_s * WAVE_SLICE_WORDS, wave_s, fresh_lane_id());
        split_partner_fetch<1>(hrs, lds + wave_s * WAVE_SLICE_WORDS, wave_s, fresh_lane_id());
    }
    // modulus index of polynomial y, carried along in SGPRs instead of y % division per iteration (a division by a runtime
    // value is a multi-instruction VALU sequence; its reciprocal sat in a VGPR across the loop and was the forward kernel's
    // spill, reloaded behind a vmcnt(0) that also waited for the next polynomial's loads)
    unsigned ymod = __builtin_amdgcn_readfirstlane(blockIdx.x % division), ystep = __builtin_amdgcn_readfirstlane(gridDim.x % division);
    asm volatile("" : "+s"(ymod), "+s"(ystep));          // in SGPRs from here on (the quotient sequence itself runs on the VALU)
    while (y < num) {
        const unsigned ynext = y + gridDim.x;
        const unsigned idx = SPLIT ? 2 * (prime_base + ymod) + h : prime_base + ymod;
        const PrimeDev p = primes[idx];
        const TwPair* twp = tw + (size_t)idx * G::N;
        const BufRsrc twr = make_rsrc(twp, G::N * 16u);
        u64* poly = a + (size_t)(half_of(y, h)) * G::N;
        if constexpr (SPLIT != 0) if (h == 0) {
            // the stage that couples the halves, eight partner rows at a time (phases 0 and 1 were requested together with this
            // polynomial's own loads, at the tail of the previous iteration; phase c + 2 goes out as soon as phase c is read)
            const BufRsrc hrs = make_rsrc(poly + G::N, G::N * 8u);
            u64* slice = lds + wave_s * WAVE_SLICE_WORDS;
            const u64 cq = (u64)Lazy<HL>::TQ * p.q;
            static_for<4>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                // hipcc does not order an LDS read behind the LDS-direct load that fills it: counted wait for phase c's four
                // loads (loads, stores and LDS-direct loads retire in issue order on one counter).  Younger than them in the
                // queue: phase 1's loads (c = 0); phase 2's loads and phase 0's eight stores (c = 1); those, phase 3's loads
                // and phase 1's stores (c = 2); the stores of phases 1 and 2 (c = 3)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (c == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if constexpr (c == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if constexpr (c == 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                u64 PV[8];
                split_partner_read<c>(PV, slice, fresh_lane_id());
                wave_lds_fence();                         // the buffer is free ...
                if constexpr (c < 2) split_partner_fetch<c + 2>(hrs, slice, wave_s, fresh_lane_id());     // ... for phase c + 2
                __builtin_amdgcn_sched_barrier(0);
                const unsigned voff = fresh_t() * 8u;
                static_for<8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, r = 8 * c + i;
                    const u64 U = v[r], Vi = PV[i];
                    const u64 Tm = Lazy<HL>::EXACT ? mul_shoup2(Vi, p.sf, p.sf_p, p.nq) : mul_shoup4m<true>(Vi, p.sf, p.sf_p, p.nq);
                    buf_store_u64(hrs, voff, ((unsigned)r << G::B0) * 8u, canon_2q(reduce_2q_sel<NEAR>(U + cq - Tm, p), p.q));
                    v[r] = reduce_2q_sel<NEAR>(U + Tm, p);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        __builtin_amdgcn_s_setprio(Tune::kPrioR1);
        ct_round<LOGN, HL, 10, 4, NEAR, Tune::kPsplitR1, Tune::kPrioR1B>(v, twp, twr, 0u, p);   // (round 1 reads no thread-derived value)
        __syncthreads();                                  // every wave has left its private slice (previous polynomial)
        exchange<LOGN, 10, 5>(v, lds, fresh_t());
        __builtin_amdgcn_s_setprio(Tune::kPrioR2);
        ct_round<LOGN, HL, 5, 4, NEAR, Tune::kPsplitR2, Tune::kPrioR2B>(v, twp, twr, fresh_t(), p);
        wave_transpose_5_to_0(v, lds + wave_s * WAVE_SLICE_WORDS, fresh_lane_id());
        __builtin_amdgcn_s_setprio(Tune::kPrioR3);
        ct_round<LOGN, HL, 0, 4, NEAR, Tune::kPsplitR3, Tune::kPrioR3B>(v, twp, twr, fresh_t(), p);
        static_for<32>([&](auto rc) { v[decltype(rc)::value] = canon_after_forward<HL, NEAR>(v[decltype(rc)::value], p); });
        // (the wave's 16 KiB chunk goes into the descriptor's base: scalar arithmetic instead of a VGPR kept live across the loop)
        wave_store_rows(v, lds + wave_s * WAVE_SLICE_WORDS, make_rsrc(poly + wave_s * 2048u, 16384u), 0u, 0u);
        if constexpr (SPLIT != 0) {
            if (h == 0) {
                // next: the upper half of the same polynomial -- what this thread stored in the coupling stage above
                load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, poly + G::N, fresh_t());
            } else if (ynext < num) {
                load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)(half_of(ynext, 0)) * G::N, fresh_t());
                const BufRsrc hrs = make_rsrc(a + (size_t)(half_of(ynext, 0) + 1) * G::N, G::N * 8u);
                split_partner_fetch<0>(hrs, lds + wave_s * WAVE_SLICE_WORDS, wave_s, fresh_lane_id());
                split_partner_fetch<1>(hrs, lds + wave_s * WAVE_SLICE_WORDS, wave_s, fresh_lane_id());
            }
        } else {
            if (ynext < num) load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)(ynext) * G::N, fresh_t());
        }
        if (SPLIT != 0 && h == 0) {
            h = 1;                       // same polynomial, upper half
        } else {
            h = 0;
            y = ynext;
            ymod = (ymod + ystep >= division ? ymod + ystep - division : ymod + ystep);
        }
    }
}

// n = 2^16 forward, cooperative: TWO workgroups per polynomial, one per output half (role 0 = lower, 1 = upper).  Both read the
// whole polynomial the same way -- the lower half U into registers, the upper half V through the LDS slices (split_partner_fetch)
// -- and form U + V w' for the stage that couples the halves: w' = psi^bitrev(1) for role 0, its negative for role 1 (the `sf` of
// the role's virtual prime, capi.cpp), so the two roles run the same code.  Nothing is written before the rounds, so the
// polynomial comes from HBM once (the second reader of a pair -- same XCD, see the mapping -- hits what the first has just pulled
// through the L2) and is written once: 1 x / 1 x instead of the 1.5 x / 1.5 x of the single-workgroup form (k_forward15 SPLIT).
// The transform is in place, so a workgroup must not store its half of the result before the partner has READ the input
// underneath: one flag per workgroup, `flags[2 pair + role]` = number of polynomials this workgroup has read completely (written
// after the barrier behind round 1, polled before the row store ~25 us later; agent scope, relaxed: what the flag protects is a
// write-after-read hazard -- the partner's loads have returned (their values are consumed in front of the barrier) before it
// writes the flag, and this workgroup's stores are issued only after it has seen the flag; no data travels with the flag, so
// there is nothing to release or acquire).  The flags are zero between launches
// (each workgroup clears the one it polls on exit); the buffer belongs to one stream at a time (kernels_fast.hip).  Grid: an even
// number of workgroups, all resident (one per CU).
template <int HL, bool NEAR>
__global__ void __launch_bounds__(1024, 4)
k_forward15_pair(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division,
                 unsigned prime_base, unsigned num, unsigned* __restrict__ flags)
{
    if (guard_says_skip(primes, prime_base)) return;      // checked raw call whose table is not the cached one
    constexpr int LOGN = 15;
    using G = Geo<LOGN>;
    __shared__ __attribute__((aligned(16))) u64 lds[G::LDS_WORDS];
    unsigned wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+s"(wave_s));
    auto fresh_t = [&]() { return (wave_s << 6) | fresh_lane_id(); };
    u64* slice = lds + wave_s * WAVE_SLICE_WORDS;
    // workgroups are dealt round-robin over the 8 XCDs: w and w + 8 share an L2 (when the grid is a multiple of 16)
    const unsigned w = blockIdx.x;
    const bool xcd_map = (gridDim.x & 15u) == 0;
    unsigned role = __builtin_amdgcn_readfirstlane(xcd_map ? (w >> 3) & 1u : w & 1u);
    unsigned pair = __builtin_amdgcn_readfirstlane(xcd_map ? ((w >> 4) << 3) | (w & 7u) : w >> 1);
    asm volatile("" : "+s"(role), "+s"(pair));
    const unsigned npairs = gridDim.x >> 1;
    // (flag addresses are rebuilt from SGPRs where they are used: as 64-bit pointers they live in VGPRs across the loop -- spills)
    auto flag_at = [&](unsigned which) {
        unsigned f = 2 * pair + which;
        asm volatile("" : "+s"(f));
        return flags + f;
    };
    unsigned it = 0;
    // the partner has read the input under this result?  (long since, normally.  A partner that never shows up -- 30 s of wall clock,
    // kernels.hpp -- means the grid is not resident as a whole, which the launch rules exclude.  The workgroup then gives up: it
    // marks the launch dead -- every workgroup that finds the mark while it waits for a partner, in this launch or in pair launches
    // queued behind it, ends within 100 us instead of waiting out its own watchdog -- writes the host-visible error word and ends; no hang, no store over words the partner still needs,
    // and no trap: the process keeps its device context and learns of the failure at its next pair call, kernels.hpp.)
    auto wait_for_partner = [&]() {
        unsigned* const partner_flag = flag_at(1u - role);
        if (__hip_atomic_load(partner_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > it) return;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // (watchdog by the constant 100 MHz clock, not by iterations)
        while (__hip_atomic_load(partner_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= it) {
            __builtin_amdgcn_s_sleep(8);
            pair_watchdog_check(flags, t0);               // (ends the wave when the launch is dead or the watchdog has expired)
        }
    };
    unsigned y = pair;
    if (y >= num) return;
    // (no check of the dead mark here: an s_load ... glc + return in front of the first loads cost the kernel a third of its time --
    // 0.281 against 0.184 ms per 512 polynomials, measured -- and a workgroup of a dead launch notices in its poll loop anyway)
    {
        const unsigned ph = (w >> 4) & 7u, units = ph * (num > npairs ? (unsigned)Tune::kStaggerFwdMulti : (unsigned)Tune::kStaggerFwd);
        for (unsigned i = 0; i < units; i++) __builtin_amdgcn_s_sleep(32);
    }
    u64 v[32];
    auto request = [&](unsigned yy) {                     // everything of polynomial yy: U into registers, V (phases 0, 1) into the slice
        load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)(2 * yy) * G::N, fresh_t());
        const BufRsrc hrs = make_rsrc(a + (size_t)(2 * yy + 1u) * G::N, G::N * 8u);
        split_partner_fetch<0>(hrs, slice, wave_s, fresh_lane_id());
        split_partner_fetch<1>(hrs, slice, wave_s, fresh_lane_id());
    };
    unsigned ymod = __builtin_amdgcn_readfirstlane(pair % division), ystep = __builtin_amdgcn_readfirstlane(npairs % division);
    asm volatile("" : "+s"(ymod), "+s"(ystep));
    // (the loop is entered one pass early, with nothing to transform yet: ONE request site -- two of them, in front of the loop and
    // at its tail, meet at the back edge with different register assignments, and the fix-up parks loaded words in scratch behind
    // a full wait)
    bool have = false;
    unsigned ynext = y;
    for (;;) {
        if (have) {
        const unsigned idx = 2 * (prime_base + ymod) + role;
        const PrimeDev p = primes[idx];
        const TwPair* twp = tw + (size_t)idx * G::N;
        const BufRsrc twr = make_rsrc(twp, G::N * 16u);
        {
            // the stage that couples the halves, eight rows of V at a time (phases 0 and 1 were requested together with U; phase
            // c + 2 goes out as soon as phase c is read).  Counted waits, see k_forward15: younger than phase c's loads are the
            // four of phase c + 1 (none behind phase 3)
            const BufRsrc hrs = make_rsrc(a + (size_t)(2 * y + 1u) * G::N, G::N * 8u);
            static_for<4>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (c < 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                u64 PV[8];
                split_partner_read<c>(PV, slice, fresh_lane_id());
                wave_lds_fence();                         // the buffer is free ...
                if constexpr (c < 2) split_partner_fetch<c + 2>(hrs, slice, wave_s, fresh_lane_id());     // ... for phase c + 2
                __builtin_amdgcn_sched_barrier(0);
                static_for<8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, r = 8 * c + i;
                    const u64 Tm = Lazy<HL>::EXACT ? mul_shoup2(PV[i], p.sf, p.sf_p, p.nq) : mul_shoup4m<true>(PV[i], p.sf, p.sf_p, p.nq);
                    v[r] = reduce_2q_sel<NEAR>(v[r] + Tm, p);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        __builtin_amdgcn_s_setprio(Tune::kPrioR1);
        ct_round<LOGN, HL, 10, 4, NEAR, Tune::kPsplitR1, Tune::kPrioR1B>(v, twp, twr, 0u, p);
        __syncthreads();                                  // every wave has left its private slice -- and holds its share of the input
        if (threadIdx.x == 0) __hip_atomic_store(flag_at(role), it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        exchange<LOGN, 10, 5>(v, lds, fresh_t());
        __builtin_amdgcn_s_setprio(Tune::kPrioR2);
        ct_round<LOGN, HL, 5, 4, NEAR, Tune::kPsplitR2, Tune::kPrioR2B>(v, twp, twr, fresh_t(), p);
        wave_transpose_5_to_0(v, slice, fresh_lane_id());
        __builtin_amdgcn_s_setprio(Tune::kPrioR3);
        ct_round<LOGN, HL, 0, 4, NEAR, Tune::kPsplitR3, Tune::kPrioR3B>(v, twp, twr, fresh_t(), p);
        static_for<32>([&](auto rc) { v[decltype(rc)::value] = canon_2q(reduce_2q_sel<NEAR>(v[decltype(rc)::value], p), p.q); });
        wait_for_partner();                               // (a wave that gives up ends here: nothing is stored; ended waves leave the workgroup's barriers)
        asm volatile("" ::: "memory");                    // (compiler-level order: nothing of the row store moves above the poll)
        wave_store_rows(v, slice, make_rsrc(a + (size_t)(2 * y + role) * G::N + wave_s * 2048u, 16384u), 0u, 0u);
        it++;
        ymod = (ymod + ystep >= division ? ymod + ystep - division : ymod + ystep);
        }
        if (ynext >= num) break;
        request(ynext);
        have = true;
        y = ynext;
        ynext = y + npairs;
    }
    __syncthreads();                                      // every wave has polled for the last time:
    if (threadIdx.x == 0) __hip_atomic_store(flag_at(1u - role), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // zero between launches
}

template <int HL, bool NEAR>
__global__ void __launch_bounds__(1024, 4)
k_inverse15(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division,
            unsigned prime_base, unsigned num)
{
    if (guard_says_skip(primes, prime_base)) return;      // checked raw call whose table is not the cached one
    constexpr int LOGN = 15;
    using G = Geo<LOGN>;
    __shared__ __attribute__((aligned(16))) u64 lds[G::LDS_WORDS];
    const bool stream_loads = (division & kStreamLoads) != 0;        // (wave-uniform: a kernel argument)
    division &= ~kStreamLoads;
    // thread-derived values are rebuilt where they are used (wave index in an SGPR, lane index from v_mbcnt), as in
    // k_forward15: kept live across the polynomial loop they are spills in the general-prime instantiations
    unsigned wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+s"(wave_s));
    auto fresh_t = [&]() { return (wave_s << 6) | fresh_lane_id(); };
    u64* slice = lds + wave_s * WAVE_SLICE_WORDS;
    u64 v[32];
    unsigned y = blockIdx.x;
    if (y >= num) return;
    stagger_start<Tune::kStaggerInv, Tune::kStaggerInvMulti>(num > gridDim.x);
    // (the wave's 16 KiB chunk goes into the descriptor's base: scalar arithmetic)
    wave_load_rows<Tune::kInv15AuxLd>(v, slice, make_rsrc(a + (size_t)INV_POS(y) * G::N + wave_s * 2048u, 16384u), 0u, 0u, stream_loads);
    // modulus index of polynomial y, carried along in SGPRs instead of y % division per iteration (a division by a runtime
    // value is a multi-instruction VALU sequence; its reciprocal sat in a VGPR across the loop and was the forward kernel's
    // spill, reloaded behind a vmcnt(0) that also waited for the next polynomial's loads)
    unsigned ymod = __builtin_amdgcn_readfirstlane(INV_POS(blockIdx.x) % division), ystep = __builtin_amdgcn_readfirstlane(gridDim.x % division);
    if (Tune::kInvDescending && ystep) ystep = division - ystep;      // walking down: -grid = division - grid (mod division)
    asm volatile("" : "+s"(ymod), "+s"(ystep));          // in SGPRs from here on (the quotient sequence itself runs on the VALU)
    for (; y < num; y += gridDim.x, ymod = (ymod + ystep >= division ? ymod + ystep - division : ymod + ystep)) {
        const unsigned ynext = y + gridDim.x;
        const unsigned idx = prime_base + ymod;
        const PrimeDev p = primes[idx];
        const TwPair* twp = tw + (size_t)idx * G::N;
        const BufRsrc twr = make_rsrc(twp, G::N * 16u);
        u64* poly = a + (size_t)INV_POS(y) * G::N;
        __builtin_amdgcn_s_setprio(Tune::kPrioI1);
        gs_round<LOGN, HL, 0, 0, NEAR, Tune::kPsplitI1, Tune::kPrioI1B>(v, twp, twr, fresh_t(), p, primes[idx].twn);
        wave_transpose_0_to_5(v, slice, fresh_lane_id());
        __builtin_amdgcn_s_setprio(Tune::kPrioI2);
        gs_round<LOGN, HL, 5, 0, NEAR, Tune::kPsplitI2, Tune::kPrioI2B>(v, twp, twr, fresh_t(), p, primes[idx].twn);
        __syncthreads();                                  // private slices are idle from here on
        exchange<LOGN, 5, 10>(v, lds, fresh_t());
        // (the exchange ends with a barrier: every slice is dead until this wave's own row staging) the next polynomial's first
        // column half starts its way from memory now and lands in the slice during the last round
        if (ynext < num)
            wave_preland_rows_half<0, Tune::kInv15AuxLd>(slice, make_rsrc(a + (size_t)INV_POS(ynext) * G::N + wave_s * 2048u, 16384u), stream_loads);
        __builtin_amdgcn_s_setprio(Tune::kPrioI3);
        gs_round<LOGN, HL, 10, 0, NEAR, Tune::kPsplitI3, Tune::kPrioI3B>(v, twp, twr, 0u, p, primes[idx].twn);   // (the last round reads no thread-derived value)
        static_for<32>([&](auto rc) { v[decltype(rc)::value] = canon_after_inverse<HL, NEAR>(v[decltype(rc)::value], p); });
        store_coalesced<LOGN, Tune::kInv15AuxSt>(v, poly, fresh_t());
        if (ynext < num) {
            // the eight LDS-direct loads are older than the result stores of store_coalesced: counted wait for everything but those
            // kStoreCoalescedOps stores (hipcc does not order an LDS read behind the LDS-direct load that fills it; on gfx950 loads,
            // stores and LDS-direct loads of a wave retire in issue order on the one vmcnt counter -- probed, tools/probe/lds_direct.hip)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kStoreCoalescedOps) : "memory");
            u64 h[16];
            wave_read_prelanded_half(h, slice);
            static_for<16>([&](auto rc) { v[decltype(rc)::value] = h[decltype(rc)::value]; });
            wave_load_rows_half<1, Tune::kInv15AuxLd>(h, slice, make_rsrc(a + (size_t)INV_POS(ynext) * G::N + wave_s * 2048u, 16384u), 0u, 0u, stream_loads);
            static_for<16>([&](auto rc) { v[16 + decltype(rc)::value] = h[decltype(rc)::value]; });
        }
    }
}

// n = 2^16 contexts (two half-size transforms per polynomial, see k_forward15 SPLIT; a kernel of its own so that k_inverse15's
// code stays what was tuned): the UPPER half of polynomial y first -- an ordinary half-size inverse transform, its
// result Y stored in place -- then the lower half, whose result X stays in registers for the stage that couples the halves (the last
// GS stage of the full-size transform: (X + Y) / 2 below, (X - Y) psi^-bitrev(1) / 2 above; the half-size transforms scale by
// (n/2)^-1).  Y comes back through the wave's LDS slice by LDS-direct loads (split_partner_fetch): the first sixteen rows are
// requested right behind the exchange and arrive during the last round, the other sixteen while the first are combined.
// Reads 1.5 x, writes 1.5 x the polynomial in ONE launch instead of 2 x / 2 x in two (a stage kernel behind).
template <int HL, bool NEAR, bool MUL = false>
__global__ void __launch_bounds__(1024, 4)
k_inverse15_split(u64* __restrict__ a, const u64* __restrict__ bhat, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division,
            unsigned prime_base, unsigned num)
{
    if (guard_says_skip(primes, prime_base)) return;      // checked raw call whose table is not the cached one
    constexpr int LOGN = 15, SPLIT = 1;
    using G = Geo<LOGN>;
    __shared__ __attribute__((aligned(16))) u64 lds[G::LDS_WORDS];
    // thread-derived values are rebuilt where they are used (wave index in an SGPR, lane index from v_mbcnt), as in
    // k_forward15: kept live across the polynomial loop they are spills in the general-prime instantiations
    unsigned wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+s"(wave_s));
    auto fresh_t = [&]() { return (wave_s << 6) | fresh_lane_id(); };
    u64* slice = lds + wave_s * WAVE_SLICE_WORDS;
    u64 v[32];
    unsigned y = blockIdx.x;
    if (y >= num) return;
    stagger_start<Tune::kStaggerInv, Tune::kStaggerInvMulti>(num > gridDim.x);
    // (SPLIT: half h of the polynomial at position pos is the half-size polynomial 2 pos + h)
    [[maybe_unused]] unsigned h = SPLIT ? 1u : 0u;
    auto half_at = [&](unsigned yy, unsigned hh) { return SPLIT ? 2 * INV_POS(yy) + hh : INV_POS(yy); };
    // (the wave's 16 KiB chunk goes into the descriptor's base: scalar arithmetic)
    wave_load_rows(v, slice, make_rsrc(a + (size_t)(half_at(y, h)) * G::N + wave_s * 2048u, 16384u), 0u, 0u);
    // modulus index of polynomial y, carried along in SGPRs instead of y % division per iteration (a division by a runtime
    // value is a multi-instruction VALU sequence; its reciprocal sat in a VGPR across the loop and was the forward kernel's
    // spill, reloaded behind a vmcnt(0) that also waited for the next polynomial's loads)
    unsigned ymod = __builtin_amdgcn_readfirstlane(INV_POS(blockIdx.x) % division), ystep = __builtin_amdgcn_readfirstlane(gridDim.x % division);
    if (Tune::kInvDescending && ystep) ystep = division - ystep;      // walking down: -grid = division - grid (mod division)
    asm volatile("" : "+s"(ymod), "+s"(ystep));          // in SGPRs from here on (the quotient sequence itself runs on the VALU)
    while (y < num) {
        const unsigned ynext = y + gridDim.x;
        h = __builtin_amdgcn_readfirstlane(h);
        asm volatile("" : "+s"(h));                      // (uniform, but carried around the loop it ends up in a VGPR -- and the prime's address with it)
        const unsigned idx = SPLIT ? 2 * (prime_base + ymod) + h : prime_base + ymod;
        const PrimeDev p = primes[idx];
        const TwPair* twp = tw + (size_t)idx * G::N;
        const BufRsrc twr = make_rsrc(twp, G::N * 16u);
        u64* poly = a + (size_t)(half_at(y, h)) * G::N;
        if constexpr (MUL) {
            // pointwise product with the same half of bhat on the way in, streamed 16 words per lane at a time (as k_polymul15)
            const BufRsrc brs = make_rsrc(bhat + (size_t)half_at(y, h) * G::N + wave_s * 2048u, 16384u);
            u64 bb[16];
            wave_load_rows_half<0>(bb, slice, brs, 0u, 0u);
            static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                v[r] = FusedMul<HL, NEAR>::mul(v[r], bb[r], p);
            });
            wave_load_rows_half<1>(bb, slice, brs, 0u, 0u);
            static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                v[16 + r] = FusedMul<HL, NEAR>::mul(v[16 + r], bb[r], p);
            });
        }
        __builtin_amdgcn_s_setprio(Tune::kPrioI1);
        gs_round<LOGN, HL, 0, 0, NEAR, Tune::kPsplitI1, Tune::kPrioI1B, MUL && FusedMul<HL, NEAR>::LAZY>(v, twp, twr, fresh_t(), p, primes[idx].twn);
        wave_transpose_0_to_5(v, slice, fresh_lane_id());
        __builtin_amdgcn_s_setprio(Tune::kPrioI2);
        gs_round<LOGN, HL, 5, 0, NEAR, Tune::kPsplitI2, Tune::kPrioI2B>(v, twp, twr, fresh_t(), p, primes[idx].twn);
        __syncthreads();                                  // private slices are idle from here on
        exchange<LOGN, 5, 10>(v, lds, fresh_t());
        if constexpr (SPLIT != 0) if (h == 0) {
            // (the exchange ends with a barrier: nobody reads or writes this wave's slice until the next iteration's row loads)
            const BufRsrc hrs = make_rsrc(poly + G::N, G::N * 8u);
            split_partner_fetch<0>(hrs, slice, wave_s, fresh_lane_id());
            split_partner_fetch<1>(hrs, slice, wave_s, fresh_lane_id());
        }
        __builtin_amdgcn_s_setprio(Tune::kPrioI3);
        gs_round<LOGN, HL, 10, 0, NEAR, Tune::kPsplitI3, Tune::kPrioI3B>(v, twp, twr, 0u, p, primes[idx].twn);   // (the last round reads no thread-derived value)
        static_for<32>([&](auto rc) { v[decltype(rc)::value] = canon_after_inverse<HL, NEAR>(v[decltype(rc)::value], p); });
        if constexpr (SPLIT != 0) {
            if (h == 0) {
                const BufRsrc lrs = make_rsrc(poly, G::N * 8u), hrs = make_rsrc(poly + G::N, G::N * 8u);
                const u64 hq = (p.q >> 1) + 1;            // 2^-1 mod q
                static_for<4>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    // counted waits (hipcc does not order an LDS read behind the LDS-direct load that fills it; loads, stores and
                    // LDS-direct loads retire in issue order on one counter): phases 0 and 1 behind the last round's twiddle loads --
                    // everything; phase 2 has 16 + 4 + 16 younger operations behind it, phase 3 has 16 + 16
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (c == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if constexpr (c == 2) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
                    else if constexpr (c == 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                    u64 PV[8];
                    split_partner_read<c>(PV, slice, fresh_lane_id());
                    wave_lds_fence();                     // the buffer is free ...
                    if constexpr (c < 2) split_partner_fetch<c + 2>(hrs, slice, wave_s, fresh_lane_id());     // ... for phase c + 2
                    __builtin_amdgcn_sched_barrier(0);
                    const unsigned voff = fresh_t() * 8u;
                    static_for<8>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, r = 8 * c + i;
                        const u64 X = v[r], Y = PV[i];
                        const u64 sm = canon_2q(X + Y, p.q);
                        const u64 lo = (sm >> 1) + ((sm & 1) ? hq : 0);
                        const u64 d = X + p.q - Y;
                        const u64 Tm = Lazy<HL>::EXACT ? mul_shoup2(d, p.si, p.si_p, p.nq) : mul_shoup4m<true>(d, p.si, p.si_p, p.nq);
                        const u64 hi = canon_2q(reduce_2q_sel<NEAR>(Tm, p), p.q);
                        v2u32 xl, xh;
                        xl.x = lo32(lo); xl.y = hi32(lo); xh.x = lo32(hi); xh.y = hi32(hi);
                        __builtin_amdgcn_raw_buffer_store_b64(xl, lrs, voff, ((unsigned)r << G::B0) * 8u, Tune::kInv15AuxSt);
                        __builtin_amdgcn_raw_buffer_store_b64(xh, hrs, voff, ((unsigned)r << G::B0) * 8u, Tune::kInv15AuxSt);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else {
                store_coalesced<LOGN, Tune::kInv15AuxSt>(v, poly, fresh_t());
            }
            // next: the lower half of the same polynomial, or the upper half of the next one (ONE load site: two of them meet at
            // the loop's back edge with different register assignments, and the fix-up spills)
            const u64* nxt = h == 1 ? poly - G::N : a + (size_t)(half_at(ynext, 1)) * G::N;
            if (h == 1 || ynext < num) wave_load_rows(v, slice, make_rsrc(nxt + wave_s * 2048u, 16384u), 0u, 0u);
        } else {
            store_coalesced<LOGN, Tune::kInv15AuxSt>(v, poly, fresh_t());
            if (ynext < num)
                wave_load_rows(v, slice, make_rsrc(a + (size_t)INV_POS(ynext) * G::N + wave_s * 2048u, 16384u), 0u, 0u);
        }
        if (SPLIT != 0 && h == 1) {
            h = 0;                       // same polynomial, lower half
        } else {
            h = SPLIT ? 1u : 0u;
            y = ynext;
            ymod = (ymod + ystep >= division ? ymod + ystep - division : ymod + ystep);
        }
    }
}

// second-operand addressing of the fused products: one bhat polynomial per polynomial of a, or (kSharedB in the division
// word, kernels.hpp) `division` of them per key group shared by the whole batch; strips the flag bits off `division`
struct SharedB {
    bool on;
    unsigned group;
    __device__ explicit SharedB(unsigned& division)
        : on((division & kSharedB) != 0), group((division & ~kSharedB) >> kSharedGroupShift)
    {
        if (on) division &= kDivisionMask;
    }
    __device__ unsigned index(unsigned y, unsigned idx, unsigned division) const
    {
        if (!on) return y;
        return group ? (y / group) * division + idx : idx;
    }
};
__host__ __device__ inline unsigned plain_division(unsigned division) { return (division & kSharedB) ? (division & kDivisionMask) : division; }

}  // namespace mi355ntt
#include "kernels_lat.cuh"      // the small-batch kernels (need SharedB)
#include "kernels_lit.cuh"      // kernel class HL_LIT: the reference's own arithmetic (contexts with a Barrett-inexact prime)
namespace mi355ntt {

template <int HL, bool NEAR>
__global__ void __launch_bounds__(1024, 4)
k_polymul15(u64* __restrict__ a, const u64* __restrict__ bhat, const TwPair* __restrict__ twf, const TwPair* __restrict__ twi,
            const PrimeDev* __restrict__ primes, unsigned division, unsigned num)
{
    const bool stream_b = (division & (kSharedB | kStreamLoads)) == kStreamLoads;      // (own second operands, batch beyond the memory-side cache)
    if (stream_b) division &= ~kStreamLoads;
    const SharedB sb(division);       // (kSharedB: bhat holds `division` polynomials per key group instead of one per polynomial)
    constexpr int LOGN = 15;
    using G = Geo<LOGN>;
    __shared__ __attribute__((aligned(16))) u64 lds[G::LDS_WORDS];
    // thread-derived values are rebuilt where they are used (wave index in an SGPR, lane index from v_mbcnt), as in
    // k_forward15: kept live across the polynomial loop they were spills
    unsigned wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+s"(wave_s));
    auto fresh_t = [&]() { return (wave_s << 6) | fresh_lane_id(); };
    u64* slice = lds + wave_s * WAVE_SLICE_WORDS;
    u64 v[32];
    unsigned y = blockIdx.x;
    stagger_start<Tune::kStaggerMul, Tune::kStaggerMulMulti>(num > gridDim.x);
    load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)y * G::N, fresh_t());
    // modulus index of polynomial y, carried along in SGPRs instead of y % division per iteration (a division by a runtime
    // value is a multi-instruction VALU sequence; its reciprocal sat in a VGPR across the loop and was the forward kernel's
    // spill, reloaded behind a vmcnt(0) that also waited for the next polynomial's loads)
    unsigned ymod = __builtin_amdgcn_readfirstlane(blockIdx.x % division), ystep = __builtin_amdgcn_readfirstlane(gridDim.x % division);
    asm volatile("" : "+s"(ymod), "+s"(ystep));          // in SGPRs from here on (the quotient sequence itself runs on the VALU)
    for (; y < num; y += gridDim.x, ymod = (ymod + ystep >= division ? ymod + ystep - division : ymod + ystep)) {
        const unsigned idx = ymod;
        const PrimeDev p = primes[idx];
        const TwPair* tf = twf + (size_t)idx * G::N;
        const TwPair* ti = twi + (size_t)idx * G::N;
        const BufRsrc tfr = make_rsrc(tf, G::N * 16u), tir = make_rsrc(ti, G::N * 16u);
        u64* poly = a + (size_t)y * G::N;
        // (the wave's 16 KiB chunk of bhat goes into the descriptor's base)
        const BufRsrc brs = make_rsrc(bhat + (size_t)sb.index(y, idx, division) * G::N + wave_s * 2048u, 16384u);
        // ---- forward ----
        __builtin_amdgcn_s_setprio(Tune::kPrioR1);
        ct_round<LOGN, HL, 10, 4, NEAR>(v, tf, tfr, 0u, p);   // (round 1 reads no thread-derived value)
        __syncthreads();
        exchange<LOGN, 10, 5>(v, lds, fresh_t());
        __builtin_amdgcn_s_setprio(Tune::kPrioR2);
        ct_round<LOGN, HL, 5, 4, NEAR>(v, tf, tfr, fresh_t(), p);
        wave_transpose_5_to_0(v, slice, fresh_lane_id());
        __builtin_amdgcn_s_setprio(Tune::kPrioR3);
        ct_round<LOGN, HL, 0, 4, NEAR>(v, tf, tfr, fresh_t(), p);
        // ---- pointwise product with bhat, streamed 16 words per lane at a time (layout 0 on both sides) ----
        {
            u64 bb[16];
            wave_load_rows_half_direct<0, Tune::kMul15BAuxLd>(bb, slice, brs, HL > 2 && stream_b);
            static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                v[r] = FusedMul<HL, NEAR>::mul(v[r], bb[r], p);
            });
            wave_load_rows_half_direct<1, Tune::kMul15BAuxLd>(bb, slice, brs, HL > 2 && stream_b);
            static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                v[16 + r] = FusedMul<HL, NEAR>::mul(v[16 + r], bb[r], p);
            });
        }
        // ---- inverse ----
        gs_round<LOGN, HL, 0, 0, NEAR, -2, 0, FusedMul<HL, NEAR>::LAZY>(v, ti, tir, fresh_t(), p, primes[idx].twn);
        wave_transpose_0_to_5(v, slice, fresh_lane_id());
        __builtin_amdgcn_s_setprio(Tune::kPrioI2);
        gs_round<LOGN, HL, 5, 0, NEAR>(v, ti, tir, fresh_t(), p, primes[idx].twn);
        __syncthreads();
        exchange<LOGN, 5, 10>(v, lds, fresh_t());
        __builtin_amdgcn_s_setprio(Tune::kPrioI3);
        gs_round<LOGN, HL, 10, 0, NEAR>(v, ti, tir, fresh_t(), p, primes[idx].twn);
        static_for<32>([&](auto rc) { v[decltype(rc)::value] = canon_after_inverse<HL, NEAR>(v[decltype(rc)::value], p); });
        store_coalesced<LOGN, Tune::kInv15AuxSt>(v, poly, fresh_t());
        if (y + gridDim.x < num) load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)(y + gridDim.x) * G::N, fresh_t());
    }
}

// Small batches run the latency kernels (kernels_lat.cuh: n/512 waves per polynomial, two launches, twice the HBM traffic),
// large ones the single-pass kernels (one workgroup of n/32 threads per polynomial, persistent).  Switching points measured
// on MI355X (profiles/r03_latency_cpp.txt): n = 2^15: fwd+inv pairs 160 polynomials, fused products 176 -- and once more just
// above one polynomial per CU (257 .. 352 / 384), where a persistent launch of 1024-thread workgroups would pay a second,
// mostly empty iteration (320 polynomials: 142 us against 162 us per pair).  Smaller n: see lat_threshold.
// MI355NTT_LATENCY_PATH_MAX in the environment replaces the rule by a plain threshold (tuning / tests: 0 = never).
// polynomials up to which the latency kernels win (fwd+inv pairs / fused products; tools/crossover_sizes.sh,
// profiles/r03_latency_cpp.txt): the smaller the ring, the more polynomials it takes to fill the chip with n/512 waves each
#define MI355NTT_LAT_T14 112u
#define MI355NTT_LAT_T14M 176u
#define MI355NTT_LAT_T13 128u
#define MI355NTT_LAT_T13M 144u
#define MI355NTT_LAT_T12 200u
#define MI355NTT_LAT_T12M 256u
#define MI355NTT_LAT_T11 256u
#define MI355NTT_LAT_T11M 448u
template <int LOGN>
constexpr unsigned lat_threshold(bool fused)
{
    if (LOGN == 15) return fused ? 176u : 160u;
    if (LOGN == 14) return fused ? MI355NTT_LAT_T14M : MI355NTT_LAT_T14;
    if (LOGN == 13) return fused ? MI355NTT_LAT_T13M : MI355NTT_LAT_T13;
    if (LOGN == 12) return fused ? MI355NTT_LAT_T12M : MI355NTT_LAT_T12;
    return fused ? MI355NTT_LAT_T11M : MI355NTT_LAT_T11;
}
template <int LOGN>
inline bool use_latency_path(unsigned num, bool fused)
{
    static const long forced = [] {
        const char* e = std::getenv("MI355NTT_LATENCY_PATH_MAX");
        return e ? (long)std::strtoul(e, nullptr, 10) : -1L;
    }();
    if (forced >= 0) return num <= (unsigned long)forced;
    if (num <= lat_threshold<LOGN>(fused)) return true;
    return LOGN == 15 && num > 256u && num <= (fused ? 384u : 352u);
}

// class 0 (kernels_lit.cuh): which of the two kernel shapes a call runs.  The literal butterflies carry 2.6 - 3.2 x the instructions of
// the lazy ones, so this class is bound by VALU issue and the second pass over memory of the small-batch kernels (a polynomial over
// n/512 waves, two launches per transform) costs next to nothing -- while their fine-grained workgroups balance any batch size over
// the chip, where the single-pass kernels (one workgroup per polynomial) take whole rounds of the persistent grid: one polynomial at
// n = 2^15 154 -> 25 us per forward + inverse, 320 polynomials 320 -> 213 us, 1024 polynomials 641 = 639 us.  Measured switching
// points (tools/probe/lit_small_ab.py, profiles/r06_literal_class.txt): n <= 2^14 the small-batch kernels up to the batch sizes below;
// n = 2^15 whenever 25 us + 0.65 us per polynomial beats the whole rounds of 157 us, up to 1024 polynomials (from 1280 polynomials on
// the batch leaves the memory-side cache and the single pass wins by 3 - 10 %).
template <int LOGN>
constexpr unsigned lit_lat_threshold(bool fused)
{
    if (LOGN == 14) return fused ? 1024u : 4096u;
    if (LOGN == 13) return fused ? 704u : 4096u;
    return 16384u;                                         // n = 2^11, 2^12
}
template <int LOGN>
inline bool lit_use_latency_path(unsigned num, bool fused, unsigned grid)
{
    static const long forced = [] {
        const char* e = std::getenv("MI355NTT_LATENCY_PATH_MAX");
        return e ? (long)std::strtoul(e, nullptr, 10) : -1L;
    }();
    if (forced >= 0) return num <= (unsigned long)forced;
    if (LOGN == 15) {
        if (num > 1024u) return false;
        const unsigned rounds = (num + grid - 1u) / grid;
        return 25000u + (fused ? 660u : 650u) * num < rounds * 156000u;      // (ns per transform pair / product)
    }
    return num <= lit_lat_threshold<LOGN>(fused);
}

// the context's kernel class (FastTables::hl: bits 0-3 headroom class, bit 4 every prime near 2^k) as compile-time arguments.
// Classes: 6 (<= 58-bit moduli: no intermediate reduction), 5 (59-bit near-2^k, round 4: one partial reduction every 7 forward / 3 inverse stages
// instead of every 3 / 2), 4 (59/60-bit: one every 2-3 stages), 3 (61-bit near-2^k, round 4: 8 q < 2^64, so the
// lazy three-product quotient estimate with values in [0, 4q) still fits -- one partial reduction per stage, but no 64 x 64 high
// product; the reference's decryption modulus gamma is 61-bit, demo.cu:93) and 2 (62-bit: exact quotients, values in [0, 2q)).
// WITH3 = false (the n = 2^16 split / pair kernels): 61-bit moduli stay in class 2 there.
template <bool WITH3 = true, class F>
inline void dispatch_class(int hl, F&& f)
{
    const bool near = (hl & 16) != 0;
    const int h = hl & 15;
    using std::integral_constant;
    if (near) {
        if (h >= 6) f(integral_constant<int, 6>{}, integral_constant<bool, true>{});
        else if (WITH3 && h == 5) f(integral_constant<int, WITH3 ? 5 : 4>{}, integral_constant<bool, true>{});
        else if (h >= 4) f(integral_constant<int, 4>{}, integral_constant<bool, true>{});
        else if (WITH3 && h == 3) f(integral_constant<int, WITH3 ? 3 : 2>{}, integral_constant<bool, true>{});
        else f(integral_constant<int, 2>{}, integral_constant<bool, true>{});
    } else {
        if (h >= 6) f(integral_constant<int, 6>{}, integral_constant<bool, false>{});
        else if (h >= 4) f(integral_constant<int, 4>{}, integral_constant<bool, false>{});
        // (general 61-bit primes, round 5: class 3 -- three-product quotient estimates instead of exact 64 x 64 high products.  Their
        // inverse rounds restore the bound by ONE conditional subtraction of 4q per reducing sum, gs_round; with the 7-instruction
        // general reduction there the class-3 inverse and fused kernels needed 28-52 bytes of scratch per lane, which kept these
        // primes in class 2 until round 4)
        else if (WITH3 && h == 3) f(integral_constant<int, WITH3 ? 3 : 2>{}, integral_constant<bool, false>{});
        else f(integral_constant<int, 2>{}, integral_constant<bool, false>{});
    }
}

// ---- fused: a = INTT( NTT(a) (.) bhat ) ---------------------------------------------------------
template <int LOGN, int HL, bool NEAR>
__global__ void __launch_bounds__(Geo<LOGN>::T, 4)
k_polymul(u64* __restrict__ a, const u64* __restrict__ bhat, const TwPair* __restrict__ twf, const TwPair* __restrict__ twi,
          const PrimeDev* __restrict__ primes, unsigned division)
{
    const SharedB sb(division);
    using G = Geo<LOGN>;
    __shared__ __attribute__((aligned(16))) u64 lds[G::LDS_WORDS];
    const unsigned y = blockIdx.x;
    unsigned idx = __builtin_amdgcn_readfirstlane(y % division);
    asm volatile("" : "+s"(idx));        // pinned in an SGPR: the table and constant addresses derived from it stay scalar (the
                                         // compiler otherwise forms them on the VALU and keeps a vector copy alive: a spill)
    const PrimeDev p = primes[idx];
    u64* poly = a + (size_t)y * G::N;
    const u64* bp = bhat + (size_t)sb.index(y, idx, division) * G::N;
    unsigned wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (thread-derived values: see k_forward)
    asm volatile("" : "+s"(wave_s));
    auto tid = [&]() { return (wave_s << 6) | fresh_lane_id(); };
    u64 v[32];
    load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, poly, tid());
    forward_core<LOGN, HL, NEAR>(v, twf + (size_t)idx * G::N, tid, p, lds);
    // layout 0: this thread holds NTT values 32t .. 32t+31; the inverse starts from the same layout
    const BufRsrc brs = make_rsrc(bp, G::N * 8u);
    const unsigned boff = tid() * 256u;
#pragma unroll
    for (int r = 0; r < 32; r += 2) {
        const TwPair bb = buf_load_tw(brs, boff, (unsigned)r * 8u);      // two consecutive words of bhat
        v[r] = FusedMul<HL, NEAR>::mul(v[r], bb.w, p);
        v[r + 1] = FusedMul<HL, NEAR>::mul(v[r + 1], bb.wp, p);
        if ((r & 6) == 6) __builtin_amdgcn_sched_barrier(0);
    }
    inverse_core<LOGN, HL, NEAR, FusedMul<HL, NEAR>::LAZY>(v, twi + (size_t)idx * G::N, tid, p, lds, primes[idx].twn);
#pragma unroll
    for (int r = 0; r < 32; r++) v[r] = canon_after_inverse<HL, NEAR>(v[r], p);
    store_coalesced<LOGN, Tune::kInvAuxSt>(v, poly, tid());
}

// Workgroups that can be resident at once: 256 CUs x (what 128 VGPRs/thread, the LDS image and 32 waves/CU admit).
// A grid of at most that size makes every workgroup persistent; more polynomials are walked in-kernel.
template <int LOGN>
inline unsigned persistent_grid(unsigned num)
{
    using G = Geo<LOGN>;
    const unsigned by_waves = 16u / (G::T / 64u);                                   // 4 waves/SIMD at 128 VGPRs
    const unsigned by_lds = 163840u / (G::LDS_WORDS * 8u);
    unsigned per_cu = by_waves < by_lds ? by_waves : by_lds;
    if (per_cu < 1) per_cu = 1;
    const unsigned cap = current_device_cus() * per_cu;
    return num < cap ? num : cap;
}

// class HL_LIT: a workgroup walks its polynomials in two passes by the kind of their prime (kernels_lit.cuh).  With a stride that
// shares a factor with `division` a workgroup would meet only some of the primes -- all the literal work on a few workgroups -- so the
// persistent grid shrinks until the two are coprime (division <= 16: a few steps).
template <int LOGN>
inline unsigned lit_grid(unsigned num, unsigned division)
{
    unsigned g = persistent_grid<LOGN>(num);
    auto gcd = [](unsigned x, unsigned y) { while (y) { const unsigned t = x % y; x = y; y = t; } return x; };
    while (g > 1 && g < num && gcd(g, division) != 1) g--;
    return g;
}

template <int LOGN>
hipError_t launch_fwd(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division, unsigned base,
                      hipStream_t s)
{
    dim3 g(persistent_grid<LOGN>(num)), b(Geo<LOGN>::T);
    using L = LatGeo<LOGN>;
    if ((hl & 15) == HL_LIT) {      // the reference's own arithmetic (kernels_lit.cuh)
        if (lit_use_latency_path<LOGN>(num, false, lit_grid<LOGN>(num, division))) {
            k_lat_fwd_a_lit<LOGN><<<dim3(num << L::GB), dim3(L::WA), 0, s>>>(d_a, tw, pr, division, base);
            k_lat_fwd_b_lit<LOGN><<<dim3(num << L::CH), dim3(64), 0, s>>>(d_a, tw, pr, division, base);
            return hipGetLastError();
        }
        const dim3 gl(lit_grid<LOGN>(num, division));
        if constexpr (LOGN == 15) k_forward15_lit<LOGN><<<gl, b, 0, s>>>(d_a, tw, pr, division, base, num);
        else k_forward_lit<LOGN><<<gl, b, 0, s>>>(d_a, tw, pr, division, base, num);
        return hipGetLastError();
    }
    if (use_latency_path<LOGN>(num, false)) {
        dispatch_class(hl, [&](auto hc, auto nc) {
            constexpr int H = decltype(hc)::value;
            constexpr bool NR = decltype(nc)::value;
            k_lat_fwd_a<LOGN, H, NR><<<dim3(num << L::GB), dim3(L::WA), 0, s>>>(d_a, tw, pr, division, base);
            k_lat_fwd_b<LOGN, H, NR><<<dim3(num << L::CH), dim3(64), 0, s>>>(d_a, tw, pr, division, base);
        });
        return hipGetLastError();
    }
    dispatch_class(hl, [&](auto hc, auto nc) {
        constexpr int H = decltype(hc)::value;
        constexpr bool NR = decltype(nc)::value;
        if constexpr (LOGN == 15) k_forward15<H, NR><<<g, b, 0, s>>>(d_a, tw, pr, division, base, num);
        else k_forward<LOGN, H, NR><<<g, b, 0, s>>>(d_a, tw, pr, division, base, num);
    });
    return hipGetLastError();
}

template <int LOGN>
hipError_t launch_inv(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division, unsigned base,
                      hipStream_t s)
{
    dim3 g(persistent_grid<LOGN>(num)), b(Geo<LOGN>::T);
    using L = LatGeo<LOGN>;
    if ((hl & 15) == HL_LIT) {
        if (lit_use_latency_path<LOGN>(num, false, lit_grid<LOGN>(num, division))) {
            k_lat_inv_b_lit<LOGN><<<dim3(num << L::CH), dim3(64), 0, s>>>(d_a, tw, pr, division, base);
            k_lat_inv_a_lit<LOGN><<<dim3(num << L::GB), dim3(L::WA), 0, s>>>(d_a, tw, pr, division, base);
            return hipGetLastError();
        }
        const dim3 gl(lit_grid<LOGN>(num, division));
        if constexpr (LOGN == 15) k_inverse15_lit<LOGN><<<gl, b, 0, s>>>(d_a, tw, pr, division, base, num);
        else k_inverse_lit<LOGN><<<gl, b, 0, s>>>(d_a, tw, pr, division, base, num);
        return hipGetLastError();
    }
    if (use_latency_path<LOGN>(num, false)) {
        dispatch_class(hl, [&](auto hc, auto nc) {
            constexpr int H = decltype(hc)::value;
            constexpr bool NR = decltype(nc)::value;
            k_lat_inv_b<LOGN, H, NR><<<dim3(num << L::CH), dim3(64), 0, s>>>(d_a, tw, pr, division, base);
            k_lat_inv_a<LOGN, H, NR><<<dim3(num << L::GB), dim3(L::WA), 0, s>>>(d_a, tw, pr, division, base);
        });
        return hipGetLastError();
    }
    dispatch_class(hl, [&](auto hc, auto nc) {
        constexpr int H = decltype(hc)::value;
        constexpr bool NR = decltype(nc)::value;
        if constexpr (LOGN == 15) k_inverse15<H, NR><<<g, b, 0, s>>>(d_a, tw, pr, inv15_division_word(division, num), base, num);
        else k_inverse<LOGN, H, NR><<<g, b, 0, s>>>(d_a, tw, pr, division, base, num);
    });
    return hipGetLastError();
}

template <int LOGN>
hipError_t launch_mul(int hl, u64* d_a, const u64* d_b, const TwPair* twf, const TwPair* twi, const PrimeDev* pr, unsigned num,
                      unsigned division, hipStream_t s)
{
    using L = LatGeo<LOGN>;
    if ((hl & 15) == HL_LIT) {
        if (lit_use_latency_path<LOGN>(num, true, lit_grid<LOGN>(num, plain_division(division)))) {
            k_lat_fwd_a_lit<LOGN><<<dim3(num << L::GB), dim3(L::WA), 0, s>>>(d_a, twf, pr, plain_division(division), 0u);
            k_lat_mul_b_lit<LOGN><<<dim3(num << L::CH), dim3(64), 0, s>>>(d_a, d_b, twf, twi, pr, division);
            k_lat_inv_a_lit<LOGN><<<dim3(num << L::GB), dim3(L::WA), 0, s>>>(d_a, twi, pr, plain_division(division), 0u);
            return hipGetLastError();
        }
        if constexpr (LOGN == 15)
            k_polymul15_lit<LOGN><<<dim3(lit_grid<LOGN>(num, plain_division(division))), dim3(Geo<LOGN>::T), 0, s>>>(d_a, d_b, twf, twi, pr, division, num);
        else
            k_polymul_lit<LOGN><<<dim3(num), dim3(Geo<LOGN>::T), 0, s>>>(d_a, d_b, twf, twi, pr, division);
        return hipGetLastError();
    }
    if (use_latency_path<LOGN>(num, true)) {
        dispatch_class(hl, [&](auto hc, auto nc) {
            constexpr int H = decltype(hc)::value;
            constexpr bool NR = decltype(nc)::value;
            k_lat_fwd_a<LOGN, H, NR><<<dim3(num << L::GB), dim3(L::WA), 0, s>>>(d_a, twf, pr, plain_division(division), 0u);
            k_lat_mul_b<LOGN, H, NR><<<dim3(num << L::CH), dim3(64), 0, s>>>(d_a, d_b, twf, twi, pr, division);
            k_lat_inv_a<LOGN, H, NR><<<dim3(num << L::GB), dim3(L::WA), 0, s>>>(d_a, twi, pr, plain_division(division), 0u);
        });
        return hipGetLastError();
    }
    dispatch_class(hl, [&](auto hc, auto nc) {
        constexpr int H = decltype(hc)::value;
        constexpr bool NR = decltype(nc)::value;
        if constexpr (LOGN == 15) {
            const unsigned dw = division | ((division & kSharedB) == 0 && num > kMulStreamLoadsAbove ? kStreamLoads : 0u);
            k_polymul15<H, NR><<<dim3(persistent_grid<LOGN>(num)), dim3(Geo<LOGN>::T), 0, s>>>(d_a, d_b, twf, twi, pr, dw, num);
        } else {
            k_polymul<LOGN, H, NR><<<dim3(num), dim3(Geo<LOGN>::T), 0, s>>>(d_a, d_b, twf, twi, pr, division);
        }
    });
    return hipGetLastError();
}


// explicit per-size entry points (defined in kernels_fast_n<LOGN>.hip)
#define MI355NTT_DECLARE_SIZE(LOGN)                                                                                               hipError_t fast_fwd_##LOGN(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division,                                     unsigned base, hipStream_t s);                                                                     hipError_t fast_inv_##LOGN(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division,                                     unsigned base, hipStream_t s);                                                                     hipError_t fast_mul_##LOGN(int hl, u64* d_a, const u64* d_b, const TwPair* twf, const TwPair* twi, const PrimeDev* pr,                                  unsigned num, unsigned division, hipStream_t s);
MI355NTT_DECLARE_SIZE(11)
MI355NTT_DECLARE_SIZE(12)
MI355NTT_DECLARE_SIZE(13)
MI355NTT_DECLARE_SIZE(14)
MI355NTT_DECLARE_SIZE(15)
// (kernels_fast_n15e.hip: the fused product with an epilogue, kernels_epi.cuh)
hipError_t fast_mul_epi_15(int kind, int hl, u64* d_a, const u64* d_b, const TwPair* twf, const TwPair* twi, const PrimeDev* pr, unsigned num,
                           unsigned division, hipStream_t s, const u64* other, const void* consts);
bool fast_split_ok_16(unsigned num, int op, bool pair);   // (kernels_fast_n16.hip; op: 0 forward, 1 inverse, 2 fused product)
hipError_t fast_fwd_pair_16(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division, unsigned base,
                            hipStream_t s, unsigned* d_flags);      // (two workgroups per polynomial; d_flags: the device's pair-flag buffer, kernels.hpp)
bool fast_fwd_pair_ok_16(int hl);                         // (headroom classes 4 and 6)
hipError_t fast_inv_split_16(int hl, u64* d_a, const u64* d_bhat, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division,
                             unsigned base, hipStream_t s);       // (d_bhat: null, or the pointwise factor applied on the way in)
hipError_t fast_fwd_split_16(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division, unsigned base,
                             hipStream_t s);

#define MI355NTT_DEFINE_SIZE(LOGN)                                                                                                hipError_t fast_fwd_##LOGN(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division,                                     unsigned base, hipStream_t s)                                                                      {                                                                                                                                 return launch_fwd<LOGN>(hl, d_a, tw, pr, num, division, base, s);                                                         }                                                                                                                             hipError_t fast_inv_##LOGN(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division,                                     unsigned base, hipStream_t s)                                                                      {                                                                                                                                 return launch_inv<LOGN>(hl, d_a, tw, pr, num, division, base, s);                                                         }                                                                                                                             hipError_t fast_mul_##LOGN(int hl, u64* d_a, const u64* d_b, const TwPair* twf, const TwPair* twi, const PrimeDev* pr,                                  unsigned num, unsigned division, hipStream_t s)                                                    {                                                                                                                                 return launch_mul<LOGN>(hl, d_a, d_b, twf, twi, pr, num, division, s);                                                    }

}  // namespace mi355ntt
