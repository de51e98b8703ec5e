// kernels_lat.cuh -- SMALL batches, n = 2^11 .. 2^15: the latency path (batch 1 is what the reference publishes: 12.5 .. 39 us
// NTT / 12.5 .. 23 us INTT on V100 for n = 2^11 .. 2^15, Article.pdf p25 Table 6; one ciphertext per BFV call, p26 Table 7).
//
// Measured on gfx950 (tools/ubench_issue.hip, profiles/r02_ubench_issue_costs.txt): ONE wave issues a v_mad_u64_u32 every
// 10.4 cycles and any other VALU instruction every 6.5-8 however independent its instructions are -- a SIMD needs 2-3 waves
// to reach its 4-cycle rate.  With few polynomials the chip is empty, so the time of a transform is the instruction count of
// its longest wave: with 32 coefficients per thread (the throughput kernels' geometry: 176-240 butterflies per lane) a single
// polynomial took 9.4 / 10.7 / 12.6 / 19.6 / 14.5 us forward at n = 2^11 .. 2^15 whatever the host did (compiled C++, Python and
// a captured hipGraph all measure the same: profiles/r03_latency_cpp.txt).
//
// These kernels spread a polynomial over n/512 waves of EIGHT coefficients per thread (3 log2(n)/... = 44-60 butterflies per lane):
//   forward : k_lat_fwd_a  the stages on index bits LOGN-1 .. 9 (up to six: two three-stage rounds with one LDS exchange that
//                          swaps the wave index with the register index; workgroups of 512 threads, 256 at n = 2^11)
//             k_lat_fwd_b  the stages on index bits 8 .. 0 (one-wave workgroups on 512 consecutive coefficients each, two
//                          wave-local 8x8 transposes, no barrier)
//   inverse : k_lat_inv_b  (bits 0..8), k_lat_inv_a (bits 9 .. LOGN-1, n^-1 folded into its last round's twiddles)
//   product : k_lat_fwd_a, k_lat_mul_b (forward bits 8..0, product with bhat, inverse bits 0..8), k_lat_inv_a
// Values travel between the launches in lazy form [0, B q); the compile-time bound tracking (fwd_reduce_mask, InvPolicy) is
// per STAGE and therefore the same as in the single-pass kernels.  Outputs are canonical: the words equal theirs.
// Twiddles come from the same device tables (per-stage blocks permuted for the 32-coefficient geometry: lat_tw_index).
#pragma once
#include "ntt_core.cuh"

namespace mi355ntt {

// entry of group p of the stage on index bit BETA in the device tables (kernels_fast.hip, fast_tables_create): the block
// [len, 2 len), len = 2^(LOGN-1-BETA), is stored as len + u * nthi + thi with p = (thi << (4 - j)) + u for the round geometry
// (B, j) of the single-pass kernels that owns the stage -- forward rounds count the bits down from LOGN-1 in fives, inverse
// rounds up from 0, so the two directions' tables are permuted differently unless LOGN is a multiple of five
template <int LOGN, bool FWD, int BETA>
__host__ __device__ constexpr unsigned lat_tw_index(unsigned p)
{
    constexpr int rho = FWD ? (LOGN - 1 - BETA) / 5 : BETA / 5;
    constexpr int top = LOGN - 1 - 5 * rho;
    constexpr int B = FWD ? (top - 4 > 0 ? top - 4 : 0) : (5 * rho < LOGN - 5 ? 5 * rho : LOGN - 5);
    constexpr int j = BETA - B;
    static_assert(j >= 0 && j <= 4, "stage outside its round");
    constexpr unsigned len = 1u << (LOGN - 1 - BETA), nthi = (1u << (LOGN - 5)) >> B;
    return len + (p & ((1u << (4 - j)) - 1u)) * nthi + (p >> (4 - j));
}

// ---- butterflies on two registers -----------------------------------------------------------------------------------
// CT stage s (index bit LOGN-1 - s): (a, b) <- (a + T, a + cq - T), T = b * w in [0, TQ q); U reduced first when the mask says so
template <int LOGN, int HL, bool NEAR, bool TWS, int S>
__device__ __forceinline__ void lat_ct(u64& a, u64& b, const TwPair w, const PrimeDev& p)
{
    if constexpr (HL == HL_LIT) {
        lit_ct_bfly(a, b, w.w, p);                        // the reference's own butterfly (kernel class 0, kernels_lit.cuh)
    } else {
        constexpr bool EX = Lazy<HL>::EXACT;
        constexpr bool red = (fwd_reduce_mask<LOGN, HL>() >> S) & 1u;
        const u64 cq = (u64)Lazy<HL>::TQ * p.q;
        u64 U = a;
        if constexpr (red) U = reduce_2q_sel<NEAR>(U, p);
        if constexpr (!EX) {
            u64 D = (U << 1) + cq;
            asm("" : "+v"(D));
            const u64 A = mul_shoup4m_acc<TWS>(b, w.w, w.wp, p.nq, U);
            a = A;
            b = D - A;
        } else {
            const u64 Tm = mul_shoup2(b, w.w, w.wp, p.nq);
            a = U + Tm;
            b = U + cq - Tm;
        }
    }
}

// GS stage on index bit BETA: (a, b) <- (a + b, (a + cq - b) * w); FIN: the sum is what leaves the transform (last stage)
template <int LOGN, int HL, bool NEAR, bool TWS, int BETA, bool IN2Q = false>
__device__ __forceinline__ void lat_gs(u64& a, u64& b, const TwPair w, const PrimeDev& p)
{
    if constexpr (HL == HL_LIT) {
        lit_gs_bfly(a, b, w.w, p, (p.q + 1) >> 1);        // the reference's own butterfly, halving included (kernel class 0)
    } else {
        static_assert(!IN2Q || !Lazy<HL>::EXACT, "lazy inputs: classes with 4q of headroom only (gs_round, ntt_core.cuh)");
        constexpr InvPolicy<LOGN, HL> POL{};
        constexpr bool EX = Lazy<HL>::EXACT;
        constexpr bool last = (BETA == LOGN - 1);
        constexpr bool red = ((POL.mask >> BETA) & 1u) || (last && !(NEAR && !EX) && (2 * POL.cmul[BETA] > Lazy<HL>::TQ));
        const u64 cq = (u64)((IN2Q && BETA == 0) ? 2 : POL.cmul[BETA]) * p.q;
        const u64 X = a, Y = b;
        u64 S = X + Y;
        const u64 D = X + cq - Y;
        if constexpr (red) {
            if constexpr (EX && !NEAR) S = csub(S, 2 * p.q);
            else S = reduce_2q_sel<NEAR>(S, p);
        }
        a = S;
        if constexpr (!EX) b = mul_shoup4m<TWS>(D, w.w, w.wp, p.nq);
        else b = mul_shoup2(D, w.w, w.wp, p.nq);
    }
}

// x * w for the one value per thread that was summed in every stage of the last inverse round (w = n^-1)
template <int HL, bool TWS>
__device__ __forceinline__ u64 lat_scale(u64 x, const TwPair w, const PrimeDev& p)
{
    if constexpr (!Lazy<HL>::EXACT) return mul_shoup4m<TWS>(x, w.w, w.wp, p.nq);
    else return mul_shoup2(x, w.w, w.wp, p.nq);
}

// ---- wave-local 8x8 transposes through a private 4608-byte LDS slice ------------------------------------------------
// layout L6: register r of lane l holds local index (r << 6) | l;  L3: ((l >> 3) << 6) | (r << 3) | (l & 7);  L0: (l << 3) | r
// Slice addressing (in 8-byte words): L6 <-> L3 through rows of 72 words (64 + 8 of padding: the eight lane groups of a
// half-wave land 16 banks apart), L3 <-> L0 through rows of 9 words per destination lane (odd stride): both directions
// conflict-free for 8-byte accesses.
constexpr unsigned LAT_SLICE_WORDS = 576;

__device__ __forceinline__ void lat_t_63(u64 (&v)[8], u64* slice, unsigned lane)       // L6 -> L3
{
    __builtin_amdgcn_sched_barrier(0);
    static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; slice[r * 72 + lane] = v[r]; });
    wave_lds_fence();
    const unsigned base = (lane >> 3) * 72 + (lane & 7);
    static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; v[r] = slice[base + r * 8]; });
    wave_lds_fence();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void lat_t_36(u64 (&v)[8], u64* slice, unsigned lane)       // L3 -> L6
{
    __builtin_amdgcn_sched_barrier(0);
    const unsigned base = (lane >> 3) * 72 + (lane & 7);
    static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; slice[base + r * 8] = v[r]; });
    wave_lds_fence();
    static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; v[r] = slice[r * 72 + lane]; });
    wave_lds_fence();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void lat_t_30(u64 (&v)[8], u64* slice, unsigned lane)       // L3 -> L0
{
    __builtin_amdgcn_sched_barrier(0);
    // element ((lh << 6) | (r << 3) | ll) goes to lane (lh << 3) | r, register ll: word (destination lane) * 9 + ll
    const unsigned wbase = (lane >> 3) * 72 + (lane & 7);
    static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; slice[wbase + r * 9] = v[r]; });
    wave_lds_fence();
    static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; v[r] = slice[lane * 9 + r]; });
    wave_lds_fence();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void lat_t_03(u64 (&v)[8], u64* slice, unsigned lane)       // L0 -> L3
{
    __builtin_amdgcn_sched_barrier(0);
    static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; slice[lane * 9 + r] = v[r]; });
    wave_lds_fence();
    const unsigned rbase = (lane >> 3) * 72 + (lane & 7);
    static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; v[r] = slice[rbase + r * 9]; });
    wave_lds_fence();
    __builtin_amdgcn_sched_barrier(0);
}

// ---- up to three in-register stages of a round ------------------------------------------------------------------------
// Forward round: the registers hold index bits TOP, TOP-1, TOP-2 (register bits 2, 1, 0); the stages on the top NST of them are
// done.  The twiddle group of the stage on index bit beta = TOP - 2 + rb is p = i >> (beta + 1) = (upper << (2 - rb)) |
// (r >> (rb + 1)) with upper = i >> (TOP + 1), the part of the index above the round -- one value per thread and round.
// UNI: `upper` is wave-uniform (scalar twiddle loads, SGPR operands); else 16-byte vector loads through `twr`.
template <int LOGN, int HL, bool NEAR, bool UNI, int TOP, int NST = 3>
__device__ __forceinline__ void lat_fwd_round(u64 (&v)[8], const TwPair* __restrict__ tw, BufRsrc twr, const PrimeDev& p, unsigned upper)
{
    static_for<NST>([&](auto jc) {
        constexpr int rb = 2 - decltype(jc)::value;                  // register bit of this stage: 2, 1, 0
        constexpr int beta = TOP - 2 + rb, s = LOGN - 1 - beta;
        TwPair W[4 >> rb];                                           // the stage's 1, 2 or 4 distinct twiddles first ...
        static_for<(4 >> rb)>([&](auto uc) {
            constexpr unsigned u = decltype(uc)::value;
            const unsigned idx = lat_tw_index<LOGN, true, beta>((upper << (2 - rb)) | u);
            if constexpr (UNI) W[u] = tw[idx];
            else W[u] = buf_load_tw(twr, idx * 16u, 0u);
        });
        static_for<4>([&](auto kc) {                                 // ... then its four butterflies
            constexpr int k = decltype(kc)::value;
            constexpr int r0 = low_reg(rb, k), r1 = r0 | (1 << rb);
            lat_ct<LOGN, HL, NEAR, UNI, s>(v[r0], v[r1], W[r0 >> (rb + 1)], p);
        });
    });
}

// Inverse round: the registers hold index bits LOW, LOW+1, LOW+2 (register bits 0, 1, 2); the stages on the top NST of them are
// done (register bits 3 - NST .. 2); upper = i >> (LOW + 3).  SCALE (the last round of the transform, upper = 0): butterflies
// whose register bits between the round's first stage and their own are zero take twiddle * n^-1 from twn (gs_round,
// ntt_core.cuh), and the registers that were summed in every stage -- those below 2^(3 - NST) -- are multiplied by n^-1.
template <int LOGN, int HL, bool NEAR, bool UNI, int LOW, int NST = 3, bool SCALE = false, bool IN2Q = false>
__device__ __forceinline__ void lat_inv_round(u64 (&v)[8], const TwPair* __restrict__ tw, BufRsrc twr, const PrimeDev& p, unsigned upper,
                                              const TwPair* __restrict__ twn = nullptr)
{
    constexpr int RB0 = 3 - NST;
    static_for<NST>([&](auto jc) {
        constexpr int rb = RB0 + decltype(jc)::value;                // register bit of this stage
        constexpr int beta = LOW + rb;
        TwPair W[4 >> rb], Wn[4 >> rb];
        static_for<(4 >> rb)>([&](auto uc) {
            constexpr unsigned u = decltype(uc)::value;
            const unsigned pg = (upper << (2 - rb)) | u;
            const unsigned idx = lat_tw_index<LOGN, false, beta>(pg);
            if constexpr (UNI) W[u] = tw[idx];
            else W[u] = buf_load_tw(twr, idx * 16u, 0u);
            if constexpr (SCALE) Wn[u] = twn[(1u << (LOGN - 1 - beta)) + u];     // reference indexing (upper = 0): entries [1, 8)
        });
        static_for<4>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int r0 = low_reg(rb, k), r1 = r0 | (1 << rb);
            constexpr bool zero_hist = SCALE && (((r0 & ((1 << rb) - 1)) >> RB0) == 0);
            lat_gs<LOGN, HL, NEAR, UNI, beta, IN2Q>(v[r0], v[r1], zero_hist ? Wn[r0 >> (rb + 1)] : W[r0 >> (rb + 1)], p);
        });
    });
    if constexpr (SCALE) {
        const TwPair ni = twn[0];
        static_for<(1 << RB0)>([&](auto rc) { constexpr int r = decltype(rc)::value; v[r] = lat_scale<HL, UNI>(v[r], ni, p); });
    }
}

// ---- "a" kernels: index bits LOGN-1 .. 9 ---------------------------------------------------------------------------------
// n/8 threads per polynomial in workgroups of WA = min(512, n/8) threads (KW = WA/64 waves); workgroup g of polynomial y,
// wave k, lane l; GB = log2(workgroups per polynomial) = max(LOGN - 12, 0):
//   layout A1 : register r holds index (r << (LOGN-3)) | (k << (6+GB)) | (g << 6) | l        (bits LOGN-1 .. LOGN-3 in the registers)
//   layout A2 : register r holds index (k << (LOGN-3)) | (r << (LOGN-6)) | (g << 6) | l      (bits LOGN-4 .. LOGN-6; n >= 2^13 only)
// n >= 2^13 has more than three stages here: the exchange between the layouts swaps the (3-bit) wave index with the register
// index through a 32 KiB image (one barrier), and the second round does the LOGN - 12 stages down to bit 9.
template <int LOGN>
struct LatGeo {
    static constexpr int N = 1 << LOGN;
    static constexpr int CH = LOGN - 9;                               // log2(chunks of 512 per polynomial) = stages of the "a" kernels
    static constexpr int WA = (N / 8 < 512) ? N / 8 : 512;            // "a" workgroup size
    static constexpr int GB = LOGN > 12 ? LOGN - 12 : 0;              // log2("a" workgroups per polynomial)
    static constexpr int NST1 = CH < 3 ? CH : 3;                      // stages of the round on the top register layout
    static constexpr int NST2 = CH - NST1;                            // stages of the second round (0: no exchange)
    static_assert(LOGN >= 11 && LOGN <= 15, "the small-batch kernels cover n = 2^11 .. 2^15");
};

__device__ __forceinline__ void lat_swap_kr(u64 (&v)[8], u64* lds, unsigned k, unsigned lane)
{
    __builtin_amdgcn_sched_barrier(0);
    static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; lds[(r * 8 + k) * 64 + lane] = v[r]; });
    __syncthreads();
    static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; v[r] = lds[(k * 8 + r) * 64 + lane]; });
    __builtin_amdgcn_sched_barrier(0);
}

template <int LOGN, int HL, bool NEAR>
__global__ void __launch_bounds__(LatGeo<LOGN>::WA, 1)
k_lat_fwd_a(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division, unsigned prime_base)
{
    if (guard_says_skip(primes, prime_base)) return;      // checked raw call whose table is not the cached one
    using L = LatGeo<LOGN>;
    __shared__ u64 lds[L::NST2 ? 4096 : 1];
    const unsigned y = blockIdx.x >> L::GB, g = blockIdx.x & ((1u << L::GB) - 1u);
    const unsigned k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    unsigned idx = __builtin_amdgcn_readfirstlane(prime_base + y % division);
    asm volatile("" : "+s"(idx));        // (pinned in an SGPR: the addresses derived from it stay scalar)
    const PrimeDev p = primes[idx];
    const TwPair* twp = tw + (size_t)idx * L::N;
    const BufRsrc twr = make_rsrc(twp, L::N * 16u), prs = make_rsrc(a + (size_t)y * L::N, L::N * 8u);
    const unsigned voff = ((g << 6) | lane) * 8u;
    u64 v[8];
    static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; v[r] = buf_load_u64(prs, voff, ((r << (LOGN - 3)) | (k << (6 + L::GB))) * 8u); });
    lat_fwd_round<LOGN, HL, NEAR, true, LOGN - 1, L::NST1>(v, twp, twr, p, 0u);
    if constexpr (L::NST2 > 0) {
        lat_swap_kr(v, lds, k, lane);
        lat_fwd_round<LOGN, HL, NEAR, true, LOGN - 4, L::NST2>(v, twp, twr, p, k);
        static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; buf_store_u64(prs, voff, ((k << (LOGN - 3)) | (r << (LOGN - 6))) * 8u, v[r]); });
    } else {
        static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; buf_store_u64(prs, voff, ((r << (LOGN - 3)) | (k << (6 + L::GB))) * 8u, v[r]); });
    }
}

template <int LOGN, int HL, bool NEAR>
__global__ void __launch_bounds__(LatGeo<LOGN>::WA, 1)
k_lat_inv_a(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division, unsigned prime_base)
{
    if (guard_says_skip(primes, prime_base)) return;
    using L = LatGeo<LOGN>;
    __shared__ u64 lds[L::NST2 ? 4096 : 1];
    const unsigned y = blockIdx.x >> L::GB, g = blockIdx.x & ((1u << L::GB) - 1u);
    const unsigned k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    unsigned idx = __builtin_amdgcn_readfirstlane(prime_base + y % division);
    asm volatile("" : "+s"(idx));        // (pinned in an SGPR: the addresses derived from it stay scalar)
    const PrimeDev p = primes[idx];
    const TwPair* twp = tw + (size_t)idx * L::N;
    const BufRsrc twr = make_rsrc(twp, L::N * 16u), prs = make_rsrc(a + (size_t)y * L::N, L::N * 8u);
    const unsigned voff = ((g << 6) | lane) * 8u;
    u64 v[8];
    if constexpr (L::NST2 > 0) {
        static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; v[r] = buf_load_u64(prs, voff, ((k << (LOGN - 3)) | (r << (LOGN - 6))) * 8u); });
        lat_inv_round<LOGN, HL, NEAR, true, LOGN - 6, L::NST2>(v, twp, twr, p, k);
        lat_swap_kr(v, lds, k, lane);
    } else {
        static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; v[r] = buf_load_u64(prs, voff, ((r << (LOGN - 3)) | (k << (6 + L::GB))) * 8u); });
    }
    lat_inv_round<LOGN, HL, NEAR, true, LOGN - 3, L::NST1, true>(v, twp, twr, p, 0u, primes[idx].twn);
    static_for<8>([&](auto rc) {
        constexpr unsigned r = decltype(rc)::value;
        buf_store_u64(prs, voff, ((r << (LOGN - 3)) | (k << (6 + L::GB))) * 8u, canon_after_inverse<HL, NEAR>(v[r], p));
    });
}

// ---- "b" kernels: index bits 8..0 on 512 consecutive coefficients per wave (chunk c of 2^(LOGN-9)) ------------------------
// layouts of the chunk-local index: L6 (r << 6) | l   (8-byte coalesced accesses, bits 8..6 in the registers),
// L3 ((l >> 3) << 6) | (r << 3) | (l & 7)  (bits 5..3),  L0 (l << 3) | r  (bits 2..0: 64 consecutive bytes per lane)
__device__ __forceinline__ void lat_load_l6(u64 (&v)[8], BufRsrc rs, unsigned c, unsigned lane)
{
    static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; v[r] = buf_load_u64(rs, lane * 8u, ((c << 9) | (r << 6)) * 8u); });
}
__device__ __forceinline__ void lat_store_l6(const u64 (&v)[8], BufRsrc rs, unsigned c, unsigned lane)
{
    static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; buf_store_u64(rs, lane * 8u, ((c << 9) | (r << 6)) * 8u, v[r]); });
}
__device__ __forceinline__ void lat_load_l0(u64 (&v)[8], BufRsrc rs, unsigned c, unsigned lane)
{
    static_for<4>([&](auto mc) {
        constexpr unsigned m = decltype(mc)::value;
        const TwPair x = buf_load_tw(rs, lane * 64u, (c << 9) * 8u + m * 16u);     // (a 16-byte load: two consecutive words)
        v[2 * m] = x.w;
        v[2 * m + 1] = x.wp;
    });
}
__device__ __forceinline__ void lat_store_l0(const u64 (&v)[8], BufRsrc rs, unsigned c, unsigned lane)
{
    static_for<4>([&](auto mc) {
        constexpr unsigned m = decltype(mc)::value;
        v4u32 x;
        x.x = lo32(v[2 * m]); x.y = hi32(v[2 * m]); x.z = lo32(v[2 * m + 1]); x.w = hi32(v[2 * m + 1]);
        __builtin_amdgcn_raw_buffer_store_b128(x, rs, lane * 64u, (c << 9) * 8u + m * 16u, 0);
    });
}

// bits 8..0 forward on registers: in L6, out L0, values in [0, B q)
template <int LOGN, int HL, bool NEAR>
__device__ __forceinline__ void lat_fwd_b_rounds(u64 (&v)[8], const TwPair* twp, BufRsrc twr, const PrimeDev& p, u64* slice, unsigned c, unsigned lane)
{
    lat_fwd_round<LOGN, HL, NEAR, true, 8>(v, twp, twr, p, c);
    lat_t_63(v, slice, lane);
    lat_fwd_round<LOGN, HL, NEAR, false, 5>(v, twp, twr, p, (c << 3) | (lane >> 3));
    lat_t_30(v, slice, lane);
    lat_fwd_round<LOGN, HL, NEAR, false, 2>(v, twp, twr, p, (c << 6) | lane);
}
// bits 0..8 inverse on registers: in L0, out L6
template <int LOGN, int HL, bool NEAR, bool IN2Q = false>
__device__ __forceinline__ void lat_inv_b_rounds(u64 (&v)[8], const TwPair* twp, BufRsrc twr, const PrimeDev& p, u64* slice, unsigned c, unsigned lane)
{
    lat_inv_round<LOGN, HL, NEAR, false, 0, 3, false, IN2Q>(v, twp, twr, p, (c << 6) | lane);
    lat_t_03(v, slice, lane);
    lat_inv_round<LOGN, HL, NEAR, false, 3>(v, twp, twr, p, (c << 3) | (lane >> 3));
    lat_t_36(v, slice, lane);
    lat_inv_round<LOGN, HL, NEAR, true, 6>(v, twp, twr, p, c);
}

template <int LOGN, int HL, bool NEAR>
__global__ void __launch_bounds__(64, 1)
k_lat_fwd_b(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division, unsigned prime_base)
{
    if (guard_says_skip(primes, prime_base)) return;
    __shared__ u64 slice[LAT_SLICE_WORDS];
    const unsigned y = blockIdx.x >> (LOGN - 9), c = blockIdx.x & ((1u << (LOGN - 9)) - 1u), lane = threadIdx.x;
    unsigned idx = __builtin_amdgcn_readfirstlane(prime_base + y % division);
    asm volatile("" : "+s"(idx));        // (pinned in an SGPR: the addresses derived from it stay scalar)
    const PrimeDev p = primes[idx];
    const TwPair* twp = tw + (size_t)idx * (1u << LOGN);
    const BufRsrc twr = make_rsrc(twp, (1u << LOGN) * 16u), prs = make_rsrc(a + (size_t)y * (1u << LOGN), (1u << LOGN) * 8u);
    u64 v[8];
    lat_load_l6(v, prs, c, lane);
    lat_fwd_b_rounds<LOGN, HL, NEAR>(v, twp, twr, p, slice, c, lane);
    static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; v[r] = canon_2q(reduce_2q_sel<NEAR>(v[r], p), p.q); });
    lat_store_l0(v, prs, c, lane);
}

template <int LOGN, int HL, bool NEAR>
__global__ void __launch_bounds__(64, 1)
k_lat_inv_b(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division, unsigned prime_base)
{
    if (guard_says_skip(primes, prime_base)) return;
    __shared__ u64 slice[LAT_SLICE_WORDS];
    const unsigned y = blockIdx.x >> (LOGN - 9), c = blockIdx.x & ((1u << (LOGN - 9)) - 1u), lane = threadIdx.x;
    unsigned idx = __builtin_amdgcn_readfirstlane(prime_base + y % division);
    asm volatile("" : "+s"(idx));        // (pinned in an SGPR: the addresses derived from it stay scalar)
    const PrimeDev p = primes[idx];
    const TwPair* twp = tw + (size_t)idx * (1u << LOGN);
    const BufRsrc twr = make_rsrc(twp, (1u << LOGN) * 16u), prs = make_rsrc(a + (size_t)y * (1u << LOGN), (1u << LOGN) * 8u);
    u64 v[8];
    lat_load_l0(v, prs, c, lane);
    lat_inv_b_rounds<LOGN, HL, NEAR>(v, twp, twr, p, slice, c, lane);
    lat_store_l6(v, prs, c, lane);
}

// fused small products: forward bits 8..0, product with bhat (Algorithm 7 on canonical operands, poly_arithmetic.cuh:36-66),
// inverse bits 0..8 -- all on the wave's own 512 coefficients:  k_lat_fwd_a -> k_lat_mul_b -> k_lat_inv_a
template <int LOGN, int HL, bool NEAR>
__global__ void __launch_bounds__(64, 1)
k_lat_mul_b(u64* __restrict__ a, const u64* __restrict__ bhat, const TwPair* __restrict__ twf, const TwPair* __restrict__ twi,
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
    const BufRsrc tfr = make_rsrc(tf, (1u << LOGN) * 16u), tir = make_rsrc(ti, (1u << LOGN) * 16u);
    const BufRsrc prs = make_rsrc(a + (size_t)y * (1u << LOGN), (1u << LOGN) * 8u);
    const BufRsrc brs = make_rsrc(bhat + (size_t)sb.index(y, idx, division) * (1u << LOGN), (1u << LOGN) * 8u);
    u64 v[8], bb[8];
    lat_load_l6(v, prs, c, lane);
    lat_load_l0(bb, brs, c, lane);
    lat_fwd_b_rounds<LOGN, HL, NEAR>(v, tf, tfr, p, slice, c, lane);
    static_for<8>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        v[r] = FusedMul<HL, NEAR>::mul(v[r], bb[r], p);
    });
    lat_inv_b_rounds<LOGN, HL, NEAR, FusedMul<HL, NEAR>::LAZY>(v, ti, tir, p, slice, c, lane);
    lat_store_l6(v, prs, c, lane);
}

}  // namespace mi355ntt
