// kernels_lat15.cuh -- n = 2^15, SMALL batches: the latency path (batch 1 is what the reference publishes: 39 us NTT /
// 23 us INTT on V100, Article.pdf p25 Table 6; one ciphertext per BFV call, p26 Table 7).
//
// Measured on gfx950 (tools/ubench_issue.hip, profiles/r02_ubench_issue_costs.txt): ONE wave issues a v_mad_u64_u32 every
// 10.4 cycles and any other VALU instruction every 6.5-8 however independent its instructions are -- a SIMD needs 2-3 waves
// to reach its 4-cycle rate.  With few polynomials the chip is empty, so the time of a transform is the instruction count of
// its longest wave: the persistent kernels (32 coefficients per thread, 240 butterflies per lane) need 75 us per polynomial
// pair, round 1/2's two-launch path with 32 coefficients per thread 14.5 / 15.8 us per transform whatever the host does
// (compiled C++, Python and a captured hipGraph all measure the same: profiles/r03_latency_cpp.txt).
//
// These kernels therefore spread a polynomial over 64 waves of EIGHT coefficients per thread (60 butterflies per lane):
//   forward : k_lat15_fwd_a  stages on index bits 14..9  (8 workgroups x 8 waves per polynomial; one LDS exchange between the
//                            two three-stage rounds swaps the wave index with the register index)
//             k_lat15_fwd_b  stages on index bits 8..0   (64 one-wave workgroups per polynomial, 512 consecutive coefficients
//                            each, two wave-local 8x8 transposes, no barrier)
//   inverse : k_lat15_inv_b  (bits 0..8), k_lat15_inv_a (bits 9..14, n^-1 folded into its last round's twiddles)
//   product : k_lat15_fwd_a, k_lat15_mul_b (forward bits 8..0, product with bhat, inverse bits 0..8), k_lat15_inv_a
// Values travel between the launches in lazy form [0, B q); the compile-time bound tracking (fwd_reduce_mask, InvPolicy) is
// per STAGE and therefore the same as in the single-pass kernels.  Outputs are canonical: the words equal theirs.
// Twiddles come from the same device tables (per-stage blocks permuted for the 32-coefficient geometry: tw_index15).
#pragma once
#include "ntt_core.cuh"

namespace mi355ntt {

// entry of group p of the stage on index bit beta in the device tables of n = 2^15 (kernels_fast.hip, fast_tables_create):
// the block [len, 2 len), len = 2^(14 - beta), is stored as len + u * nthi + thi with p = (thi << (4 - j)) + u for the round
// geometry (B, j) of the single-pass kernels that owns the stage
template <int BETA>
__host__ __device__ constexpr unsigned tw_index15(unsigned p)
{
    constexpr int B = BETA >= 10 ? 10 : BETA >= 5 ? 5 : 0, j = BETA - B;
    constexpr unsigned len = 1u << (14 - BETA), nthi = 1024u >> B;
    return len + (p & ((1u << (4 - j)) - 1u)) * nthi + (p >> (4 - j));
}

// ---- butterflies on two registers -----------------------------------------------------------------------------------
// CT stage s (index bit 14 - s): (a, b) <- (a + T, a + cq - T), T = b * w in [0, TQ q); U reduced first when the mask says so
template <int HL, bool NEAR, bool TWS, int S>
__device__ __forceinline__ void lat_ct(u64& a, u64& b, const TwPair w, const PrimeDev& p)
{
    constexpr bool EX = Lazy<HL>::EXACT;
    constexpr bool red = (fwd_reduce_mask<15, HL>() >> S) & 1u;
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

// GS stage on index bit BETA: (a, b) <- (a + b, (a + cq - b) * w); FIN: the sum is what leaves the transform (last stage)
template <int HL, bool NEAR, bool TWS, int BETA, bool IN2Q = false>
__device__ __forceinline__ void lat_gs(u64& a, u64& b, const TwPair w, const PrimeDev& p)
{
    static_assert(!IN2Q || !Lazy<HL>::EXACT, "lazy inputs: classes with 4q of headroom only (gs_round, ntt_core.cuh)");
    constexpr InvPolicy<15, HL> POL{};
    constexpr bool EX = Lazy<HL>::EXACT;
    constexpr bool last = (BETA == 14);
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

// ---- three in-register stages of a round ------------------------------------------------------------------------------
// Forward round on index bits TOP, TOP-1, TOP-2 = register bits 2, 1, 0.  The twiddle group of the stage on index bit
// beta = TOP - 2 + rb is p = i >> (beta + 1) = (upper << (2 - rb)) | (r >> (rb + 1)) with upper = i >> (TOP + 1), the part of
// the index above the round -- one value per thread and round.
// UNI: `upper` is wave-uniform (scalar twiddle loads, SGPR operands); else 16-byte vector loads through `twr`.
template <int HL, bool NEAR, bool UNI, int TOP>
__device__ __forceinline__ void lat_fwd_round(u64 (&v)[8], const TwPair* __restrict__ tw, BufRsrc twr, const PrimeDev& p, unsigned upper)
{
    static_for<3>([&](auto jc) {
        constexpr int rb = 2 - decltype(jc)::value;                  // register bit of this stage: 2, 1, 0
        constexpr int beta = TOP - 2 + rb, s = 14 - beta;
        TwPair W[4 >> rb];                                           // the stage's 1, 2 or 4 distinct twiddles first ...
        static_for<(4 >> rb)>([&](auto uc) {
            constexpr unsigned u = decltype(uc)::value;
            const unsigned idx = tw_index15<beta>((upper << (2 - rb)) | u);
            if constexpr (UNI) W[u] = tw[idx];
            else W[u] = buf_load_tw(twr, idx * 16u, 0u);
        });
        static_for<4>([&](auto kc) {                                 // ... then its four butterflies
            constexpr int k = decltype(kc)::value;
            constexpr int r0 = low_reg(rb, k), r1 = r0 | (1 << rb);
            lat_ct<HL, NEAR, UNI, s>(v[r0], v[r1], W[r0 >> (rb + 1)], p);
        });
    });
}

// Inverse round on index bits LOW, LOW+1, LOW+2 = register bits 0, 1, 2; upper = i >> (LOW + 3).  SCALE (the last round,
// bits 12..14, upper = 0): butterflies whose lower register bits are zero take twiddle * n^-1 from twn (gs_round,
// ntt_core.cuh), and register 0 -- summed in all three stages -- is multiplied by n^-1 itself.
template <int HL, bool NEAR, bool UNI, int LOW, bool SCALE = false, bool IN2Q = false>
__device__ __forceinline__ void lat_inv_round(u64 (&v)[8], const TwPair* __restrict__ tw, BufRsrc twr, const PrimeDev& p, unsigned upper,
                                              const TwPair* __restrict__ twn = nullptr)
{
    static_for<3>([&](auto jc) {
        constexpr int rb = decltype(jc)::value;                      // register bit of this stage: 0, 1, 2
        constexpr int beta = LOW + rb;
        TwPair W[4 >> rb], Wn[4 >> rb];
        static_for<(4 >> rb)>([&](auto uc) {
            constexpr unsigned u = decltype(uc)::value;
            const unsigned pg = (upper << (2 - rb)) | u;
            const unsigned idx = tw_index15<beta>(pg);
            if constexpr (UNI) W[u] = tw[idx];
            else W[u] = buf_load_tw(twr, idx * 16u, 0u);
            if constexpr (SCALE) Wn[u] = twn[(1u << (14 - beta)) + u];            // reference indexing (upper = 0): entries [1, 8)
        });
        static_for<4>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int r0 = low_reg(rb, k), r1 = r0 | (1 << rb);
            constexpr bool zero_hist = SCALE && ((r0 & ((1 << rb) - 1)) == 0);
            lat_gs<HL, NEAR, UNI, beta, IN2Q>(v[r0], v[r1], zero_hist ? Wn[r0 >> (rb + 1)] : W[r0 >> (rb + 1)], p);
        });
    });
    if constexpr (SCALE) v[0] = lat_scale<HL, UNI>(v[0], twn[0], p);
}

// ---- "a" kernels: index bits 14..9 -------------------------------------------------------------------------------------
// 512 threads; workgroup g (0..7) of polynomial y, wave k, lane l:
//   layout A1 : register r holds index (r << 12) | (k << 9) | (g << 6) | l   (bits 14..12 in the registers)
//   layout A2 : register r holds index (k << 12) | (r << 9) | (g << 6) | l   (bits 11..9 in the registers)
// The exchange between them swaps the wave index with the register index through a 32 KiB image (one barrier).
__device__ __forceinline__ void lat_swap_kr(u64 (&v)[8], u64* lds, unsigned k, unsigned lane)
{
    __builtin_amdgcn_sched_barrier(0);
    static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; lds[(r * 8 + k) * 64 + lane] = v[r]; });
    __syncthreads();
    static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; v[r] = lds[(k * 8 + r) * 64 + lane]; });
    __builtin_amdgcn_sched_barrier(0);
}

template <int HL, bool NEAR>
__global__ void __launch_bounds__(512, 1)
k_lat15_fwd_a(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division, unsigned prime_base)
{
    if (guard_says_skip(primes, prime_base)) return;      // checked raw call whose table is not the cached one
    __shared__ u64 lds[4096];
    const unsigned y = blockIdx.x >> 3, g = blockIdx.x & 7u;
    const unsigned k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const unsigned idx = __builtin_amdgcn_readfirstlane(prime_base + y % division);      // (uniform: addresses stay in SGPRs)
    const PrimeDev p = primes[idx];
    const TwPair* twp = tw + (size_t)idx * 32768u;
    const BufRsrc twr = make_rsrc(twp, 32768u * 16u), prs = make_rsrc(a + (size_t)y * 32768u, 32768u * 8u);
    const unsigned voff = ((g << 6) | lane) * 8u;
    u64 v[8];
    static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; v[r] = buf_load_u64(prs, voff, ((r << 12) | (k << 9)) * 8u); });
    lat_fwd_round<HL, NEAR, true, 14>(v, twp, twr, p, 0u);
    lat_swap_kr(v, lds, k, lane);
    lat_fwd_round<HL, NEAR, true, 11>(v, twp, twr, p, k);
    static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; buf_store_u64(prs, voff, ((k << 12) | (r << 9)) * 8u, v[r]); });
}

template <int HL, bool NEAR>
__global__ void __launch_bounds__(512, 1)
k_lat15_inv_a(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division, unsigned prime_base)
{
    if (guard_says_skip(primes, prime_base)) return;
    __shared__ u64 lds[4096];
    const unsigned y = blockIdx.x >> 3, g = blockIdx.x & 7u;
    const unsigned k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const unsigned idx = __builtin_amdgcn_readfirstlane(prime_base + y % division);      // (uniform: addresses stay in SGPRs)
    const PrimeDev p = primes[idx];
    const TwPair* twp = tw + (size_t)idx * 32768u;
    const BufRsrc twr = make_rsrc(twp, 32768u * 16u), prs = make_rsrc(a + (size_t)y * 32768u, 32768u * 8u);
    const unsigned voff = ((g << 6) | lane) * 8u;
    u64 v[8];
    static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; v[r] = buf_load_u64(prs, voff, ((k << 12) | (r << 9)) * 8u); });
    lat_inv_round<HL, NEAR, true, 9>(v, twp, twr, p, k);
    lat_swap_kr(v, lds, k, lane);
    lat_inv_round<HL, NEAR, true, 12, true>(v, twp, twr, p, 0u, primes[idx].twn);
    static_for<8>([&](auto rc) {
        constexpr unsigned r = decltype(rc)::value;
        buf_store_u64(prs, voff, ((r << 12) | (k << 9)) * 8u, canon_after_inverse<HL, NEAR>(v[r], p));
    });
}

// ---- "b" kernels: index bits 8..0 on 512 consecutive coefficients per wave --------------------------------------------
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
template <int HL, bool NEAR>
__device__ __forceinline__ void lat_fwd_b_rounds(u64 (&v)[8], const TwPair* twp, BufRsrc twr, const PrimeDev& p, u64* slice, unsigned c, unsigned lane)
{
    lat_fwd_round<HL, NEAR, true, 8>(v, twp, twr, p, c);
    lat_t_63(v, slice, lane);
    lat_fwd_round<HL, NEAR, false, 5>(v, twp, twr, p, (c << 3) | (lane >> 3));
    lat_t_30(v, slice, lane);
    lat_fwd_round<HL, NEAR, false, 2>(v, twp, twr, p, (c << 6) | lane);
}
// bits 0..8 inverse on registers: in L0, out L6
template <int HL, bool NEAR, bool IN2Q = false>
__device__ __forceinline__ void lat_inv_b_rounds(u64 (&v)[8], const TwPair* twp, BufRsrc twr, const PrimeDev& p, u64* slice, unsigned c, unsigned lane)
{
    lat_inv_round<HL, NEAR, false, 0, false, IN2Q>(v, twp, twr, p, (c << 6) | lane);
    lat_t_03(v, slice, lane);
    lat_inv_round<HL, NEAR, false, 3>(v, twp, twr, p, (c << 3) | (lane >> 3));
    lat_t_36(v, slice, lane);
    lat_inv_round<HL, NEAR, true, 6>(v, twp, twr, p, c);
}

template <int HL, bool NEAR>
__global__ void __launch_bounds__(64, 1)
k_lat15_fwd_b(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division, unsigned prime_base)
{
    if (guard_says_skip(primes, prime_base)) return;
    __shared__ u64 slice[LAT_SLICE_WORDS];
    const unsigned y = blockIdx.x >> 6, c = blockIdx.x & 63u, lane = threadIdx.x;
    const unsigned idx = __builtin_amdgcn_readfirstlane(prime_base + y % division);      // (uniform: addresses stay in SGPRs)
    const PrimeDev p = primes[idx];
    const TwPair* twp = tw + (size_t)idx * 32768u;
    const BufRsrc twr = make_rsrc(twp, 32768u * 16u), prs = make_rsrc(a + (size_t)y * 32768u, 32768u * 8u);
    u64 v[8];
    lat_load_l6(v, prs, c, lane);
    lat_fwd_b_rounds<HL, NEAR>(v, twp, twr, p, slice, c, lane);
    static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; v[r] = canon_2q(reduce_2q_sel<NEAR>(v[r], p), p.q); });
    lat_store_l0(v, prs, c, lane);
}

template <int HL, bool NEAR>
__global__ void __launch_bounds__(64, 1)
k_lat15_inv_b(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division, unsigned prime_base)
{
    if (guard_says_skip(primes, prime_base)) return;
    __shared__ u64 slice[LAT_SLICE_WORDS];
    const unsigned y = blockIdx.x >> 6, c = blockIdx.x & 63u, lane = threadIdx.x;
    const unsigned idx = __builtin_amdgcn_readfirstlane(prime_base + y % division);      // (uniform: addresses stay in SGPRs)
    const PrimeDev p = primes[idx];
    const TwPair* twp = tw + (size_t)idx * 32768u;
    const BufRsrc twr = make_rsrc(twp, 32768u * 16u), prs = make_rsrc(a + (size_t)y * 32768u, 32768u * 8u);
    u64 v[8];
    lat_load_l0(v, prs, c, lane);
    lat_inv_b_rounds<HL, NEAR>(v, twp, twr, p, slice, c, lane);
    lat_store_l6(v, prs, c, lane);
}

// fused small products: forward bits 8..0, product with bhat (Algorithm 7 on canonical operands, poly_arithmetic.cuh:36-66),
// inverse bits 0..8 -- all on the wave's own 512 coefficients:  k_lat15_fwd_a -> k_lat15_mul_b -> k_lat15_inv_a
template <int HL, bool NEAR>
__global__ void __launch_bounds__(64, 1)
k_lat15_mul_b(u64* __restrict__ a, const u64* __restrict__ bhat, const TwPair* __restrict__ twf, const TwPair* __restrict__ twi,
              const PrimeDev* __restrict__ primes, unsigned division)
{
    const SharedB sb(division);
    __shared__ u64 slice[LAT_SLICE_WORDS];
    const unsigned y = blockIdx.x >> 6, c = blockIdx.x & 63u, lane = threadIdx.x;
    const unsigned idx = __builtin_amdgcn_readfirstlane(y % division);
    const PrimeDev p = primes[idx];
    const TwPair* tf = twf + (size_t)idx * 32768u;
    const TwPair* ti = twi + (size_t)idx * 32768u;
    const BufRsrc tfr = make_rsrc(tf, 32768u * 16u), tir = make_rsrc(ti, 32768u * 16u);
    const BufRsrc prs = make_rsrc(a + (size_t)y * 32768u, 32768u * 8u);
    const BufRsrc brs = make_rsrc(bhat + (size_t)sb.index(y, idx, division) * 32768u, 32768u * 8u);
    u64 v[8], bb[8];
    lat_load_l6(v, prs, c, lane);
    lat_load_l0(bb, brs, c, lane);
    lat_fwd_b_rounds<HL, NEAR>(v, tf, tfr, p, slice, c, lane);
    static_for<8>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        v[r] = FusedMul<HL, NEAR>::mul(v[r], bb[r], p);
    });
    lat_inv_b_rounds<HL, NEAR, FusedMul<HL, NEAR>::LAZY>(v, ti, tir, p, slice, c, lane);
    lat_store_l6(v, prs, c, lane);
}

}  // namespace mi355ntt
