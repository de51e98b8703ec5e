// kernels_epi.cuh -- the fused product of n = 2^15 with a per-prime element-wise EPILOGUE in its store path (round 6, VERDICT r05
// item 4): the batched BFV drivers of configs[4] made element-wise passes over memory that need nothing but the polynomial the fused
// product already holds in registers.
//   decryption (bfv_decryption.cuh:98-122): c1 <- ((c1 s_hat + c0, `>`) (t gamma mod q)) (q~_i^-1 mod q) -- poly_add_xq_d,
//           poly_mul_int_xq_prodtgamma, poly_mul_int_xq_invpq: the three launches k_decrypt_scale had already fused into one pass.  Every
//           modulus of a context that runs these kernels is Barrett-exact, so the two Barrett products by constants are ONE exact product
//           by k = (t gamma) q~_i^-1 mod q_i -- a Shoup product with a precomputed companion, canonicalised: the same words for a third of
//           the multiplies.  (The sum may equal q, `>`: the reference's products then return 0, as does this one; bfv_host.cpp checks the
//           exactness bound for that operand as well before it hands the constants out.)
//   (The encryption's `+ e` (bfv_encryption.cuh:279) was built the same way and measured: 378.7 against 364.7 us per 64 ciphertexts --
//   the loads of e in front of a plain store cost the product more than k_encrypt_tail saves by not reading e.  Not shipped:
//   profiles/r06_bfv_batch.txt.)
// The second polynomial (c0 / e: the same position in a buffer of the same shape) arrives like the partner rows of the n = 2^16
// kernels: LDS-direct loads into the wave's slice (split_partner_fetch: two 512-byte rows of layout 10 per instruction, no VGPRs),
// sixteen rows requested right behind the workgroup-wide exchange -- they land during the last inverse round -- and the other sixteen
// while the first are combined, behind counted vmcnt waits.  A kernel of its own (the loop of k_polymul15 with the epilogue in place of
// the plain store) so that k_polymul15 itself stays the code that was tuned.
#pragma once

namespace mi355ntt {

struct EpiPrime {             // per prime of the call (index y % division); read by scalar loads
    u64 k1, k2;               // k = (t gamma) q~_i^-1 mod q_i and its Shoup companion floor(k 2^64 / q_i)
    unsigned on;              // 0: this polynomial is stored as the product leaves it (the dropped prime's slot of a decryption batch)
    unsigned pad;
};
struct PolymulEpi {
    const u64* other;         // c0 / e: polynomial y at other + y n
    const EpiPrime* consts;
};

// x k mod q, canonical, for x <= q: the exact-quotient Shoup product (result in [0, 2q)) and one conditional subtraction
__device__ __forceinline__ u64 epi_scale(u64 x, const EpiPrime& ec, const PrimeDev& p)
{
    return canon_2q(mul_shoup2(x, ec.k1, ec.k2, p.nq), p.q);
}

template <int HL, bool NEAR, int EPI>
__global__ void __launch_bounds__(1024, 4)
k_polymul15_epi(u64* __restrict__ a, const u64* __restrict__ bhat, const TwPair* __restrict__ twf, const TwPair* __restrict__ twi,
                const PrimeDev* __restrict__ primes, unsigned division, unsigned num, PolymulEpi epi)
{
    static_assert(EPI == 1, "1: decryption scale");
    const bool stream_b = (division & (kSharedB | kStreamLoads)) == kStreamLoads;
    if (stream_b) division &= ~kStreamLoads;
    const SharedB sb(division);
    constexpr int LOGN = 15;
    using G = Geo<LOGN>;
    __shared__ __attribute__((aligned(16))) u64 lds[G::LDS_WORDS];
    unsigned wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+s"(wave_s));
    auto fresh_t = [&]() { return (wave_s << 6) | fresh_lane_id(); };
    u64* slice = lds + wave_s * WAVE_SLICE_WORDS;
    u64 v[32];
    unsigned y = blockIdx.x;
    stagger_start<Tune::kStaggerMul, Tune::kStaggerMulMulti>(num > gridDim.x);
    load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)y * G::N, fresh_t());
    unsigned ymod = __builtin_amdgcn_readfirstlane(blockIdx.x % division), ystep = __builtin_amdgcn_readfirstlane(gridDim.x % division);
    asm volatile("" : "+s"(ymod), "+s"(ystep));
    for (; y < num; y += gridDim.x, ymod = (ymod + ystep >= division ? ymod + ystep - division : ymod + ystep)) {
        const unsigned idx = ymod;
        const PrimeDev p = primes[idx];
        const EpiPrime ec = epi.consts[idx];
        const TwPair* tf = twf + (size_t)idx * G::N;
        const TwPair* ti = twi + (size_t)idx * G::N;
        const BufRsrc tfr = make_rsrc(tf, G::N * 16u), tir = make_rsrc(ti, G::N * 16u);
        u64* poly = a + (size_t)y * G::N;
        const BufRsrc brs = make_rsrc(bhat + (size_t)sb.index(y, idx, division) * G::N + wave_s * 2048u, 16384u);
        // ---- forward ----
        __builtin_amdgcn_s_setprio(Tune::kPrioR1);
        ct_round<LOGN, HL, 10, 4, NEAR>(v, tf, tfr, 0u, p);
        __syncthreads();
        exchange<LOGN, 10, 5>(v, lds, fresh_t());
        __builtin_amdgcn_s_setprio(Tune::kPrioR2);
        ct_round<LOGN, HL, 5, 4, NEAR>(v, tf, tfr, fresh_t(), p);
        wave_transpose_5_to_0(v, slice, fresh_lane_id());
        __builtin_amdgcn_s_setprio(Tune::kPrioR3);
        ct_round<LOGN, HL, 0, 4, NEAR>(v, tf, tfr, fresh_t(), p);
        // ---- pointwise product with bhat ----
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
        // (the exchange ends with a barrier: nobody touches this wave's slice until the next polynomial's transposes) rows 0..15 of the
        // second polynomial start their way from memory now and land during the last round
        const BufRsrc ors = make_rsrc(epi.other + (size_t)y * G::N, G::N * 8u);
        if (ec.on) {
            split_partner_fetch<0>(ors, slice, wave_s, fresh_lane_id());
            split_partner_fetch<1>(ors, slice, wave_s, fresh_lane_id());
        }
        __builtin_amdgcn_s_setprio(Tune::kPrioI3);
        gs_round<LOGN, HL, 10, 0, NEAR>(v, ti, tir, fresh_t(), p, primes[idx].twn);      // (twiddles through the scalar cache: no vector-memory operation)
        static_for<32>([&](auto rc) { v[decltype(rc)::value] = canon_after_inverse<HL, NEAR>(v[decltype(rc)::value], p); });
        if (ec.on) {
            const BufRsrc prs = make_rsrc(poly, G::N * 8u);
            static_for<4>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                // counted waits (hipcc does not order an LDS read behind the LDS-direct load that fills it; loads, stores and LDS-direct
                // loads retire in issue order on one counter) for phase c's four loads.  Younger than them in the queue: phase 1's loads
                // (c = 0); phase 2's loads and phase 0's eight stores (c = 1); phase 0's stores, phase 3's loads and phase 1's stores
                // (c = 2); the stores of phases 1 and 2 (c = 3) -- the sequence of k_forward15's coupling stage (SPLIT)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (c == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if constexpr (c == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if constexpr (c == 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                u64 O[8];
                split_partner_read<c>(O, slice, fresh_lane_id());
                wave_lds_fence();                         // the buffer is free ...
                if constexpr (c < 2) split_partner_fetch<c + 2>(ors, slice, wave_s, fresh_lane_id());     // ... for phase c + 2
                __builtin_amdgcn_sched_barrier(0);
                const unsigned voff = fresh_t() * 8u;
                static_for<8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, r = 8 * c + i;
                    u64 ra = v[r] + O[i];
                    if (ra > p.q) ra -= p.q;              // poly_add_xq_d: `>`, not `>=` (a sum equal to q stays q)
                    ra = epi_scale(ra, ec, p);
                    v2u32 x;
                    x.x = lo32(ra); x.y = hi32(ra);
                    __builtin_amdgcn_raw_buffer_store_b64(x, prs, voff, ((unsigned)r << G::B0) * 8u, Tune::kInv15AuxSt);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
            store_coalesced<LOGN, Tune::kInv15AuxSt>(v, poly, fresh_t());
        }
        if (y + gridDim.x < num) load_coalesced<LOGN, Tune::kFwdLoadPair16>(v, a + (size_t)(y + gridDim.x) * G::N, fresh_t());
    }
}

// The same epilogue behind the small-batch product (kernels_lat.cuh: k_lat_fwd_a -> k_lat_mul_b -> k_lat_inv_a): the last kernel of the
// three with the second polynomial's eight words per thread requested at its start -- a batch of 64 ciphertexts on 4 + 1 primes
// decrypts 320 polynomials per call, which is the window just above one polynomial per CU where the small-batch kernels run.
template <int LOGN, int HL, bool NEAR, int EPI>
__global__ void __launch_bounds__(LatGeo<LOGN>::WA, 1)
k_lat_inv_a_epi(u64* __restrict__ a, const TwPair* __restrict__ tw, const PrimeDev* __restrict__ primes, unsigned division, PolymulEpi epi)
{
    using L = LatGeo<LOGN>;
    static_assert(L::NST2 > 0, "n = 2^15 only");
    __shared__ u64 lds[4096];
    const unsigned y = blockIdx.x >> L::GB, g = blockIdx.x & ((1u << L::GB) - 1u);
    const unsigned k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    unsigned idx = __builtin_amdgcn_readfirstlane(y % division);
    asm volatile("" : "+s"(idx));
    const PrimeDev p = primes[idx];
    const EpiPrime ec = epi.consts[idx];
    const TwPair* twp = tw + (size_t)idx * L::N;
    const BufRsrc twr = make_rsrc(twp, L::N * 16u), prs = make_rsrc(a + (size_t)y * L::N, L::N * 8u);
    const BufRsrc ors = make_rsrc(epi.other + (size_t)y * L::N, L::N * 8u);
    const unsigned voff = ((g << 6) | lane) * 8u;
    u64 v[8], o[8];
    static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; v[r] = buf_load_u64(prs, voff, ((k << (LOGN - 3)) | (r << (LOGN - 6))) * 8u); });
    if (ec.on) static_for<8>([&](auto rc) { constexpr unsigned r = decltype(rc)::value; o[r] = buf_load_u64(ors, voff, ((r << (LOGN - 3)) | (k << (6 + L::GB))) * 8u); });
    lat_inv_round<LOGN, HL, NEAR, true, LOGN - 6, L::NST2>(v, twp, twr, p, k);
    lat_swap_kr(v, lds, k, lane);
    lat_inv_round<LOGN, HL, NEAR, true, LOGN - 3, L::NST1, true>(v, twp, twr, p, 0u, primes[idx].twn);
    static_for<8>([&](auto rc) {
        constexpr unsigned r = decltype(rc)::value;
        u64 ra = canon_after_inverse<HL, NEAR>(v[r], p);
        if (ec.on) {
            ra += o[r];
            if (ra > p.q) ra -= p.q;                      // poly_add_xq_d: `>`
            ra = epi_scale(ra, ec, p);
        }
        buf_store_u64(prs, voff, ((r << (LOGN - 3)) | (k << (6 + L::GB))) * 8u, ra);
    });
}

// Classes whose kernel holds the epilogue without leaving the register file (compiler remarks, tools/kernel_resources.py): every
// near-2^k class with 4q of headroom and the general class 6.  The exact-quotient classes (62-bit moduli) and the general classes 4 and 3
// take 8-48 bytes of scratch with it and keep the two separate steps.
template <int H, bool NR, int EPI>
constexpr bool epi_class() { return NR ? H >= 3 : H == 6; }

// the fused product of `num` polynomials with the epilogue, one persistent launch (the callers have asked fast_polymul_epi_ok)
template <int EPI>
inline hipError_t launch_mul_epi15(int hl, u64* d_a, const u64* d_b, const TwPair* twf, const TwPair* twi, const PrimeDev* pr, unsigned num,
                                   unsigned division, hipStream_t s, const PolymulEpi& epi)
{
    bool launched = false;
    if (use_latency_path<15>(num, true)) {
        using L = LatGeo<15>;
        dispatch_class(hl, [&](auto hc, auto nc) {
            constexpr int H = decltype(hc)::value;
            constexpr bool NR = decltype(nc)::value;
            k_lat_fwd_a<15, H, NR><<<dim3(num << L::GB), dim3(L::WA), 0, s>>>(d_a, twf, pr, plain_division(division), 0u);
            k_lat_mul_b<15, H, NR><<<dim3(num << L::CH), dim3(64), 0, s>>>(d_a, d_b, twf, twi, pr, division);
            k_lat_inv_a_epi<15, H, NR, EPI><<<dim3(num << L::GB), dim3(L::WA), 0, s>>>(d_a, twi, pr, plain_division(division), epi);
        });
        return hipGetLastError();
    }
    dispatch_class(hl, [&](auto hc, auto nc) {
        constexpr int H = decltype(hc)::value;
        constexpr bool NR = decltype(nc)::value;
        if constexpr (epi_class<H, NR, EPI>()) {
            const unsigned dw = division | ((division & kSharedB) == 0 && num > kMulStreamLoadsAbove ? kStreamLoads : 0u);
            k_polymul15_epi<H, NR, EPI><<<dim3(persistent_grid<15>(num)), dim3(1024), 0, s>>>(d_a, d_b, twf, twi, pr, dw, num, epi);
            launched = true;
        }
    });
    return launched ? hipGetLastError() : hipErrorNotSupported;
}

}  // namespace mi355ntt
