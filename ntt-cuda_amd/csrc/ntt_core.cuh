// ntt_core.cuh -- single-pass, register-resident NTT / INTT for gfx950.
//
// One workgroup transforms one polynomial with ONE read and ONE write of HBM (the reference makes 4
// (forward) / 5 (inverse) full passes, ntt_60bit.cuh:318-324,354-359).  n/32 threads each keep 32
// coefficients in VGPRs; log2(n) stages are done as rounds of up to five in-register radix-2 stages,
// with an LDS transposition between rounds that moves the next five index bits into the register
// position.  Arithmetic is the lazy Harvey/Shoup form (values kept in [0, B*q), B tracked at compile
// time, one cheap partial reduction every few stages), and outputs are canonicalised at the end, so
// the words written are exactly those the reference's canonical-every-stage Barrett butterflies
// (ntt_60bit.cuh:86-110,151-178) produce.
//
// Index conventions: CT stage s works on index bit LOGN-1-s with twiddle
// tab[2^s + (i >> (LOGN-s))]; GS stage on index bit beta uses tab[2^(LOGN-1-beta) + (i >> (beta+1))].
#pragma once
#include <hip/hip_runtime.h>

#include "modarith.cuh"

namespace mi355ntt {

// Per-prime constants, read with scalar loads (replaces __constant__ q_cons/mu_cons/q_bit_cons).
struct PrimeDev {
    u64 q, nq;        // modulus and 2^64 - q
    u64 ninv, ninv_p; // n^-1 mod q and its Shoup companion
    u64 w1n, w1n_p;   // psi^-bitrev(1) * n^-1 (last GS stage with the scaling folded in) and companion
    u64 mu;           // Barrett mu = floor(2^(2k)/q), reference convention (pointwise products)
    u32 red_c;        // floor(2^(31+k) / q), 32 bits
    u32 red_sh1;      // k - 1 - g
    u32 red_sh2;      // g = min(16, k-1)
    u32 k;            // bit length
};

struct TwPair {       // {w, floor(w * 2^64 / q)}
    u64 w, wp;
};

// Buffer-descriptor loads/stores: one 32-bit lane offset VGPR serves every access of a thread, the
// per-register displacement rides in the scalar offset / immediate (no 64-bit address arithmetic in VGPRs).
typedef u32 v2u32 __attribute__((ext_vector_type(2)));
typedef u32 v4u32 __attribute__((ext_vector_type(4)));
using BufRsrc = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ BufRsrc make_rsrc(const void* base, u32 bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u64 buf_load_u64(BufRsrc r, u32 voff, u32 soff)
{
    const v2u32 x = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return (u64)x.x | ((u64)x.y << 32);
}
__device__ __forceinline__ void buf_store_u64(BufRsrc r, u32 voff, u32 soff, u64 v)
{
    v2u32 x;
    x.x = lo32(v);
    x.y = hi32(v);
    __builtin_amdgcn_raw_buffer_store_b64(x, r, voff, soff, 0);
}
__device__ __forceinline__ TwPair buf_load_tw(BufRsrc r, u32 voff, u32 soff)
{
    const v4u32 x = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    TwPair t;
    t.w = (u64)x.x | ((u64)x.y << 32);
    t.wp = (u64)x.z | ((u64)x.w << 32);
    return t;
}

// ------------------------------------------------------------------------------------------------
// lazy modular primitives
// ------------------------------------------------------------------------------------------------

// y*w mod q, result congruent and in [0, 4q).  y: any 64-bit value.  wp = floor(w*2^64/q).
// Quotient estimate from three of the four partial products (error <= 2), remainder as y*w + h*(2^64-q).
__device__ __forceinline__ u64 mul_shoup4(u64 y, u64 w, u64 wp, u64 nq)
{
    u32 y0 = lo32(y), y1 = hi32(y), p0 = lo32(wp), p1 = hi32(wp);
    u64 h = mad32(y1, p1, (u64)__umulhi(y0, p1)) + (u64)__umulhi(y1, p0);
    return y * w + h * nq;
}

// exact-quotient variant, result in [0, 2q)  (needed only when 4q does not fit: q >= 2^61... see policy)
__device__ __forceinline__ u64 mul_shoup2(u64 y, u64 w, u64 wp, u64 nq)
{
    u64 h = mul_hi(y, wp);
    return y * w + h * nq;
}

template <bool EXACT>
__device__ __forceinline__ u64 mul_shoup(u64 y, u64 w, u64 wp, u64 nq)
{
    if constexpr (EXACT) return mul_shoup2(y, w, wp, nq);
    else return mul_shoup4(y, w, wp, nq);
}

// x in [0, B*q) (B*q < 2^64, B <= 66 or B*q < 2^(k+5)) -> congruent value in [0, 2q).
// e = floor(x/q) or one less, from the top bits of x and a 32-bit reciprocal.
__device__ __forceinline__ u64 reduce_2q(u64 x, const PrimeDev& p)
{
    u32 t = (u32)(x >> p.red_sh1);
    u32 e = __umulhi(t, p.red_c) >> p.red_sh2;
    return x + (u64)e * p.nq;      // e < 2^32: low 64 bits of e*nq added = x - e*q
}

// [0, 2q) -> [0, q)
__device__ __forceinline__ u64 canon_2q(u64 x, u64 q) { return x >= q ? x - q : x; }

// ------------------------------------------------------------------------------------------------
// compile-time bound tracking
// ------------------------------------------------------------------------------------------------
// HL = log2 of the headroom class: every value must stay below 2^HL * q <= 2^64 (HL = 64 - bit length,
// capped at 6 = "never needs an intermediate reduction for n <= 2^15").
template <int HL>
struct Lazy {
    static constexpr bool EXACT = (HL <= 2);          // 4q does not fit below 2^64 for 62-bit moduli
    static constexpr int TQ = EXACT ? 2 : 4;          // product range [0, TQ*q)
    static constexpr long H = 1L << HL;
};

// Forward: bit s of the mask = "reduce the U inputs to [0,2q) before CT stage s".
template <int LOGN, int HL>
constexpr unsigned fwd_reduce_mask()
{
    unsigned m = 0;
    long B = 1;
    for (int s = 0; s < LOGN; s++) {
        if (B + Lazy<HL>::TQ > Lazy<HL>::H) {
            m |= 1u << s;
            B = 2;
        }
        B += Lazy<HL>::TQ;
    }
    return m;
}

// Inverse: bit s of the mask = "reduce the sum outputs to [0,2q) right after GS stage s" (s counts
// from 0 = first GS stage); cmul[s] = multiple of q added to (x - y) in stage s.
template <int LOGN, int HL>
struct InvPolicy {
    unsigned mask = 0;
    int cmul[16] = {};
    constexpr InvPolicy()
    {
        long B = 1;
        for (int s = 0; s < LOGN; s++) {
            cmul[s] = (int)B;                         // y < B*q
            long Bn = 2 * B > Lazy<HL>::TQ ? 2 * B : Lazy<HL>::TQ;   // sums < 2B q, products < TQ q
            if (s + 1 < LOGN && 2 * Bn > Lazy<HL>::H) {
                mask |= 1u << s;
                Bn = Lazy<HL>::TQ > 2 ? Lazy<HL>::TQ : 2;
            }
            B = Bn;
        }
    }
};

// ------------------------------------------------------------------------------------------------
// geometry
// ------------------------------------------------------------------------------------------------
template <int LOGN>
struct Geo {
    static constexpr int N = 1 << LOGN;
    static constexpr int T = N / 32;                  // threads per polynomial
    static constexpr int B0 = LOGN - 5;               // coalesced layout: i = (r << B0) | t
    static constexpr int NR = (LOGN + 4) / 5;         // rounds
    static constexpr bool TWO_PHASE = (N * 8 > 131072);   // LDS holds half a 2^15 polynomial at a time
    static constexpr int PB = LOGN - 1;               // index bit that selects the phase
    static constexpr int ROWS = (TWO_PHASE ? N / 2 : N) / 32;
    static constexpr int LDS_WORDS = ROWS * 34;       // 32 columns + 2 words (16 B) of padding per row
};

// element index held by thread t in register r for a layout whose register field sits at bit B
template <int B>
__device__ __forceinline__ unsigned elem_index(unsigned t, unsigned r)
{
    return ((t >> B) << (B + 5)) | (r << B) | (t & ((1u << B) - 1u));
}

// LDS image: element i lives at row (i' >> 5), column (i' & 31) of a [ROWS][34] array of u64 (two words of
// padding per row keep 8- and 16-byte accesses conflict-free for every layout), where i' = i without the
// phase bit.  For a layout with register field at bit B the slot splits into a per-thread base and a
// per-register COMPILE-TIME offset, so every access is base VGPR + immediate.
template <int LOGN, int B>
__device__ __forceinline__ unsigned slot_base(unsigned t)
{
    using G = Geo<LOGN>;
    unsigned thi = t >> B, tlo = t & ((1u << B) - 1u);
    if constexpr (B >= 5) {
        unsigned i = (thi << (B + 5)) | tlo;                      // register field zero
        if constexpr (G::TWO_PHASE) i &= (1u << G::PB) - 1u;
        return (i >> 5) * 34u + (i & 31u);
    } else {
        unsigned row = thi << B;                                  // + (r >> (5-B)) from the register
        if constexpr (G::TWO_PHASE) row &= (1u << (G::PB - 5)) - 1u;
        return row * 34u + tlo;
    }
}

template <int LOGN, int B>
constexpr unsigned slot_off(unsigned r)
{
    using G = Geo<LOGN>;
    if (B >= 5) {
        unsigned c = r << B;
        if (G::TWO_PHASE) c &= (1u << G::PB) - 1u;
        return (c >> 5) * 34u;
    }
    return (r >> (5 - B)) * 34u + ((r & ((1u << (5 - B)) - 1u)) << B);
}

// which phase a (thread, register) element belongs to: the top index bit
template <int LOGN, int B>
__device__ __forceinline__ unsigned phase_of_thread(unsigned t)
{
    return (elem_index<B>(t, 0) >> Geo<LOGN>::PB) & 1u;
}
template <int LOGN, int B>
constexpr unsigned phase_of_reg(unsigned r)
{
    return ((r << B) >> Geo<LOGN>::PB) & 1u;
}

// Transposition through LDS: registers hold layout BO on entry, layout BN on exit.
template <int LOGN, int BO, int BN>
__device__ __forceinline__ void exchange(u64 (&v)[32], u64* lds, unsigned t)
{
    using G = Geo<LOGN>;
    constexpr int PH = G::TWO_PHASE ? 2 : 1;
    constexpr bool W_REG_SPLIT = G::TWO_PHASE && (G::PB >= BO && G::PB < BO + 5);   // phase bit is a writer register bit
    constexpr bool R_REG_SPLIT = G::TWO_PHASE && (G::PB >= BN && G::PB < BN + 5);   // phase bit is a reader register bit
    const unsigned w_phase = G::TWO_PHASE ? phase_of_thread<LOGN, BO>(t) : 0u;
    const unsigned r_phase = G::TWO_PHASE ? phase_of_thread<LOGN, BN>(t) : 0u;
    u64* wbase = lds + slot_base<LOGN, BO>(t);
    const u64* rbase = lds + slot_base<LOGN, BN>(t);
    u64 nv[32];
#pragma unroll
    for (int ph = 0; ph < PH; ph++) {
        // ---- write ----
        if (W_REG_SPLIT || !G::TWO_PHASE || w_phase == (unsigned)ph) {
#pragma unroll
            for (int r = 0; r < 32; r += (BO == 0 ? 2 : 1)) {
                if (W_REG_SPLIT && phase_of_reg<LOGN, BO>(r) != (unsigned)ph) continue;
                if constexpr (BO == 0)
                    *reinterpret_cast<ulonglong2*>(wbase + slot_off<LOGN, BO>(r)) = make_ulonglong2(v[r], v[r + 1]);
                else
                    wbase[slot_off<LOGN, BO>(r)] = v[r];
            }
        }
        __syncthreads();
        // ---- read ----
        if (R_REG_SPLIT || !G::TWO_PHASE || r_phase == (unsigned)ph) {
#pragma unroll
            for (int r = 0; r < 32; r += (BN == 0 ? 2 : 1)) {
                if (R_REG_SPLIT && phase_of_reg<LOGN, BN>(r) != (unsigned)ph) continue;
                if constexpr (BN == 0) {
                    const ulonglong2 pr = *reinterpret_cast<const ulonglong2*>(rbase + slot_off<LOGN, BN>(r));
                    nv[r] = pr.x;
                    nv[r + 1] = pr.y;
                } else {
                    nv[r] = rbase[slot_off<LOGN, BN>(r)];
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 32; r++) v[r] = nv[r];
}

// ------------------------------------------------------------------------------------------------
// in-register rounds
// ------------------------------------------------------------------------------------------------
// The whole transform is straight-line code; without fences the scheduler hoists dozens of twiddle loads
// (4 VGPRs each) and spills.  A scheduling fence every SCHED_GROUP butterflies bounds the live set; the
// other three waves of the SIMD cover the load latency.
#ifndef MI355NTT_SCHED_GROUP
#define MI355NTT_SCHED_GROUP 4
#endif
constexpr int SCHED_GROUP = MI355NTT_SCHED_GROUP;

// Forward (CT) stages on register bits JHI..0 of a layout with register field at bit B.
// S0 = global stage number of the first stage in this round.
template <int LOGN, int HL, int B, int JHI>
__device__ __forceinline__ void ct_round(u64 (&v)[32], const TwPair* __restrict__ tw, BufRsrc twr, unsigned t, const PrimeDev& p)
{
    constexpr unsigned RMASK = fwd_reduce_mask<LOGN, HL>();
    constexpr bool EX = Lazy<HL>::EXACT;
    const u64 cq = (u64)Lazy<HL>::TQ * p.q;
    const unsigned thi = t >> B;
    int cnt = 0;
#pragma unroll
    for (int j = JHI; j >= 0; j--) {
        const int s = LOGN - 1 - (B + j);
        const bool red = (RMASK >> s) & 1u;
#pragma unroll
        for (int r0 = 0; r0 < 32; r0++) {
            if (r0 & (1 << j)) continue;
            const int r1 = r0 | (1 << j);
            TwPair W;
            if constexpr (B == Geo<LOGN>::B0)      // twiddle index does not depend on the thread: scalar load
                W = tw[(1u << s) + ((unsigned)r0 >> (j + 1))];
            else
                W = buf_load_tw(twr, (thi << (4 - j)) * 16u, ((1u << s) + ((unsigned)r0 >> (j + 1))) * 16u);
            u64 U = v[r0];
            if (red) U = reduce_2q(U, p);
            const u64 Tm = mul_shoup<EX>(v[r1], W.w, W.wp, p.nq);
            v[r0] = U + Tm;
            v[r1] = U + cq - Tm;
            if ((++cnt % SCHED_GROUP) == 0) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Inverse (GS) stages on register bits JLO..4 of a layout with register field at bit B.
template <int LOGN, int HL, int B, int JLO>
__device__ __forceinline__ void gs_round(u64 (&v)[32], const TwPair* __restrict__ tw, BufRsrc twr, unsigned t, const PrimeDev& p)
{
    constexpr InvPolicy<LOGN, HL> POL{};
    constexpr bool EX = Lazy<HL>::EXACT;
    const unsigned thi = t >> B;
    int cnt = 0;
#pragma unroll
    for (int j = JLO; j <= 4; j++) {
        const int beta = B + j;                 // index bit of this stage
        const int s = beta;                     // GS stage number (0 = first)
        const bool last = (beta == LOGN - 1);
        const bool red = (POL.mask >> s) & 1u;
        const u64 cq = (u64)POL.cmul[s] * p.q;
#pragma unroll
        for (int r0 = 0; r0 < 32; r0++) {
            if (r0 & (1 << j)) continue;
            const int r1 = r0 | (1 << j);
            const u64 X = v[r0], Y = v[r1];
            u64 S = X + Y;
            const u64 D = X + cq - Y;
            if (last) {
                // length = 1: single twiddle, n^-1 folded into both outputs (the reference halves every stage)
                v[r0] = mul_shoup<EX>(S, p.ninv, p.ninv_p, p.nq);
                v[r1] = mul_shoup<EX>(D, p.w1n, p.w1n_p, p.nq);
            } else {
                TwPair W;
                if constexpr (B == Geo<LOGN>::B0)
                    W = tw[(1u << (LOGN - 1 - beta)) + ((unsigned)r0 >> (j + 1))];
                else
                    W = buf_load_tw(twr, (thi << (4 - j)) * 16u, ((1u << (LOGN - 1 - beta)) + ((unsigned)r0 >> (j + 1))) * 16u);
                if (red) S = reduce_2q(S, p);
                v[r0] = S;
                v[r1] = mul_shoup<EX>(D, W.w, W.wp, p.nq);
            }
            if ((++cnt % SCHED_GROUP) == 0) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// whole transforms on registers.  Entry and exit layout: B0 (coalesced: i = (r << B0) | t).
// ------------------------------------------------------------------------------------------------
template <int LOGN, int HL, int RHO>
__device__ __forceinline__ void fwd_rounds(u64 (&v)[32], const TwPair* tw, BufRsrc twr, unsigned t, const PrimeDev& p, u64* lds)
{
    using G = Geo<LOGN>;
    if constexpr (RHO < G::NR) {
        constexpr int TOP = LOGN - 1 - 5 * RHO;
        constexpr int B = TOP - 4 > 0 ? TOP - 4 : 0;
        if constexpr (RHO > 0) {
            constexpr int TOPP = LOGN - 1 - 5 * (RHO - 1);
            constexpr int BP = TOPP - 4 > 0 ? TOPP - 4 : 0;
            exchange<LOGN, BP, B>(v, lds, t);
        }
        ct_round<LOGN, HL, B, TOP - B>(v, tw, twr, t, p);
        fwd_rounds<LOGN, HL, RHO + 1>(v, tw, twr, t, p, lds);
    }
}

// natural-order coefficients in (layout B0, canonical) -> bit-reversed NTT values, left in layout 0, in [0, B*q)
template <int LOGN, int HL>
__device__ __forceinline__ void forward_core(u64 (&v)[32], const TwPair* tw, unsigned t, const PrimeDev& p, u64* lds)
{
    fwd_rounds<LOGN, HL, 0>(v, tw, make_rsrc(tw, Geo<LOGN>::N * 16u), t, p, lds);
}

template <int LOGN, int HL, int RHO>
__device__ __forceinline__ void inv_rounds(u64 (&v)[32], const TwPair* tw, BufRsrc twr, unsigned t, const PrimeDev& p, u64* lds)
{
    using G = Geo<LOGN>;
    if constexpr (RHO < G::NR) {
        constexpr int LOW = 5 * RHO;
        constexpr int B = LOW < G::B0 ? LOW : G::B0;
        if constexpr (RHO > 0) {
            constexpr int LOWP = 5 * (RHO - 1);
            constexpr int BP = LOWP < G::B0 ? LOWP : G::B0;
            exchange<LOGN, BP, B>(v, lds, t);
        }
        gs_round<LOGN, HL, B, LOW - B>(v, tw, twr, t, p);
        inv_rounds<LOGN, HL, RHO + 1>(v, tw, twr, t, p, lds);
    }
}

// bit-reversed values in layout 0 (any representative below 2q... see callers) -> coefficients in layout B0, in [0, TQ*q)
template <int LOGN, int HL>
__device__ __forceinline__ void inverse_core(u64 (&v)[32], const TwPair* tw, unsigned t, const PrimeDev& p, u64* lds)
{
    inv_rounds<LOGN, HL, 0>(v, tw, make_rsrc(tw, Geo<LOGN>::N * 16u), t, p, lds);
}

template <int HL>
__device__ __forceinline__ u64 canon_after_inverse(u64 x, const PrimeDev& p)
{
    if constexpr (!Lazy<HL>::EXACT) {
        const u64 twoq = 2 * p.q;
        x = x >= twoq ? x - twoq : x;
    }
    return canon_2q(x, p.q);
}

}  // namespace mi355ntt
