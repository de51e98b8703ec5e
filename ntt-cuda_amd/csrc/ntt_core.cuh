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

#include <utility>

#include "modarith.cuh"
#include "tune.hpp"      // the tuning constants of these kernels: ONE struct (measurement builds substitute their own, tools/build_kbench.sh)

namespace mi355ntt {

// Compile-time loop: the body receives std::integral_constant<int, I>, so every register-array index below
// is a constant expression (a runtime-indexed u64[32] would be placed in scratch memory).
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct TwPair {       // {w, floor(w * 2^64 / q)}
    u64 w, wp;
};

// Per-prime constants, read with scalar loads (replaces __constant__ q_cons/mu_cons/q_bit_cons).
struct PrimeDev {
    u64 q, nq;        // modulus and 2^64 - q
    u64 si, si_p;     // n = 2^16 contexts only: psi^-bitrev(1) / 2 of the full-size table (the last GS stage, which couples the halves
                      // and halves once more) and its Shoup companion (k_inverse15, SPLIT); else 0
    u64 sf, sf_p;     // n = 2^16 contexts only (two half-size transforms per polynomial, capi.cpp): the twiddle of the stage that
                      // couples the halves, psi^bitrev(1) of the full-size table, and its Shoup companion (k_forward15, SPLIT)
    u64 mu;           // Barrett mu = floor(2^(2k)/q), reference convention (pointwise products)
    u32 red_c;        // floor(2^(31+k) / q), 32 bits
    u32 red_sh1;      // k - 1 - g
    u32 red_sh2;      // g = min(16, k-1)
    u32 k;            // bit length
    u32 delta;        // 2^k - q when that is small ("near-2^k" prime, the shape of every SEAL-style modulus), else 0
    u32 near_sh;      // k - 32
    u32 near_mask;    // 2^(k-32) - 1
    u32 lit;          // contexts of kernel class HL_LIT only: 1 = this prime is Barrett-inexact -- its polynomials take the literal
                      // butterflies; 0 = the reference's words are the exact transform's, its polynomials take the lazy ones (else 0)
    // n^-1 folded into the LAST inverse round (gs_round, SCALE): twn[0] = {n^-1, companion}, twn[i] = psi^-bitrev(i) * n^-1
    // for i = 1..31 -- the twiddles of the top five GS stages (reference table entries [1, 32)) times n^-1.
    TwPair twn[32];
};

// Checked raw calls (kernels.hpp, kGuardBit): the record in front of the PrimeDev array holds {current epoch, epoch of the
// last table mismatch}.  Returns true when this launch must not touch the data; clears the flag bit of prime_base.
__device__ __forceinline__ bool guard_says_skip(const PrimeDev* primes, unsigned& prime_base)
{
    if (!(prime_base & 0x80000000u)) return false;
    prime_base &= 0x7fffffffu;
    const unsigned* g = reinterpret_cast<const unsigned*>(primes - 1);
    return g[0] == g[1];
}

// Device twiddle layout.  The reference table keeps stage `len` in entries [len, 2 len) indexed by the
// butterfly group p = i >> (bit + 1).  A thread of a round whose register field sits at bit B needs, for the
// stage on register bit j, the groups p = (thi << (4-j)) + u, u = 0 .. 2^(4-j)-1, thi = t >> B: for the
// last round (B = 0) that is 2^(4-j) consecutive entries per lane, i.e. lanes 256 bytes apart.  The device
// copy therefore stores each stage block transposed, entry len + u * nthi + thi (nthi = threads >> B), so
// that consecutive lanes read consecutive 16-byte entries and the per-register part is a compile-time offset.
__host__ __device__ constexpr unsigned tw_dev_index(int logn, int B, int j, unsigned len, unsigned thi, unsigned u)
{
    const unsigned nthi = (1u << (logn - 5)) >> B;
    (void)j;
    return len + u * nthi + thi;
}

// Buffer-descriptor loads/stores: one 32-bit lane offset VGPR serves every access of a thread, the
// per-register displacement rides in the scalar offset / immediate (no 64-bit address arithmetic in VGPRs).
typedef u32 v2u32 __attribute__((ext_vector_type(2)));
typedef u32 v4u32 __attribute__((ext_vector_type(4)));
using BufRsrc = __amdgpu_buffer_rsrc_t;

// `base` must be wave-uniform (it always derives from kernel arguments and the polynomial index).  The
// readfirstlane makes that provable to the compiler; otherwise every buffer access is wrapped in a
// serialising "waterfall" loop.
__device__ __forceinline__ BufRsrc make_rsrc(const void* base, u32 bytes)
{
    const u64 b = reinterpret_cast<u64>(base);
    const u32 lo = __builtin_amdgcn_readfirstlane(lo32(b)), hi = __builtin_amdgcn_readfirstlane(hi32(b));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((u64)hi << 32) | lo), 0, bytes, 0x00020000);
}
// (cache policy of the polynomial stream: Tune::kStreamAuxLd / kStreamAuxSt, tune.hpp)
__device__ __forceinline__ u64 buf_load_u64(BufRsrc r, u32 voff, u32 soff)
{
    const v2u32 x = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, Tune::kStreamAuxLd);
    return (u64)x.x | ((u64)x.y << 32);
}
__device__ __forceinline__ void buf_store_u64(BufRsrc r, u32 voff, u32 soff, u64 v)
{
    v2u32 x;
    x.x = lo32(v);
    x.y = hi32(v);
    __builtin_amdgcn_raw_buffer_store_b64(x, r, voff, soff, Tune::kStreamAuxSt);
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
// Quotient estimate h from three of the four partial products (error <= 2), remainder as y*w + h*(2^64-q): mul_shoup4m below.
//
// The four cross terms y0*w1 + y1*w0 + h0*n1 + h1*n0 of that remainder (only their low 32 bits matter) are accumulated
// by a chain of v_mad_u64_u32 instead of four v_mul_lo_u32 + two v_add3_u32.  Measured on gfx950 with every wave busy
// until a common deadline (tools/ubench_issue.hip, profiles/r02_ubench_issue_costs.txt): v_mad_u64_u32 issues in 4.1
// cycles per wave-instruction, the same as v_mul_lo_u32 / v_mul_hi_u32 / v_add3_u32 / v_lshl_add_u64 (4.0 ... 4.3; only
// plain 32-bit add / sub / and / xor / lshr / mov are full rate, 2.2), so the chain costs 4 x 4.1 + one full-rate add
// = 18.6 cycles against 24.3.  Inline asm, because the compiler -- knowing that only 32 bits of the chain are used --
// rewrites it into the multiply/add3 form.  `base` rides in the 64-bit addend of the lo*lo multiply-adds (0, or the
// butterfly's other input: the sum U + T then costs nothing).  TWS: the twiddle is wave-uniform (SGPR operands; a VOP3
// instruction of gfx9 may read one scalar register, which every mad below respects).
template <bool BS>
__device__ __forceinline__ u64 mad32_chain(u32 a, u32 b, u64 c)        // a: VGPR, b: VGPR or (BS) SGPR, c: VGPR pair
{
    u64 d, carry;
    if constexpr (BS) asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "s"(b), "v"(c));
    else asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
    return d;
}
template <bool BS>
__device__ __forceinline__ u64 mad32_chain0(u32 a, u32 b)
{
    u64 d, carry;
    if constexpr (BS) asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "s"(b));
    else asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "v"(b));
    return d;
}
// y*w + h*(2^64 - q) + base  (mod 2^64) for a given quotient estimate h
template <bool TWS>
__device__ __forceinline__ u64 shoup_rem_chain(u64 y, u64 w, u64 h, u64 nq, u64 base)
{
    const u32 y0 = lo32(y), y1 = hi32(y), w0 = lo32(w), w1 = hi32(w), n0 = lo32(nq), n1 = hi32(nq);
    const u32 h0 = lo32(h), h1 = hi32(h);
    const u64 acc = mad32(h0, n0, mad32(y0, w0, base));
    const u64 c = mad32_chain<true>(h1, n0, mad32_chain<true>(h0, n1, mad32_chain<TWS>(y1, w0, mad32_chain0<TWS>(y0, w1))));
    u32 xh = hi32(acc);
    asm("v_add_u32 %0, %0, %1" : "+v"(xh) : "v"(lo32(c)));     // (as C++ the compiler re-associates it into a 64-bit add of {0, c})
    return ((u64)xh << 32) | lo32(acc);
}
// y*w + h*(2^64 - q) + base  (mod 2^64); without base: congruent to y*w and in [0, 4q).  nq is always scalar (PrimeDev).
template <bool TWS>
__device__ __forceinline__ u64 mul_shoup4m_acc(u64 y, u64 w, u64 wp, u64 nq, u64 base)
{
    const u32 y0 = lo32(y), y1 = hi32(y), p0 = lo32(wp), p1 = hi32(wp);
    const u64 h = mad32(y1, p1, (u64)__umulhi(y0, p1)) + (u64)__umulhi(y1, p0);
    return shoup_rem_chain<TWS>(y, w, h, nq, base);
}
template <bool TWS>
__device__ __forceinline__ u64 mul_shoup4m(u64 y, u64 w, u64 wp, u64 nq) { return mul_shoup4m_acc<TWS>(y, w, wp, nq, 0); }

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
    else return mul_shoup4m<false>(y, w, wp, nq);
}

// x in [0, B*q) (B*q < 2^64, B <= 66 or B*q < 2^(k+5)) -> congruent value in [0, 2q).
// e = floor(x/q) or one less, from the top bits of x and a 32-bit reciprocal.
__device__ __forceinline__ u64 reduce_2q(u64 x, const PrimeDev& p)
{
    u32 t = (u32)(x >> p.red_sh1);
    u32 e = __umulhi(t, p.red_c) >> p.red_sh2;
    // e < 2^32: low 64 bits of e*nq added = x - e*q
    return x + (u64)e * p.nq;
}

// Same contract for primes q = 2^k - delta with small delta (k > 32, 2^(64-k) * delta + 2 delta < 2^k, checked on
// the host): with e = x >> k,  x - e*2^k = x mod 2^k  and  e*2^k = e*q + e*delta, so  x = (x mod 2^k) + e*delta (mod q).
// Three instructions (shift, and, multiply-add) instead of seven.
__device__ __forceinline__ u64 reduce_2q_near(u64 x, const PrimeDev& p)
{
    const u32 e = hi32(x) >> p.near_sh;
    const u64 xm = ((u64)(hi32(x) & p.near_mask) << 32) | lo32(x);
    return mad32(e, p.delta, xm);
}
template <bool NEAR>
__device__ __forceinline__ u64 reduce_2q_sel(u64 x, const PrimeDev& p)
{
    if constexpr (NEAR) return reduce_2q_near(x, p);
    else return reduce_2q(x, p);
}

// x * b mod q for DATA operands (no Shoup companion: the second operand of the fused products) and q = 2^k - delta:
// x in [0, 2q), b < 2^k  ->  congruent value in [0, 2q).  The 128-bit product is folded twice with 2^k = delta (mod q):
//   P = Phi 2^k + Plo,  F = Phi delta + Plo < 2^(k+26),  F = T 2^k + R,  result = T delta + R
// 7 multiply-adds, 3 funnel shifts, 2 masks -- Algorithm 7 on canonical operands (barrett_mul, poly_arithmetic.cuh:36-66) costs
// about twice that plus the canonicalisation of x.  Needs 2 delta^2 + 3 delta < 2^k, part of the near-2^k test on the host
// (fast_tables_create); every step is exact, so the transform's canonical output is the reference's.
__device__ __forceinline__ u64 mul_fold_near(u64 x, u64 b, const PrimeDev& p)
{
    u64 lo, hi;
    mul_wide(x, b, lo, hi);
    const u32 w0 = lo32(lo), w1 = hi32(lo), w2 = lo32(hi), w3 = hi32(hi);
    const u32 phi0 = __builtin_amdgcn_alignbit(w2, w1, p.near_sh), phi1 = __builtin_amdgcn_alignbit(w3, w2, p.near_sh);   // P >> k
    const u64 plo = ((u64)(w1 & p.near_mask) << 32) | w0;                                                                 // P mod 2^k
    const u64 f0 = mad32(phi0, p.delta, plo);                     // < 2^56 + 2^k: no carry out of 64 bits
    const u64 f1 = mad32(phi1, p.delta, (u64)hi32(f0));           // bits 32.. of F
    const u32 t = __builtin_amdgcn_alignbit(hi32(f1), lo32(f1), p.near_sh);                                               // F >> k  (< 2^26)
    const u64 r = ((u64)(lo32(f1) & p.near_mask) << 32) | lo32(f0);
    return mad32(t, p.delta, r);
}

// [0, 2q) -> [0, q)
__device__ __forceinline__ u64 canon_2q(u64 x, u64 q)
{
    return x >= q ? x - q : x;
}

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
    static constexpr bool TWO_PHASE = (LOGN >= Tune::kTwoPhaseMinLogN);   // LDS image holds half a polynomial at a time
    static constexpr int PB = LOGN - 1;               // index bit that selects the phase
    static constexpr int ROWS = (TWO_PHASE ? N / 2 : N) / 32;
    static constexpr int LDS_WORDS = (LOGN == 15) ? 16 * 1152 : ROWS * 34;   // image (32 + 2 pad words per row); n = 2^15: 16 wave slices of 9216 B
};

// element index held by thread t in register r for a layout whose register field sits at bit B
template <int B>
__device__ __forceinline__ unsigned elem_index(unsigned t, unsigned r)
{
    return ((t >> B) << (B + 5)) | (r << B) | (t & ((1u << B) - 1u));
}

// LDS image: element i lives at row (i' >> 5), column (i' & 31) of a [ROWS][34] array of u64 (two words of
// padding per row keep 8- and 16-byte accesses conflict-free for every layout).  When the image holds only
// half a polynomial (n = 2^15) the exchange runs in two phases selected by index bit PB, and i' is i with
// that bit removed.  Because the thread part and the register part of an index occupy disjoint bits, the
// slot splits into a per-thread base and a per-register COMPILE-TIME offset: base VGPR + immediate.
template <int PB>
__host__ __device__ constexpr unsigned drop_bit(unsigned i)
{
    if (PB < 0) return i;
    return ((i >> (PB + 1)) << PB) | (i & ((1u << PB) - 1u));
}
__host__ __device__ constexpr unsigned slot_of(unsigned ip) { return (ip >> 5) * 34u + (ip & 31u); }

__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Two adjacent 64-bit words from two INDEPENDENT register pairs (ds_write2_b64, offsets in units of 8 bytes).  A 16-byte
// store of (v[r], v[r+1]) as one ds_write_b128 needs the four VGPRs consecutive; the allocator cannot arrange that for all
// sixteen pairs of a round's outputs and assembles the tuples with copies -- in the general-prime kernels with spills.
// Inline asm: callers fence with wave_lds_fence() / a barrier before the words are read (the compiler's own waitcnt
// insertion does not see these stores).
template <int OFF8>
__device__ __forceinline__ void lds_write2_u64(__attribute__((address_space(3))) u64* addr, u64 a, u64 b)
{
    static_assert(OFF8 >= 0 && OFF8 + 1 < 256, "ds_write2_b64 offsets are 8-bit, in units of 8 bytes");
    const u32 la = (u32)reinterpret_cast<uintptr_t>(addr);
    asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(la), "v"(a), "v"(b), "n"(OFF8), "n"(OFF8 + 1) : "memory");
}
template <int OFF8>
__device__ __forceinline__ void lds_write2_u64(u64* addr, u64 a, u64 b)
{
    static_assert(OFF8 >= 0 && OFF8 + 1 < 256, "ds_write2_b64 offsets are 8-bit, in units of 8 bytes");
    const u32 la = (u32)reinterpret_cast<uintptr_t>(addr);      // LDS addresses are 32-bit (the low half of the flat address)
    asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(la), "v"(a), "v"(b), "n"(OFF8), "n"(OFF8 + 1) : "memory");
}

// Transposition through LDS: registers hold layout BO on entry, layout BN on exit.
// Two-phase case: the phase bit is the TOP REGISTER BIT OF THE READER layout, so every thread reads half of its
// registers in each phase (no divergent register definitions); writers take part in the phase their element
// belongs to (by register when the bit is in their register field, otherwise by thread).
template <int LOGN, int BO, int BN>
__device__ __forceinline__ void exchange(u64 (&v)[32], u64* lds, unsigned t)
{
    __builtin_amdgcn_sched_barrier(0);   // keep the next round's twiddle loads (and anything else) out of the exchange
    using G = Geo<LOGN>;
    constexpr int PH = G::TWO_PHASE ? 2 : 1;
    constexpr int PB = G::TWO_PHASE ? BN + 4 : -1;
    constexpr bool W_REG_SPLIT = G::TWO_PHASE && (PB >= BO && PB < BO + 5);
    const unsigned w_phase = G::TWO_PHASE ? ((elem_index<BO>(t, 0) >> (PB < 0 ? 0 : PB)) & 1u) : 0u;
    // DS instructions carry a 16-bit byte offset: address the (up to 136 KiB) image through up to three bases
    // 64 KiB apart.  The asm fences keep the bases local to this exchange.
    constexpr unsigned SEG = 8192;                      // words per 64 KiB
    unsigned ws0 = slot_of(drop_bit<PB>(elem_index<BO>(t, 0))), rs0 = slot_of(drop_bit<PB>(elem_index<BN>(t, 0)));
    asm volatile("" : "+v"(ws0), "+v"(rs0));
    unsigned ws1 = ws0 + SEG, ws2 = ws0 + 2 * SEG, rs1 = rs0 + SEG, rs2 = rs0 + 2 * SEG;
    asm volatile("" : "+v"(ws1), "+v"(ws2), "+v"(rs1), "+v"(rs2));
    // (bases as LDS-address-space pointers: formed as generic pointers they drag the aperture's high word along in a VGPR pair --
    // the last 12 bytes of scratch of k_inverse<13|14, 4, false>)
    typedef __attribute__((address_space(3))) u64 LdsWord;
    typedef u64 v2u64 __attribute__((ext_vector_type(2)));       // (a builtin vector: the HIP vector classes have no LDS-qualified members)
    typedef __attribute__((address_space(3))) v2u64 LdsPair;
    LdsWord* const lds3 = (LdsWord*)lds;
    LdsWord* const wb[3] = {lds3 + ws0, lds3 + ws1, lds3 + ws2};
    const LdsWord* const rb[3] = {lds3 + rs0, lds3 + rs1, lds3 + rs2};
    u64 nv[32];
    static_for<PH>([&](auto phc) {
        constexpr unsigned ph = decltype(phc)::value;
        // ---- write ----
        if (W_REG_SPLIT || !G::TWO_PHASE || w_phase == ph) {
            static_for<(BO == 0 ? 16 : 32)>([&](auto rc) {
                constexpr int r = decltype(rc)::value * (BO == 0 ? 2 : 1);
                if constexpr (!W_REG_SPLIT || ((((unsigned)r << BO) >> (PB < 0 ? 0 : PB)) & 1u) == ph) {
                    constexpr unsigned off = slot_of(drop_bit<PB>((unsigned)r << BO));
                    if constexpr (BO == 0 && (off % SEG) + 1 < 256)
                        lds_write2_u64<off % SEG>(wb[off / SEG], v[r], v[r + 1]);       // (pair store without a register tuple)
                    else if constexpr (BO == 0)
                        *reinterpret_cast<LdsPair*>(wb[off / SEG] + off % SEG) = v2u64{v[r], v[r + 1]};
                    else
                        wb[off / SEG][off % SEG] = v[r];
                }
            });
        }
        __syncthreads();
        // ---- read: registers whose top bit equals the phase (all of them in the single-phase case) ----
        static_for<(BN == 0 ? 16 : 32)>([&](auto rc) {
            constexpr int r = decltype(rc)::value * (BN == 0 ? 2 : 1);
            if constexpr (!G::TWO_PHASE || (unsigned)(r >> 4) == ph) {
                constexpr unsigned off = slot_of(drop_bit<PB>((unsigned)r << BN));
                if constexpr (BN == 0) {
                    const v2u64 pr = *reinterpret_cast<const LdsPair*>(rb[off / SEG] + off % SEG);
                    nv[r] = pr.x;
                    nv[r + 1] = pr.y;
                } else {
                    nv[r] = rb[off / SEG][off % SEG];
                }
            }
        });
        __syncthreads();
    });
#pragma unroll
    for (int r = 0; r < 32; r++) v[r] = nv[r];
    __builtin_amdgcn_sched_barrier(0);
}

// ------------------------------------------------------------------------------------------------
// wave-local transposes (n = 2^15 path): no workgroup barrier, every wave uses a private 8704-byte slice
// of the LDS image, so between two workgroup-wide exchanges the 16 waves run freely and their memory,
// LDS and VALU phases overlap.
// ------------------------------------------------------------------------------------------------
constexpr unsigned WAVE_SLICE_WORDS = 1152;     // 2 half-waves x 32 rows x 18 words = 9216 B per wave (16 slices = 144 KiB)

// layout 5 -> layout 0.  Per half-wave (32 lanes) this is a 32x32 transpose: lane (h, c) holds M[r][c] in register r
// and ends with row (its own c): M[c][0..31].  Two steps of 16 COLUMNS each: the lanes owning those columns store
// all their registers (an exec-masked store), then every lane loads 16 words of its row.
__device__ __forceinline__ void wave_transpose_5_to_0(u64 (&v)[32], u64* slice, unsigned lane)
{
    __builtin_amdgcn_sched_barrier(0);
    const unsigned h = lane >> 5, c = lane & 31;
    u64* wcol = slice + h * 576 + (c & 15);                // [h][row 0..31][16 cols], row stride 18 words
    const u64* rrow = slice + h * 576 + c * 18;
    u64 nv[32];
    static_for<2>([&](auto sc) {
        constexpr int step = decltype(sc)::value;
        if ((c >> 4) == (unsigned)step) {
            static_for<32>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                wcol[r * 18] = v[r];
            });
        }
        wave_lds_fence();
        static_for<8>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const ulonglong2 pr = *reinterpret_cast<const ulonglong2*>(rrow + 2 * m);
            nv[16 * step + 2 * m] = pr.x;
            nv[16 * step + 2 * m + 1] = pr.y;
        });
        wave_lds_fence();
    });
    static_for<32>([&](auto rc) { v[decltype(rc)::value] = nv[decltype(rc)::value]; });
    __builtin_amdgcn_sched_barrier(0);
}

// layout 0 -> layout 5 (the inverse direction)
__device__ __forceinline__ void wave_transpose_0_to_5(u64 (&v)[32], u64* slice, unsigned lane)
{
    __builtin_amdgcn_sched_barrier(0);
    const unsigned h = lane >> 5, c = lane & 31;
    u64* wrow = slice + h * 544 + (c & 15) * 34;
    const u64* rcol = slice + h * 544 + c;
    u64 nv[32];
    static_for<2>([&](auto sc) {
        constexpr int step = decltype(sc)::value;
        if ((c >> 4) == (unsigned)step) {
            static_for<16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                lds_write2_u64<2 * m>(wrow, v[2 * m], v[2 * m + 1]);
            });
        }
        wave_lds_fence();
        static_for<16>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            nv[16 * step + r] = rcol[r * 34];
        });
        wave_lds_fence();
    });
    static_for<32>([&](auto rc) { v[decltype(rc)::value] = nv[decltype(rc)::value]; });
    __builtin_amdgcn_sched_barrier(0);
}

// Layout 0 <-> global memory with 16-byte accesses.  A wave owns 2048 consecutive coefficients (16 KiB): lane L
// holds row L (32 words).  Going through the slice in two column halves of 128 B per row lets every global
// instruction move 8 rows x 128 contiguous bytes.  Rows are 128 B apart and the eight 16-byte pieces of row R sit
// at slot piece ^ ((R >> 1) & 7): with that swizzle both the row accesses (lane = row: ds_*_b128 serviced in
// 16-lane groups {0-3,12-15,20-27}, ...) and the transposed accesses (8 consecutive lanes per row) touch every
// bank once.  Lane-derived address parts are fenced per call so they are recomputed (2 instructions each)
// instead of being hoisted out of the polynomial loop and spilled.
__device__ __forceinline__ unsigned row_swz(unsigned row) { return (row >> 1) & 7u; }
// The store direction (rows written with ds_write_b128, read back transposed with ds_read_b128) needs another swizzle
// than the load direction: ds_write_b128 is serviced in 8 groups of 8 CONTIGUOUS lanes on 32 banks (MI355X_MICROARCH.md,
// LDS), so lanes 2j and 2j + 1 of a group must not share a slot -- with (row >> 1) & 7 they did: 2-way, exactly the 8
// extra cycles per store that SQ_LDS_BANK_CONFLICT showed on k_forward15 (128 per wave and polynomial).  row & 7 keeps
// both the writes and the transposed reads (16-lane groups {0-3,12-15,20-27}, ... on 64 banks) conflict-free.
__device__ __forceinline__ unsigned row_swz_store(unsigned row) { return row & 7u; }

// Lane index from the execution mask (v_mbcnt_lo/hi: two instructions, no register kept live).  The callers below take it
// fresh each time: the thread index would otherwise have to survive the whole polynomial loop in a VGPR the kernels do not
// have (it was one of their spills, reloaded from scratch in front of the row store).
__device__ __forceinline__ unsigned fresh_lane_id()
{
    unsigned l;                          // (volatile asm: emitted at every use site -- the builtins are computed once, hoisted
                                         // out of the loop and spilled by the general-prime kernels)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

__device__ __forceinline__ void wave_store_rows(const u64 (&v)[32], u64* slice, BufRsrc dst, unsigned wave_byte_off, unsigned)
{
    __builtin_amdgcn_sched_barrier(0);
    const unsigned lane = fresh_lane_id();
    char* base = reinterpret_cast<char*>(slice);
    const unsigned sw = lane & 7, rr = lane >> 3;
    static_for<2>([&](auto cc) {
        constexpr int ch = decltype(cc)::value;
        static_for<8>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            lds_write2_u64<0>(reinterpret_cast<u64*>(base + lane * 128 + ((m ^ row_swz_store(lane)) << 4)), v[16 * ch + 2 * m], v[16 * ch + 2 * m + 1]);
        });
        wave_lds_fence();
        static_for<8>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            // row 8k + rr, piece sw
            const ulonglong2 pr = *reinterpret_cast<const ulonglong2*>(base + (8 * k + rr) * 128 + ((sw ^ row_swz_store(8 * k + rr)) << 4));
            v4u32 x;
            x.x = lo32(pr.x); x.y = hi32(pr.x); x.z = lo32(pr.y); x.w = hi32(pr.y);
            __builtin_amdgcn_raw_buffer_store_b128(x, dst, wave_byte_off + rr * 256u + sw * 16u, k * 2048u + ch * 128u, Tune::kStreamAuxSt);
        });
        wave_lds_fence();
    });
}

template <int CH, int AUX>
__device__ __forceinline__ void issue_row_loads(v4u32 (&x)[8], BufRsrc src, unsigned voff)
{
    static_for<8>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        x[k] = __builtin_amdgcn_raw_buffer_load_b128(src, voff, k * 2048u + CH * 128u, AUX);
    });
}
// one column half (CH = 0/1): 16 words of this lane's row into out[0..15].  AUX_ALT: cache-policy bits of the global loads when the
// (wave-uniform) run-time flag `alt` is set -- only the eight load instructions are issued twice in the code, the rest is shared
template <int CH, int AUX_ALT = Tune::kRowsAuxLd>
__device__ __forceinline__ void wave_load_rows_half(u64 (&out)[16], u64* slice, BufRsrc src, unsigned wave_byte_off, unsigned, bool alt = false)
{
    const unsigned lane = fresh_lane_id();
    char* base = reinterpret_cast<char*>(slice);
    const unsigned sw = lane & 7, rr = lane >> 3;
    v4u32 x[8];
    if (AUX_ALT != Tune::kRowsAuxLd && alt) issue_row_loads<CH, AUX_ALT>(x, src, wave_byte_off + rr * 256u + sw * 16u);
    else issue_row_loads<CH, Tune::kRowsAuxLd>(x, src, wave_byte_off + rr * 256u + sw * 16u);
    if constexpr (AUX_ALT != Tune::kRowsAuxLd) __builtin_amdgcn_sched_barrier(0);     // (a convergent join: keeps the compiler from duplicating the staging code below into both arms)
    static_for<8>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        *reinterpret_cast<v4u32*>(base + (8 * k + rr) * 128 + ((sw ^ row_swz(8 * k + rr)) << 4)) = x[k];
    });
    wave_lds_fence();
    static_for<8>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const ulonglong2 pr = *reinterpret_cast<const ulonglong2*>(base + lane * 128 + ((m ^ row_swz(lane)) << 4));
        out[2 * m] = pr.x;
        out[2 * m + 1] = pr.y;
    });
    wave_lds_fence();
}

// Pre-landing (round 5): column half CH of this wave's rows requested by LDS-direct loads -- global memory -> LDS without a VGPR in
// between, so the request can go out while the registers still hold the previous polynomial (k_inverse15: right behind the
// workgroup-wide exchange, when the slices are dead and the last round reads its twiddles through the scalar cache: nothing else of
// this wave sits in the vector-memory queue until the result stores).  Lane L of instruction k lands at slice + 1024 k + 16 L, i.e.
// row 8 k + (L >> 3), slot L & 7 of the staging layout of wave_load_rows_half; the swizzle moves to the global side: the lane fetches
// piece slot ^ row_swz(row) of its row (still eight whole 128-byte lines per instruction).
typedef __attribute__((address_space(3))) void* LdsVoidPtr;
template <int CH, int AUX>
__device__ __forceinline__ void issue_preland_loads(u64* slice, BufRsrc src, unsigned voff_even, unsigned voff_odd)
{
    static_for<8>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (LdsVoidPtr)(slice + k * 128), 16, (k & 1) ? voff_odd : voff_even, k * 2048u + CH * 128u, 0, AUX);
    });
}
template <int CH, int AUX_ALT = Tune::kRowsAuxLd>
__device__ __forceinline__ void wave_preland_rows_half(u64* slice, BufRsrc src, bool alt = false)
{
    const unsigned lane = fresh_lane_id();
    const unsigned sw = lane & 7, rr = lane >> 3;
    const unsigned voff_even = rr * 256u + ((sw ^ (rr >> 1)) << 4);       // row_swz(8 k + rr) = 4 (k & 1) | (rr >> 1)
    const unsigned voff_odd = voff_even ^ 64u;
    if (AUX_ALT != Tune::kRowsAuxLd && alt) issue_preland_loads<CH, AUX_ALT>(slice, src, voff_even, voff_odd);
    else issue_preland_loads<CH, Tune::kRowsAuxLd>(slice, src, voff_even, voff_odd);
    if constexpr (AUX_ALT != Tune::kRowsAuxLd) __builtin_amdgcn_sched_barrier(0);
}
// (the caller has waited for the eight LDS-direct loads with a counted s_waitcnt vmcnt)
__device__ __forceinline__ void wave_read_prelanded_half(u64 (&out)[16], const u64* slice)
{
    const unsigned lane = fresh_lane_id();
    const char* base = reinterpret_cast<const char*>(slice);
    static_for<8>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const ulonglong2 pr = *reinterpret_cast<const ulonglong2*>(base + lane * 128 + ((m ^ row_swz(lane)) << 4));
        out[2 * m] = pr.x;
        out[2 * m + 1] = pr.y;
    });
    wave_lds_fence();
}

// One column half through LDS-direct loads end to end: request, wait, read-out -- no VGPRs and no ds_write pass between memory and the
// slice (wave_load_rows_half moves the same bytes memory -> VGPR -> ds_write_b128 -> slice), and the run-time choice of the cache policy
// (alt) is a branch around eight instructions that define no register.  The slice must be idle (every earlier LDS access of this wave
// retired); waits for everything this wave has in the vector-memory queue.
template <int CH, int AUX_ALT = Tune::kRowsAuxLd>
__device__ __forceinline__ void wave_load_rows_half_direct(u64 (&out)[16], u64* slice, BufRsrc src, bool alt = false)
{
    __builtin_amdgcn_sched_barrier(0);
    wave_preland_rows_half<CH, AUX_ALT>(slice, src, alt);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (hipcc does not order an LDS read behind the LDS-direct load that fills it)
    wave_read_prelanded_half(out, slice);
}

template <int AUX_ALT = Tune::kRowsAuxLd>
__device__ __forceinline__ void wave_load_rows(u64 (&v)[32], u64* slice, BufRsrc src, unsigned wave_byte_off, unsigned lane, bool alt = false)
{
    u64 h[16];
    wave_load_rows_half<0, AUX_ALT>(h, slice, src, wave_byte_off, lane, alt);
    static_for<16>([&](auto rc) { v[decltype(rc)::value] = h[decltype(rc)::value]; });
    wave_load_rows_half<1, AUX_ALT>(h, slice, src, wave_byte_off, lane, alt);
    static_for<16>([&](auto rc) { v[16 + decltype(rc)::value] = h[decltype(rc)::value]; });
}

// ------------------------------------------------------------------------------------------------
// in-register rounds
// ------------------------------------------------------------------------------------------------
// The whole transform is straight-line code; without fences the scheduler hoists dozens of twiddle loads
// (4 VGPRs each) and spills.  A scheduling fence every SCHED_GROUP butterflies bounds the live set; the
// other three waves of the SIMD cover the load latency.
constexpr int SCHED_GROUP = Tune::kSchedGroup;

// k-th register index (k = 0..15) whose bit j is clear
__host__ __device__ constexpr int low_reg(int j, int k) { return ((k >> j) << (j + 1)) | (k & ((1 << j) - 1)); }

// Twiddle ring of a round: GROUP butterflies per scheduling group, DEPTH buffers (the loads run DEPTH - 1 groups ahead).
// VGPRs = 4 * GROUP * DEPTH.  The round at bit 0 reads lane-distinct entries that come from L2: smaller groups and a
// deeper ring buy prefetch distance for the same registers.
// TIGHT (exact-quotient general-prime inverse at n = 2^15: a full 64x64 high product and a 7-instruction partial reduction in
// every stage need the registers): single butterflies, four buffers -- a prefetch distance of three butterflies out of 16
// VGPRs instead of 32.
template <int LOGN, int B, bool TIGHT = false>
struct Ring {
    static constexpr int GROUP = TIGHT ? 1 : (B == 0 && LOGN == 15) ? Tune::kRingGroupB0 : SCHED_GROUP;
    static constexpr int DEPTH = TIGHT ? 4 : (B == 0 && LOGN == 15) ? Tune::kRingDepthB0 : 2;
};

// Twiddles of butterfly group G of a round (GROUP butterflies per group, 16 / GROUP groups per stage).
// FWD: stages run j = JA, JA-1, ...; INV: j = JA, JA+1, ...
// SCALE (inverse, last round only): the butterflies whose register bits JA .. j-1 are all zero have not met a twiddle in this
// round yet -- they take theirs from twn (PrimeDev::twn of the modulus), i.e. times n^-1 (see gs_round).
__host__ __device__ constexpr bool zero_history(int r0, int j, int jlo) { return ((r0 & ((1 << j) - 1)) >> jlo) == 0; }
template <int LOGN, int B, int JA, bool FWD, int GROUP, int G, bool SCALE = false>
__device__ __forceinline__ void load_tw_group(TwPair (&W)[GROUP], const TwPair* __restrict__ tw, BufRsrc twr, unsigned thi, const TwPair* __restrict__ twn)
{
    constexpr int GPS = 16 / GROUP;                         // groups per stage
    constexpr int j = FWD ? JA - G / GPS : JA + G / GPS;
    constexpr int beta = B + j;                             // index bit of the stage
    constexpr unsigned len = 1u << (LOGN - 1 - beta);
    static_for<GROUP>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int r0 = low_reg(j, (G % GPS) * GROUP + k);
        constexpr unsigned u = (unsigned)r0 >> (j + 1);
        if constexpr (B == Geo<LOGN>::B0) {                 // group index independent of the thread: scalar load
            if constexpr (SCALE && !FWD && zero_history(r0, j, JA)) W[k] = twn[len + u];
            else W[k] = tw[len + u];
        } else
            W[k] = buf_load_tw(twr, thi * 16u, tw_dev_index(LOGN, B, j, len, 0, u) * 16u);
    });
}

// ------------------------------------------------------------------------------------------------
// wave-priority hooks of the n = 2^15 kernels (s_setprio; priority outranks age in the SIMD's issue arbitration)
// ------------------------------------------------------------------------------------------------
// A round may lower the wave's priority from scheduling group PSPLIT on: the phases of a polynomial get descending
// priorities so that the waves that are behind win the issue slots and the 16 waves reach the workgroup exchange closer
// together.  (Round 2 measured a time-sliced rotation of the priorities as well -- priority ((s_memtime >> k) + rank of the
// wave on its SIMD) & 3, re-evaluated every group: a wave that is starved never reaches the instruction that would raise
// its priority, the youngest wave of each SIMD fell further behind and the inverse lost 8 %:
// profiles/r02_priority_schemes_wg_timeline.txt, dyn8..dyn13.)
template <int PSPLIT, int PAFTER, int G>
__device__ __forceinline__ void prio_hook(unsigned)
{
    if constexpr (PSPLIT >= 0 && G == PSPLIT) __builtin_amdgcn_s_setprio(PAFTER);
}

// ------------------------------------------------------------------------------------------------
// kernel class HL_LIT: the reference's own arithmetic in the single-pass kernel shape (round 6)
// ------------------------------------------------------------------------------------------------
// The lazy classes above compute the exact transform; the reference reduces every product with singleBarrett's ONE conditional
// subtraction (ntt_60bit.cuh:44-61), and for a Barrett-inexact modulus (hostparams.cpp, barrett_single_subtraction_exact) that leaves
// q + r now and then, which the next butterfly's unsigned compares (:102-110, :166-178) turn into other words than the exact
// transform's.  A drop-in has to return THOSE words, so the polynomials of such moduli -- and everything a raw call cannot verify --
// used to run the stage-per-launch kernels of kernels_compat.hip: 2 passes over memory at n = 2^15, 0.10 of the HBM roofline.  Class
// HL_LIT runs the same register-resident rounds with the reference's butterflies written out literally: three wide products per
// butterfly (a psi, x1 mu, s q: `barrett_mul`, modarith.cuh -- the function the literal stage kernels call), the value carried from
// stage to stage exactly as the reference's global / shared memory carries it (canonical, or q + r, or whatever a wrapped
// subtraction made of it: every step is the same 64-bit operation on the same operands as ct_stage_kernel / gs_stage_kernel), a
// halving in every GS stage instead of one scaling by n^-1, no canonicalisation at the end.  The butterfly network, the index
// conventions and the device tables are those of the lazy classes (the .w half of a TwPair IS the reference's table entry); only the
// order of the butterflies inside a stage differs, which no word depends on.  One read and one write of HBM per transform.
constexpr int HL_LIT = 0;

__device__ __forceinline__ u64 buf_load_w(BufRsrc r, u32 voff, u32 soff)          // the .w half of a TwPair entry (8 bytes)
{
    const v2u32 x = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return (u64)x.x | ((u64)x.y << 32);
}

// twiddles (table entries, no companions) of butterfly group G of a literal round; indexing as load_tw_group
template <int LOGN, int B, int JA, bool FWD, int GROUP, int G>
__device__ __forceinline__ void load_w_group(u64 (&W)[GROUP], const TwPair* __restrict__ tw, BufRsrc twr, unsigned thi)
{
    constexpr int GPS = 16 / GROUP;
    constexpr int j = FWD ? JA - G / GPS : JA + G / GPS;
    constexpr int beta = B + j;
    constexpr unsigned len = 1u << (LOGN - 1 - beta);
    static_for<GROUP>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int r0 = low_reg(j, (G % GPS) * GROUP + k);
        constexpr unsigned u = (unsigned)r0 >> (j + 1);
        if constexpr (B == Geo<LOGN>::B0) W[k] = tw[len + u].w;                  // group index independent of the thread: scalar load
        else W[k] = buf_load_w(twr, thi * 16u, tw_dev_index(LOGN, B, j, len, 0, u) * 16u);
    });
}

// singleBarrett on the product y w (ntt_60bit.cuh:44-61; modarith.cuh barrett_mul is the plain form of the same function) for moduli of
// 34 ... 61 bits, where both 128-bit shifts are funnel shifts of 32-bit words: x1 = P >> (k - 2) from words 1..3 of the product,
// s = (x1 mu) >> (k + 2) from words 1..3 of that product (v_alignbit_b32 by k - 34 and k - 30), r = P - s q as P + s (2^64 - q) with
// only the low 64 bits formed, ONE conditional subtraction.  Every step is the reference's 64/128-bit operation on the same
// operands -- for ANY 64-bit y, not only residues -- so the words are the reference's whatever it was fed.  11 multiply-adds.
__device__ __forceinline__ u64 lit_barrett_mul(u64 y, u64 w, const PrimeDev& p)
{
    const u32 y0 = lo32(y), y1 = hi32(y), w0 = lo32(w), w1 = hi32(w);
    const u64 A = mad32(y0, w0, 0);                                   // P0 = lo32(A)
    const u64 T = mad32(y0, w1, (u64)hi32(A));
    const u64 U = mad32(y1, w0, (u64)lo32(T));                        // P1 = lo32(U)
    const u64 V = mad32(y1, w1, (u64)hi32(T)) + hi32(U);              // P2, P3
    const u32 sh = p.near_sh - 2u, sh2 = p.near_sh + 2u;              // k - 34, k - 30  (near_sh = k - 32)
    const u32 x0 = __builtin_amdgcn_alignbit(lo32(V), lo32(U), sh), x1 = __builtin_amdgcn_alignbit(hi32(V), lo32(V), sh);      // (P >> (k - 2)).low
    const u32 m0 = lo32(p.mu), m1 = hi32(p.mu);
    const u64 A2 = mad32(x0, m0, 0);
    const u64 T2 = mad32(x0, m1, (u64)hi32(A2));
    const u64 U2 = mad32(x1, m0, (u64)lo32(T2));                      // word 1 of x1 mu = lo32(U2)
    const u64 V2 = mad32(x1, m1, (u64)hi32(T2)) + hi32(U2);           // words 2, 3
    const u32 s0 = __builtin_amdgcn_alignbit(lo32(V2), lo32(U2), sh2), s1 = __builtin_amdgcn_alignbit(hi32(V2), lo32(V2), sh2);   // ((x1 mu) >> (k + 2)).low
    const u32 n0 = lo32(p.nq), n1 = hi32(p.nq);
    const u64 acc = mad32(s0, n0, A);                                 // low word: P0 + s0 n0; high word so far: hi32(A) + carries
    const u64 c = mad32(s1, n0, mad32(s0, n1, 0));                    // cross terms of s (2^64 - q): only their low 32 bits count
    const u32 r1 = hi32(acc) + (lo32(U) - hi32(A)) + lo32(c);         // P1 = hi32(A) + (lo32(U) - hi32(A))
    const u64 r = ((u64)r1 << 32) | lo32(acc);
    return r >= p.q ? r - p.q : r;
}

// CTBasedNTTInner's butterfly (ntt_60bit.cuh:199-222): V = singleBarrett(a[j + step] psi); a[j] = U + V - q (U + V >= q);
// a[j + step] = U + q (U < V) - V
__device__ __forceinline__ void lit_ct_bfly(u64& a, u64& b, u64 w, const PrimeDev& p)
{
    const u64 U = a;
    const u64 V = lit_barrett_mul(b, w, p);
    a = add_mod(U, V, p.q);
    b = sub_mod(U, V, p.q);
}
// GSBasedINTTInner's butterfly (:232-264): a[j] = half((U + V) mod q); a[j + step] = half(singleBarrett((U + q (U < V) - V) psiinv))
__device__ __forceinline__ void lit_gs_bfly(u64& a, u64& b, u64 w, const PrimeDev& p, u64 q2)
{
    const u64 U = a, V = b;
    a = half_mod(add_mod(U, V, p.q), q2);
    b = half_mod(lit_barrett_mul(sub_mod(U, V, p.q), w, p), q2);
}

template <int LOGN, int B, int JHI, int PSPLIT = -2, int PAFTER = 0>
__device__ __forceinline__ void ct_round_lit(u64 (&v)[32], const TwPair* __restrict__ tw, BufRsrc twr, unsigned t, const PrimeDev& p)
{
    constexpr bool VEC = (B != Geo<LOGN>::B0);
    constexpr int GROUP = Ring<LOGN, B>::GROUP, DEPTH = Ring<LOGN, B>::DEPTH, GPS = 16 / GROUP, NG = (JHI + 1) * GPS;
    const unsigned thi = t >> B;
    u64 W[DEPTH][GROUP];
    static_for<DEPTH - 1>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        if constexpr (d < NG) load_w_group<LOGN, B, JHI, true, GROUP, d>(W[d], tw, twr, thi);
    });
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int j = JHI - g / GPS;
        u64 (&Wc)[GROUP] = W[g % DEPTH];
        prio_hook<PSPLIT, PAFTER, g>(t);
        if constexpr (g + DEPTH - 1 < NG) load_w_group<LOGN, B, JHI, true, GROUP, g + DEPTH - 1>(W[(g + DEPTH - 1) % DEPTH], tw, twr, thi);
        if constexpr (VEC) __builtin_amdgcn_sched_barrier(0);
        static_for<GROUP>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int r0 = low_reg(j, (g % GPS) * GROUP + k);
            constexpr int r1 = r0 | (1 << j);
            lit_ct_bfly(v[r0], v[r1], Wc[k], p);
        });
        if constexpr (VEC) __builtin_amdgcn_sched_barrier(0);
    });
}

template <int LOGN, int B, int JLO, int PSPLIT = -2, int PAFTER = 0>
__device__ __forceinline__ void gs_round_lit(u64 (&v)[32], const TwPair* __restrict__ tw, BufRsrc twr, unsigned t, const PrimeDev& p)
{
    constexpr bool VEC = (B != Geo<LOGN>::B0);
    using RingT = Ring<LOGN, B, false>;
    constexpr int GROUP = RingT::GROUP, DEPTH = RingT::DEPTH, GPS = 16 / GROUP, NG = (5 - JLO) * GPS;
    const unsigned thi = t >> B;
    const u64 q2 = (p.q + 1) >> 1;
    u64 W[DEPTH][GROUP];
    static_for<DEPTH - 1>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        if constexpr (d < NG) load_w_group<LOGN, B, JLO, false, GROUP, d>(W[d], tw, twr, thi);
    });
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int j = JLO + g / GPS;
        u64 (&Wc)[GROUP] = W[g % DEPTH];
        prio_hook<PSPLIT, PAFTER, g>(t);
        if constexpr (g + DEPTH - 1 < NG) load_w_group<LOGN, B, JLO, false, GROUP, g + DEPTH - 1>(W[(g + DEPTH - 1) % DEPTH], tw, twr, thi);
        if constexpr (VEC) __builtin_amdgcn_sched_barrier(0);
        static_for<GROUP>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int r0 = low_reg(j, (g % GPS) * GROUP + k);
            constexpr int r1 = r0 | (1 << j);
            lit_gs_bfly(v[r0], v[r1], Wc[k], p, q2);
        });
        if constexpr (VEC) __builtin_amdgcn_sched_barrier(0);
    });
}

// Forward (CT) stages on register bits JHI..0 of a layout with register field at bit B.
// PSPLIT / PAFTER: see prio_hook (-2 = no hook at all: kernels other than the n = 2^15 persistent ones).
template <int LOGN, int HL, int B, int JHI, bool NEAR = false, int PSPLIT = -2, int PAFTER = 0>
__device__ __forceinline__ void ct_round(u64 (&v)[32], const TwPair* __restrict__ tw, BufRsrc twr, unsigned t, const PrimeDev& p)
{
    if constexpr (HL == HL_LIT) {
        ct_round_lit<LOGN, B, JHI, PSPLIT, PAFTER>(v, tw, twr, t, p);      // the reference's own butterflies
    } else {
        constexpr unsigned RMASK = fwd_reduce_mask<LOGN, HL>();
        constexpr bool EX = Lazy<HL>::EXACT;
        constexpr bool VEC = (B != Geo<LOGN>::B0);              // twiddles arrive in VGPRs: software-pipelined DEPTH - 1 groups ahead
        constexpr int GROUP = Ring<LOGN, B>::GROUP, DEPTH = Ring<LOGN, B>::DEPTH, GPS = 16 / GROUP, NG = (JHI + 1) * GPS;
        const u64 cq = (u64)Lazy<HL>::TQ * p.q;
        const unsigned thi = t >> B;
        TwPair W[DEPTH][GROUP];
        static_for<DEPTH - 1>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d < NG) load_tw_group<LOGN, B, JHI, true, GROUP, d>(W[d], tw, twr, thi, nullptr);
        });
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int j = JHI - g / GPS;
            constexpr int s = LOGN - 1 - (B + j);
            constexpr bool red = (RMASK >> s) & 1u;
            TwPair (&Wc)[GROUP] = W[g % DEPTH];
            prio_hook<PSPLIT, PAFTER, g>(t);
            if constexpr (g + DEPTH - 1 < NG) load_tw_group<LOGN, B, JHI, true, GROUP, g + DEPTH - 1>(W[(g + DEPTH - 1) % DEPTH], tw, twr, thi, nullptr);
            if constexpr (VEC) __builtin_amdgcn_sched_barrier(0);
            static_for<GROUP>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int r0 = low_reg(j, (g % GPS) * GROUP + k);
                constexpr int r1 = r0 | (1 << j);
                u64 U = v[r0];
                if constexpr (red) U = reduce_2q_sel<NEAR>(U, p);
                if constexpr (!EX) {
                    // (a, b) <- (U + T, U + cq - T): the sum comes out of the multiply-add accumulator, the difference is
                    // (2U + cq) - (U + T) (exact mod 2^64 even where 2U + cq wraps, because U + cq - T itself is below 2^64)
                    u64 D = (U << 1) + cq;
                    asm("" : "+v"(D));
                    const u64 A = mul_shoup4m_acc<!VEC>(v[r1], Wc[k].w, Wc[k].wp, p.nq, U);
                    v[r0] = A;
                    v[r1] = D - A;
                } else {
                    const u64 Tm = mul_shoup<EX>(v[r1], Wc[k].w, Wc[k].wp, p.nq);
                    v[r0] = U + Tm;
                    v[r1] = U + cq - Tm;
                }
            });
            if constexpr (VEC) __builtin_amdgcn_sched_barrier(0);
        });
    }
}

// Inverse (GS) stages on register bits JLO..4 of a layout with register field at bit B.
// IN2Q: the inputs are in [0, 2q) instead of canonical (the fused products hand over lazily reduced values); for the classes
// with 4q of headroom only the first stage's difference changes (x + 2q - y), every later bound is the same.
template <int LOGN, int HL, int B, int JLO, bool NEAR = false, int PSPLIT = -2, int PAFTER = 0, bool IN2Q = false>
__device__ __forceinline__ void gs_round(u64 (&v)[32], const TwPair* __restrict__ tw, BufRsrc twr, unsigned t, const PrimeDev& p,
                                         const TwPair* __restrict__ twn)      // twn: &primes[idx].twn[0] -- read where it is used (last round only)
{
    if constexpr (HL == HL_LIT) {
        gs_round_lit<LOGN, B, JLO, PSPLIT, PAFTER>(v, tw, twr, t, p);      // the reference's own butterflies
    } else {
        constexpr InvPolicy<LOGN, HL> POL{};
        static_assert(!IN2Q || (!Lazy<HL>::EXACT && B == 0 && JLO == 0), "lazy inputs: first round of a class with 4q of headroom only");
        constexpr bool EX = Lazy<HL>::EXACT;
        constexpr bool VEC = (B != Geo<LOGN>::B0);
        using RingT = Ring<LOGN, B, false>;
        constexpr int GROUP = RingT::GROUP, DEPTH = RingT::DEPTH, GPS = 16 / GROUP, NG = (5 - JLO) * GPS;
        // The scaling by n^-1 (the reference halves in every stage, ntt_60bit.cuh:132,166,178) is folded into the twiddles of the
        // LAST round: in stage j the butterflies whose register bits JLO .. j-1 are zero hold values that have only been summed in
        // this round so far; their difference output takes twiddle * n^-1 (twn), every later butterfly of that output uses the
        // plain table.  What has been summed in all stages -- the registers below 2^JLO -- is multiplied by n^-1 in the last stage:
        // 2^JLO products per thread instead of 16.
        constexpr bool SCALE = !VEC;
        const unsigned thi = t >> B;
        TwPair W[DEPTH][GROUP];
        static_for<DEPTH - 1>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d < NG) load_tw_group<LOGN, B, JLO, false, GROUP, d, SCALE>(W[d], tw, twr, thi, twn);
        });
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int j = JLO + g / GPS;
            constexpr int beta = B + j;             // index bit of this stage = GS stage number (0 = first)
            constexpr bool last = (beta == LOGN - 1);
            constexpr bool red = (POL.mask >> beta) & 1u;
            const u64 cq = (u64)((IN2Q && beta == 0) ? 2 : POL.cmul[beta]) * p.q;
            TwPair (&Wc)[GROUP] = W[g % DEPTH];
            prio_hook<PSPLIT, PAFTER, g>(t);
            if constexpr (g + DEPTH - 1 < NG) load_tw_group<LOGN, B, JLO, false, GROUP, g + DEPTH - 1, SCALE>(W[(g + DEPTH - 1) % DEPTH], tw, twr, thi, twn);
            if constexpr (VEC) __builtin_amdgcn_sched_barrier(0);
            static_for<GROUP>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int r0 = low_reg(j, (g % GPS) * GROUP + k);
                constexpr int r1 = r0 | (1 << j);
                const u64 X = v[r0], Y = v[r1];
                u64 S = X + Y;
                const u64 D = X + cq - Y;
                // values entering canon_after_inverse: any multiple range for NEAR (it folds), below TQ*q otherwise
                constexpr bool fin_red = last && !(NEAR && !EX) && (2 * POL.cmul[beta] > Lazy<HL>::TQ);
                if constexpr (last && zero_history(r0, 5, JLO)) {
                    // summed in every stage of this round: the only values that still need an explicit n^-1
                    const TwPair ni = twn[0];
                    if constexpr (!EX) v[r0] = mul_shoup4m<true>(S, ni.w, ni.wp, p.nq);
                    else v[r0] = mul_shoup<EX>(S, ni.w, ni.wp, p.nq);
                } else {
                    if constexpr (red || fin_red) {
                        // exact-quotient class: S < 4q always, one conditional subtraction of 2q (5 instructions, no multiply, fewer
                        // temporaries than the general partial reduction -- with it these kernels spilled)
                        if constexpr (EX && !NEAR) S = csub(S, 2 * p.q);
                        // general 61-bit primes (class 3, 8q < 2^64; round 5): a reducing stage sums two values below 4q, one conditional
                        // subtraction of 4q restores the bound the policy assumes behind a reduction (max(TQ, 2) q = 4q) -- no multiply, no
                        // reciprocal constants: with the 7-instruction general reduction the class-3 inverse and fused kernels spilled
                        else if constexpr (HL == 3 && !NEAR) S = csub(S, 4 * p.q);
                        else S = reduce_2q_sel<NEAR>(S, p);
                    }
                    v[r0] = S;
                }
                if constexpr (!EX) v[r1] = mul_shoup4m<!VEC>(D, Wc[k].w, Wc[k].wp, p.nq);
                else v[r1] = mul_shoup<EX>(D, Wc[k].w, Wc[k].wp, p.nq);
            });
            if constexpr (VEC) __builtin_amdgcn_sched_barrier(0);
        });
    }
}

// (ct_round / gs_round above: class HL_LIT -> the literal butterflies, every other class -> the lazy ones.)
// A context of class HL_LIT may hold Barrett-EXACT primes next to the inexact ones (the reference's own decryption_test.cu:47-48 set:
// two exact, one not).  For those the reference's words are the exact transform's, so their polynomials take the lazy butterflies of
// class HL_LIT_EXACT = 2 (exact quotients: valid for every q < 2^62; general partial reductions) -- the kernels branch per POLYNOMIAL
// on PrimeDev::lit (wave-uniform, scalar) around the whole body of the polynomial loop (kernels_fast_impl.cuh, MI355NTT_BODY_PER_CLASS).
constexpr int HL_LIT_EXACT = 2;
// The pointwise step of the fused products: forward output x in [0, B q) times a word of bhat.  Near-2^k classes with 4q of
// headroom: fold product, result in [0, 2q) (the inverse's first round then runs with IN2Q); otherwise Algorithm 7 on the
// canonicalised value, result canonical.
template <int HL, bool NEAR>
struct FusedMul {
    static constexpr bool LAZY = HL != HL_LIT && NEAR && !Lazy<HL>::EXACT;
    __device__ static __forceinline__ u64 mul(u64 x, u64 b, const PrimeDev& p)
    {
        // class HL_LIT: barrett_batch on what forwardNTT_batch left behind (poly_arithmetic.cuh:36-66; bfv_encryption.cuh:268-271) --
        // the literal forward rounds' words as they are, the exact primes' lazy values canonicalised first
        if constexpr (HL == HL_LIT) return barrett_mul(x, b, p.q, p.mu, p.k);
        else if constexpr (LAZY) return mul_fold_near(reduce_2q_near(x, p), b, p);
        else return barrett_mul(canon_2q(reduce_2q_sel<NEAR>(x, p), p.q), b, p.q, p.mu, p.k);      // poly_arithmetic.cuh:36-66
    }
};

// ------------------------------------------------------------------------------------------------
// whole transforms on registers.  Entry and exit layout: B0 (coalesced: i = (r << B0) | t).
// ------------------------------------------------------------------------------------------------
// TID: callable returning the thread index -- rebuilt at every use (wave index from an SGPR, lane index from v_mbcnt) instead of
// one VGPR kept live across the whole transform
template <int LOGN, int HL, int RHO, bool NEAR = false, class TID>
__device__ __forceinline__ void fwd_rounds(u64 (&v)[32], const TwPair* tw, BufRsrc twr, TID t, const PrimeDev& p, u64* lds)
{
    using G = Geo<LOGN>;
    if constexpr (RHO < G::NR) {
        constexpr int TOP = LOGN - 1 - 5 * RHO;
        constexpr int B = TOP - 4 > 0 ? TOP - 4 : 0;
        if constexpr (RHO > 0) {
            constexpr int TOPP = LOGN - 1 - 5 * (RHO - 1);
            constexpr int BP = TOPP - 4 > 0 ? TOPP - 4 : 0;
            exchange<LOGN, BP, B>(v, lds, t());
        }
        ct_round<LOGN, HL, B, TOP - B, NEAR>(v, tw, twr, t(), p);
        fwd_rounds<LOGN, HL, RHO + 1, NEAR>(v, tw, twr, t, p, lds);
    }
}

// natural-order coefficients in (layout B0, canonical) -> bit-reversed NTT values, left in layout 0, in [0, B*q)
template <int LOGN, int HL, bool NEAR = false, class TID>
__device__ __forceinline__ void forward_core(u64 (&v)[32], const TwPair* tw, TID t, const PrimeDev& p, u64* lds)
{
    fwd_rounds<LOGN, HL, 0, NEAR>(v, tw, make_rsrc(tw, Geo<LOGN>::N * 16u), t, p, lds);
}

template <int LOGN, int HL, int RHO, bool NEAR = false, bool IN2Q = false, class TID>
__device__ __forceinline__ void inv_rounds(u64 (&v)[32], const TwPair* tw, BufRsrc twr, TID t, const PrimeDev& p, u64* lds, const TwPair* twn)
{
    using G = Geo<LOGN>;
    if constexpr (RHO < G::NR) {
        constexpr int LOW = 5 * RHO;
        constexpr int B = LOW < G::B0 ? LOW : G::B0;
        if constexpr (RHO > 0) {
            constexpr int LOWP = 5 * (RHO - 1);
            constexpr int BP = LOWP < G::B0 ? LOWP : G::B0;
            exchange<LOGN, BP, B>(v, lds, t());
        }
        gs_round<LOGN, HL, B, LOW - B, NEAR, -2, 0, (IN2Q && RHO == 0)>(v, tw, twr, t(), p, twn);
        inv_rounds<LOGN, HL, RHO + 1, NEAR, IN2Q>(v, tw, twr, t, p, lds, twn);
    }
}

// bit-reversed values in layout 0 (any representative below 2q... see callers) -> coefficients in layout B0, in [0, TQ*q)
template <int LOGN, int HL, bool NEAR = false, bool IN2Q = false, class TID>
__device__ __forceinline__ void inverse_core(u64 (&v)[32], const TwPair* tw, TID t, const PrimeDev& p, u64* lds, const TwPair* twn)
{
    inv_rounds<LOGN, HL, 0, NEAR, IN2Q>(v, tw, make_rsrc(tw, Geo<LOGN>::N * 16u), t, p, lds, twn);
}

// [0, TQ*q) -> [0, q).  Near-2^k primes: the 3-instruction fold brings [0, 4q) below 2q, so one compare/select pair
// instead of two (a compare + 64-bit subtract + two selects cost ~21 issue cycles, the fold 8.5).
template <int HL, bool NEAR = false>
__device__ __forceinline__ u64 canon_after_inverse(u64 x, const PrimeDev& p)
{
    if constexpr (HL == HL_LIT) return x;           // (the reference stores what its last stage produced)
    else if constexpr (!Lazy<HL>::EXACT) {
        if constexpr (NEAR) {
            x = reduce_2q_near(x, p);
        } else {
            const u64 twoq = 2 * p.q;
            x = x >= twoq ? x - twoq : x;
        }
    }
    return canon_2q(x, p.q);
}
// forward outputs: [0, B q) -> [0, q)
template <int HL, bool NEAR = false>
__device__ __forceinline__ u64 canon_after_forward(u64 x, const PrimeDev& p)
{
    if constexpr (HL == HL_LIT) return x;
    else return canon_2q(reduce_2q_sel<NEAR>(x, p), p.q);
}

}  // namespace mi355ntt
