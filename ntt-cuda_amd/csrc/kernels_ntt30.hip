// kernels_ntt30.hip -- the reference's 30-bit path (old/ntt_30bit.cuh): 32-bit coefficients, one prime
// q < 2^30 per call, the caller's psi table and Barrett constant.
//
// Two kernel families behind the same entry points:
//   * k_ntt30x (native): one workgroup of n/32 threads per polynomial, 32 coefficients per thread in 32 VGPRs, rounds of five
//     in-register radix-2 stages with the polynomial exchanged through a padded LDS image between rounds (two exchanges at
//     n = 2^15 instead of the reference's 15 synchronised stages), lazy 32-bit Harvey/Shoup butterflies (values in [0, 4q),
//     9 VALU instructions), canonical at the end.  The Shoup companions floor(w * 2^32 / q) of the CALLER's table are
//     recomputed by a small kernel in front of every call (k_ntt30_prepare: n exact divisions into a per-stream scratch
//     table), so the transform always follows the table the caller passed; words equal the reference's because its
//     single-subtraction Barrett is exact for the moduli this path is taken for (checked on the host) and both are then
//     the exact transform.  n = 2^16 (old/ntt_30bit.cuh:271-283,323-331): one stage in memory + two independent 2^15
//     transforms whose stage L reads table entries [2L + hL, 2L + (h+1)L) -- the same kernel with a table multiplier.
//   * k_ntt30 / k_ntt30_stage (literal): the reference's butterflies on the reference's indices with the caller's mu
//     (Barrett-inexact moduli, hand-made mu, table entries >= q): every stage in LDS (n <= 2^15) or one launch per stage.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdlib>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "kernels.hpp"
#include "ntt_core.cuh"

namespace mi355ntt {

namespace {

using u32 = unsigned;

// singleBarrett, old/ntt_30bit.cuh:52-68
__device__ __forceinline__ u32 barrett30(u64 a, u32 q, u32 mu, int qbit)
{
    u64 rx = a >> (qbit - 2);
    rx *= mu;
    rx >>= qbit + 2;
    rx *= q;
    a -= rx;
    if (a >= q) a -= q;
    return (u32)a;
}

// guard words of the native path: {epoch of this call, epoch of the last call whose table had an entry >= q}
__device__ __forceinline__ bool literal_leg_skips(const unsigned* guard) { return guard && guard[0] != guard[1]; }

// all stages of CTBasedNTTInner(Single) (old/ntt_30bit.cuh:70-129, 199-227) / GSBasedINTTInner(Single) (:131-196, 229-267)
template <int LOGN, bool FWD>
__global__ void __launch_bounds__(1024)
k_ntt30(u32* __restrict__ a, const u32* __restrict__ tab, u32 q, u32 mu, int qbit, unsigned num, const unsigned* __restrict__ guard,
        unsigned split)
{
    // split: the polynomials are the halves of 2^(LOGN+1)-word polynomials whose first (forward) / last (inverse) stage runs as a
    // stage launch: stage L of half h is the sub-block [L (2 + h), L (2 + h) + L) of the caller's stage block 2 L -- the same
    // butterflies on the same words as that stage of the full-size transform, so the words are the reference's
    if (literal_leg_skips(guard)) return;
    constexpr unsigned n = 1u << LOGN, T = n / 2 < 1024 ? n / 2 : 1024, PER = n / 2 / T;
    __shared__ u32 s[n];
    const unsigned t = threadIdx.x;
    const u32 q2 = (q + 1) >> 1;
    for (unsigned y = blockIdx.x; y < num; y += gridDim.x) {
        u32* poly = a + (size_t)y * n;
        const unsigned tm = split ? 2u + (y & 1u) : 1u;
        for (unsigned i = t; i < n; i += T) s[i] = poly[i];
        __syncthreads();
        if constexpr (FWD) {
            for (unsigned length = 1; length < n; length *= 2) {
                const unsigned step = (n / length) / 2;
                for (unsigned it = 0; it < PER; it++) {
                    const unsigned g = t + it * T;
                    const unsigned psi_step = g / step;
                    const unsigned j = psi_step * step * 2 + g % step;
                    const u32 psi = tab[length * tm + psi_step];
                    u32 U = s[j];
                    const u32 V = barrett30((u64)s[j + step] * psi, q, mu, qbit);
                    u32 r = U + V;
                    r -= q * (r >= q);
                    s[j] = r;
                    U += q * (U < V);
                    s[j + step] = U - V;
                }
                __syncthreads();
            }
        } else {
            for (unsigned length = n / 2; length >= 1; length /= 2) {
                const unsigned step = (n / length) / 2;
                for (unsigned it = 0; it < PER; it++) {
                    const unsigned g = t + it * T;
                    const unsigned psi_step = g / step;
                    const unsigned j = psi_step * step * 2 + g % step;
                    const u32 psiinv = tab[length * tm + psi_step];
                    u32 U = s[j];
                    const u32 V = s[j + step];
                    u32 r = U + V;
                    r -= q * (r >= q);
                    s[j] = (r >> 1) + q2 * (r & 1);
                    U += q * (U < V);
                    const u32 d = barrett30((u64)(U - V) * psiinv, q, mu, qbit);
                    s[j + step] = (d >> 1) + q2 * (d & 1);
                }
                __syncthreads();
            }
        }
        for (unsigned i = t; i < n; i += T) poly[i] = s[i];
        __syncthreads();
    }
}

// one literal stage in memory (CTBasedNTTInner / GSBasedINTTInner, old/ntt_30bit.cuh:199-267): n = 2^16 and the fallback
template <bool FWD>
__global__ void __launch_bounds__(256)
k_ntt30_stage(u32* __restrict__ a, const u32* __restrict__ tab, unsigned n, unsigned length, u32 q, u32 mu, int qbit, unsigned num,
              const unsigned* __restrict__ guard)
{
    if (literal_leg_skips(guard)) return;
    const unsigned half = n / 2, step = (n / length) / 2;
    const u32 q2 = (q + 1) >> 1;
    const size_t total = (size_t)num * half;
    for (size_t x = (size_t)blockIdx.x * 256 + threadIdx.x; x < total; x += (size_t)gridDim.x * 256) {
        const unsigned y = (unsigned)(x / half), g = (unsigned)(x % half);
        const unsigned p = g / step, j = p * step * 2 + g % step;
        u32* poly = a + (size_t)y * n;
        const u32 w = tab[length + p];
        u32 U = poly[j];
        if constexpr (FWD) {
            const u32 V = barrett30((u64)poly[j + step] * w, q, mu, qbit);
            u32 r = U + V;
            r -= q * (r >= q);
            poly[j] = r;
            U += q * (U < V);
            poly[j + step] = U - V;
        } else {
            const u32 V = poly[j + step];
            u32 r = U + V;
            r -= q * (r >= q);
            poly[j] = (r >> 1) + q2 * (r & 1);
            U += q * (U < V);
            const u32 d = barrett30((u64)(U - V) * w, q, mu, qbit);
            poly[j + step] = (d >> 1) + q2 * (d & 1);
        }
    }
}

// barrett_30bit, old/ntt_30bit.cuh:10-35
__global__ void __launch_bounds__(256)
k_barrett30(u32* __restrict__ a, const u32* __restrict__ b, size_t count, u32 q, u32 mu, int qbit)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        u64 rc = (u64)a[i] * b[i];
        u64 rx = rc >> (qbit - 2);
        rx *= mu;
        rx >>= qbit + 2;
        rx *= q;
        rc -= rx;
        a[i] = rc < q ? (u32)rc : (u32)(rc - q);
    }
}

// ================================================================================================ native path
// scratch of one (device, stream): header + the caller's table with Shoup companions
struct Scratch30 {
    unsigned guard[2];       // {epoch, epoch of the last table with an entry >= q}
    u32 ninv, ninv_p;        // m^-1 mod q for the transform size m the kernel runs (n, or n/2 for the split) + companion
    u32 w1n[2], w1n_p[2];    // psiinv-table entry of the last GS stage (index A: 1, or 2 + h for half h) times ninv + companions
    unsigned pad[8];
    unsigned reserved_[1024];   // (round 3 kept the PAIR flags here, per (device, stream); they are the device's now: kernels.hpp, pair_acquire)
    uint2 tw[65536];         // {w, floor(w * 2^32 / q)}
};

__device__ __forceinline__ u32 shoup32_companion(u32 w, u32 q) { return (u32)(((u64)w << 32) / q); }

__global__ void __launch_bounds__(256)
k_ntt30_prepare(const u32* __restrict__ tab, unsigned n, u32 q, u32 ninv, unsigned split, unsigned fwd, Scratch30* __restrict__ sc, unsigned epoch)
{
    // The round on index bits <= 4 (register field at bit 0: the forward's last round, the inverse's first) gives every
    // thread its own twiddles: in the caller's order thread t reads cnt = 16 >> j consecutive entries of stage block j,
    // i.e. lanes are 8 cnt bytes apart and every load instruction touches 64 cache lines.  The scratch copy stores those
    // blocks transposed (entry k of thread t at k T + t, T = m / 32 threads), so that a load instruction reads 512
    // consecutive bytes.  m: size the native kernel transforms (n, or n / 2 when the first stage is split off: block L of
    // half h then is the sub-block [L (2 + h), L (2 + h) + L) of the caller's block 2 L).
    const unsigned m = split ? n / 2 : n, logm = 31u - __clz(m), T = m / 32u;
    const unsigned top_b0 = fwd ? (logm - 1u) % 5u : 4u;   // highest index bit of the round with the register field at bit 0
    bool bad = false;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const u32 w = tab[i];
        if (w >= q && i != 0) bad = true;                  // (entry 0 is never read)
        unsigned dst = i;
        if (i >= (split ? 2u : 1u)) {
            const unsigned l0 = 1u << (31u - __clz(i));
            const unsigned len = split ? l0 / 2 : l0, base = l0 + ((i - l0) / len) * len, pos = (i - l0) % len;
            const unsigned beta = logm - 1u - (31u - __clz(len));
            if (beta <= top_b0) {
                const unsigned cnt = 16u >> beta;
                dst = base + (pos % cnt) * T + pos / cnt;
            }
        }
        sc->tw[dst] = make_uint2(w, shoup32_companion(w < q ? w : 0, q));
    }
    if (bad) sc->guard[1] = epoch;       // (every writer stores the same value; a plain store stays right when the host epoch wraps)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc->guard[0] = epoch;
        sc->ninv = ninv;
        sc->ninv_p = shoup32_companion(ninv, q);
        for (unsigned h = 0; h < 2; h++) {
            const u32 w1 = tab[split ? 2 + h : 1] % q;
            const u32 v = (u32)(((u64)w1 * ninv) % q);
            sc->w1n[h] = v;
            sc->w1n_p[h] = shoup32_companion(v, q);
        }
    }
}

__device__ __forceinline__ u32 min_u32(u32 a, u32 b) { return a < b ? a : b; }
// y * w mod q in [0, 2q) for ANY 32-bit y (w < q, wp = floor(w 2^32 / q))
__device__ __forceinline__ u32 shoup32(u32 y, u32 w, u32 wp, u32 q) { return y * w - __umulhi(y, wp) * q; }

// LDS image of the native kernel: element i at word i + (i >> 5) (one pad word per 32: every access pattern of the three
// layouts is conflict-free or 2-way).  Thread part and register part of an index occupy disjoint bit fields, so the slot
// splits into a per-thread base and a compile-time offset per register.
constexpr unsigned pad32(unsigned i) { return i + (i >> 5); }
// n = 2^16: polynomials per call from which the pair launches win over the stage launch + one workgroup per half (tools/crossover30.py,
// profiles/r03_ntt30_n65536_pair.txt): forward from 32 (63.6 against 68.0 us), inverse -- whose lower workgroup carries the last stage alone --
// from a few hundred (256: 116.8 = 116.8 us, 512: 169.6 against 176.1 us)
constexpr unsigned kPair30MinPolysFwd = 32, kPair30MinPolysInv = 384;

template <int BO, int BN>
__device__ __forceinline__ void exchange32(u32 (&v)[32], u32* img, unsigned t)
{
#ifdef NTT30_NOEX                                         // timing experiment: no workgroup exchange
    return;
#endif
    __builtin_amdgcn_sched_barrier(0);
    // DS instructions carry a 16-bit byte offset: the image (up to 132 KiB) is addressed through three bases 64 KiB apart,
    // pinned so that the compiler does not materialise one address per register
    constexpr unsigned SEG = 16384;                     // words per 64 KiB
    unsigned wb0 = pad32(elem_index<BO>(t, 0)), rb0 = pad32(elem_index<BN>(t, 0));
    asm volatile("" : "+v"(wb0), "+v"(rb0));
    unsigned wb1 = wb0 + SEG, wb2 = wb0 + 2 * SEG, rb1 = rb0 + SEG, rb2 = rb0 + 2 * SEG;
    asm volatile("" : "+v"(wb1), "+v"(wb2), "+v"(rb1), "+v"(rb2));
    u32* const wb[3] = {img + wb0, img + wb1, img + wb2};
    const u32* const rb[3] = {img + rb0, img + rb1, img + rb2};
    // The barrier that protects the image from the previous use (exchange reads or the wave-local row staging) sits in
    // front of the writes, not behind the reads: by the time a wave gets here the others have long finished reading, so
    // of the two barriers only the one between writes and reads is a real rendezvous.
    __syncthreads();
    static_for<32>([&](auto rc) {
        constexpr unsigned off = pad32((unsigned)decltype(rc)::value << BO);
        wb[off / SEG][off % SEG] = v[decltype(rc)::value];
    });
    __syncthreads();
    static_for<32>([&](auto rc) {
        constexpr unsigned off = pad32((unsigned)decltype(rc)::value << BN);
        v[decltype(rc)::value] = rb[off / SEG][off % SEG];
    });
    __builtin_amdgcn_sched_barrier(0);
}

// Layout 0 (thread = 32 consecutive words = one 128-byte row; a wave = 8 KiB contiguous) <-> global memory with 16-byte
// accesses, staged through the wave's own 8 KiB of the image: no workgroup barrier.  The eight 16-byte pieces of row R
// sit at slot piece ^ ((R >> 1) & 7), so that both the row accesses (lane = row) and the transposed accesses (8 lanes
// per row, 8 rows = 1 KiB contiguous per instruction) spread over all banks.
__device__ __forceinline__ unsigned row_swz32(unsigned row) { return (row >> 1) & 7u; }
__device__ __forceinline__ unsigned row_swz32_store(unsigned row) { return row & 7u; }      // (ntt_core.cuh, row_swz_store)
__device__ __forceinline__ void lds_fence32() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void wave_store_rows32(const u32 (&v)[32], u32* img, BufRsrc dst, unsigned t)
{
    __builtin_amdgcn_sched_barrier(0);
    const unsigned lane = t & 63u, wave = t >> 6;
    char* base = reinterpret_cast<char*>(img) + wave * 8192u;
    static_for<8>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        v4u32 x;
        x.x = v[4 * m]; x.y = v[4 * m + 1]; x.z = v[4 * m + 2]; x.w = v[4 * m + 3];
        *reinterpret_cast<v4u32*>(base + lane * 128u + ((m ^ row_swz32_store(lane)) << 4)) = x;
    });
    lds_fence32();
    const unsigned sw = lane & 7u, rr = lane >> 3;
    static_for<8>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const unsigned row = 8 * k + rr;
        const v4u32 x = *reinterpret_cast<const v4u32*>(base + row * 128u + ((sw ^ row_swz32_store(row)) << 4));
        __builtin_amdgcn_raw_buffer_store_b128(x, dst, wave * 8192u + rr * 128u + sw * 16u, k * 1024u, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
}

// ---- prefetch of the next polynomial ---------------------------------------------------------------------------------
// Vector-memory results return in order and share one counter (vmcnt) with the stores, and the compiler's waitcnt pass
// merges the loop entry and the back edge conservatively.  With "load the first polynomial, then conditionally prefetch
// inside the loop" it drained vmcnt to 0 inside round 1 of EVERY iteration -- waiting for the prefetch it had just issued
// and for the previous polynomial's stores.  The loop below therefore has ONE shape on both paths: the prefetch is
// unconditional (past the end it loads through a zero-length descriptor: out of range, returns 0, no memory traffic) and
// the prologue issues the same number of (zero-length, dropped) stores behind the first loads as an iteration does, so
// that "wait for the prefetch, leave the stores in flight" is the same count on entry and in steady state.
// the row-pattern loads (x[4k .. 4k+3] = piece (lane & 7) of row 8k + (lane >> 3)), and their way into layout 0
__device__ __forceinline__ void issue_row_loads32(u32 (&x)[32], BufRsrc src, unsigned t)
{
    const unsigned lane = t & 63u, wave = t >> 6;
    const unsigned sw = lane & 7u, rr = lane >> 3;
    static_for<8>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const v4u32 p = __builtin_amdgcn_raw_buffer_load_b128(src, wave * 8192u + rr * 128u + sw * 16u, k * 1024u, 0);
        x[4 * k] = p.x; x[4 * k + 1] = p.y; x[4 * k + 2] = p.z; x[4 * k + 3] = p.w;
    });
}

__device__ __forceinline__ void rows_to_layout0_32(u32 (&v)[32], u32* img, unsigned t)
{
    __builtin_amdgcn_sched_barrier(0);
    const unsigned lane = t & 63u, wave = t >> 6;
    char* base = reinterpret_cast<char*>(img) + wave * 8192u;
    const unsigned sw = lane & 7u, rr = lane >> 3;
    static_for<8>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const unsigned row = 8 * k + rr;
        v4u32 x;
        x.x = v[4 * k]; x.y = v[4 * k + 1]; x.z = v[4 * k + 2]; x.w = v[4 * k + 3];
        *reinterpret_cast<v4u32*>(base + row * 128u + ((sw ^ row_swz32(row)) << 4)) = x;
    });
    lds_fence32();
    static_for<8>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const v4u32 x = *reinterpret_cast<const v4u32*>(base + lane * 128u + ((m ^ row_swz32(lane)) << 4));
        v[4 * m] = x.x; v[4 * m + 1] = x.y; v[4 * m + 2] = x.z; v[4 * m + 3] = x.w;
    });
    lds_fence32();
    __builtin_amdgcn_sched_barrier(0);
}

// Twiddles of butterfly group G of a round: 8 butterflies per group, two groups per stage, the loads run one group ahead
// (a ring of 2 x 8 pairs = 32 VGPRs); scheduling fences around each group keep the compiler from hoisting a whole round's
// loads (which cost the first version 144 VGPRs and spills).
#ifndef NTT30_GROUP
#define NTT30_GROUP 8
#endif
constexpr int GROUP32 = NTT30_GROUP, GPS32 = 16 / GROUP32;      // butterflies per twiddle group, groups per stage

// n = 2^15: the twiddles of the middle round (register field at bit 5: stage blocks 32 ... 512, 992 pairs = 8 KiB, two
// distinct addresses per wave) stay in LDS for the life of the persistent workgroup -- one modulus per call, so every
// polynomial of the workgroup uses the same ones.  That takes half of the vector twiddle loads off the vector-memory
// counter: the middle round then never waits behind the prefetch of the next polynomial (results return in order).
#ifndef NTT30_LDS_TW
#define NTT30_LDS_TW 1
#endif
__device__ __forceinline__ uint2* tw2_lds()
{
    __shared__ uint2 buf[1024];
    return buf;
}
template <int LOGN, int B, int JA, bool FWD, int G>
__device__ __forceinline__ void load_tw32(uint2 (&W)[GROUP32], const uint2* __restrict__ tw, BufRsrc twr, unsigned tmul, unsigned thi)
{
    constexpr int j = FWD ? JA - G / GPS32 : JA + G / GPS32;
    constexpr unsigned len = 1u << (LOGN - 1 - (B + j));
#ifdef NTT30_NOTW                                         // timing experiment (tools/kbench30.hip): no twiddle loads
    static_for<GROUP32>([&](auto kc) { W[decltype(kc)::value] = make_uint2(12345u + thi, 54321u + tmul); });
    return;
#endif
    if constexpr (B == Geo<LOGN>::B0) {                  // first / last round: the group index does not depend on the thread -> scalar loads
        static_for<GROUP32>([&](auto kc) {
            constexpr int r0 = low_reg(j, (G % GPS32) * GROUP32 + decltype(kc)::value);
            W[decltype(kc)::value] = tw[len * tmul + ((unsigned)r0 >> (j + 1))];
        });
    } else if constexpr (NTT30_LDS_TW && LOGN == 15 && B == 5) {      // the workgroup's LDS copy (k_ntt30x fills it once): index as in the table, tmul folded in
        const uint2* lt = tw2_lds() + len + (thi << (4 - j));
        static_for<GROUP32>([&](auto kc) {
            constexpr int r0 = low_reg(j, (G % GPS32) * GROUP32 + decltype(kc)::value);
            W[decltype(kc)::value] = lt[(unsigned)r0 >> (j + 1)];
        });
    } else if constexpr (B == 0) {                       // per-thread twiddles, stored transposed by k_ntt30_prepare: entry k of thread t at k T + t
        const unsigned voff = thi * 8u;
        const unsigned soff = len * tmul * 8u;
        static_for<GROUP32>([&](auto kc) {
            constexpr int r0 = low_reg(j, (G % GPS32) * GROUP32 + decltype(kc)::value);
            constexpr unsigned koff = ((unsigned)r0 >> (j + 1)) * (unsigned)Geo<LOGN>::T * 8u;
            const v2u32 x = __builtin_amdgcn_raw_buffer_load_b64(twr, voff, soff + koff, 0);
            W[decltype(kc)::value] = make_uint2(x.x, x.y);
        });
    } else {                                             // one 32-bit lane offset, the rest in the scalar offset / immediate
        const unsigned voff = (thi << (4 - j)) * 8u;
        const unsigned soff = len * tmul * 8u;
        static_for<GROUP32>([&](auto kc) {
            constexpr int r0 = low_reg(j, (G % GPS32) * GROUP32 + decltype(kc)::value);
            const v2u32 x = __builtin_amdgcn_raw_buffer_load_b64(twr, voff + ((unsigned)r0 >> (j + 1)) * 8u, soff, 0);
            W[decltype(kc)::value] = make_uint2(x.x, x.y);
        });
    }
}

// CT stages on register bits JHI..0 of the layout with register field at bit B.  Values in [0, 4q): the first operand is
// brought below 2q (one v_sub + v_min), the Shoup product lies in [0, 2q) for any 32-bit multiplicand.
// tmul: table multiplier (1; 2 + h for half h of a split transform: stage L then reads entries [L tmul, L tmul + L)).
template <int LOGN, int B, int JHI>
__device__ __forceinline__ void ct_round32(u32 (&v)[32], const uint2* __restrict__ tw, BufRsrc twr, unsigned tmul, unsigned t, u32 q)
{
    const u32 twoq = 2 * q;
    const unsigned thi = t >> B;
    constexpr int NG = (JHI + 1) * GPS32;
    uint2 W[2][GROUP32];
    load_tw32<LOGN, B, JHI, true, 0>(W[0], tw, twr, tmul, thi);
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int j = JHI - g / GPS32;
        if constexpr (g + 1 < NG) load_tw32<LOGN, B, JHI, true, g + 1>(W[(g + 1) % 2], tw, twr, tmul, thi);
        __builtin_amdgcn_sched_barrier(0);
        static_for<GROUP32>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int r0 = low_reg(j, (g % GPS32) * GROUP32 + k), r1 = r0 | (1 << j);
            const uint2 w = W[g % 2][k];
            const u32 X = min_u32(v[r0], v[r0] - twoq);
            const u32 T = shoup32(v[r1], w.x, w.y, q);
            v[r0] = X + T;
            v[r1] = X + twoq - T;
        });
        __builtin_amdgcn_sched_barrier(0);
    });
}

// GS stages on register bits JLO..4.  Values in [0, 2q) between stages; the stage on index bit LOGN - 1 (length 1) carries
// m^-1: both outputs are Shoup products (the reference halves in every stage instead, old/ntt_30bit.cuh:131-196).
// first: the round's first twiddle group when the caller keeps it in registers across polynomials (k_ntt30x: the first
// round of every polynomial of a workgroup uses the same twiddles, and nothing is there to hide that first load behind)
template <int LOGN, int B, int JLO>
__device__ __forceinline__ void gs_round32(u32 (&v)[32], const uint2* __restrict__ tw, BufRsrc twr, unsigned tmul, unsigned t, u32 q, const Scratch30* sc_, unsigned h,
                                           const uint2 (*first)[GROUP32] = nullptr)
{
    const u32 twoq = 2 * q;
    const unsigned thi = t >> B;
    constexpr int NG = (5 - JLO) * GPS32;
    uint2 W[2][GROUP32];
    if (first) {
        static_for<GROUP32>([&](auto kc) { W[0][decltype(kc)::value] = (*first)[decltype(kc)::value]; });
    } else {
        load_tw32<LOGN, B, JLO, false, 0>(W[0], tw, twr, tmul, thi);
    }
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int j = JLO + g / GPS32;
        constexpr int beta = B + j;
        constexpr bool last = (beta == LOGN - 1);
        if constexpr (g + 1 < NG) load_tw32<LOGN, B, JLO, false, g + 1>(W[(g + 1) % 2], tw, twr, tmul, thi);
        __builtin_amdgcn_sched_barrier(0);
        u32 ninv = 0, ninv_p = 0, w1n = 0, w1n_p = 0;
        if constexpr (last) {
            ninv = sc_->ninv; ninv_p = sc_->ninv_p; w1n = sc_->w1n[h]; w1n_p = sc_->w1n_p[h];
        }
        static_for<GROUP32>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int r0 = low_reg(j, (g % GPS32) * GROUP32 + k), r1 = r0 | (1 << j);
            const u32 X = v[r0], Y = v[r1];
            u32 S = X + Y;
            S = min_u32(S, S - twoq);
            const u32 D = X + twoq - Y;
            if constexpr (last) {
                v[r0] = shoup32(S, ninv, ninv_p, q);
                v[r1] = shoup32(D, w1n, w1n_p, q);
            } else {
                const uint2 w = W[g % 2][k];
                v[r0] = S;
                v[r1] = shoup32(D, w.x, w.y, q);
            }
        });
        __builtin_amdgcn_sched_barrier(0);
    });
}

// after_first_exchange: called behind the first workgroup-wide exchange (every wave has passed its barriers, i.e. has finished
// whatever it did with its inputs in front of round 1)
template <int LOGN, int RHO, class F>
__device__ __forceinline__ void fwd_rounds32(u32 (&v)[32], const uint2* tw, BufRsrc twr, unsigned tmul, unsigned t, u32 q, u32* img, F&& after_first_exchange)
{
    using G = Geo<LOGN>;
    if constexpr (RHO < G::NR) {
        constexpr int TOP = LOGN - 1 - 5 * RHO;
        constexpr int B = TOP - 4 > 0 ? TOP - 4 : 0;
        if constexpr (RHO > 0) {
            constexpr int TOPP = LOGN - 1 - 5 * (RHO - 1);
            constexpr int BP = TOPP - 4 > 0 ? TOPP - 4 : 0;
            exchange32<BP, B>(v, img, t);
            if constexpr (RHO == 1) after_first_exchange();
        }
        ct_round32<LOGN, B, TOP - B>(v, tw, twr, tmul, t, q);
        fwd_rounds32<LOGN, RHO + 1>(v, tw, twr, tmul, t, q, img, after_first_exchange);
    }
}

// before_last: called in front of the last round (the one whose twiddles come through scalar loads) -- the place where the
// inverse kernel issues the next polynomial's loads: vector-memory results return in order, so a prefetch issued earlier
// would sit in front of every twiddle load of the rounds before and turn their waits into waits for HBM
template <int LOGN, int RHO, class F>
__device__ __forceinline__ void inv_rounds32(u32 (&v)[32], const uint2* tw, BufRsrc twr, unsigned tmul, unsigned t, u32 q, u32* img, const Scratch30* sc, unsigned h,
                                             F&& before_last, const uint2 (*first)[GROUP32])
{
    using G = Geo<LOGN>;
    if constexpr (RHO < G::NR) {
        constexpr int LOW = 5 * RHO;
        constexpr int B = LOW < G::B0 ? LOW : G::B0;
        if constexpr (RHO == G::NR - 1) before_last();
        if constexpr (RHO > 0) {
            constexpr int LOWP = 5 * (RHO - 1);
            constexpr int BP = LOWP < G::B0 ? LOWP : G::B0;
            exchange32<BP, B>(v, img, t);
        }
        gs_round32<LOGN, B, LOW - B>(v, tw, twr, tmul, t, q, sc, h, RHO == 0 ? first : nullptr);
        inv_rounds32<LOGN, RHO + 1>(v, tw, twr, tmul, t, q, img, sc, h, before_last, first);
    }
}

// start stagger of the persistent workgroups (forward kernel only; measured at 4096 polynomials of 2^15 words: +3 % with
// 2 units, nothing on the inverse)
#ifndef NTT30_STAGGER
#define NTT30_STAGGER 2
#endif
// One workgroup of 2^LOGN / 32 threads per polynomial of 2^LOGN words, persistent over the batch.  split: the polynomials
// are the halves of 2^(LOGN+1)-word polynomials whose first (forward) / last (inverse) stage runs as a stage launch.
// PAIR (forward, split: the polynomials are the halves of 2^(LOGN+1)-word polynomials): no stage launch in front.  Two
// workgroups per full-size polynomial, one per output half (role = the half; on CUs of one XCD when the grid is a multiple of
// 16), both read the lower half U (prefetch set) and the upper half V (second set, requested behind the last round of the
// previous polynomial) and enter round 1 with U + V w' -- w' the table's entry 1 for the lower, its negative for the upper
// half -- so the polynomial is read once from HBM (the second reader hits the L2) and written once, instead of twice each.
// In place: a workgroup stores only after its partner has read the input underneath -- flags[2 pair + role] counts the
// polynomials a workgroup has read completely (written behind the first exchange, polled in front of the stores, cleared on
// exit); the buffer is the device's pair-flag slot (kernels.hpp: one pair kernel of either word size in flight per device), the grid is
// resident as a whole (one workgroup per CU).  As
// k_forward15_pair of the 60-bit path (kernels_fast_impl.cuh).
template <int LOGN, bool FWD, bool PAIR = false>
// (two workgroups of 1024 threads per CU would need 64 VGPRs per thread; without the prefetch set and with twiddle groups
// of 4 the forward kernel still wants 98 -- measured with a waves-per-SIMD bound of 8 -- so n = 2^15 stays at one)
__global__ void __launch_bounds__(Geo<LOGN>::T, 4)       // 128 VGPRs: four waves per SIMD, i.e. as many workgroups per CU as the LDS image admits
k_ntt30x(u32* __restrict__ a, const Scratch30* __restrict__ sc, u32 q, unsigned num, unsigned split, unsigned* __restrict__ flags)
{
    if (sc->guard[0] == sc->guard[1]) return;            // a table entry >= q: the literal leg transforms the data
    using G = Geo<LOGN>;
    constexpr unsigned n = G::N;
    __shared__ u32 img[pad32(n)];
    const unsigned t = threadIdx.x;
    const uint2* tw = sc->tw;
    const BufRsrc twr = make_rsrc(tw, 65536u * 8u);
    // first half-size polynomial of this workgroup and the distance to its next one
    [[maybe_unused]] unsigned role = 0, pair = 0;
    if constexpr (PAIR) {
        const unsigned w = blockIdx.x;
        const bool xcd_map = (gridDim.x & 15u) == 0;      // workgroups are dealt round-robin over the 8 XCDs: w and w + 8 share an L2
        role = __builtin_amdgcn_readfirstlane(xcd_map ? (w >> 3) & 1u : w & 1u);
        pair = __builtin_amdgcn_readfirstlane(xcd_map ? ((w >> 4) << 3) | (w & 7u) : w >> 1);
    }
    const unsigned first = PAIR ? 2 * pair + role : blockIdx.x, stride = gridDim.x;      // (PAIR: an even grid, 2 x pairs)
    auto flag_at = [&](unsigned which) {                  // (rebuilt from SGPRs at its uses: a pointer kept across the loop is a VGPR pair)
        unsigned f = __builtin_amdgcn_readfirstlane(2 * pair + which);
        asm volatile("" : "+s"(f));
        return flags + f;
    };
    // the next polynomial's coefficients are loaded into a second register set while the current one is transformed (one
    // workgroup per CU at n = 2^15: nothing else would cover the memory latency)
    // Forward: coalesced 4-byte loads (thread t, register r = word (r << B0) | t, the layout round 1 wants), results leave
    // from layout 0 through the wave-local row staging (16-byte stores); inverse: the mirror image.
    u32 v[32], nx[32];
    [[maybe_unused]] u32 nv[PAIR ? 32 : 1];
    auto issue_loads = [&](unsigned y, bool real) {
        const BufRsrc rs = make_rsrc(a + (size_t)(PAIR && FWD ? y & ~1u : y) * n, real ? n * 4u : 0u);      // (PAIR forward: the lower half U)
#if defined(NTT30_NOMEM) || defined(NTT30_NOLOAD)         // timing experiment: no polynomial traffic
        static_for<32>([&](auto rc) { nx[decltype(rc)::value] = (t + decltype(rc)::value + y) & 0xffffu; });
        return;
#endif
        if constexpr (FWD)
            static_for<32>([&](auto rc) { nx[decltype(rc)::value] = __builtin_amdgcn_raw_buffer_load_b32(rs, t * 4u, ((unsigned)decltype(rc)::value << G::B0) * 4u, 0); });
        else
            issue_row_loads32(nx, rs, t);
    };
    auto issue_loads_v = [&](unsigned y, bool real) {     // PAIR forward: the upper half V of the same full-size polynomial
        if constexpr (PAIR && FWD) {
            const BufRsrc rs = make_rsrc(a + (size_t)(y | 1u) * n, real ? n * 4u : 0u);
            static_for<32>([&](auto rc) { nv[decltype(rc)::value] = __builtin_amdgcn_raw_buffer_load_b32(rs, t * 4u, ((unsigned)decltype(rc)::value << G::B0) * 4u, 0); });
        }
    };
    auto issue_stores = [&](BufRsrc prs) {                // forward: 8 x 16 bytes per thread from layout 0; inverse: 32 x 4 bytes, coalesced layout
        if constexpr (FWD)
            wave_store_rows32(v, img, prs, t);
        else
            static_for<32>([&](auto rc) { __builtin_amdgcn_raw_buffer_store_b32(v[decltype(rc)::value], prs, t * 4u, ((unsigned)decltype(rc)::value << G::B0) * 4u, 0); });
    };
#if NTT30_STAGGER > 0
    // every workgroup runs the same schedule; 8 phase groups per XCD start NTT30_STAGGER x 2048 cycles apart
    if (FWD && gridDim.x >= 256u)                          // (PAIR: partners start together)
        for (unsigned i = 0; i < ((blockIdx.x >> (PAIR ? 4 : 3)) & 7u) * NTT30_STAGGER; i++) __builtin_amdgcn_s_sleep(32);
#endif
    if (first >= num) return;
    if constexpr (NTT30_LDS_TW && LOGN == 15) {          // middle-round twiddles into LDS (read only after the first exchange's barriers)
        const unsigned tm = split ? 2u + (first & 1u) : 1u;
        for (unsigned i = 32u + t; i < 1024u; i += G::T) {
            const unsigned l0 = 1u << (31u - __clz(i));
            tw2_lds()[i] = tw[l0 * tm + (i - l0)];
        }
    }
    // inverse: the first twiddle group of the first round stays in registers (every polynomial of this workgroup belongs to
    // the same half when the transform is split: the grid is even whenever a workgroup sees more than one polynomial)
    [[maybe_unused]] uint2 W0[GROUP32];
    if constexpr (!FWD) load_tw32<LOGN, 0, 0, false, 0>(W0, tw, twr, split ? 2u + (first & 1u) : 1u, t);
    issue_loads(first, true);
    issue_loads_v(first, true);
    static_for<32>([&](auto rc) { v[decltype(rc)::value] = 0; });
    issue_stores(make_rsrc(a, 0u));                       // (zero-length descriptor: dropped; see above)
    [[maybe_unused]] unsigned it = 0;
    for (unsigned y = first; y < num; y += stride) {
        const BufRsrc prs = make_rsrc(a + (size_t)y * n, n * 4u);
        const unsigned h = split ? (y & 1u) : 0u, tmul = split ? 2u + h : 1u;
        const bool more = y + stride < num;
        static_for<32>([&](auto rc) { v[decltype(rc)::value] = nx[decltype(rc)::value]; });
        if constexpr (PAIR) {
            // the stage that couples the halves: U + V w' in [0, 3q) (round 1 takes [0, 4q)); -w = q - w has the companion ~wp
            // (floor((q - w) 2^32 / q) = 2^32 - 1 - floor(w 2^32 / q): w 2^32 is no multiple of the prime q)
            const uint2 w1 = tw[1];
            const u32 cw = h ? q - w1.x : w1.x, cwp = h ? ~w1.y : w1.y;
            static_for<32>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                v[r] += shoup32(nv[r], cw, cwp, q);
            });
        }
        // the prefetch of the next polynomial (unconditional: one instruction stream; past the end it loads nothing) goes in
        // front of the round with scalar twiddle loads: the forward's first, the inverse's last
        auto prefetch = [&]() { issue_loads(more ? y + stride : y, more); };
        if constexpr (FWD) {
            prefetch();
            fwd_rounds32<LOGN, 0>(v, tw, twr, tmul, t, q, img, [&]() {
                if constexpr (PAIR) {                    // every wave holds its share of the input: the partner may store
                    unsigned* const mf = flag_at(h);
                    if (t == 0) __hip_atomic_store(mf, it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            });
            static_for<32>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                u32 x = min_u32(v[r], v[r] - 2 * q);
                v[r] = min_u32(x, x - q);
            });
            __syncthreads();                             // the last exchange has been read by every wave: the image is free
            if constexpr (PAIR) {
                // V of the next polynomial in front of the stores (results return in order: behind them it would wait for their
                // drain), then: has the partner read the input under this result?  (long since, normally; a partner that never
                // shows up -- 30 s of wall clock -- means the grid is not resident as a whole: the workgroup gives up, kernels.hpp, rather than hang)
                issue_loads_v(more ? y + stride : y, more);
                unsigned* const pf = flag_at(1u - h);
                if (__hip_atomic_load(pf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= it) {
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // (watchdog by the constant 100 MHz clock)
                    while (__hip_atomic_load(pf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= it) {
                        __builtin_amdgcn_s_sleep(8);
                        pair_watchdog_check(flags, t0);      // (gives up -- ends the wave -- without storing: kernels.hpp)
                    }
                }
                asm volatile("" ::: "memory");            // (compiler-level order: no store of the result moves above the poll)
                it++;
            }
#if !defined(NTT30_NOMEM) && !defined(NTT30_NOSTORE)
            issue_stores(prs);
#else
            {
                u32 x = 0;                                // (keeps every output alive)
                static_for<32>([&](auto rc) { x ^= v[decltype(rc)::value] + decltype(rc)::value; });
                if (x == 0xdeadbeefu) a[t] = x;
            }
#endif
        } else {
            __syncthreads();                             // (the previous polynomial's last exchange has been read)
            rows_to_layout0_32(v, img, t);
            inv_rounds32<LOGN, 0>(v, tw, twr, tmul, t, q, img, sc, h, prefetch, &W0);
            static_for<32>([&](auto rc) { v[decltype(rc)::value] = min_u32(v[decltype(rc)::value], v[decltype(rc)::value] - q); });
            if constexpr (PAIR) {
                // PAIR inverse: no stage launch behind.  The half-size results X (lower) and Y (upper) carry n^-1 already (the host
                // passes (2m)^-1); the last GS stage is X + Y below and (X - Y) psi^-bitrev(1) above.  The upper workgroup writes Y
                // through to memory, drains, and counts it in flags[2 pair + 1]; the lower one waits for that count, reads Y back
                // (system-scope loads: the two may sit on different XCDs) and stores both halves of the result -- the upper
                // workgroup is done with its half by then, so one flag suffices.  The hand-off is the guide's write-through form
                // (MI355X_MICROARCH.md, inter-workgroup visibility): every handed-off byte stored sc0 sc1, every storing wave drained
                // (vmcnt(0)) and the workgroup's barrier passed before the count is published, every load of those bytes sc0 sc1 --
                // no L2 write-back / L1 invalidate per polynomial.
                const BufRsrc urs = make_rsrc(a + (size_t)(y | 1u) * n, n * 4u);
                if (h == 1) {
                    static_for<32>([&](auto rc) { __builtin_amdgcn_raw_buffer_store_b32(v[decltype(rc)::value], urs, t * 4u, ((unsigned)decltype(rc)::value << G::B0) * 4u, 17); });
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();                      // every wave's words are in memory
                    asm volatile("" ::: "memory");        // (compiler-level order: the count is published behind the drain and the barrier)
                    unsigned* const mf = flag_at(1u);
                    if (t == 0) __hip_atomic_store(mf, it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    unsigned* const pf = flag_at(1u);
                    if (__hip_atomic_load(pf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= it) {
                        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                        while (__hip_atomic_load(pf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= it) {
                            __builtin_amdgcn_s_sleep(8);
                            pair_watchdog_check(flags, t0);      // (the partner never became resident: see the forward form)
                        }
                    }
                    // compiler-level acquire: the (non-volatile) buffer loads of Y below must not be hoisted above the poll -- the
                    // hardware issues them in order behind it, and they are sc0 sc1 (they do not hit a stale line)
                    asm volatile("" ::: "memory");
                    __atomic_signal_fence(__ATOMIC_ACQUIRE);
                    u32 yy[32];
                    static_for<32>([&](auto rc) { yy[decltype(rc)::value] = __builtin_amdgcn_raw_buffer_load_b32(urs, t * 4u, ((unsigned)decltype(rc)::value << G::B0) * 4u, 17); });
                    const uint2 w1 = tw[1];
                    static_for<32>([&](auto rc) {
                        constexpr int r = decltype(rc)::value;
                        const u32 X = v[r], Y = yy[r];
                        const u32 lo = min_u32(X + Y, X + Y - q);
                        u32 hi = shoup32(X + q - Y, w1.x, w1.y, q);
                        hi = min_u32(hi, hi - q);
                        __builtin_amdgcn_raw_buffer_store_b32(lo, prs, t * 4u, ((unsigned)r << G::B0) * 4u, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(hi, urs, t * 4u, ((unsigned)r << G::B0) * 4u, 0);
                    });
                }
                it++;
            } else {
#if !defined(NTT30_NOMEM) && !defined(NTT30_NOSTORE)
            issue_stores(prs);
#else
            {
                u32 x = 0;                                // (keeps every output alive)
                static_for<32>([&](auto rc) { x ^= v[decltype(rc)::value] + decltype(rc)::value; });
                if (x == 0xdeadbeefu) a[t] = x;
            }
#endif
            }
        }
    }
    if constexpr (PAIR) {
        __syncthreads();                                  // every wave has polled for the last time:
        unsigned* const pf = flag_at(FWD ? 1u - (first & 1u) : 1u);
        if (t == 0 && (FWD || (first & 1u) == 0)) __hip_atomic_store(pf, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // zero between launches
    }
}

// ---- scratch per (device, stream) -----------------------------------------------------------------------------------
std::mutex g_scratch_mutex;
std::mutex g_launch30_mutex;      // one for forward AND inverse calls: both directions share the (device, stream) scratch table
std::map<std::pair<int, hipStream_t>, Scratch30*> g_scratch;
unsigned g_epoch = 0;

Scratch30* scratch_for(hipStream_t s)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    auto key = std::make_pair(dev, s);
    auto it = g_scratch.find(key);
    if (it != g_scratch.end()) return it->second;
    Scratch30* p = nullptr;
    if (hipMalloc((void**)&p, sizeof(Scratch30)) != hipSuccess) return nullptr;
    // The header (guard words) is zeroed ON THE CALLER'S STREAM, in front of the first k_ntt30_prepare.  Up to round 6 this was a
    // hipMemset: asynchronous to the host for device memory and enqueued on the NULL stream, with which a non-blocking stream (every
    // torch stream) is not ordered -- with the device busy (another stream's pair launch holding every CU) it could land after the
    // first call's prepare kernel had written the guard words, and the guarded literal leg then transformed the data a second time.
    if (hipMemsetAsync(p, 0, offsetof(Scratch30, tw), s) != hipSuccess) {
        (void)hipFree(p);
        return nullptr;
    }
    g_scratch[key] = p;
    return p;
}

unsigned next_epoch()
{
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (++g_epoch == 0) g_epoch = 1;
    return g_epoch;
}

template <int LOGN, bool FWD>
void launch_native(u32* d_a, const Scratch30* sc, u32 q, unsigned num, unsigned split, hipStream_t s)
{
    const unsigned lds = pad32(1u << LOGN) * 4u, per_cu_lds = 163840u / lds, per_cu_waves = 16u / (Geo<LOGN>::T / 64u);   // 4 waves/SIMD at 128 VGPRs
    unsigned per_cu = per_cu_lds < per_cu_waves ? per_cu_lds : per_cu_waves;
    if (per_cu < 1) per_cu = 1;
    const unsigned cap = current_device_cus() * per_cu;
    k_ntt30x<LOGN, FWD><<<num < cap ? num : cap, Geo<LOGN>::T, 0, s>>>(d_a, sc, q, num, split, nullptr);
}

// n = 2^16 without the stage launch: two workgroups per polynomial (k_ntt30x PAIR); halves = 2 x polynomials
inline unsigned pair_grid(unsigned halves)
{
    const unsigned cus = current_device_cus() & ~1u;      // one workgroup per CU: the grid is resident as a whole
    const unsigned grid = halves < cus ? halves : cus;
    return grid < 2 || grid > 1024u ? 0u : grid;
}
template <bool FWD>
void launch_native_pair(u32* d_a, Scratch30* sc, u32 q, unsigned halves, unsigned grid, hipStream_t s, unsigned* d_flags)
{
    k_ntt30x<15, FWD, true><<<grid, Geo<15>::T, 0, s>>>(d_a, sc, q, halves, 1u, d_flags);
}

template <bool FWD>
void launch_literal_lds(unsigned* d_a, unsigned n, const unsigned* d_tab, unsigned num, unsigned q, unsigned mu, int bits, hipStream_t s,
                        const unsigned* guard)
{
    const unsigned gmax = guard ? 256u : 512u, g = num < gmax ? num : gmax;      // (guarded: see launch_stage)
    switch (n) {
    case 2048: k_ntt30<11, FWD><<<g, 1024, 0, s>>>(d_a, d_tab, q, mu, bits, num, guard, 0u); break;
    case 4096: k_ntt30<12, FWD><<<g, 1024, 0, s>>>(d_a, d_tab, q, mu, bits, num, guard, 0u); break;
    case 8192: k_ntt30<13, FWD><<<g, 1024, 0, s>>>(d_a, d_tab, q, mu, bits, num, guard, 0u); break;
    case 16384: k_ntt30<14, FWD><<<g, 1024, 0, s>>>(d_a, d_tab, q, mu, bits, num, guard, 0u); break;
    default: k_ntt30<15, FWD><<<g < 256u ? g : 256u, 1024, 0, s>>>(d_a, d_tab, q, mu, bits, num, guard, 0u); break;
    }
}

template <bool FWD>
void launch_stage(unsigned* d_a, unsigned n, const unsigned* d_tab, unsigned length, unsigned num, unsigned q, unsigned mu, int bits,
                  hipStream_t s, const unsigned* guard)
{
    const size_t total = (size_t)num * (n / 2), cap = guard ? 1024 : 8192;      // (a guarded leg normally returns at once: its price is the dispatch)
    const unsigned g = (unsigned)((total + 255) / 256 < cap ? (total + 255) / 256 : cap);
    k_ntt30_stage<FWD><<<g, 256, 0, s>>>(d_a, d_tab, n, length, q, mu, bits, num, guard);
}

// the literal transform (the reference's arithmetic with the caller's mu); guard: run only as the fallback leg
template <bool FWD>
void launch_literal(unsigned* d_a, unsigned n, const unsigned* d_tab, unsigned num, unsigned q, unsigned mu, int bits, hipStream_t s,
                    const unsigned* guard, bool skip_split_stage)
{
    if (n <= 32768) {
        launch_literal_lds<FWD>(d_a, n, d_tab, num, q, mu, bits, s, guard);
        return;
    }
    // n = 2^16: the stage that couples the halves in memory, every other stage out of LDS on the two halves (three launches instead
    // of sixteen: as the guarded fallback leg they are enqueued with every call)
    const unsigned halves = 2 * num, g = halves < 256u ? halves : 256u;
    if (FWD) {
        if (!skip_split_stage) launch_stage<true>(d_a, n, d_tab, 1, num, q, mu, bits, s, guard);
        k_ntt30<15, true><<<g, 1024, 0, s>>>(d_a, d_tab, q, mu, bits, halves, guard, 1u);
    } else {
        k_ntt30<15, false><<<g, 1024, 0, s>>>(d_a, d_tab, q, mu, bits, halves, guard, 1u);
        if (!skip_split_stage) launch_stage<false>(d_a, n, d_tab, 1, num, q, mu, bits, s, guard);
    }
}

template <bool FWD>
hipError_t run30(unsigned* d_a, unsigned n, const unsigned* d_tab, unsigned num, unsigned q, unsigned mu, int bits, unsigned ninv_native,
                 hipStream_t s)
{
    if (ninv_native == 0) {                               // literal kernels only (hand-made mu / Barrett-inexact modulus)
        launch_literal<FWD>(d_a, n, d_tab, num, q, mu, bits, s, nullptr, false);
        return hipGetLastError();
    }
    Scratch30* sc = scratch_for(s);
    if (!sc) return hipErrorOutOfMemory;
    // the scratch table belongs to the (device, stream) pair: one call's prepare -> transform -> fallback sequence must
    // reach the stream as a unit even when several host threads share the stream
    std::lock_guard<std::mutex> launch_lock(g_launch30_mutex);
    const unsigned epoch = next_epoch();
    const unsigned split = n == 65536 ? 1u : 0u;
    const unsigned m = split ? n / 2 : n, cnt = split ? 2 * num : num;
    // n = 2^16, large enough calls (kPair30MinPolys*) outside stream capture (the flags of the scratch belong to live launches of
    // this stream only): the stage that couples the halves rides in a pair launch -- in its loads (forward) / behind its last
    // round (inverse, which then scales by n^-1 = m^-1 / 2 itself)
    // The flags are the DEVICE's (kernels.hpp, pair_acquire), shared with the 60-bit pair kernel: at most one pair kernel of either
    // word size is in flight per device, on one stream -- a call on another stream while one is still running, on a capturing stream or on
    // a stream restricted to part of the CUs takes the stage launch below.
    unsigned pgrid = 0;
    PairSlot* slot = nullptr;
    if (split && num >= (FWD ? kPair30MinPolysFwd : kPair30MinPolysInv) && (pgrid = pair_grid(cnt)) != 0) {
        hipError_t pst = hipSuccess;
        if ((slot = pair_acquire(s, &pst)) == nullptr) pgrid = 0;
        if (pst != hipSuccess) return pst;                // (an earlier pair launch gave up on a partner: reported here, nothing launched)
    }
    static const bool trace = std::getenv("MI355NTT_TRACE30") != nullptr;
    if (trace) std::fprintf(stderr, "ntt30 %s n=%u num=%u stream=%p pair_grid=%u epoch=%u sc=%p\n", FWD ? "fwd" : "inv", n, num, (void*)s, pgrid, epoch, (void*)sc);
    const unsigned ninv_k = (pgrid && !FWD) ? (unsigned)(((u64)ninv_native * ((q + 1) / 2)) % q) : ninv_native;
    if (pgrid) {
        k_ntt30_prepare<<<64, 256, 0, s>>>(d_tab, n, q, ninv_k, split, FWD ? 1u : 0u, sc, epoch);
        launch_native_pair<FWD>(d_a, sc, q, cnt, pgrid, s, pair_flags(slot));
        pair_release(slot, s);
        launch_literal<FWD>(d_a, n, d_tab, num, q, mu, bits, s, sc->guard, false);              // fallback leg: every stage
        return hipGetLastError();
    }
    k_ntt30_prepare<<<64, 256, 0, s>>>(d_tab, n, q, ninv_k, split, FWD ? 1u : 0u, sc, epoch);
    if (split && FWD) launch_stage<true>(d_a, n, d_tab, 1, num, q, mu, bits, s, nullptr);       // stage 1 couples the two halves
    switch (m) {
    case 2048: launch_native<11, FWD>(d_a, sc, q, cnt, split, s); break;
    case 4096: launch_native<12, FWD>(d_a, sc, q, cnt, split, s); break;
    case 8192: launch_native<13, FWD>(d_a, sc, q, cnt, split, s); break;
    case 16384: launch_native<14, FWD>(d_a, sc, q, cnt, split, s); break;
    case 32768: launch_native<15, FWD>(d_a, sc, q, cnt, split, s); break;
    default: return hipErrorInvalidValue;
    }
    launch_literal<FWD>(d_a, n, d_tab, num, q, mu, bits, s, sc->guard, split != 0);             // fallback leg (a table entry >= q)
    if (split && !FWD) launch_stage<false>(d_a, n, d_tab, 1, num, q, mu, bits, s, nullptr);
    return hipGetLastError();
}

}  // namespace

// ninv_native: m^-1 mod q for the size m the native kernel transforms (n, or n / 2 at n = 2^16) when the call may take the
// native kernels (mu and bits are the canonical ones and the single-subtraction Barrett is exact for q), else 0.
hipError_t ntt30_forward(unsigned* d_a, unsigned n, const unsigned* d_tab, unsigned num, unsigned q, unsigned mu, int bits, unsigned ninv_native,
                         hipStream_t s)
{
    return run30<true>(d_a, n, d_tab, num, q, mu, bits, ninv_native, s);
}

hipError_t ntt30_inverse(unsigned* d_a, unsigned n, const unsigned* d_tab, unsigned num, unsigned q, unsigned mu, int bits, unsigned ninv_native,
                         hipStream_t s)
{
    return run30<false>(d_a, n, d_tab, num, q, mu, bits, ninv_native, s);
}

hipError_t ntt30_barrett(unsigned* d_a, const unsigned* d_b, size_t count, unsigned q, unsigned mu, int bits, hipStream_t s)
{
    const size_t blocks = (count + 255) / 256;
    k_barrett30<<<(unsigned)(blocks < 65536 ? blocks : 65536), 256, 0, s>>>(d_a, d_b, count, q, mu, bits);
    return hipGetLastError();
}

}  // namespace mi355ntt
