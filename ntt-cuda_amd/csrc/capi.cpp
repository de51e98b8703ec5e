// capi.cpp -- implementation of the C ABI declared in include/mi355ntt.h.
#include "../../include/mi355ntt.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "device_scope.hpp"
#include "hostparams.hpp"
#include "kernels.hpp"

using namespace mi355ntt;

static thread_local int g_last_hip_error = 0;

// shared with bfv_host.cpp so that mi355ntt_last_hip_error() covers every entry point (internal, not part of the ABI)
namespace mi355ntt {
void record_hip_error(int e) { g_last_hip_error = e; }
}

#define HIP_TRY(expr)                         \
    do {                                      \
        hipError_t e__ = (expr);              \
        if (e__ != hipSuccess) {              \
            g_last_hip_error = (int)e__;      \
            return MI355NTT_EHIP;             \
        }                                     \
    } while (0)

#define ON_CTX_DEVICE(c)                 \
    DeviceScope scope__((c)->device);    \
    HIP_TRY(scope__.err)

struct mi355ntt_ctx {
    unsigned n = 0;
    unsigned log_n = 0;
    unsigned num_primes = 0;
    int device = 0;
    PrimeParams prime[kMaxPrimes];
    ModSet mods;                 // Barrett constants, reference convention
    u64* d_psi = nullptr;        // [P][n]  psi^bitrev(i)      (reference format, demo.cu:188-196)
    u64* d_psiinv = nullptr;     // [P][n]  psi^-bitrev(i)
    FastTables fast;             // tables of the throughput kernels (kernels_fast.hip)
    // Barrett-inexact primes (hostparams.cpp, barrett_single_subtraction_exact) without MI355NTT_CTX_EXACT_ON_INEXACT_PRIMES: the
    // reference's single-subtraction Barrett leaves q + r now and then for such a modulus, so the transforms of this context must
    // return the reference's own words, not the exact transform's.  inexact_mask: bit i = prime i is such a prime.
    //   literal        -- inexact_mask != 0: every transform of the context runs literal arithmetic.  Up to n = 2^15 that is kernel
    //                     class HL_LIT of the single-pass kernels (round 6: one read and one write per transform, every prime of the
    //                     context -- for the exact ones the literal words ARE the exact transform's; no gather buffer, no per-prime
    //                     routing, nothing special under stream capture);
    //   literal_stages -- ... with an inexact prime narrower than 34 or wider than 61 bits, or at n = 2^16 with more than 8 primes: the
    //                     stage-per-launch kernels of kernels_compat.hip (n = 2^16 otherwise: `split16` below with class-0 tables);
    //   mixed          -- some, not all, primes inexact (reported by mi355ntt_ctx_uses_literal_kernels as 2; routing as `literal`).
    bool literal = false, literal_stages = false, mixed = false;
    unsigned inexact_mask = 0;
    // n = 2^16 (beyond the reference's dispatch): stage 1 splits the transform into two independent half-size ones whose
    // stage `L` reads table entries [2L + h L, 2L + (h + 1) L) -- an ordinary 2^15 transform on a derived table.  `fast`
    // then holds 2 P "virtual primes" (2 i + h) for n/2, polynomial y's half h is virtual polynomial 2 y + h.
    bool split16 = false;
};

static ModSet mods_from(const mi355ntt_ctx* c, unsigned base, unsigned division)
{
    ModSet r;
    std::memset(&r, 0, sizeof(r));
    for (unsigned i = 0; i < division && base + i < kMaxPrimes; i++) {
        r.q[i] = c->mods.q[base + i];
        r.mu[i] = c->mods.mu[base + i];
        r.k[i] = c->mods.k[base + i];
    }
    return r;
}

// `base`: first prime of the call; checked raw calls carry kGuardBit in it (kernels.hpp) -- the throughput kernels strip it, the
// table / modulus indexing of the stage-launch legs below uses the plain index pb.
static hipError_t run_forward(const mi355ntt_ctx* c, u64* d_a, unsigned num, unsigned division, unsigned base, hipStream_t s)
{
    const unsigned pb = base & ~kGuardBit;
    if (c->literal_stages) return compat_forward_batch(d_a, c->n, c->d_psi + (size_t)pb * c->n, num, division, mods_from(c, pb, division), s);
    if (c->split16) {
        // large batches: the coupling stage rides in the loads of the lower halves' launch (1.5 passes over memory instead of 2)
        if (fast_forward_split16_ok(c->fast, num)) return fast_forward_split16(c->fast, d_a, num, division, base, s);
        hipError_t e = compat_ct_stage(d_a, c->n, c->d_psi + (size_t)pb * c->n, 1, num, division, mods_from(c, pb, division), s);
        if (e != hipSuccess) return e;
        return fast_forward_batch(c->fast, d_a, 2 * num, 2 * division, 2 * base, s);
    }
    return fast_forward_batch(c->fast, d_a, num, division, base, s);      // (literal contexts up to n = 2^15: kernel class HL_LIT)
}

static hipError_t run_inverse(const mi355ntt_ctx* c, u64* d_a, unsigned num, unsigned division, unsigned base, hipStream_t s)
{
    const unsigned pb = base & ~kGuardBit;
    if (c->literal_stages) return compat_inverse_batch(d_a, c->n, c->d_psiinv + (size_t)pb * c->n, num, division, mods_from(c, pb, division), s);
    if (c->split16) {
        // large batches: the coupling stage rides behind the lower halves' last round (1.5 passes over memory instead of 2)
        if (fast_inverse_split16_ok(c->fast, num, false)) return fast_inverse_split16(c->fast, d_a, num, division, base, s);
        hipError_t e = fast_inverse_batch(c->fast, d_a, 2 * num, 2 * division, 2 * base, s);
        if (e != hipSuccess) return e;
        return compat_gs_stage(d_a, c->n, c->d_psiinv + (size_t)pb * c->n, 1, num, division, mods_from(c, pb, division), s);
    }
    return fast_inverse_batch(c->fast, d_a, num, division, base, s);
}

// The fused product with an element-wise epilogue for the batched BFV drivers (bfv_host.cpp; kernels_epi.cuh): MI355NTT_OK, a negative
// error, or 1 = "not on this context / batch": the caller then runs the product and its element-wise kernel one after the other.
namespace mi355ntt {
int ctx_polymul_epi(const mi355ntt_ctx* c, int kind, u64* d_a, const u64* d_bhat, unsigned num, unsigned division, unsigned group,
                    const u64* d_other, const void* d_consts, hipStream_t s)
{
    if (!c || !d_a || !d_bhat || !d_other || !d_consts || division == 0 || division > c->num_primes) return MI355NTT_EINVAL;
    if (c->literal || c->split16 || !fast_polymul_epi_ok(c->fast, num, division)) return 1;
    const hipError_t e = fast_polymul_batch_epi(c->fast, kind, d_a, d_bhat, num, division, s, true, group, d_other, d_consts);
    if (e == hipErrorNotSupported) return 1;
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return MI355NTT_EHIP;
    }
    return MI355NTT_OK;
}
}  // namespace mi355ntt

static bool is_pow2(unsigned n) { return n && !(n & (n - 1)); }

static int check_n(unsigned n)
{
    if (!is_pow2(n) || n < 2048 || n > 65536) return MI355NTT_EUNSUPPORTED;
    return MI355NTT_OK;
}

extern "C" {

const char* mi355ntt_strerror(int code)
{
    switch (code) {
    case MI355NTT_OK: return "ok";
    case MI355NTT_EINVAL: return "invalid argument";
    case MI355NTT_EUNSUPPORTED: return "unsupported ring degree, prime count or modulus size";
    case MI355NTT_EHIP: return "HIP runtime error";
    case MI355NTT_ENOMEM: return "out of memory";
    case MI355NTT_EPARAM: return "inconsistent NTT parameters (q, psi, n)";
    default: return "unknown error";
    }
}

int mi355ntt_last_hip_error(void) { return g_last_hip_error; }
unsigned long long mi355ntt_pair_fault_count(int device) { return pair_fault_count(device); }
const char* mi355ntt_version(void) { return "mi355ntt 0.1 (gfx950)"; }

/* ---------------- host-only helpers ---------------- */
unsigned mi355ntt_bit_length(mi355ntt_u64 q) { return bit_length(q); }
mi355ntt_u64 mi355ntt_barrett_mu(mi355ntt_u64 q, unsigned k) { return (k == 0 || k > 63 || q == 0) ? 0 : barrett_mu(q, k); }
int mi355ntt_barrett_is_exact(mi355ntt_u64 q)
{
    const unsigned k = bit_length(q);
    return (k >= 3 && k <= 62 && barrett_single_subtraction_exact(q, k, barrett_mu(q, k))) ? 1 : 0;
}
mi355ntt_u64 mi355ntt_mulmod(mi355ntt_u64 a, mi355ntt_u64 b, mi355ntt_u64 m) { return m ? mulmod(a, b, m) : 0; }
mi355ntt_u64 mi355ntt_modpow(mi355ntt_u64 a, mi355ntt_u64 e, mi355ntt_u64 m) { return m ? modpow(a, e, m) : 0; }
mi355ntt_u64 mi355ntt_modinv(mi355ntt_u64 a, mi355ntt_u64 q) { return q > 2 ? modinv(a, q) : 0; }
mi355ntt_u64 mi355ntt_bit_reverse(mi355ntt_u64 a, int bits) { return bit_reverse(a, bits); }

int mi355ntt_fill_tables(mi355ntt_u64 psi, mi355ntt_u64 psiinv, mi355ntt_u64 q, unsigned n, mi355ntt_u64* tp, mi355ntt_u64* ti)
{
    if (!is_pow2(n) || q < 2) return MI355NTT_EINVAL;
    if (tp) fill_table(psi, q, n, tp);
    if (ti) fill_table(psiinv, q, n, ti);
    return MI355NTT_OK;
}

int mi355ntt_get_params(unsigned n, mi355ntt_u64* q, mi355ntt_u64* psi, mi355ntt_u64* psiinv, mi355ntt_u64* ninv,
                        unsigned* bits)
{
    // parameter.h:31-79 (active sets)
    struct Row { unsigned n; u64 q, psi, psiinv, ninv; unsigned bits; };
    static const Row rows[] = {
        {2048, 137438691329ULL, 22157790ULL, 88431458764ULL, 137371582593ULL, 37},
        {4096, 33538049ULL, 2386ULL, 26102329ULL, 33529861ULL, 25},
        {8192, 8796092858369ULL, 1734247217ULL, 5727406356888ULL, 8795019116565ULL, 43},
        {16384, 281474976546817ULL, 23720796222ULL, 129310633907832ULL, 281457796677643ULL, 48},
        {32768, 36028797017456641ULL, 1155186985540ULL, 31335194304461613ULL, 36027697505828911ULL, 55},
    };
    for (const Row& r : rows)
        if (r.n == n) {
            if (q) *q = r.q;
            if (psi) *psi = r.psi;
            if (psiinv) *psiinv = r.psiinv;
            if (ninv) *ninv = r.ninv;
            if (bits) *bits = r.bits;
            return MI355NTT_OK;
        }
    return MI355NTT_EUNSUPPORTED;
}

/* ---------------- context ---------------- */
int mi355ntt_ctx_create(mi355ntt_ctx** out, unsigned n, unsigned num_primes, const mi355ntt_u64* q, const mi355ntt_u64* psi,
                        int device)
{
    return mi355ntt_ctx_create_ex(out, n, num_primes, q, psi, device, 0);
}

int mi355ntt_ctx_uses_literal_kernels(const mi355ntt_ctx* c) { return !c ? 0 : c->mixed ? 2 : c->literal ? 1 : 0; }
int mi355ntt_ctx_kernel_class(const mi355ntt_ctx* c) { return !c ? MI355NTT_EINVAL : c->fast.hl; }

int mi355ntt_ctx_create_ex(mi355ntt_ctx** out, unsigned n, unsigned num_primes, const mi355ntt_u64* q, const mi355ntt_u64* psi,
                           int device, unsigned flags)
{
    if (!out || !q || !psi) return MI355NTT_EINVAL;
    *out = nullptr;
    int rc = check_n(n);
    if (rc) return rc;
    if (num_primes == 0 || num_primes > kMaxPrimes) return MI355NTT_EUNSUPPORTED;

    mi355ntt_ctx* c = new (std::nothrow) mi355ntt_ctx();
    if (!c) return MI355NTT_ENOMEM;
    c->n = n;
    c->num_primes = num_primes;
    c->device = device;
    while ((1u << c->log_n) < n) c->log_n++;
    std::memset(&c->mods, 0, sizeof(c->mods));
    for (unsigned i = 0; i < num_primes; i++) {
        rc = derive_prime(n, q[i], psi[i], &c->prime[i]);
        if (rc) {
            delete c;
            return rc;
        }
        c->mods.q[i] = c->prime[i].q;
        c->mods.mu[i] = c->prime[i].mu;
        c->mods.k[i] = c->prime[i].k;
        if (!c->prime[i].barrett_exact && !(flags & MI355NTT_CTX_EXACT_ON_INEXACT_PRIMES)) c->inexact_mask |= 1u << i;
    }
    if (c->inexact_mask) {
        c->literal = true;
        // the class-0 kernels serve inexact moduli of 34 ... 61 bits (ntt_core.cuh, lit_barrett_mul: both 128-bit shifts of
        // singleBarrett as funnel shifts of 32-bit words) -- n = 2^16 as the literal coupling stage in memory around two half-size
        // class-0 transforms on the derived tables (up to 8 primes, as for the lazy classes); anything else keeps the
        // stage-per-launch kernels
        c->literal_stages = (n == 65536 && 2 * num_primes > kMaxPrimes);
        for (unsigned i = 0; i < num_primes; i++)
            if (((c->inexact_mask >> i) & 1u) && (c->prime[i].k < 34 || c->prime[i].k > 61)) c->literal_stages = true;
        c->mixed = !c->literal_stages && c->inexact_mask != (num_primes >= 32 ? ~0u : (1u << num_primes) - 1u);
    }

    auto fail = [&](int code) {
        mi355ntt_ctx_destroy(c);
        return code;
    };
    DeviceScope scope(device);           // the caller's current device is restored on every return path
    hipError_t e = scope.err;
    if (e != hipSuccess) { g_last_hip_error = (int)e; return fail(MI355NTT_EHIP); }

    size_t words = (size_t)num_primes * n;
    std::vector<u64> hp(words), hi(words);
    for (unsigned i = 0; i < num_primes; i++) {
        fill_table(c->prime[i].psi, c->prime[i].q, n, hp.data() + (size_t)i * n);
        fill_table(c->prime[i].psiinv, c->prime[i].q, n, hi.data() + (size_t)i * n);
    }
    if ((e = hipMalloc((void**)&c->d_psi, words * sizeof(u64))) != hipSuccess ||
        (e = hipMalloc((void**)&c->d_psiinv, words * sizeof(u64))) != hipSuccess ||
        (e = hipMemcpy(c->d_psi, hp.data(), words * sizeof(u64), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(c->d_psiinv, hi.data(), words * sizeof(u64), hipMemcpyHostToDevice)) != hipSuccess) {
        g_last_hip_error = (int)e;
        return fail(e == hipErrorOutOfMemory ? MI355NTT_ENOMEM : MI355NTT_EHIP);
    }
    if (n == 65536 && !c->literal_stages && 2 * num_primes <= kMaxPrimes) {
        const unsigned h_n = n / 2;
        std::vector<u64> vp((size_t)2 * num_primes * h_n), vi((size_t)2 * num_primes * h_n);
        PrimeParams vprime[kMaxPrimes];
        u64 split_fwd[kMaxPrimes], split_inv[kMaxPrimes];
        for (unsigned i = 0; i < num_primes; i++)
            for (unsigned h = 0; h < 2; h++) {
                const unsigned v = 2 * i + h;
                vprime[v] = c->prime[i];
                // psi^bitrev(1): stage 1 of the full-size transform (k_forward15 SPLIT reads the lower half's); negated for the
                // upper half, which forms U - V w as U + V (-w) (k_forward15_pair)
                split_fwd[v] = h ? c->prime[i].q - hp[(size_t)i * n + 1] : hp[(size_t)i * n + 1];
                split_inv[v] = hi[(size_t)i * n + 1];                               // psi^-bitrev(1): its last GS stage
                vprime[v].ninv = modinv(h_n % c->prime[i].q, c->prime[i].q);       // the half-size transform scales by (n/2)^-1 ...
                u64* tp = vp.data() + (size_t)v * h_n;                              // ... and the last GS stage halves once more
                u64* ti = vi.data() + (size_t)v * h_n;
                tp[0] = ti[0] = 1;
                for (unsigned L = 1; L < h_n; L *= 2)
                    for (unsigned p = 0; p < L; p++) {
                        tp[L + p] = hp[(size_t)i * n + 2 * L + h * L + p];
                        ti[L + p] = hi[(size_t)i * n + 2 * L + h * L + p];
                    }
            }
        e = fast_tables_create(&c->fast, h_n, 2 * num_primes, vprime, vp.data(), vi.data(), nullptr, nullptr, split_fwd, split_inv, c->literal);
        c->split16 = (e == hipSuccess);
    } else {
        e = fast_tables_create(&c->fast, n, num_primes, c->prime, hp.data(), hi.data(), c->d_psi, c->d_psiinv, nullptr, nullptr,
                               c->literal && !c->literal_stages);
    }
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return fail(e == hipErrorOutOfMemory ? MI355NTT_ENOMEM : MI355NTT_EHIP);
    }
    *out = c;
    return MI355NTT_OK;
}

int mi355ntt_ctx_destroy(mi355ntt_ctx* c)
{
    if (!c) return MI355NTT_OK;
    DeviceScope scope(c->device);
    if (c->d_psi) (void)hipFree(c->d_psi);
    if (c->d_psiinv) (void)hipFree(c->d_psiinv);
    fast_tables_destroy(&c->fast);
    delete c;
    return MI355NTT_OK;
}

unsigned mi355ntt_ctx_n(const mi355ntt_ctx* c) { return c ? c->n : 0; }
unsigned mi355ntt_ctx_num_primes(const mi355ntt_ctx* c) { return c ? c->num_primes : 0; }
int mi355ntt_ctx_device(const mi355ntt_ctx* c) { return c ? c->device : -1; }

int mi355ntt_ctx_prime(const mi355ntt_ctx* c, unsigned i, mi355ntt_u64* q, mi355ntt_u64* mu, unsigned* bits,
                       mi355ntt_u64* psi, mi355ntt_u64* psiinv)
{
    if (!c || i >= c->num_primes) return MI355NTT_EINVAL;
    if (q) *q = c->prime[i].q;
    if (mu) *mu = c->prime[i].mu;
    if (bits) *bits = c->prime[i].k;
    if (psi) *psi = c->prime[i].psi;
    if (psiinv) *psiinv = c->prime[i].psiinv;
    return MI355NTT_OK;
}

const mi355ntt_u64* mi355ntt_ctx_psi_tables(const mi355ntt_ctx* c) { return c ? c->d_psi : nullptr; }
const mi355ntt_u64* mi355ntt_ctx_psiinv_tables(const mi355ntt_ctx* c) { return c ? c->d_psiinv : nullptr; }

/* ---------------- transforms on a context ---------------- */
static int check_batch(const mi355ntt_ctx* c, const void* p, unsigned num, unsigned division)
{
    if (!c || !p) return MI355NTT_EINVAL;
    if (division == 0 || division > c->num_primes) return MI355NTT_EINVAL;
    (void)num;
    return MI355NTT_OK;
}

int mi355ntt_forward_batch(const mi355ntt_ctx* c, mi355ntt_u64* d_a, unsigned num, unsigned division, mi355ntt_stream s)
{
    int rc = check_batch(c, d_a, num, division);
    if (rc) return rc;
    if (num == 0) return MI355NTT_OK;
    ON_CTX_DEVICE(c);
    HIP_TRY(run_forward(c, d_a, num, division, 0, (hipStream_t)s));
    return MI355NTT_OK;
}

int mi355ntt_inverse_batch(const mi355ntt_ctx* c, mi355ntt_u64* d_a, unsigned num, unsigned division, mi355ntt_stream s)
{
    int rc = check_batch(c, d_a, num, division);
    if (rc) return rc;
    if (num == 0) return MI355NTT_OK;
    ON_CTX_DEVICE(c);
    HIP_TRY(run_inverse(c, d_a, num, division, 0, (hipStream_t)s));
    return MI355NTT_OK;
}

int mi355ntt_forward(const mi355ntt_ctx* c, mi355ntt_u64* d_a, unsigned prime_idx, mi355ntt_stream s)
{
    if (!c || !d_a || prime_idx >= c->num_primes) return MI355NTT_EINVAL;
    ON_CTX_DEVICE(c);
    HIP_TRY(run_forward(c, d_a, 1, 1, prime_idx, (hipStream_t)s));
    return MI355NTT_OK;
}

int mi355ntt_inverse(const mi355ntt_ctx* c, mi355ntt_u64* d_a, unsigned prime_idx, mi355ntt_stream s)
{
    if (!c || !d_a || prime_idx >= c->num_primes) return MI355NTT_EINVAL;
    ON_CTX_DEVICE(c);
    HIP_TRY(run_inverse(c, d_a, 1, 1, prime_idx, (hipStream_t)s));
    return MI355NTT_OK;
}

int mi355ntt_forward_double(const mi355ntt_ctx* c, mi355ntt_u64* d_a, mi355ntt_u64* d_b, unsigned prime_idx,
                            mi355ntt_stream s1, mi355ntt_stream s2)
{
    int rc = mi355ntt_forward(c, d_a, prime_idx, s1);
    if (rc) return rc;
    return mi355ntt_forward(c, d_b, prime_idx, s2);
}

int mi355ntt_pointwise_mul(const mi355ntt_ctx* c, mi355ntt_u64* d_c, const mi355ntt_u64* d_a, const mi355ntt_u64* d_b,
                           unsigned num, unsigned division, mi355ntt_stream s)
{
    int rc = check_batch(c, d_c, num, division);
    if (rc) return rc;
    if (!d_a || !d_b) return MI355NTT_EINVAL;
    if (num == 0) return MI355NTT_OK;
    ON_CTX_DEVICE(c);
    if (c->split16) HIP_TRY(compat_pointwise(d_c, d_a, d_b, c->n, num, division, mods_from(c, 0, division), (hipStream_t)s));
    else HIP_TRY(fast_pointwise(c->fast, d_c, d_a, d_b, num, division, (hipStream_t)s));
    return MI355NTT_OK;
}

int mi355ntt_pointwise_mul_scalar(const mi355ntt_ctx* c, mi355ntt_u64* d_a, mi355ntt_u64 b, unsigned prime_idx,
                                  mi355ntt_stream s)
{
    if (!c || !d_a || prime_idx >= c->num_primes) return MI355NTT_EINVAL;
    const PrimeParams& p = c->prime[prime_idx];
    ON_CTX_DEVICE(c);
    HIP_TRY(compat_pointwise_scalar(d_a, b, c->n, p.q, p.mu, p.k, (hipStream_t)s));
    return MI355NTT_OK;
}

int mi355ntt_polymul_batch(const mi355ntt_ctx* c, mi355ntt_u64* d_a, const mi355ntt_u64* d_bhat, unsigned num,
                           unsigned division, mi355ntt_stream s)
{
    int rc = check_batch(c, d_a, num, division);
    if (rc) return rc;
    if (!d_bhat) return MI355NTT_EINVAL;
    if (num == 0) return MI355NTT_OK;
    ON_CTX_DEVICE(c);
    if (c->split16 && fast_inverse_split16_ok(c->fast, num, true)) {      // n = 2^16, large batch: the product rides in the inverse launch's loads
        HIP_TRY(fast_forward_split16(c->fast, d_a, num, division, 0, (hipStream_t)s));
        HIP_TRY(fast_inverse_split16(c->fast, d_a, num, division, 0, (hipStream_t)s, d_bhat));
        return MI355NTT_OK;
    }
    if (c->literal_stages || c->split16) {   // the reference's own sequence (bfv_encryption.cuh:268-271), three calls
        HIP_TRY(run_forward(c, d_a, num, division, 0, (hipStream_t)s));
        HIP_TRY(compat_pointwise(d_a, d_a, d_bhat, c->n, num, division, mods_from(c, 0, division), (hipStream_t)s));
        HIP_TRY(run_inverse(c, d_a, num, division, 0, (hipStream_t)s));
        return MI355NTT_OK;
    }
    HIP_TRY(fast_polymul_batch(c->fast, d_a, d_bhat, num, division, (hipStream_t)s));
    return MI355NTT_OK;
}

int mi355ntt_polymul_batch_shared(const mi355ntt_ctx* c, mi355ntt_u64* d_a, const mi355ntt_u64* d_bhat, unsigned num,
                                  unsigned division, unsigned group, mi355ntt_stream s)
{
    int rc = check_batch(c, d_a, num, division);
    if (rc) return rc;
    if (!d_bhat || (group && group % division) || group >= (1u << 23)) return MI355NTT_EINVAL;
    if (num == 0) return MI355NTT_OK;
    ON_CTX_DEVICE(c);
    if (!c->literal_stages && !c->split16) {
        const hipError_t e = fast_polymul_batch(c->fast, d_a, d_bhat, num, division, (hipStream_t)s, true, group);
        if (e == hipSuccess) return MI355NTT_OK;
        if (e != hipErrorNotSupported) HIP_TRY(e);
    }
    // the three steps (literal kernels, n = 65536, and the sizes without a fused kernel)
    HIP_TRY(run_forward(c, d_a, num, division, 0, (hipStream_t)s));
    HIP_TRY(compat_pointwise(d_a, d_a, d_bhat, c->n, num, division, mods_from(c, 0, division), (hipStream_t)s, true, group));
    HIP_TRY(run_inverse(c, d_a, num, division, 0, (hipStream_t)s));
    return MI355NTT_OK;
}

/* ---------------- measurement helpers ---------------- */
/* synthetic inputs of SURVEY.md 4.2 / 8d on the context's device: polynomial y = splitmix64(seed_base + y) mod q[y % division] */
int mi355ntt_synth_splitmix(const mi355ntt_ctx* c, mi355ntt_u64* d_a, unsigned num, unsigned division, mi355ntt_u64 seed_base,
                            mi355ntt_stream s)
{
    int rc = check_batch(c, d_a, num, division);
    if (rc) return rc;
    ON_CTX_DEVICE(c);
    HIP_TRY(compat_synth_splitmix(d_a, c->n, num, division, mods_from(c, 0, division), seed_base, (hipStream_t)s));
    return MI355NTT_OK;
}

/* Clock probe: mi355ntt_ctx_clock_probe enqueues a one-wave kernel on `stream` that counts shader cycles (s_memtime) over 20 us
 * of the 100 MHz constant clock (s_memrealtime); mi355ntt_ctx_probed_clock_mhz returns the last probe's result -- the shader clock
 * the work enqueued in front of the probe left the chip at.  The second call synchronises the device (a host read).  *mhz = 0
 * until a probe has run. */
int mi355ntt_ctx_clock_probe(const mi355ntt_ctx* c, mi355ntt_stream s)
{
    if (!c) return MI355NTT_EINVAL;
    ON_CTX_DEVICE(c);
    HIP_TRY(fast_clock_probe(c->fast, (hipStream_t)s));
    return MI355NTT_OK;
}

int mi355ntt_ctx_occupy(const mi355ntt_ctx* c, unsigned workgroups, unsigned microseconds, mi355ntt_stream s)
{
    if (!c || workgroups > 65535u || microseconds > 10000000u) return MI355NTT_EINVAL;
    ON_CTX_DEVICE(c);
    HIP_TRY(fast_occupy(workgroups, microseconds, (hipStream_t)s));
    return MI355NTT_OK;
}

int mi355ntt_ctx_probed_clock_mhz(const mi355ntt_ctx* c, double* mhz)
{
    if (!c || !mhz) return MI355NTT_EINVAL;
    ON_CTX_DEVICE(c);
    HIP_TRY(fast_probed_clock_mhz(c->fast, mhz));
    return MI355NTT_OK;
}

/* ---------------- the element-wise wrappers of poly_arithmetic.cuh:312-352 ---------------- */
static int elementwise(int op, mi355ntt_u64* d_a, const mi355ntt_u64* d_b, bool need_b, mi355ntt_u64 scalar, mi355ntt_u64 q, unsigned n, mi355ntt_stream s)
{
    if (!d_a || (need_b && !d_b) || q == 0) return MI355NTT_EINVAL;
    if (((uintptr_t)d_a & 7u) || (need_b && ((uintptr_t)d_b & 7u))) return MI355NTT_EINVAL;        /* words; 16-byte aligned pointers take the 16-byte kernel */
    (void)hipGetLastError();
    HIP_TRY(compat_elementwise(op, d_a, d_b, scalar, q, n, (hipStream_t)s));
    return MI355NTT_OK;
}
int mi355ntt_poly_add_raw(mi355ntt_u64* d_a, const mi355ntt_u64* d_b, unsigned n, mi355ntt_stream s, mi355ntt_u64 q)
{
    return elementwise(kEwAdd, d_a, d_b, true, 0, q, n, s);
}
int mi355ntt_poly_sub_raw(mi355ntt_u64* d_a, const mi355ntt_u64* d_b, unsigned n, mi355ntt_stream s, mi355ntt_u64 q)
{
    return elementwise(kEwSub, d_a, d_b, true, 0, q, n, s);
}
int mi355ntt_poly_negate_raw(mi355ntt_u64* d_a, unsigned n, mi355ntt_stream s, mi355ntt_u64 q)
{
    return elementwise(kEwNegate, d_a, nullptr, false, 0, q, n, s);
}
int mi355ntt_poly_add_integer_raw(mi355ntt_u64* d_a, mi355ntt_u64 b, unsigned n, mi355ntt_stream s, mi355ntt_u64 q)
{
    return elementwise(kEwAddInteger, d_a, nullptr, false, b, q, n, s);
}
int mi355ntt_poly_mul_int_t_raw(mi355ntt_u64* d_a, mi355ntt_u64 b, unsigned n, mi355ntt_stream s, mi355ntt_u64 t)
{
    return elementwise(kEwMulIntT, d_a, nullptr, false, b, t, n, s);
}

/* ---------------- raw-parameter entry points ---------------- */
static int fill_modset(ModSet* m, unsigned division, const mi355ntt_u64* q, const mi355ntt_u64* mu, const unsigned* bits)
{
    if (!q || !mu || !bits || division == 0 || division > kMaxPrimes) return MI355NTT_EINVAL;
    std::memset(m, 0, sizeof(*m));
    for (unsigned i = 0; i < division; i++) {
        if (bits[i] < 3 || bits[i] > 62 || q[i] == 0) return MI355NTT_EUNSUPPORTED;
        m->q[i] = q[i];
        m->mu[i] = mu[i];
        m->k[i] = bits[i];
    }
    return MI355NTT_OK;
}

static int check_raw_n(unsigned n) { return (is_pow2(n) && n >= 2 && n <= (1u << 20)) ? MI355NTT_OK : MI355NTT_EUNSUPPORTED; }

}  // extern "C" (reopened below)

// The reference's launchers take (q, mu, bit_length, device table) on every call (ntt_60bit.cuh:314,350,608,652 with the
// __constant__ moduli of :8-10).  A caller who passes what the reference's own bootstrap computes -- mu = floor(2^(2k)/q),
// bits = bitlen(q), tables = fillTablePsi128(psi) (60bit_ntt_test.cu:47-49, demo.cu:69,188-196) -- with moduli for which the
// reference's single-subtraction Barrett is exact gets, word for word, the exact transform: those calls run the throughput
// kernels on a context the library derives once per (device, n, moduli, table address) and keeps.  Anything else (a
// hand-made mu, a table that is not psi^bitrev(i), a Barrett-inexact modulus) runs the literal kernels, which follow the
// caller's numbers step by step.  First sight of a table costs two small synchronous copies and one compare kernel;
// the table is assumed not to be rewritten at the same address afterwards (mi355ntt_raw_cache_clear drops everything,
// MI355NTT_RAW_LITERAL=1 in the environment disables the routing).
namespace {

struct RawStreamSlot {
    hipStream_t stream = nullptr;
    void* d_primes_alloc = nullptr;    // (num_primes + 1) PrimeDev records: [0] = guard record, [1..] = the context's
};
constexpr size_t kRawStreamSlots = 16;

struct RawEntry {
    int device = 0;
    unsigned n = 0, division = 0;
    bool inverse = false;
    const u64* tab = nullptr;
    u64 q[kMaxPrimes], mu[kMaxPrimes];
    unsigned bits[kMaxPrimes];
    mi355ntt_ctx* ctx = nullptr;       // null: literal kernels
    unsigned long long stamp = 0;
    // checked mode (default): the table is compared with the context's in front of every transform (guard words, kernels.hpp)
    bool trusted = false;              // the caller promised not to rewrite the table: no per-call comparison
    unsigned epoch = 0;
    // The guard words of the checked calls belong to ONE stream: every stream that calls with this entry gets a copy of the context's
    // PrimeDev array with a guard record of its own in front (a few KB; the tables are shared, they are read-only).  Nothing is handed
    // from stream to stream, so no call ever touches a stream other than the one it was given -- which the caller may have destroyed
    // by then (until round 6 an event was recorded on the previous caller's stream: a use-after-free when that caller's thread had
    // finished, found by tests/cpp/threads_test.cpp).  Streams beyond kRawStreamSlots run the literal kernels (they share nothing).
    std::vector<RawStreamSlot> slots;
    // host-mapped word the comparison kernel sets when the table no longer holds what the context was derived from: the next call
    // derives a new context (twice; an entry whose table keeps changing then stays on the literal kernels)
    volatile unsigned* h_changed = nullptr;
    unsigned* d_changed = nullptr;
    unsigned rederived = 0;
};

std::mutex g_raw_mutex;
std::vector<RawEntry> g_raw_cache;
unsigned long long g_raw_clock = 0;
constexpr size_t kRawCacheMax = 32;

bool raw_routing_enabled()
{
    static const bool on = [] {
        const char* e = std::getenv("MI355NTT_RAW_LITERAL");
        return !(e && e[0] && e[0] != '0');
    }();
    return on;
}

bool raw_key_equal(const RawEntry& e, int device, unsigned n, unsigned division, bool inverse, const u64* tab, const u64* q, const u64* mu,
                   const unsigned* bits)
{
    if (e.device != device || e.n != n || e.division != division || e.inverse != inverse || e.tab != tab) return false;
    for (unsigned i = 0; i < division; i++)
        if (e.q[i] != q[i] || e.mu[i] != mu[i] || e.bits[i] != bits[i]) return false;
    return true;
}

// Derive a context from what the caller passed, or return null when the call has to follow the caller's numbers literally.
mi355ntt_ctx* raw_derive(int device, unsigned n, unsigned division, bool inverse, const u64* d_tab, const u64* q, const u64* mu,
                         const unsigned* bits)
{
    if (check_n(n) != MI355NTT_OK || ((uintptr_t)d_tab & 15u) != 0) return nullptr;
    u64 root[kMaxPrimes];
    for (unsigned i = 0; i < division; i++) {
        if (q[i] < 3 || !(q[i] & 1) || bits[i] != bit_length(q[i]) || mu[i] != barrett_mu(q[i], bits[i])) return nullptr;
        // (a Barrett-inexact modulus does not send the call to the literal stage kernels: the derived context is a class-0 one --
        // the reference's own decryption_test.cu:47-48 set through forwardNTT_batch runs single-pass kernels on all three primes)
        // entry n/2 of a table is root^bitrev(n/2) = root^1
        if (hipMemcpy(&root[i], d_tab + (size_t)i * n + n / 2, sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
        if (root[i] == 0 || root[i] >= q[i]) return nullptr;
        if (inverse) root[i] = modinv(root[i], q[i]);
    }
    mi355ntt_ctx* c = nullptr;
    if (mi355ntt_ctx_create_ex(&c, n, division, q, root, device, 0) != MI355NTT_OK || !c) return nullptr;
    bool same = false;
    if (!c->literal_stages) {
        unsigned* d_flag = nullptr;
        unsigned h_flag = 1;
        if (hipMalloc((void**)&d_flag, sizeof(unsigned)) == hipSuccess) {
            if (hipMemset(d_flag, 0, sizeof(unsigned)) == hipSuccess &&
                compat_tables_differ(d_tab, inverse ? c->d_psiinv : c->d_psi, n, division, d_flag, nullptr) == hipSuccess &&
                hipMemcpy(&h_flag, d_flag, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess)
                same = (h_flag == 0);
            (void)hipFree(d_flag);
        }
    }
    if (!same) {
        mi355ntt_ctx_destroy(c);
        return nullptr;
    }
    return c;
}

void raw_slots_release(RawEntry& e)
{
    for (RawStreamSlot& sl : e.slots)
        if (sl.d_primes_alloc) (void)hipFree(sl.d_primes_alloc);       // (hipFree waits for the launches that still read it)
    e.slots.clear();
}

// The guard record + PrimeDev copy of entry e for stream s (g_raw_mutex held); null: no slot left or no memory -- the call then runs
// the literal kernels.  Filled on s itself, in front of the first launch that reads it.
void* raw_slot_for(RawEntry& e, hipStream_t s)
{
    for (RawStreamSlot& sl : e.slots)
        if (sl.stream == s) return sl.d_primes_alloc;
    if (e.slots.size() >= kRawStreamSlots) return nullptr;
    const mi355ntt_ctx* c = e.ctx;
    const size_t rec = fast_prime_record_bytes(), bytes = (size_t)(c->num_primes + 1) * rec;
    void* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemsetAsync(d, 0, rec, s) != hipSuccess ||
        hipMemcpyAsync(static_cast<char*>(d) + rec, c->fast.d_primes, bytes - rec, hipMemcpyDeviceToDevice, s) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(d);
        return nullptr;
    }
    RawStreamSlot sl;
    sl.stream = s;
    sl.d_primes_alloc = d;
    e.slots.push_back(sl);
    return d;
}

void raw_entry_release(RawEntry& e)
{
    raw_slots_release(e);
    if (e.ctx) mi355ntt_ctx_destroy(e.ctx);
    if (e.h_changed) (void)hipHostFree(const_cast<unsigned*>(e.h_changed));
    e.ctx = nullptr;
    e.h_changed = nullptr;
    e.d_changed = nullptr;
}

// The cache entry of this raw call (g_raw_mutex held by the caller); entry->ctx is the context that serves it with the
// throughput kernels, or null (literal kernels).  Null when the routing is off.
// `s` / `have_stream`: the stream the caller's table was (possibly asynchronously) written on -- first sight reads the table
// from the host, so that stream is synchronised first (a table still being filled on a non-blocking stream would otherwise
// fail the comparison and pin the entry to the literal kernels for good); without a stream the whole device is.
RawEntry* raw_lookup(unsigned n, unsigned division, bool inverse, const u64* d_tab, const u64* q, const u64* mu, const unsigned* bits,
                     hipStream_t s = nullptr, bool have_stream = false)
{
    if (!raw_routing_enabled()) return nullptr;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    for (RawEntry& e : g_raw_cache)
        if (raw_key_equal(e, device, n, division, inverse, d_tab, q, mu, bits)) {
            e.stamp = ++g_raw_clock;
            return &e;
        }
    RawEntry e;
    e.device = device; e.n = n; e.division = division; e.inverse = inverse; e.tab = d_tab;
    for (unsigned i = 0; i < division; i++) { e.q[i] = q[i]; e.mu[i] = mu[i]; e.bits[i] = bits[i]; }
    if (have_stream) (void)hipStreamSynchronize(s);
    else (void)hipDeviceSynchronize();
    e.ctx = raw_derive(device, n, division, inverse, d_tab, q, mu, bits);
    (void)hipGetLastError();
    e.stamp = ++g_raw_clock;
    if (g_raw_cache.size() >= kRawCacheMax) {          // evict the least recently used entry
        size_t victim = 0;
        for (size_t i = 1; i < g_raw_cache.size(); i++)
            if (g_raw_cache[i].stamp < g_raw_cache[victim].stamp) victim = i;
        raw_entry_release(g_raw_cache[victim]);
        g_raw_cache[victim] = e;
        return &g_raw_cache[victim];
    }
    g_raw_cache.push_back(e);
    return &g_raw_cache.back();
}

// One transform of a raw call.  Trusted table: the throughput kernels.  Otherwise the checked sequence: compare the
// caller's table with the context's (device side, ~2 MB of reads for 4 primes at n = 2^15), the throughput kernel guarded by
// the result, and the literal kernels guarded the other way round -- whichever the table calls for transforms the data.
hipError_t raw_run(RawEntry* e, bool inverse, u64* d_a, unsigned n, const u64* d_tab, unsigned num, unsigned division, const ModSet& m,
                   hipStream_t s)
{
    // The comparison kernel of an earlier call found the table changed (host-mapped word, read without a synchronisation: a call or two
    // late at worst): derive a new context from what the table holds now, instead of sending every later call through the guarded
    // literal leg.  Launches on the old context may be in flight on any stream, hence the device-wide wait; an entry whose table has
    // changed three times keeps the literal kernels (a buffer the caller rewrites for every call).
    if (e && e->ctx && e->h_changed && *e->h_changed) {
        (void)hipDeviceSynchronize();
        *e->h_changed = 0;
        raw_slots_release(*e);                            // (copies of the old context's per-prime records)
        mi355ntt_ctx_destroy(e->ctx);
        e->ctx = ++e->rederived > 2 ? nullptr : raw_derive(e->device, e->n, e->division, e->inverse, e->tab, e->q, e->mu, e->bits);
        (void)hipGetLastError();
        e->trusted = false;
    }
    const mi355ntt_ctx* c = e ? e->ctx : nullptr;
    if (!c || c->literal_stages || (c->split16 && !e->trusted))
        return inverse ? compat_inverse_batch(d_a, n, d_tab, num, division, m, s) : compat_forward_batch(d_a, n, d_tab, num, division, m, s);
    if (e->trusted) return inverse ? run_inverse(c, d_a, num, division, 0, s) : run_forward(c, d_a, num, division, 0, s);
    hipError_t err;
    // A capturing stream gets no guard record: the graph may be replayed on any stream, next to direct calls that use the record of
    // the stream it was captured on.  Its checked calls therefore follow the caller's table with the literal kernels -- always the
    // reference's words; a caller who wants the throughput kernels inside a graph promises the table with mi355ntt_raw_trust_tables
    // (then nothing is written at all).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
        return inverse ? compat_inverse_batch(d_a, n, d_tab, num, division, m, s) : compat_forward_batch(d_a, n, d_tab, num, division, m, s);
    // The guard words this call uses are its stream's own (raw_slot_for): no event, no wait, nothing shared with calls on other streams
    // (until round 5 every call recorded an event behind itself -- a barrier packet and 6 us of idle GPU per call in the kernel trace,
    // profiles/r06_raw_checked_calls.txt).
    void* const slot = raw_slot_for(*e, s);
    if (!slot) return inverse ? compat_inverse_batch(d_a, n, d_tab, num, division, m, s) : compat_forward_batch(d_a, n, d_tab, num, division, m, s);
    if (!e->h_changed) {
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
            std::memset(h, 0, 64);
            e->h_changed = static_cast<volatile unsigned*>(h);
            e->d_changed = static_cast<unsigned*>(d);
        } else {
            if (h) (void)hipHostFree(h);
            (void)hipGetLastError();
        }
    }
    unsigned* guard = static_cast<unsigned*>(slot);
    FastTables t = c->fast;                               // the context's tables with this stream's guard record + per-prime records
    t.d_primes_alloc = slot;
    t.d_primes = static_cast<char*>(slot) + fast_prime_record_bytes();
    if (++e->epoch == 0) e->epoch = 1;
    if ((err = compat_tables_check(d_tab, inverse ? c->d_psiinv : c->d_psi, n, division, guard, e->epoch, s, e->d_changed)) != hipSuccess) return err;
    err = inverse ? fast_inverse_batch(t, d_a, num, division, kGuardBit, s) : fast_forward_batch(t, d_a, num, division, kGuardBit, s);
    if (err != hipSuccess) return err;
    return inverse ? compat_inverse_batch(d_a, n, d_tab, num, division, m, s, guard) : compat_forward_batch(d_a, n, d_tab, num, division, m, s, guard);
}

}  // namespace

extern "C" {

int mi355ntt_raw_cache_clear(void)
{
    std::lock_guard<std::mutex> lock(g_raw_mutex);
    for (RawEntry& e : g_raw_cache) raw_entry_release(e);
    g_raw_cache.clear();
    return MI355NTT_OK;
}

/* The caller promises that the table at d_table keeps its contents (until mi355ntt_raw_cache_clear): calls with these
 * arguments skip the per-call comparison.  1: such calls now run the throughput kernels unchecked, 0: they run the literal
 * kernels (nothing to trust). */
int mi355ntt_raw_trust_tables(unsigned n, const mi355ntt_u64* d_table, int inverse, unsigned division, const mi355ntt_u64* q,
                              const mi355ntt_u64* mu, const unsigned* bits)
{
    ModSet m;
    if (!d_table || check_raw_n(n) != MI355NTT_OK || fill_modset(&m, division, q, mu, bits) != MI355NTT_OK) return 0;
    std::lock_guard<std::mutex> lock(g_raw_mutex);
    RawEntry* e = raw_lookup(n, division, inverse != 0, d_table, q, mu, bits);
    if (!e || !e->ctx) return 0;
    e->trusted = true;
    return 1;
}

/* 1: calls with these arguments run the throughput kernels, 0: the literal kernels (derives and caches on first use) */
int mi355ntt_raw_uses_fast_kernels(unsigned n, const mi355ntt_u64* d_table, int inverse, unsigned division, const mi355ntt_u64* q,
                                   const mi355ntt_u64* mu, const unsigned* bits)
{
    ModSet m;
    if (!d_table || check_raw_n(n) != MI355NTT_OK || fill_modset(&m, division, q, mu, bits) != MI355NTT_OK) return 0;
    std::lock_guard<std::mutex> lock(g_raw_mutex);
    const RawEntry* e = raw_lookup(n, division, inverse != 0, d_table, q, mu, bits);
    return (e && e->ctx && (!e->ctx->split16 || e->trusted)) ? 1 : 0;
}

int mi355ntt_forward_batch_raw(mi355ntt_u64* d_a, unsigned n, const mi355ntt_u64* d_tabs, unsigned num, unsigned division,
                               const mi355ntt_u64* q, const mi355ntt_u64* mu, const unsigned* bits, mi355ntt_stream s)
{
    if (!d_a || !d_tabs) return MI355NTT_EINVAL;
    int rc = check_raw_n(n);
    if (rc) return rc;
    ModSet m;
    rc = fill_modset(&m, division, q, mu, bits);
    if (rc) return rc;
    if (num == 0) return MI355NTT_OK;
    (void)hipGetLastError();
    std::lock_guard<std::mutex> lock(g_raw_mutex);       // one call's launches stay together on the stream
    HIP_TRY(raw_run(raw_lookup(n, division, false, d_tabs, q, mu, bits, (hipStream_t)s, true), false, d_a, n, d_tabs, num, division, m, (hipStream_t)s));
    return MI355NTT_OK;
}

int mi355ntt_inverse_batch_raw(mi355ntt_u64* d_a, unsigned n, const mi355ntt_u64* d_tabs, unsigned num, unsigned division,
                               const mi355ntt_u64* q, const mi355ntt_u64* mu, const unsigned* bits, mi355ntt_stream s)
{
    if (!d_a || !d_tabs) return MI355NTT_EINVAL;
    int rc = check_raw_n(n);
    if (rc) return rc;
    ModSet m;
    rc = fill_modset(&m, division, q, mu, bits);
    if (rc) return rc;
    if (num == 0) return MI355NTT_OK;
    (void)hipGetLastError();
    std::lock_guard<std::mutex> lock(g_raw_mutex);
    HIP_TRY(raw_run(raw_lookup(n, division, true, d_tabs, q, mu, bits, (hipStream_t)s, true), true, d_a, n, d_tabs, num, division, m, (hipStream_t)s));
    return MI355NTT_OK;
}

int mi355ntt_forward_raw(mi355ntt_u64* d_a, unsigned n, mi355ntt_stream s, mi355ntt_u64 q, mi355ntt_u64 mu, int bits,
                         const mi355ntt_u64* d_tab)
{
    unsigned b = (unsigned)bits;
    return mi355ntt_forward_batch_raw(d_a, n, d_tab, 1, 1, &q, &mu, &b, s);
}

int mi355ntt_inverse_raw(mi355ntt_u64* d_a, unsigned n, mi355ntt_stream s, mi355ntt_u64 q, mi355ntt_u64 mu, int bits,
                         const mi355ntt_u64* d_tab)
{
    unsigned b = (unsigned)bits;
    return mi355ntt_inverse_batch_raw(d_a, n, d_tab, 1, 1, &q, &mu, &b, s);
}

int mi355ntt_barrett_raw(mi355ntt_u64* d_c, const mi355ntt_u64* d_a, const mi355ntt_u64* d_b, unsigned n, unsigned num,
                         unsigned division, const mi355ntt_u64* q, const mi355ntt_u64* mu, const unsigned* bits,
                         mi355ntt_stream s)
{
    if (!d_c || !d_a || !d_b || n == 0) return MI355NTT_EINVAL;
    ModSet m;
    int rc = fill_modset(&m, division, q, mu, bits);
    if (rc) return rc;
    if (num == 0) return MI355NTT_OK;
    (void)hipGetLastError();
    HIP_TRY(compat_pointwise(d_c, d_a, d_b, n, num, division, m, (hipStream_t)s));
    return MI355NTT_OK;
}

int mi355ntt_barrett_int_raw(mi355ntt_u64* d_a, mi355ntt_u64 b, unsigned n, mi355ntt_u64 q, mi355ntt_u64 mu, int bits,
                             mi355ntt_stream s)
{
    if (!d_a || n == 0) return MI355NTT_EINVAL;
    if (bits < 3 || bits > 62 || q == 0) return MI355NTT_EUNSUPPORTED;
    HIP_TRY(compat_pointwise_scalar(d_a, b, n, q, mu, (unsigned)bits, (hipStream_t)s));
    return MI355NTT_OK;
}

/* ---------------- the reference's 30-bit path (old/ntt_30bit.cuh) ---------------- */
static int check30(const void* a, const void* tab, unsigned n, mi355ntt_u32 q, int bits)
{
    if (!a || !tab) return MI355NTT_EINVAL;
    if (!is_pow2(n) || n < 2048 || n > 65536) return MI355NTT_EUNSUPPORTED;      /* forwardNTT dispatch, old/ntt_30bit.cuh:271-283,321-359 */
    if (bits < 3 || bits > 30 || q < 3 || (q >> bits) != 0) return MI355NTT_EUNSUPPORTED;
    return MI355NTT_OK;
}

/* m^-1 mod q when a 30-bit call may take the native kernels (the caller's mu and bit_length are the canonical ones and the
 * reference's single-subtraction Barrett is exact for q, so the exact transform is the reference's words), else 0 */
static unsigned ninv30_if_native(unsigned n, mi355ntt_u32 q, mi355ntt_u32 mu, int bits)
{
    static const bool literal_only = [] { const char* e = std::getenv("MI355NTT_RAW_LITERAL"); return e && e[0] && e[0] != '0'; }();
    if (literal_only || !(q & 1u) || (unsigned)bits != bit_length(q)) return 0;
    const u64 mu_ref = (((u128)1) << (2 * bits)) / q;
    if (mu_ref != (u64)mu || !barrett_single_subtraction_exact(q, (unsigned)bits, mu_ref)) return 0;
    const unsigned m = n == 65536 ? n / 2 : n;
    if (m % q == 0) return 0;
    return (unsigned)modinv(m % q, q);
}

int mi355ntt_forward30_batch_raw(mi355ntt_u32* d_a, unsigned n, const mi355ntt_u32* d_psi_table, unsigned num, mi355ntt_u32 q,
                                 mi355ntt_u32 mu, int bits, mi355ntt_stream s)
{
    int rc = check30(d_a, d_psi_table, n, q, bits);
    if (rc) return rc;
    if (num == 0) return MI355NTT_OK;
    (void)hipGetLastError();
    HIP_TRY(ntt30_forward(d_a, n, d_psi_table, num, q, mu, bits, ninv30_if_native(n, q, mu, bits), (hipStream_t)s));
    return MI355NTT_OK;
}

int mi355ntt_inverse30_batch_raw(mi355ntt_u32* d_a, unsigned n, const mi355ntt_u32* d_psiinv_table, unsigned num, mi355ntt_u32 q,
                                 mi355ntt_u32 mu, int bits, mi355ntt_stream s)
{
    int rc = check30(d_a, d_psiinv_table, n, q, bits);
    if (rc) return rc;
    if (num == 0) return MI355NTT_OK;
    (void)hipGetLastError();
    HIP_TRY(ntt30_inverse(d_a, n, d_psiinv_table, num, q, mu, bits, ninv30_if_native(n, q, mu, bits), (hipStream_t)s));
    return MI355NTT_OK;
}

int mi355ntt_forward30_raw(mi355ntt_u32* d_a, unsigned n, mi355ntt_stream s, mi355ntt_u32 q, mi355ntt_u32 mu, int bits,
                           const mi355ntt_u32* d_psi_table)
{
    return mi355ntt_forward30_batch_raw(d_a, n, d_psi_table, 1, q, mu, bits, s);
}

int mi355ntt_inverse30_raw(mi355ntt_u32* d_a, unsigned n, mi355ntt_stream s, mi355ntt_u32 q, mi355ntt_u32 mu, int bits,
                           const mi355ntt_u32* d_psiinv_table)
{
    return mi355ntt_inverse30_batch_raw(d_a, n, d_psiinv_table, 1, q, mu, bits, s);
}

int mi355ntt_barrett30_raw(mi355ntt_u32* d_a, const mi355ntt_u32* d_b, size_t count, mi355ntt_u32 q, mi355ntt_u32 mu, int bits,
                           mi355ntt_stream s)
{
    if (!d_a || !d_b) return MI355NTT_EINVAL;
    if (bits < 3 || bits > 30 || q < 3 || (q >> bits) != 0) return MI355NTT_EUNSUPPORTED;
    if (count == 0) return MI355NTT_OK;
    HIP_TRY(ntt30_barrett(d_a, d_b, count, q, mu, bits, (hipStream_t)s));
    return MI355NTT_OK;
}

}  // extern "C"
