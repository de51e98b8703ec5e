// capi.cpp -- implementation of the C ABI declared in include/mi355ntt.h.
#include "../../include/mi355ntt.h"

#include <hip/hip_runtime.h>

#include <cstring>
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
    bool literal = false;        // some prime is not barrett_exact and the caller did not ask for exact results:
                                 // transforms run the stage-per-launch kernels with the reference's arithmetic
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

static hipError_t run_forward(const mi355ntt_ctx* c, u64* d_a, unsigned num, unsigned division, unsigned base, hipStream_t s)
{
    if (c->literal) return compat_forward_batch(d_a, c->n, c->d_psi + (size_t)base * c->n, num, division, mods_from(c, base, division), s);
    if (c->split16) {
        hipError_t e = compat_ct_stage(d_a, c->n, c->d_psi + (size_t)base * c->n, 1, num, division, mods_from(c, base, division), s);
        if (e != hipSuccess) return e;
        return fast_forward_batch(c->fast, d_a, 2 * num, 2 * division, 2 * base, s);
    }
    return fast_forward_batch(c->fast, d_a, num, division, base, s);
}

static hipError_t run_inverse(const mi355ntt_ctx* c, u64* d_a, unsigned num, unsigned division, unsigned base, hipStream_t s)
{
    if (c->literal) return compat_inverse_batch(d_a, c->n, c->d_psiinv + (size_t)base * c->n, num, division, mods_from(c, base, division), s);
    if (c->split16) {
        hipError_t e = fast_inverse_batch(c->fast, d_a, 2 * num, 2 * division, 2 * base, s);
        if (e != hipSuccess) return e;
        return compat_gs_stage(d_a, c->n, c->d_psiinv + (size_t)base * c->n, 1, num, division, mods_from(c, base, division), s);
    }
    return fast_inverse_batch(c->fast, d_a, num, division, base, s);
}

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

int mi355ntt_ctx_uses_literal_kernels(const mi355ntt_ctx* c) { return (c && c->literal) ? 1 : 0; }

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
        if (!c->prime[i].barrett_exact && !(flags & MI355NTT_CTX_EXACT_ON_INEXACT_PRIMES)) c->literal = true;
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
    if (n == 65536 && !c->literal && 2 * num_primes <= kMaxPrimes) {
        const unsigned h_n = n / 2;
        std::vector<u64> vp((size_t)2 * num_primes * h_n), vi((size_t)2 * num_primes * h_n);
        PrimeParams vprime[kMaxPrimes];
        for (unsigned i = 0; i < num_primes; i++)
            for (unsigned h = 0; h < 2; h++) {
                const unsigned v = 2 * i + h;
                vprime[v] = c->prime[i];
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
        e = fast_tables_create(&c->fast, h_n, 2 * num_primes, vprime, vp.data(), vi.data(), nullptr, nullptr);
        c->split16 = (e == hipSuccess);
    } else {
        e = fast_tables_create(&c->fast, n, num_primes, c->prime, hp.data(), hi.data(), c->d_psi, c->d_psiinv);
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
    if (c->literal || c->split16) {   // the reference's own sequence (bfv_encryption.cuh:268-271), three calls
        HIP_TRY(run_forward(c, d_a, num, division, 0, (hipStream_t)s));
        HIP_TRY(compat_pointwise(d_a, d_a, d_bhat, c->n, num, division, mods_from(c, 0, division), (hipStream_t)s));
        HIP_TRY(run_inverse(c, d_a, num, division, 0, (hipStream_t)s));
        return MI355NTT_OK;
    }
    HIP_TRY(fast_polymul_batch(c->fast, d_a, d_bhat, num, division, (hipStream_t)s));
    return MI355NTT_OK;
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
    HIP_TRY(compat_forward_batch(d_a, n, d_tabs, num, division, m, (hipStream_t)s));
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
    HIP_TRY(compat_inverse_batch(d_a, n, d_tabs, num, division, m, (hipStream_t)s));
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
    if (!is_pow2(n) || n < 2048 || n > 32768) return MI355NTT_EUNSUPPORTED;      /* forwardNTT dispatch, old/ntt_30bit.cuh:321-359 */
    if (bits < 3 || bits > 30 || q < 3 || (q >> bits) != 0) return MI355NTT_EUNSUPPORTED;
    return MI355NTT_OK;
}

int mi355ntt_forward30_batch_raw(mi355ntt_u32* d_a, unsigned n, const mi355ntt_u32* d_psi_table, unsigned num, mi355ntt_u32 q,
                                 mi355ntt_u32 mu, int bits, mi355ntt_stream s)
{
    int rc = check30(d_a, d_psi_table, n, q, bits);
    if (rc) return rc;
    if (num == 0) return MI355NTT_OK;
    HIP_TRY(ntt30_forward(d_a, n, d_psi_table, num, q, mu, bits, (hipStream_t)s));
    return MI355NTT_OK;
}

int mi355ntt_inverse30_batch_raw(mi355ntt_u32* d_a, unsigned n, const mi355ntt_u32* d_psiinv_table, unsigned num, mi355ntt_u32 q,
                                 mi355ntt_u32 mu, int bits, mi355ntt_stream s)
{
    int rc = check30(d_a, d_psiinv_table, n, q, bits);
    if (rc) return rc;
    if (num == 0) return MI355NTT_OK;
    HIP_TRY(ntt30_inverse(d_a, n, d_psiinv_table, num, q, mu, bits, (hipStream_t)s));
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
