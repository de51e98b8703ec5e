// kernels_fast.hip -- throughput path of the NTT engine (context entry points).
#include "kernels.hpp"
#include "modarith.cuh"

#include <cstring>
#include <vector>

namespace mi355ntt {

hipError_t fast_tables_create(FastTables* t, unsigned n, unsigned num_primes, const PrimeParams* prime, const u64* h_psi,
                              const u64* h_psiinv, const u64* d_psi, const u64* d_psiinv)
{
    t->n = n;
    t->log_n = 0;
    while ((1u << t->log_n) < n) t->log_n++;
    t->num_primes = num_primes;
    std::memset(&t->mods, 0, sizeof(t->mods));
    for (unsigned i = 0; i < num_primes; i++) {
        t->prime[i] = prime[i];
        t->mods.q[i] = prime[i].q;
        t->mods.mu[i] = prime[i].mu;
        t->mods.k[i] = prime[i].k;
    }
    t->d_psi = d_psi;
    t->d_psiinv = d_psiinv;
    (void)h_psi;
    (void)h_psiinv;
    return hipSuccess;
}

void fast_tables_destroy(FastTables* t)
{
    if (t->d_fwd) (void)hipFree(t->d_fwd);
    if (t->d_inv) (void)hipFree(t->d_inv);
    if (t->d_ninv) (void)hipFree(t->d_ninv);
    t->d_fwd = t->d_inv = t->d_ninv = nullptr;
}

static ModSet shifted(const ModSet& m, unsigned base, unsigned division)
{
    ModSet r;
    std::memset(&r, 0, sizeof(r));
    for (unsigned i = 0; i < division && base + i < kMaxPrimes; i++) {
        r.q[i] = m.q[base + i];
        r.mu[i] = m.mu[base + i];
        r.k[i] = m.k[base + i];
    }
    return r;
}

hipError_t fast_forward_batch(const FastTables& t, u64* d_a, unsigned num, unsigned division, unsigned prime_base, hipStream_t s)
{
    return compat_forward_batch(d_a, t.n, t.d_psi + (size_t)prime_base * t.n, num, division, shifted(t.mods, prime_base, division), s);
}

hipError_t fast_inverse_batch(const FastTables& t, u64* d_a, unsigned num, unsigned division, unsigned prime_base, hipStream_t s)
{
    return compat_inverse_batch(d_a, t.n, t.d_psiinv + (size_t)prime_base * t.n, num, division, shifted(t.mods, prime_base, division), s);
}

hipError_t fast_pointwise(const FastTables& t, u64* d_c, const u64* d_a, const u64* d_b, unsigned num, unsigned division,
                          hipStream_t s)
{
    return compat_pointwise(d_c, d_a, d_b, t.n, num, division, t.mods, s);
}

hipError_t fast_polymul_batch(const FastTables& t, u64* d_a, const u64* d_bhat, unsigned num, unsigned division, hipStream_t s)
{
    hipError_t e = fast_forward_batch(t, d_a, num, division, 0, s);
    if (e != hipSuccess) return e;
    e = fast_pointwise(t, d_a, d_a, d_bhat, num, division, s);
    if (e != hipSuccess) return e;
    return fast_inverse_batch(t, d_a, num, division, 0, s);
}

}  // namespace mi355ntt
