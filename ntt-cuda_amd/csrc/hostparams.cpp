// hostparams.cpp -- host-only parameter derivation (no HIP).
#include "hostparams.hpp"

#include <cmath>

#include "../../include/mi355ntt.h"

namespace mi355ntt {

// demo.cu:69 computes (unsigned)(log2((double)q) + 1); this is the exact integer bit length, which
// agrees with that expression for every modulus the reference ships (tests/test_host_params.py).
unsigned bit_length(u64 q) { return q ? 64u - (unsigned)__builtin_clzll(q) : 0u; }

// 60bit_ntt_test.cu:47-49: mu = floor(2^(2k) / q)
u64 barrett_mu(u64 q, unsigned k) { return (u64)((((u128)1) << (2 * k)) / q); }

// The reference reduces a = x*y (x, y < q) as  s = ((a >> (k-2)) * mu) >> (k+2),  r = a - s*q,  r -= q if r >= q.
// With 2^(2k)/q = mu + f (0 <= f < 1) and a/2^(k-2) = x1 + g (0 <= g < 1):  a/q - x1*mu/2^(k+2) = a*f/2^(2k) + g*mu/2^(k+2),
// so the quotient estimate s is at most one short -- and the single subtraction enough -- whenever
//     (q-1)^2 * f / 2^(2k)  +  mu / 2^(k+2)  <  1.
// That holds for every modulus the reference ships and for every q = 2^k - d with d^2 << 2^k (then f ~ d^2/2^k), but
// not for all primes: q = 68719230977 = 2^36 - 245759 (second prime of decryption_test.cu) has f = 0.88 and the
// reference's transform returns q + r, or a wrong residue after the next butterfly's unsigned compare, for about
// 6e-5 of uniform operand pairs (1.8 % when one operand is q - 1).  The exact lazy kernels return the canonical value
// there, which is NOT what the reference prints, so contexts holding such a prime run the literal kernels by default.
bool barrett_single_subtraction_exact(u64 q, unsigned k, u64 mu)
{
    if (k < 3 || k > 62 || q < 2) return false;
    const u128 two2k = ((u128)1) << (2 * k);
    const long double f = (long double)(u64)(two2k % q) / (long double)q;
    const long double top = (long double)(q - 1) / ldexpl(1.0L, (int)k);
    const long double bound = top * top * f + (long double)mu / ldexpl(1.0L, (int)k + 2);
    return bound < 1.0L - 1e-9L;
}

// the same bound for products x y with x = q allowed (y < q): the sums the BFV drivers form with `>` instead of `>=` may equal q
// (bfv_decryption.cuh:13-23), and a product by a constant then has to come out 0 for a one-product form to stand for it
bool barrett_exact_for_operand_q(u64 q, unsigned k, u64 mu)
{
    if (k < 3 || k > 62 || q < 2) return false;
    const u128 two2k = ((u128)1) << (2 * k);
    const long double f = (long double)(u64)(two2k % q) / (long double)q;
    const long double top = (long double)(q - 1) / ldexpl(1.0L, (int)k), topq = (long double)q / ldexpl(1.0L, (int)k);
    const long double bound = topq * top * f + (long double)mu / ldexpl(1.0L, (int)k + 2);
    return bound < 1.0L - 1e-9L;
}

u64 mulmod(u64 a, u64 b, u64 m) { return (u64)(((u128)a * b) % m); }

// helper.h:8-28 (square-and-multiply, LSB first); reduced result for every exponent
u64 modpow(u64 a, u64 e, u64 m)
{
    u64 res = 1 % m;
    a %= m;
    while (e) {
        if (e & 1) res = mulmod(res, a, m);
        a = mulmod(a, a, m);
        e >>= 1;
    }
    return res;
}

// helper.h:52-56: Fermat inverse a^(q-2) (q prime)
u64 modinv(u64 a, u64 q) { return modpow(a, q - 2, q); }

u64 bit_reverse(u64 a, int bits)
{
    u64 r = 0;
    for (int i = 0; i < bits; i++) {
        r = (r << 1) | (a & 1);
        a >>= 1;
    }
    return r;
}

u64 shoup(u64 w, u64 q) { return (u64)((((u128)w) << 64) / q); }

// parameter.h:5-12: tab[i] = root^bitrev(i, log2 n).  Built by repeated multiplication in natural
// order and scattered, instead of n modpows.
void fill_table(u64 root, u64 q, unsigned n, u64* tab)
{
    int lg = 0;
    while ((1u << lg) < n) lg++;
    u64 p = 1 % q;
    for (unsigned e = 0; e < n; e++) {
        tab[bit_reverse(e, lg)] = p;
        p = mulmod(p, root, q);
    }
}

int derive_prime(unsigned n, u64 q, u64 psi, PrimeParams* out)
{
    if (q < 3 || (q & 1) == 0 || (q >> 62) != 0) return MI355NTT_EUNSUPPORTED;
    if (psi == 0 || psi >= q) return MI355NTT_EPARAM;
    if ((q - 1) % (2ull * n) != 0) return MI355NTT_EPARAM;
    // psi must be a primitive 2n-th root of unity: psi^n = -1 (n a power of two makes this sufficient)
    if (modpow(psi, n, q) != q - 1) return MI355NTT_EPARAM;
    out->q = q;
    out->psi = psi;
    out->k = bit_length(q);
    out->mu = barrett_mu(q, out->k);
    out->barrett_exact = barrett_single_subtraction_exact(q, out->k, out->mu);
    out->psiinv = modinv(psi, q);       // demo.cu:96-97
    if (mulmod(out->psiinv, psi, q) != 1) return MI355NTT_EPARAM;  // q not prime
    out->ninv = modinv(n % q, q);
    if (mulmod(out->ninv, n % q, q) != 1) return MI355NTT_EPARAM;
    return MI355NTT_OK;
}

}  // namespace mi355ntt
