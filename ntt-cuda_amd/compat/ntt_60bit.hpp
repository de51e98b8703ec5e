// ntt_60bit.hpp -- C++ host mirror of the reference's NTT call surface on top of the C ABI (libmi355ntt.so).
//
// Same names, argument order and meaning as ozgunozerk/NTT-Cuda BFV_Scheme/ntt_60bit.cuh:267-386,608-697, with
// hipStream_t in place of cudaStream_t.  The reference's `__constant__ q_cons / q_bit_cons / mu_cons`
// (ntt_60bit.cuh:8-10), which callers fill with cudaMemcpyToSymbolAsync (demo.cu:72,127,172), become plain host
// arrays in this header: assign them (or call set_moduli) before the *_batch launchers, exactly where the
// reference uploads the symbols.  Unlike the reference, every launcher returns the C-ABI status (0 = ok).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mi355ntt.h"

namespace mi355 {

inline unsigned long long q_cons[MI355NTT_MAX_PRIMES];
inline unsigned q_bit_cons[MI355NTT_MAX_PRIMES];
inline unsigned long long mu_cons[MI355NTT_MAX_PRIMES];

// replaces the three cudaMemcpyToSymbolAsync calls at demo.cu:72,127,172 (and fixes the 8-byte-per-entry copy
// into the 4-byte q_bit_cons array there)
inline void set_moduli(const unsigned long long* q, const unsigned* bits, const unsigned long long* mu, unsigned count)
{
    for (unsigned i = 0; i < count && i < MI355NTT_MAX_PRIMES; i++) {
        q_cons[i] = q[i];
        q_bit_cons[i] = bits[i];
        mu_cons[i] = mu[i];
    }
}

// ntt_60bit.cuh:314
inline int forwardNTT(unsigned long long* device_a, unsigned n, hipStream_t& stream1, unsigned long long q, unsigned long long mu,
                      int bit_length, unsigned long long* psi_powers)
{
    return mi355ntt_forward_raw(device_a, n, stream1, q, mu, bit_length, psi_powers);
}

// ntt_60bit.cuh:267
inline int forwardNTTdouble(unsigned long long* device_a, unsigned long long* device_b, unsigned n, hipStream_t& stream1,
                            hipStream_t& stream2, unsigned long long q, unsigned long long mu, int bit_length,
                            unsigned long long* psi_powers)
{
    int rc = mi355ntt_forward_raw(device_a, n, stream1, q, mu, bit_length, psi_powers);
    return rc ? rc : mi355ntt_forward_raw(device_b, n, stream2, q, mu, bit_length, psi_powers);
}

// ntt_60bit.cuh:350
inline int inverseNTT(unsigned long long* device_a, unsigned n, hipStream_t& stream1, unsigned long long q, unsigned long long mu,
                      int bit_length, unsigned long long* psiinv_powers)
{
    return mi355ntt_inverse_raw(device_a, n, stream1, q, mu, bit_length, psiinv_powers);
}

// ntt_60bit.cuh:608 -- stream 0 like the reference (pass a stream to overlap)
inline int forwardNTT_batch(unsigned long long* device_a, unsigned n, unsigned long long* psi_powers, unsigned num, unsigned division,
                            hipStream_t stream = nullptr)
{
    return mi355ntt_forward_batch_raw(device_a, n, psi_powers, num, division, q_cons, mu_cons, q_bit_cons, stream);
}

// ntt_60bit.cuh:652
inline int inverseNTT_batch(unsigned long long* device_a, unsigned n, unsigned long long* psiinv_powers, unsigned num,
                            unsigned division, hipStream_t stream = nullptr)
{
    return mi355ntt_inverse_batch_raw(device_a, n, psiinv_powers, num, division, q_cons, mu_cons, q_bit_cons, stream);
}

}  // namespace mi355
