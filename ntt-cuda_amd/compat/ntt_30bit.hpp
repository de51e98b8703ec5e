// ntt_30bit.hpp -- C++ mirror of the reference's 30-bit launchers (old/ntt_30bit.cuh) on libmi355ntt.
// Same names and argument order as the reference (unsigned* data and tables, caller's q / mu / bit_length);
// hipStream_t for cudaStream_t; every function returns an int status instead of void.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mi355ntt.h"

namespace mi355 {

// old/ntt_30bit.cuh:321-359
inline int forwardNTT(unsigned* device_a, unsigned n, hipStream_t& stream1, unsigned q, unsigned mu, int bit_length, unsigned* psi_powers)
{
    return mi355ntt_forward30_raw(device_a, n, stream1, q, mu, bit_length, psi_powers);
}

// old/ntt_30bit.cuh:269-319: two polynomials on two streams
inline int forwardNTTdouble(unsigned* device_a, unsigned* device_b, unsigned n, hipStream_t& stream1, hipStream_t& stream2, unsigned q,
                            unsigned mu, int bit_length, unsigned* psi_powers)
{
    int rc = mi355ntt_forward30_raw(device_a, n, stream1, q, mu, bit_length, psi_powers);
    return rc ? rc : mi355ntt_forward30_raw(device_b, n, stream2, q, mu, bit_length, psi_powers);
}

// old/ntt_30bit.cuh:361-405
inline int inverseNTT(unsigned* device_a, unsigned n, hipStream_t& stream1, unsigned q, unsigned mu, int bit_length, unsigned* psiinv_powers)
{
    return mi355ntt_inverse30_raw(device_a, n, stream1, q, mu, bit_length, psiinv_powers);
}

// barrett_30bit<<<N / 256, 256, 0, stream>>>(a, b, q, mu, qbit)  (old/ntt_30bit.cuh:10-35)
inline int barrett_30bit(unsigned* a, const unsigned* b, unsigned n, unsigned q, unsigned mu, int qbit, hipStream_t stream = nullptr)
{
    return mi355ntt_barrett30_raw(a, b, n, q, mu, qbit, stream);
}

}  // namespace mi355
