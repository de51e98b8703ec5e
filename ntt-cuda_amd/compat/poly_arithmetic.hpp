// poly_arithmetic.hpp -- C++ host mirror of the pointwise-product part and of the element-wise host wrappers (:312-352) of
// BFV_Scheme/poly_arithmetic.cuh.
//
// The reference exposes `barrett*` as __global__ kernels that callers launch themselves with
// <<<n/256[, num], 256, 0, stream>>> (60bit_ntt_test.cu:76, bfv_encryption.cuh:269-270).  Here each is a host
// function taking the grid's meaning (n, num) instead of the launch configuration.
#pragma once
#include <cstdlib>

#include "ntt_60bit.hpp"

namespace mi355 {

// barrett<<<n/256, 256, 0, stream>>>(a, b, q, mu, qbit)            poly_arithmetic.cuh:9
inline int barrett(unsigned long long* a, const unsigned long long* b, unsigned n, unsigned long long q, unsigned long long mu,
                   int qbit, hipStream_t stream = nullptr)
{
    unsigned bits = (unsigned)qbit;
    return mi355ntt_barrett_raw(a, a, b, n, 1, 1, &q, &mu, &bits, stream);
}

// barrett_batch<<<dim3(n/256, num), 256>>>(a, b, n, division)      poly_arithmetic.cuh:36  (moduli from q_cons/mu_cons/q_bit_cons)
inline int barrett_batch(unsigned long long* a, const unsigned long long* b, unsigned n, unsigned num, unsigned division,
                         hipStream_t stream = nullptr)
{
    return mi355ntt_barrett_raw(a, a, b, n, num, division, q_cons, mu_cons, q_bit_cons, stream);
}

// barrett_batch_3param<<<dim3(n/256, num), 256>>>(c, a, b, n, division)   poly_arithmetic.cuh:68
inline int barrett_batch_3param(unsigned long long* c, const unsigned long long* a, const unsigned long long* b, unsigned n,
                                unsigned num, unsigned division, hipStream_t stream = nullptr)
{
    return mi355ntt_barrett_raw(c, a, b, n, num, division, q_cons, mu_cons, q_bit_cons, stream);
}

// barrett_int<<<n/256, 256, 0, stream>>>(a, b, q, mu, qbit) / poly_mul_int   poly_arithmetic.cuh:100,317
inline int poly_mul_int(unsigned long long* device_a, const unsigned long long b, unsigned n, hipStream_t& stream,
                        unsigned long long q, unsigned long long mu, int bit_length)
{
    return mi355ntt_barrett_int_raw(device_a, b, n, q, mu, bit_length, stream);
}

// poly_add_device   poly_arithmetic.cuh:312  (a[i] = a[i] + b[i], minus q where the sum is > q)
inline int poly_add_device(unsigned long long* device_a, const unsigned long long* device_b, unsigned n, hipStream_t& stream, unsigned long long q)
{
    return mi355ntt_poly_add_raw(device_a, device_b, n, stream, q);
}

// poly_mul_int_t   poly_arithmetic.cuh:322  (mod_t: low 64 bits of a[i] b, masked with the 32-bit t - 1)
inline int poly_mul_int_t(unsigned long long* device_a, const unsigned long long b, unsigned n, hipStream_t& stream, unsigned long long t)
{
    return mi355ntt_poly_mul_int_t_raw(device_a, b, n, stream, t);
}

// poly_sub_device   poly_arithmetic.cuh:327  (the reference's poly_sub adds q where a[i] < b[i] and never subtracts b: mirrored)
inline int poly_sub_device(unsigned long long* device_a, const unsigned long long* device_b, unsigned n, hipStream_t& stream, unsigned long long q)
{
    return mi355ntt_poly_sub_raw(device_a, device_b, n, stream, q);
}

// poly_negate_device   poly_arithmetic.cuh:340
inline int poly_negate_device(unsigned long long* device_a, unsigned n, hipStream_t& stream, unsigned long long q)
{
    return mi355ntt_poly_negate_raw(device_a, n, stream, q);
}

// poly_add_integer_device / poly_add_integer_device_default   poly_arithmetic.cuh:345-352
inline int poly_add_integer_device(unsigned long long* device_a, unsigned long long b, unsigned n, hipStream_t& stream, unsigned long long q)
{
    return mi355ntt_poly_add_integer_raw(device_a, b, n, stream, q);
}
inline int poly_add_integer_device_default(unsigned long long* device_a, unsigned long long b, unsigned n, unsigned long long q)
{
    return mi355ntt_poly_add_integer_raw(device_a, b, n, nullptr, q);
}

// half_poly_mul_device   poly_arithmetic.cuh:303-310: a = INTT(NTT(a) (.) b), b already in the NTT domain
inline int half_poly_mul_device(unsigned long long* device_a, unsigned long long* device_b, unsigned n, hipStream_t& stream,
                                unsigned long long q, unsigned long long mu, int bit_length, unsigned long long* psi_powers,
                                unsigned long long* psiinv_powers)
{
    int rc = forwardNTT(device_a, n, stream, q, mu, bit_length, psi_powers);
    if (!rc) rc = barrett(device_a, device_b, n, q, mu, bit_length, stream);
    if (!rc) rc = inverseNTT(device_a, n, stream, q, mu, bit_length, psiinv_powers);
    return rc;
}

// full_poly_mul_device   poly_arithmetic.cuh:296-301
inline int full_poly_mul_device(unsigned long long* device_a, unsigned long long* device_b, unsigned n, hipStream_t& stream1,
                                hipStream_t& stream2, unsigned long long q, unsigned long long mu, int bit_length,
                                unsigned long long* psi_powers)
{
    int rc = forwardNTTdouble(device_a, device_b, n, stream1, stream2, q, mu, bit_length, psi_powers);
    if (rc) return rc;
    if (stream1 != stream2) {           // the reference leans on legacy default-stream ordering here
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return MI355NTT_EHIP;
        (void)hipEventRecord(e, stream1);
        (void)hipStreamWaitEvent(stream2, e, 0);
        (void)hipEventDestroy(e);
    }
    return barrett(device_a, device_b, n, q, mu, bit_length, stream2);
}

// full_poly_mul   poly_arithmetic.cuh:277-294: returns a malloc'ed host array the caller frees (copy completes on stream2)
inline unsigned long long* full_poly_mul(unsigned long long* host_a, unsigned long long* host_b, unsigned long long* device_a,
                                         unsigned long long* device_b, unsigned n, hipStream_t& stream1, hipStream_t& stream2,
                                         unsigned long long q, unsigned long long mu, int bit_length,
                                         unsigned long long* psi_powers, unsigned long long* psiinv_powers)
{
    const size_t array_size = sizeof(unsigned long long) * n;
    unsigned long long* result = (unsigned long long*)std::malloc(array_size);
    (void)hipMemcpyAsync(device_a, host_a, array_size, hipMemcpyHostToDevice, stream1);
    (void)hipMemcpyAsync(device_b, host_b, array_size, hipMemcpyHostToDevice, stream2);
    if (full_poly_mul_device(device_a, device_b, n, stream1, stream2, q, mu, bit_length, psi_powers) ||
        inverseNTT(device_a, n, stream2, q, mu, bit_length, psiinv_powers)) {
        std::free(result);
        return nullptr;
    }
    (void)hipMemcpyAsync(result, device_a, array_size, hipMemcpyDeviceToHost, stream2);
    return result;
}

}  // namespace mi355
