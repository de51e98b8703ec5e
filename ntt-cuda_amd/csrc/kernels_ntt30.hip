// kernels_ntt30.hip -- the reference's 30-bit path (old/ntt_30bit.cuh): 32-bit coefficients, products
// in 64 bits, the same single-subtraction Barrett written on one machine word.  One 1024-thread workgroup per
// polynomial with the whole polynomial in LDS (n * 4 B <= 128 KiB): every stage is the reference's butterfly on the
// reference's indices, so the words are the reference's for every input; twiddles come from the caller's table.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

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

// all stages of CTBasedNTTInner(Single) (old/ntt_30bit.cuh:70-129, 199-227) / GSBasedINTTInner(Single) (:131-196, 229-267)
template <int LOGN, bool FWD>
__global__ void __launch_bounds__(1024)
k_ntt30(u32* __restrict__ a, const u32* __restrict__ tab, u32 q, u32 mu, int qbit)
{
    constexpr unsigned n = 1u << LOGN, T = n / 2 < 1024 ? n / 2 : 1024, PER = n / 2 / T;
    __shared__ u32 s[n];
    u32* poly = a + (size_t)blockIdx.x * n;
    const unsigned t = threadIdx.x;
    for (unsigned i = t; i < n; i += T) s[i] = poly[i];
    __syncthreads();
    const u32 q2 = (q + 1) >> 1;
    if constexpr (FWD) {
        for (unsigned length = 1; length < n; length *= 2) {
            const unsigned step = (n / length) / 2;
            for (unsigned it = 0; it < PER; it++) {
                const unsigned g = t + it * T;
                const unsigned psi_step = g / step;
                const unsigned j = psi_step * step * 2 + g % step;
                const u32 psi = tab[length + psi_step];
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
                const u32 psiinv = tab[length + psi_step];
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
}

// barrett_30bit, old/ntt_30bit.cuh:10-35
__global__ void __launch_bounds__(256)
k_barrett30(u32* __restrict__ a, const u32* __restrict__ b, size_t count, u32 q, u32 mu, int qbit)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    u64 rc = (u64)a[i] * b[i];
    u64 rx = rc >> (qbit - 2);
    rx *= mu;
    rx >>= qbit + 2;
    rx *= q;
    rc -= rx;
    a[i] = rc < q ? (u32)rc : (u32)(rc - q);
}

template <bool FWD>
hipError_t launch30(unsigned* d_a, unsigned n, const unsigned* d_tab, unsigned num, unsigned q, unsigned mu, int bits, hipStream_t s)
{
    dim3 g(num);
    switch (n) {
    case 2048: k_ntt30<11, FWD><<<g, 1024, 0, s>>>(d_a, d_tab, q, mu, bits); break;
    case 4096: k_ntt30<12, FWD><<<g, 1024, 0, s>>>(d_a, d_tab, q, mu, bits); break;
    case 8192: k_ntt30<13, FWD><<<g, 1024, 0, s>>>(d_a, d_tab, q, mu, bits); break;
    case 16384: k_ntt30<14, FWD><<<g, 1024, 0, s>>>(d_a, d_tab, q, mu, bits); break;
    case 32768: k_ntt30<15, FWD><<<g, 1024, 0, s>>>(d_a, d_tab, q, mu, bits); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace

hipError_t ntt30_forward(unsigned* d_a, unsigned n, const unsigned* d_tab, unsigned num, unsigned q, unsigned mu, int bits, hipStream_t s)
{
    return launch30<true>(d_a, n, d_tab, num, q, mu, bits, s);
}

hipError_t ntt30_inverse(unsigned* d_a, unsigned n, const unsigned* d_tab, unsigned num, unsigned q, unsigned mu, int bits, hipStream_t s)
{
    return launch30<false>(d_a, n, d_tab, num, q, mu, bits, s);
}

hipError_t ntt30_barrett(unsigned* d_a, const unsigned* d_b, size_t count, unsigned q, unsigned mu, int bits, hipStream_t s)
{
    k_barrett30<<<dim3((unsigned)((count + 255) / 256)), 256, 0, s>>>(d_a, d_b, count, q, mu, bits);
    return hipGetLastError();
}

}  // namespace mi355ntt
