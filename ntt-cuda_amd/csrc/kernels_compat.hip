// kernels_compat.hip -- literal, stage-per-launch kernels behind the raw-parameter entry points.
//
// These follow the reference's arithmetic step by step (Algorithm 7 with the caller's mu/bit_length,
// canonical residues after every stage, halving in every GS stage) so that a caller who brings its
// own tables and Barrett constants gets exactly what ntt_60bit.cuh would have produced.  For n <= 2^14
// the stages of one polynomial run in one launch out of LDS; larger (or other) n one launch per stage.  They are the
// device-side ground truth the fast kernels in kernels_fast.hip are checked against; they are not the
// throughput path.
//
// Reference counterparts (BFV_Scheme/): CTBasedNTTInner(_batch) ntt_60bit.cuh:192-223,527-561;
// GSBasedINTTInner(_batch) :225-265,563-606; barrett* poly_arithmetic.cuh:9-126.
#include "kernels.hpp"
#include "modarith.cuh"

namespace mi355ntt {

namespace {

constexpr int kBlock = 256;

// One CT stage over a batch.  blockIdx.y = polynomial, modulus index = y % division
// (ntt_60bit.cuh:391-394), data offset y*n (:404), table offset index*n (:422).
__global__ void __launch_bounds__(kBlock) ct_stage_kernel(u64* __restrict__ a, const u64* __restrict__ tabs, unsigned n,
                                                          unsigned length, unsigned division, ModSet m)
{
    unsigned y = blockIdx.y;
    unsigned idx = y % division;
    u64 q = m.q[idx], mu = m.mu[idx];
    u32 k = m.k[idx];
    unsigned g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n / 2) return;
    unsigned step = (n / length) / 2;
    unsigned p = g / step;
    unsigned j = p * step * 2 + (g % step);
    u64* poly = a + (size_t)y * n;
    u64 psi = tabs[(size_t)idx * n + length + p];
    u64 U = poly[j];
    u64 V = barrett_mul(poly[j + step], psi, q, mu, k);
    poly[j] = add_mod(U, V, q);
    poly[j + step] = sub_mod(U, V, q);
}

// One GS stage with the n^-1 halving (ntt_60bit.cuh:225-265)
__global__ void __launch_bounds__(kBlock) gs_stage_kernel(u64* __restrict__ a, const u64* __restrict__ tabs, unsigned n,
                                                          unsigned length, unsigned division, ModSet m)
{
    unsigned y = blockIdx.y;
    unsigned idx = y % division;
    u64 q = m.q[idx], mu = m.mu[idx];
    u32 k = m.k[idx];
    unsigned g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n / 2) return;
    unsigned step = (n / length) / 2;
    unsigned p = g / step;
    unsigned j = p * step * 2 + (g % step);
    u64* poly = a + (size_t)y * n;
    u64 psiinv = tabs[(size_t)idx * n + length + p];
    u64 q2 = (q + 1) >> 1;
    u64 U = poly[j];
    u64 V = poly[j + step];
    poly[j] = half_mod(add_mod(U, V, q), q2);
    u64 d = barrett_mul(sub_mod(U, V, q), psiinv, q, mu, k);
    poly[j + step] = half_mod(d, q2);
}

// c[i] = a[i] * b[i] mod q[y % division]   (barrett / barrett_batch / barrett_batch_3param)
__global__ void __launch_bounds__(kBlock) pointwise_kernel(u64* __restrict__ c, const u64* __restrict__ a,
                                                           const u64* __restrict__ b, unsigned n, unsigned division, ModSet m)
{
    unsigned y = blockIdx.y;
    unsigned idx = y % division;
    u64 q = m.q[idx], mu = m.mu[idx];
    u32 k = m.k[idx];
    unsigned x = blockIdx.x * kBlock + threadIdx.x;
    if (x >= n) return;
    size_t i = (size_t)y * n + x;
    c[i] = barrett_mul(a[i], b[i], q, mu, k);
}

// a[i] = a[i] * b mod q   (barrett_int)
__global__ void __launch_bounds__(kBlock) pointwise_scalar_kernel(u64* __restrict__ a, u64 b, unsigned n, u64 q, u64 mu, u32 k)
{
    unsigned x = blockIdx.x * kBlock + threadIdx.x;
    if (x >= n) return;
    a[x] = barrett_mul(a[x], b, q, mu, k);
}

// All stages of one polynomial in one launch, polynomial resident in LDS (n * 8 B <= 128 KiB): the same butterflies on the
// same indices as the stage kernels above (the reference's CTBasedNTTInnerSingle / GSBasedINTTInnerSingle do the same
// inside their block-private slices), so the same words.  One 1024-thread workgroup per polynomial.
template <int LOGN, bool FWD>
__global__ void __launch_bounds__(1024) literal_lds_kernel(u64* __restrict__ a, const u64* __restrict__ tabs, unsigned division, ModSet m)
{
    constexpr unsigned n = 1u << LOGN, T = n / 2 < 1024 ? n / 2 : 1024, PER = n / 2 / T;
    __shared__ u64 sh[n];
    const unsigned y = blockIdx.x, idx = y % division, t = threadIdx.x;
    const u64 q = m.q[idx], mu = m.mu[idx];
    const u32 k = m.k[idx];
    u64* poly = a + (size_t)y * n;
    const u64* tab = tabs + (size_t)idx * n;
    if (t < T)
        for (unsigned i = t; i < n; i += T) sh[i] = poly[i];
    __syncthreads();
    if constexpr (FWD) {
        for (unsigned length = 1; length < n; length *= 2) {
            const unsigned step = (n / length) / 2;
            if (t < T)
                for (unsigned it = 0; it < PER; it++) {
                    const unsigned g = t + it * T, p = g / step, j = p * step * 2 + (g % step);
                    const u64 U = sh[j];
                    const u64 V = barrett_mul(sh[j + step], tab[length + p], q, mu, k);
                    sh[j] = add_mod(U, V, q);
                    sh[j + step] = sub_mod(U, V, q);
                }
            __syncthreads();
        }
    } else {
        const u64 q2 = (q + 1) >> 1;
        for (unsigned length = n / 2; length >= 1; length /= 2) {
            const unsigned step = (n / length) / 2;
            if (t < T)
                for (unsigned it = 0; it < PER; it++) {
                    const unsigned g = t + it * T, p = g / step, j = p * step * 2 + (g % step);
                    const u64 U = sh[j], V = sh[j + step];
                    sh[j] = half_mod(add_mod(U, V, q), q2);
                    sh[j + step] = half_mod(barrett_mul(sub_mod(U, V, q), tab[length + p], q, mu, k), q2);
                }
            __syncthreads();
        }
    }
    if (t < T)
        for (unsigned i = t; i < n; i += T) poly[i] = sh[i];
}

template <bool FWD>
bool launch_literal_lds(u64* d_a, unsigned n, const u64* d_tabs, unsigned num, unsigned division, const ModSet& m, hipStream_t s)
{
    switch (n) {
    case 2048: literal_lds_kernel<11, FWD><<<num, 1024, 0, s>>>(d_a, d_tabs, division, m); return true;
    case 4096: literal_lds_kernel<12, FWD><<<num, 1024, 0, s>>>(d_a, d_tabs, division, m); return true;
    case 8192: literal_lds_kernel<13, FWD><<<num, 1024, 0, s>>>(d_a, d_tabs, division, m); return true;
    case 16384: literal_lds_kernel<14, FWD><<<num, 1024, 0, s>>>(d_a, d_tabs, division, m); return true;
    default: return false;
    }
}

}  // namespace

hipError_t compat_forward_batch(u64* d_a, unsigned n, const u64* d_tabs, unsigned num, unsigned division, const ModSet& m,
                                hipStream_t s)
{
    if (launch_literal_lds<true>(d_a, n, d_tabs, num, division, m, s)) return hipGetLastError();
    dim3 grid((n / 2 + kBlock - 1) / kBlock, num);
    for (unsigned length = 1; length < n; length *= 2)
        ct_stage_kernel<<<grid, kBlock, 0, s>>>(d_a, d_tabs, n, length, division, m);
    return hipGetLastError();
}

hipError_t compat_inverse_batch(u64* d_a, unsigned n, const u64* d_tabs, unsigned num, unsigned division, const ModSet& m,
                                hipStream_t s)
{
    if (launch_literal_lds<false>(d_a, n, d_tabs, num, division, m, s)) return hipGetLastError();
    dim3 grid((n / 2 + kBlock - 1) / kBlock, num);
    for (unsigned length = n / 2; length >= 1; length /= 2)
        gs_stage_kernel<<<grid, kBlock, 0, s>>>(d_a, d_tabs, n, length, division, m);
    return hipGetLastError();
}

hipError_t compat_ct_stage(u64* d_a, unsigned n, const u64* d_tabs, unsigned length, unsigned num, unsigned division, const ModSet& m,
                           hipStream_t s)
{
    ct_stage_kernel<<<dim3((n / 2 + kBlock - 1) / kBlock, num), kBlock, 0, s>>>(d_a, d_tabs, n, length, division, m);
    return hipGetLastError();
}

hipError_t compat_gs_stage(u64* d_a, unsigned n, const u64* d_tabs, unsigned length, unsigned num, unsigned division, const ModSet& m,
                           hipStream_t s)
{
    gs_stage_kernel<<<dim3((n / 2 + kBlock - 1) / kBlock, num), kBlock, 0, s>>>(d_a, d_tabs, n, length, division, m);
    return hipGetLastError();
}

hipError_t compat_pointwise(u64* d_c, const u64* d_a, const u64* d_b, unsigned n, unsigned num, unsigned division,
                            const ModSet& m, hipStream_t s)
{
    dim3 grid((n + kBlock - 1) / kBlock, num);
    pointwise_kernel<<<grid, kBlock, 0, s>>>(d_c, d_a, d_b, n, division, m);
    return hipGetLastError();
}

hipError_t compat_pointwise_scalar(u64* d_a, u64 b, unsigned n, u64 q, u64 mu, unsigned k, hipStream_t s)
{
    pointwise_scalar_kernel<<<(n + kBlock - 1) / kBlock, kBlock, 0, s>>>(d_a, b, n, q, mu, k);
    return hipGetLastError();
}

}  // namespace mi355ntt
