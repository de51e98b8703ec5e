// kernels_bfv.hip -- element-wise kernels of the BFV drivers.  Streaming integer work: every
// kernel reads and writes each word once, 256 threads per block, per-prime constants by scalar loads.  The reference
// issues one launch per step; the steps of one driver touch the same index (or the same column across the RNS
// polynomials), so they are fused here -- each fused kernel performs the reference's steps in the reference's order on
// its element and leaves the same words behind.
#include "bfv.hpp"
#include "modarith.cuh"

namespace mi355ntt {

namespace {

constexpr unsigned kBlock = 256;

// poly_add_negate_xq, bfv_keygen.cuh:80-93
__global__ void __launch_bounds__(kBlock)
k_add_negate(u64* __restrict__ a, const u64* __restrict__ b, unsigned n, const BfvPrime* __restrict__ primes)
{
    const unsigned y = blockIdx.y;
    const u64 q = primes[y].q;
    const size_t i = (size_t)y * n + blockIdx.x * kBlock + threadIdx.x;
    u64 ra = a[i] + b[i];
    if (ra >= q) ra -= q;
    ra = q - ra;
    a[i] = ra * (ra != q);
}

// one column i of one half h of the ciphertext: poly_add_xq on all R polynomials (note `>`, bfv_encryption.cuh:180),
// +half on the last one (:110-124), subtract-and-scale on the others (:126-171), message term on c0 (:186-208)
__global__ void __launch_bounds__(kBlock)
k_encrypt_tail(u64* __restrict__ c, const u64* __restrict__ e, const u64* __restrict__ m, unsigned n, unsigned R, u64 t,
               const BfvPrime* __restrict__ primes)
{
    const unsigned h = blockIdx.y;
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    const unsigned r = R - 1;
    u64* ch = c + (size_t)h * R * n;
    const u64* eh = e + (size_t)h * R * n;
    const u64 q_last = primes[r].q, half_last = q_last >> 1;
    u64 last = ch[(size_t)r * n + i] + eh[(size_t)r * n + i];
    if (last > q_last) last -= q_last;                         // poly_add_xq
    last += half_last;                                         // ..._add_x2
    if (last >= q_last) last -= q_last;
    ch[(size_t)r * n + i] = last;
    u64 mi = 0, fix = 0;
    if (h == 0) {
        mi = m[i];
        fix = (mi + ((t + 1) >> 1)) / t;                       // weird_m_stuff: numerator / t
    }
    for (unsigned j = 0; j < r; j++) {
        const BfvPrime p = primes[j];
        u64 x = ch[(size_t)j * n + i] + eh[(size_t)j * n + i];
        if (x > p.q) x -= p.q;                                 // poly_add_xq
        u64 tmp = last % p.q;                                  // ..._loop_xq
        if (tmp < p.half_last_mod_q) tmp += p.q;
        tmp -= p.half_last_mod_q;
        if (x < tmp) x += p.q;
        x -= tmp;
        x = barrett_mul(x, p.inv_q_last_mod_q, p.q, p.mu, p.k);
        if (h == 0) x = (x + (mi * p.q_div_t + fix)) % p.q;    // weird_m_stuff
        ch[(size_t)j * n + i] = x;
    }
}

// c1[i] = ((c1[i] + c0[i], `>`) * prod_t_gamma) * inv_punctured_q, bfv_decryption.cuh:13-57
__global__ void __launch_bounds__(kBlock)
k_decrypt_scale(u64* __restrict__ c, unsigned n, unsigned R, const BfvPrime* __restrict__ primes)
{
    const unsigned y = blockIdx.y;
    const BfvPrime p = primes[y];
    const size_t i = (size_t)y * n + blockIdx.x * kBlock + threadIdx.x;
    u64* c1 = c + (size_t)R * n;
    u64 ra = c1[i] + c[i];
    if (ra > p.q) ra -= p.q;
    ra = barrett_mul(ra, p.prod_t_gamma_mod_q, p.q, p.mu, p.k);
    ra = barrett_mul(ra, p.inv_punctured_q, p.q, p.mu, p.k);
    c1[i] = ra;
}

// poly_arithmetic.cuh:221-268, :128-142, barrett_int (:100-126) per column k
__global__ void __launch_bounds__(kBlock)
k_decrypt_round(u64* __restrict__ c, unsigned n, unsigned R, u64 t, u64 gamma, u64 mu_gamma, unsigned gamma_bits, u64 gamma_div_2,
                u64 neg_inv_t, u64 neg_inv_gamma, const u64* __restrict__ bcm)
{
    const unsigned k = blockIdx.x * kBlock + threadIdx.x;
    const unsigned r = R - 1;
    const u64* c1 = c + (size_t)R * n;
    const unsigned mask32 = (unsigned)(t - 1);                 // `unsigned mask = t - 1`
    u64 acc_t = 0, acc_g = 0;
    for (unsigned i = 0; i < r; i++) {
        const u64 v = c1[k + (size_t)i * n];
        acc_t += (v * bcm[i]) & mask32;                                            // fast_convert_array_kernel_t
        const u64 tg = barrett_mul(v, bcm[i + r], gamma, mu_gamma, gamma_bits);   // fast_convert_array_kernel_gamma
        acc_g = (acc_g + tg) % gamma;
    }
    u64 x0 = acc_t & mask32;
    u64 x1 = acc_g % gamma;
    x0 = (x0 * neg_inv_t) & mask32;                                                // poly_mul_int_t -> mod_t
    x1 = barrett_mul(x1, neg_inv_gamma, gamma, mu_gamma, gamma_bits);             // poly_mul_int -> barrett_int
    c[k] = x0;
    c[k + n] = x1;
    const u64 mask = t - 1;                                                        // dec_round_kernel
    u64 res;
    if (x1 > gamma_div_2) res = (x0 + (gamma - x1)) & mask;
    else res = (x0 - x1) & mask;
    c[k + (size_t)n * (r - 1)] = res;
}

}  // namespace

hipError_t bfv_add_negate(const BfvParams& p, const BfvDevice& d, u64* pk0, const u64* e, hipStream_t s)
{
    k_add_negate<<<dim3(p.n / kBlock, p.R), kBlock, 0, s>>>(pk0, e, p.n, d.d_prime);
    return hipGetLastError();
}

hipError_t bfv_encrypt_tail(const BfvParams& p, const BfvDevice& d, u64* c, const u64* e, const u64* m, hipStream_t s)
{
    k_encrypt_tail<<<dim3(p.n / kBlock, 2), kBlock, 0, s>>>(c, e, m, p.n, p.R, p.t, d.d_prime);
    return hipGetLastError();
}

hipError_t bfv_decrypt_scale(const BfvParams& p, const BfvDevice& d, u64* c, hipStream_t s)
{
    k_decrypt_scale<<<dim3(p.n / kBlock, p.r), kBlock, 0, s>>>(c, p.n, p.R, d.d_prime);
    return hipGetLastError();
}

hipError_t bfv_decrypt_round(const BfvParams& p, const BfvDevice& d, u64* c, hipStream_t s)
{
    k_decrypt_round<<<dim3(p.n / kBlock), kBlock, 0, s>>>(c, p.n, p.R, p.t, p.gamma, p.mu_gamma, p.gamma_bits, p.gamma_div_2,
                                                          p.neg_inv_q_mod_t, p.neg_inv_q_mod_gamma, d.d_base_change);
    return hipGetLastError();
}

}  // namespace mi355ntt
