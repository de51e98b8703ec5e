// kernels_bfv.hip -- element-wise kernels of the BFV drivers.  Streaming integer work: every
// kernel reads and writes each word once, 256 threads per block, per-prime constants by scalar loads.  The reference
// issues one launch per step; the steps of one driver touch the same index (or the same column across the RNS
// polynomials), so they are fused here -- each fused kernel performs the reference's steps in the reference's order on
// its element and leaves the same words behind.
#include "bfv.hpp"
#include "modarith.cuh"

using u32 = unsigned;

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

// keygen in the NTT domain: pk0 <- -(a_hat (.) s_hat + pk0) with pk0 holding NTT(e) on entry.  The reference forms
// NTT(-(INTT(a_hat (.) s_hat) + e)) (bfv_keygen.cuh:131-145: barrett_batch_3param, inverseNTT_batch, poly_add_negate_xq,
// forwardNTT_batch); the transform is linear and every step exact, so the canonical words are the same -- without the inverse
// transform.
__global__ void __launch_bounds__(kBlock)
k_keygen_pk0(u64* __restrict__ pk0, const u64* __restrict__ a_hat, const u64* __restrict__ s_hat, unsigned n,
             const BfvPrime* __restrict__ primes)
{
    const unsigned y = blockIdx.y;
    const BfvPrime p = primes[y];
    const size_t i = (size_t)y * n + blockIdx.x * kBlock + threadIdx.x;
    u64 ra = barrett_mul(a_hat[i], s_hat[i], p.q, p.mu, p.k) + pk0[i];
    if (ra >= p.q) ra -= p.q;
    ra = p.q - ra;
    pk0[i] = ra * (ra != p.q);
}

// one column i of one half h of the ciphertext: poly_add_xq on all R polynomials (note `>`, bfv_encryption.cuh:180),
// +half on the last one (:110-124), subtract-and-scale on the others (:126-171), message term on c0 (:186-208)
__global__ void __launch_bounds__(kBlock)
k_encrypt_tail(u64* __restrict__ c, const u64* __restrict__ e, const u64* __restrict__ m, unsigned n, unsigned R, u64 t,
               const BfvPrime* __restrict__ primes, size_t half_stride)
{
    // blockIdx.z: ciphertext of a batch laid out [2][count][R][n] (half_stride = count R n; one ciphertext: R n, gridDim.z = 1)
    const unsigned h = blockIdx.y;
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    const unsigned r = R - 1;
    u64* ch = c + (size_t)blockIdx.z * R * n + h * half_stride;
    const u64* eh = e + (size_t)blockIdx.z * R * n + h * half_stride;
    m += (size_t)blockIdx.z * n;
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
k_decrypt_scale(u64* __restrict__ c, unsigned n, unsigned R, const BfvPrime* __restrict__ primes, size_t half_stride)
{
    const unsigned y = blockIdx.y;
    const BfvPrime p = primes[y];
    const size_t i = (size_t)y * n + blockIdx.x * kBlock + threadIdx.x;
    c += (size_t)blockIdx.z * R * n;
    u64* c1 = c + half_stride;
    u64 ra = c1[i] + c[i];
    if (ra > p.q) ra -= p.q;
    ra = barrett_mul(ra, p.prod_t_gamma_mod_q, p.q, p.mu, p.k);
    ra = barrett_mul(ra, p.inv_punctured_q, p.q, p.mu, p.k);
    c1[i] = ra;
}

// poly_arithmetic.cuh:221-268, :128-142, barrett_int (:100-126) per column k
__global__ void __launch_bounds__(kBlock)
k_decrypt_round(u64* __restrict__ c, unsigned n, unsigned R, u64 t, u64 gamma, u64 mu_gamma, unsigned gamma_bits, u64 gamma_div_2,
                u64 neg_inv_t, u64 neg_inv_gamma, const u64* __restrict__ bcm, size_t half_stride)
{
    const unsigned k = blockIdx.x * kBlock + threadIdx.x;
    const unsigned r = R - 1;
    c += (size_t)blockIdx.z * R * n;
    const u64* c1 = c + half_stride;
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

// ---- samplers --------------------------------------------------------------------------------

__device__ __forceinline__ u32 rotl32(u32 u, int c) { return (u << c) | (u >> (32 - c)); }

// VecCrypt with one block per thread over a zeroed buffer = the Salsa20/20 keystream (distributions.cuh:48-155):
// constants "expand 32-byte k", key words k[0..7], 64-bit nonce, block counter = block index.  16 B stores.
__global__ void __launch_bounds__(128)
k_salsa20_keystream(uint4* __restrict__ out, unsigned long long nblocks, BfvSalsaKey key, u64 nonce)
{
    const unsigned long long blockno = (unsigned long long)blockIdx.x * 128 + threadIdx.x;
    if (blockno >= nblocks) return;
    u32 j[16], x[16];
    j[0] = 0x61707865u; j[5] = 0x3320646eu; j[10] = 0x79622d32u; j[15] = 0x6b206574u;      // "expa" "nd 3" "2-by" "te k"
    j[1] = key.k[0]; j[2] = key.k[1]; j[3] = key.k[2]; j[4] = key.k[3];
    j[11] = key.k[4]; j[12] = key.k[5]; j[13] = key.k[6]; j[14] = key.k[7];
    j[6] = (u32)nonce; j[7] = (u32)(nonce >> 32);
    j[8] = (u32)blockno; j[9] = (u32)(blockno >> 32);
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = j[i];
#pragma unroll 1
    for (int i = 20; i > 0; i -= 2) {
        x[4] ^= rotl32(x[0] + x[12], 7);   x[8] ^= rotl32(x[4] + x[0], 9);    x[12] ^= rotl32(x[8] + x[4], 13);   x[0] ^= rotl32(x[12] + x[8], 18);
        x[9] ^= rotl32(x[5] + x[1], 7);    x[13] ^= rotl32(x[9] + x[5], 9);   x[1] ^= rotl32(x[13] + x[9], 13);   x[5] ^= rotl32(x[1] + x[13], 18);
        x[14] ^= rotl32(x[10] + x[6], 7);  x[2] ^= rotl32(x[14] + x[10], 9);  x[6] ^= rotl32(x[2] + x[14], 13);   x[10] ^= rotl32(x[6] + x[2], 18);
        x[3] ^= rotl32(x[15] + x[11], 7);  x[7] ^= rotl32(x[3] + x[15], 9);   x[11] ^= rotl32(x[7] + x[3], 13);   x[15] ^= rotl32(x[11] + x[7], 18);
        x[1] ^= rotl32(x[0] + x[3], 7);    x[2] ^= rotl32(x[1] + x[0], 9);    x[3] ^= rotl32(x[2] + x[1], 13);    x[0] ^= rotl32(x[3] + x[2], 18);
        x[6] ^= rotl32(x[5] + x[4], 7);    x[7] ^= rotl32(x[6] + x[5], 9);    x[4] ^= rotl32(x[7] + x[6], 13);    x[5] ^= rotl32(x[4] + x[7], 18);
        x[11] ^= rotl32(x[10] + x[9], 7);  x[8] ^= rotl32(x[11] + x[10], 9);  x[9] ^= rotl32(x[8] + x[11], 13);   x[10] ^= rotl32(x[9] + x[8], 18);
        x[12] ^= rotl32(x[15] + x[14], 7); x[13] ^= rotl32(x[12] + x[15], 9); x[14] ^= rotl32(x[13] + x[12], 13); x[15] ^= rotl32(x[14] + x[13], 18);
    }
    uint4* o = out + blockno * 4;
    o[0] = make_uint4(x[0] + j[0], x[1] + j[1], x[2] + j[2], x[3] + j[3]);
    o[1] = make_uint4(x[4] + j[4], x[5] + j[5], x[6] + j[6], x[7] + j[7]);
    o[2] = make_uint4(x[8] + j[8], x[9] + j[9], x[10] + j[10], x[11] + j[11]);
    o[3] = make_uint4(x[12] + j[12], x[13] + j[13], x[14] + j[14], x[15] + j[15]);
}

// the conversion of one byte / one 32-bit word, exactly as the reference's kernels write it
__device__ __forceinline__ u64 ternary_from_byte(unsigned char byte, u64 q)
{
    float d = (float)byte;
    d /= (255.0f / 3);
    const int b = int(d) - 1;
    return (u64)(b < 0) * q + (u64)(long long)b;
}
__device__ __forceinline__ u64 gaussian_from_word(u32 w, u64 q)
{
    float d = (float)w;
    d /= 4294967295;
    if (d == 0) d += 1.192092896e-07F;
    else if (d == 1) d -= 1.192092896e-07F;
    d = normcdfinvf(d);
    d = d * (float)3.2 + 0;                 // dstdev, dmean (salsa_common.h:31-32)
    if (d > 19.2) d = 19.2;
    else if (d < -19.2) d = -19.2;
    const int dd = (int)d;
    return dd < 0 ? q + (u64)(long long)dd : (u64)dd;
}

// ternary_dist_xq + uniform_dist_xq + gaussian_dist_xq (bfv_keygen.cuh:14-79): grid (n / 256, R)
__global__ void __launch_bounds__(kBlock)
k_sample_keygen(const unsigned char* __restrict__ in, u64* __restrict__ secret_key, u64* __restrict__ pk1, u64* __restrict__ temp,
                unsigned n, unsigned R, const BfvPrime* __restrict__ primes)
{
    const unsigned y = blockIdx.y, i = blockIdx.x * kBlock + threadIdx.x;
    const u64 q = primes[y].q;
    const size_t x = (size_t)y * n + i;
    secret_key[x] = ternary_from_byte(in[i], q);
    const u64* inl = reinterpret_cast<const u64*>(in + n);
    double d = (double)inl[x];
    d /= 18446744073709551615ULL;           // UINT64_MAX
    d *= (double)(q - 1);
    pk1[x] = (u64)d;
    const u32* inw = reinterpret_cast<const u32*>(in + n + (size_t)R * n * 8);
    temp[x] = gaussian_from_word(inw[i], q);
}

// convert_ternary_gaussian_x2 (bfv_encryption.cuh:17-109): grid (n / 256, R)
__global__ void __launch_bounds__(kBlock)
k_sample_encrypt(const unsigned char* __restrict__ in, u64* __restrict__ c, u64* __restrict__ e, unsigned n, unsigned R,
                 const BfvPrime* __restrict__ primes)
{
    const unsigned y = blockIdx.y, i = blockIdx.x * kBlock + threadIdx.x;
    const u64 q = primes[y].q;
    const size_t x = (size_t)y * n + i, half = (size_t)R * n;
    const u64 tv = ternary_from_byte(in[i], q);
    c[x] = tv;
    c[x + half] = tv;
    e[x] = gaussian_from_word(reinterpret_cast<const u32*>(in + n)[i], q);
    e[x + half] = gaussian_from_word(reinterpret_cast<const u32*>(in + (size_t)n * 5)[i], q);
}

}  // namespace

hipError_t bfv_salsa20_keystream(void* d_out, size_t nbytes, const BfvSalsaKey& key, u64 nonce, hipStream_t s)
{
    const unsigned long long nblocks = nbytes / 64;                  // NBLKS = n / 64, distributions.cuh:200,227
    if (nblocks == 0) return hipSuccess;
    k_salsa20_keystream<<<dim3((unsigned)((nblocks + 127) / 128)), 128, 0, s>>>(static_cast<uint4*>(d_out), nblocks, key, nonce);
    return hipGetLastError();
}

hipError_t bfv_sample_keygen(const BfvParams& p, const BfvDevice& d, const unsigned char* in, u64* secret_key, u64* public_key,
                             u64* temp, hipStream_t s)
{
    k_sample_keygen<<<dim3(p.n / kBlock, p.R), kBlock, 0, s>>>(in, secret_key, public_key + (size_t)p.R * p.n, temp, p.n, p.R, d.d_prime);
    return hipGetLastError();
}

hipError_t bfv_sample_encrypt(const BfvParams& p, const BfvDevice& d, const unsigned char* in, u64* c, u64* e, hipStream_t s)
{
    k_sample_encrypt<<<dim3(p.n / kBlock, p.R), kBlock, 0, s>>>(in, c, e, p.n, p.R, d.d_prime);
    return hipGetLastError();
}

hipError_t bfv_add_negate(const BfvParams& p, const BfvDevice& d, u64* pk0, const u64* e, hipStream_t s)
{
    k_add_negate<<<dim3(p.n / kBlock, p.R), kBlock, 0, s>>>(pk0, e, p.n, d.d_prime);
    return hipGetLastError();
}

hipError_t bfv_keygen_pk0(const BfvParams& p, const BfvDevice& d, u64* pk0, const u64* a_hat, const u64* s_hat, hipStream_t s)
{
    k_keygen_pk0<<<dim3(p.n / kBlock, p.R), kBlock, 0, s>>>(pk0, a_hat, s_hat, p.n, d.d_prime);
    return hipGetLastError();
}

// count > 1: a batch of ciphertexts laid out [2][count][R][n] (all first halves, then all second halves)
hipError_t bfv_encrypt_tail(const BfvParams& p, const BfvDevice& d, u64* c, const u64* e, const u64* m, hipStream_t s, unsigned count)
{
    k_encrypt_tail<<<dim3(p.n / kBlock, 2, count), kBlock, 0, s>>>(c, e, m, p.n, p.R, p.t, d.d_prime, (size_t)count * p.R * p.n);
    return hipGetLastError();
}

hipError_t bfv_decrypt_scale(const BfvParams& p, const BfvDevice& d, u64* c, hipStream_t s, unsigned count)
{
    k_decrypt_scale<<<dim3(p.n / kBlock, p.r, count), kBlock, 0, s>>>(c, p.n, p.R, d.d_prime, (size_t)count * p.R * p.n);
    return hipGetLastError();
}

hipError_t bfv_decrypt_round(const BfvParams& p, const BfvDevice& d, u64* c, hipStream_t s, unsigned count)
{
    k_decrypt_round<<<dim3(p.n / kBlock, 1, count), kBlock, 0, s>>>(c, p.n, p.R, p.t, p.gamma, p.mu_gamma, p.gamma_bits, p.gamma_div_2,
                                                                    p.neg_inv_q_mod_t, p.neg_inv_q_mod_gamma, d.d_base_change,
                                                                    (size_t)count * p.R * p.n);
    return hipGetLastError();
}

}  // namespace mi355ntt
