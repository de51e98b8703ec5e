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

// V consecutive words per lane: V = 2 moves 16 bytes per lane and instruction (pointers 16-byte aligned: every device allocation
// and every whole-polynomial offset is), V = 1 is the form for callers that hand over merely 8-byte aligned pointers.
template <int V>
__device__ __forceinline__ void ldv(u64 (&x)[V], const u64* __restrict__ p)
{
    if constexpr (V == 2) {
        const ulonglong2 t = *reinterpret_cast<const ulonglong2*>(p);
        x[0] = t.x; x[1] = t.y;
    } else {
        x[0] = p[0];
    }
}
template <int V>
__device__ __forceinline__ void stv(u64* __restrict__ p, const u64 (&x)[V])
{
    if constexpr (V == 2) *reinterpret_cast<ulonglong2*>(p) = make_ulonglong2(x[0], x[1]);
    else p[0] = x[0];
}

// x mod q for ANY 64-bit x, exact -- the reference writes `%` (bfv_encryption.cuh:150,204; poly_arithmetic.cuh:252,262), which
// gfx950 runs as a software division of some sixty instructions.  m64 = floor((2^64 - 1) / q) from the host: the estimate
// e = floor(x m64 / 2^64) is the quotient or up to two less (x m64 / 2^64 > x/q - x/(q 2^64) - x/2^64 > x/q - 2), and
// x - e q <= x never leaves 64 bits, so two conditional subtractions finish it for every q.
__device__ __forceinline__ u64 reduce64(u64 x, u64 q, u64 m64)
{
    u64 r = x - mul_hi(x, m64) * q;
    r = r >= q ? r - q : r;
    return r >= q ? r - q : r;
}

// poly_add_negate_xq, bfv_keygen.cuh:80-93
template <int V>
__global__ void __launch_bounds__(kBlock)
k_add_negate(u64* __restrict__ a, const u64* __restrict__ b, unsigned n, const BfvPrime* __restrict__ primes)
{
    const unsigned y = blockIdx.y;
    const u64 q = primes[y].q;
    const size_t i = (size_t)y * n + (size_t)(blockIdx.x * kBlock + threadIdx.x) * V;
    u64 va[V], vb[V];
    ldv<V>(va, a + i);
    ldv<V>(vb, b + i);
#pragma unroll
    for (int v = 0; v < V; v++) {
        u64 ra = va[v] + vb[v];
        if (ra >= q) ra -= q;
        ra = q - ra;
        va[v] = ra * (ra != q);
    }
    stv<V>(a + i, va);
}

// keygen in the NTT domain: pk0 <- -(a_hat (.) s_hat + pk0) with pk0 holding NTT(e) on entry.  The reference forms
// NTT(-(INTT(a_hat (.) s_hat) + e)) (bfv_keygen.cuh:131-145: barrett_batch_3param, inverseNTT_batch, poly_add_negate_xq,
// forwardNTT_batch); the transform is linear and every step exact, so the canonical words are the same -- without the inverse
// transform.
template <int V>
__global__ void __launch_bounds__(kBlock)
k_keygen_pk0(u64* __restrict__ pk0, const u64* __restrict__ a_hat, const u64* __restrict__ s_hat, unsigned n,
             const BfvPrime* __restrict__ primes)
{
    const unsigned y = blockIdx.y;
    const BfvPrime p = primes[y];
    const size_t i = (size_t)y * n + (size_t)(blockIdx.x * kBlock + threadIdx.x) * V;
    u64 va[V], vs[V], vp[V];
    ldv<V>(va, a_hat + i);
    ldv<V>(vs, s_hat + i);
    ldv<V>(vp, pk0 + i);
#pragma unroll
    for (int v = 0; v < V; v++) {
        u64 ra = barrett_mul(va[v], vs[v], p.q, p.mu, p.k) + vp[v];
        if (ra >= p.q) ra -= p.q;
        ra = p.q - ra;
        vp[v] = ra * (ra != p.q);
    }
    stv<V>(pk0 + i, vp);
}

// V columns i.. of one half h of the ciphertext: poly_add_xq on all R polynomials (note `>`, bfv_encryption.cuh:180),
// +half on the last one (:110-124), subtract-and-scale on the others (:126-171), message term on c0 (:186-208).
// The reference's two `%` (:150 last % q_j, :204 the message term) are reduce64 -- the same words for every input; its
// `/ t` (:203) is 0 or 1 for a message below t and a real division otherwise.
template <int V>
__global__ void __launch_bounds__(kBlock)
k_encrypt_tail(u64* __restrict__ c, const u64* __restrict__ e, const u64* __restrict__ m, unsigned n, unsigned R, u64 t,
               const BfvPrime* __restrict__ primes, size_t half_stride)
{
    // blockIdx.z: ciphertext of a batch laid out [2][count][R][n] (half_stride = count R n; one ciphertext: R n, gridDim.z = 1)
    const unsigned h = blockIdx.y;
    const unsigned i = (blockIdx.x * kBlock + threadIdx.x) * V;
    const unsigned r = R - 1;
    u64* ch = c + (size_t)blockIdx.z * R * n + h * half_stride;
    const u64* eh = e + (size_t)blockIdx.z * R * n + h * half_stride;
    m += (size_t)blockIdx.z * n;
    const u64 q_last = primes[r].q, half_last = q_last >> 1;
    u64 last[V], el[V];
    ldv<V>(last, ch + (size_t)r * n + i);
    ldv<V>(el, eh + (size_t)r * n + i);
#pragma unroll
    for (int v = 0; v < V; v++) {
        last[v] += el[v];
        if (last[v] > q_last) last[v] -= q_last;               // poly_add_xq
        last[v] += half_last;                                  // ..._add_x2
        if (last[v] >= q_last) last[v] -= q_last;
    }
    stv<V>(ch + (size_t)r * n + i, last);
    u64 mi[V] = {}, fix[V] = {};
    if (h == 0) {
        ldv<V>(mi, m + i);
#pragma unroll
        for (int v = 0; v < V; v++) {
            const u64 num = mi[v] + ((t + 1) >> 1);            // weird_m_stuff: numerator / t
            fix[v] = num < t ? 0 : (num - t < t ? 1 : num / t);
        }
    }
    for (unsigned j = 0; j < r; j++) {
        const BfvPrime p = primes[j];
        u64 x[V], ex[V];
        ldv<V>(x, ch + (size_t)j * n + i);
        ldv<V>(ex, eh + (size_t)j * n + i);
#pragma unroll
        for (int v = 0; v < V; v++) {
            x[v] += ex[v];
            if (x[v] > p.q) x[v] -= p.q;                       // poly_add_xq
            u64 tmp = reduce64(last[v], p.q, p.m64);           // ..._loop_xq
            if (tmp < p.half_last_mod_q) tmp += p.q;
            tmp -= p.half_last_mod_q;
            if (x[v] < tmp) x[v] += p.q;
            x[v] -= tmp;
            x[v] = barrett_mul(x[v], p.inv_q_last_mod_q, p.q, p.mu, p.k);
            if (h == 0) x[v] = reduce64(x[v] + (mi[v] * p.q_div_t + fix[v]), p.q, p.m64);    // weird_m_stuff
        }
        stv<V>(ch + (size_t)j * n + i, x);
    }
}

// c1[i] = ((c1[i] + c0[i], `>`) * prod_t_gamma) * inv_punctured_q, bfv_decryption.cuh:13-57
template <int V>
__global__ void __launch_bounds__(kBlock)
k_decrypt_scale(u64* __restrict__ c, unsigned n, unsigned R, const BfvPrime* __restrict__ primes, size_t half_stride)
{
    const unsigned y = blockIdx.y;
    const BfvPrime p = primes[y];
    const size_t i = (size_t)y * n + (size_t)(blockIdx.x * kBlock + threadIdx.x) * V;
    c += (size_t)blockIdx.z * R * n;
    u64* c1 = c + half_stride;
    u64 a1[V], a0[V];
    ldv<V>(a1, c1 + i);
    ldv<V>(a0, c + i);
#pragma unroll
    for (int v = 0; v < V; v++) {
        u64 ra = a1[v] + a0[v];
        if (ra > p.q) ra -= p.q;
        ra = barrett_mul(ra, p.prod_t_gamma_mod_q, p.q, p.mu, p.k);
        a1[v] = barrett_mul(ra, p.inv_punctured_q, p.q, p.mu, p.k);
    }
    stv<V>(c1 + i, a1);
}

// poly_arithmetic.cuh:221-268, :128-142, barrett_int (:100-126) per column k.  The reference reduces its running sum with `% gamma`
// after every term (:252); every step is exact arithmetic mod gamma, so the sum may as well run LAZILY: a Barrett product is below
// 2 gamma (the quotient estimate of Algorithm 7 is at most two short), so `lazy` = floor((2^64 - 1 - gamma) / (2 gamma)) terms fit on top
// of a reduced accumulator without leaving 64 bits (3 for the reference's 61-bit gamma, never less than 1) -- one reduce64 per `lazy`
// terms instead of one per term, the same residue in the end; the final `% gamma` (:262) then finds a reduced value.
template <int V>
__global__ void __launch_bounds__(kBlock)
k_decrypt_round(u64* __restrict__ c, unsigned n, unsigned R, u64 t, u64 gamma, u64 mu_gamma, unsigned gamma_bits, u64 gamma_div_2,
                u64 neg_inv_t, u64 neg_inv_gamma, const u64* __restrict__ bcm, size_t half_stride, u64 m64_gamma, unsigned lazy)
{
    const unsigned k = (blockIdx.x * kBlock + threadIdx.x) * V;
    const unsigned r = R - 1;
    c += (size_t)blockIdx.z * R * n;
    const u64* c1 = c + half_stride;
    const unsigned mask32 = (unsigned)(t - 1);                 // `unsigned mask = t - 1`
    const u64 mask = t - 1;                                    // dec_round_kernel
    u64 acc_t[V] = {}, acc_g[V] = {};
    unsigned pending = 0;
    for (unsigned i = 0; i < r; i++) {
        u64 val[V];
        ldv<V>(val, c1 + k + (size_t)i * n);
        const u64 bt = bcm[i], bg = bcm[i + r];
#pragma unroll
        for (int v = 0; v < V; v++) {
            acc_t[v] += (val[v] * bt) & mask32;                                          // fast_convert_array_kernel_t
            acc_g[v] += barrett_mul(val[v], bg, gamma, mu_gamma, gamma_bits);            // fast_convert_array_kernel_gamma (sum: lazily)
        }
        if (++pending == lazy || i + 1 == r) {
#pragma unroll
            for (int v = 0; v < V; v++) acc_g[v] = reduce64(acc_g[v], gamma, m64_gamma);
            pending = 0;
        }
    }
    u64 x0[V], x1[V], res[V];
#pragma unroll
    for (int v = 0; v < V; v++) {
        x0[v] = acc_t[v] & mask32;
        x1[v] = acc_g[v];                                                                // (acc_g % gamma: already reduced)
        x0[v] = (x0[v] * neg_inv_t) & mask32;                                            // poly_mul_int_t -> mod_t
        x1[v] = barrett_mul(x1[v], neg_inv_gamma, gamma, mu_gamma, gamma_bits);          // poly_mul_int -> barrett_int
        if (x1[v] > gamma_div_2) res[v] = (x0[v] + (gamma - x1[v])) & mask;
        else res[v] = (x0[v] - x1[v]) & mask;
    }
    stv<V>(c + k, x0);                     // (in this order: for R <= 3 the three destinations coincide pairwise, as in the reference)
    stv<V>(c + k + n, x1);
    stv<V>(c + k + (size_t)n * (r - 1), res);
}

template <class... P>
__host__ bool aligned16(P... p) { return ((reinterpret_cast<uintptr_t>(p) | ...) & 15u) == 0; }

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
    if (aligned16(pk0, e)) k_add_negate<2><<<dim3(p.n / (2 * kBlock), p.R), kBlock, 0, s>>>(pk0, e, p.n, d.d_prime);
    else k_add_negate<1><<<dim3(p.n / kBlock, p.R), kBlock, 0, s>>>(pk0, e, p.n, d.d_prime);
    return hipGetLastError();
}

hipError_t bfv_keygen_pk0(const BfvParams& p, const BfvDevice& d, u64* pk0, const u64* a_hat, const u64* s_hat, hipStream_t s)
{
    if (aligned16(pk0, a_hat, s_hat)) k_keygen_pk0<2><<<dim3(p.n / (2 * kBlock), p.R), kBlock, 0, s>>>(pk0, a_hat, s_hat, p.n, d.d_prime);
    else k_keygen_pk0<1><<<dim3(p.n / kBlock, p.R), kBlock, 0, s>>>(pk0, a_hat, s_hat, p.n, d.d_prime);
    return hipGetLastError();
}

// count > 1: a batch of ciphertexts laid out [2][count][R][n] (all first halves, then all second halves)
hipError_t bfv_encrypt_tail(const BfvParams& p, const BfvDevice& d, u64* c, const u64* e, const u64* m, hipStream_t s, unsigned count)
{
    const size_t hs = (size_t)count * p.R * p.n;
    if (aligned16(c, e, m)) k_encrypt_tail<2><<<dim3(p.n / (2 * kBlock), 2, count), kBlock, 0, s>>>(c, e, m, p.n, p.R, p.t, d.d_prime, hs);
    else k_encrypt_tail<1><<<dim3(p.n / kBlock, 2, count), kBlock, 0, s>>>(c, e, m, p.n, p.R, p.t, d.d_prime, hs);
    return hipGetLastError();
}

hipError_t bfv_decrypt_scale(const BfvParams& p, const BfvDevice& d, u64* c, hipStream_t s, unsigned count)
{
    const size_t hs = (size_t)count * p.R * p.n;
    if (aligned16(c)) k_decrypt_scale<2><<<dim3(p.n / (2 * kBlock), p.r, count), kBlock, 0, s>>>(c, p.n, p.R, d.d_prime, hs);
    else k_decrypt_scale<1><<<dim3(p.n / kBlock, p.r, count), kBlock, 0, s>>>(c, p.n, p.R, d.d_prime, hs);
    return hipGetLastError();
}

hipError_t bfv_decrypt_round(const BfvParams& p, const BfvDevice& d, u64* c, hipStream_t s, unsigned count)
{
    const size_t hs = (size_t)count * p.R * p.n;
    if (aligned16(c))
        k_decrypt_round<2><<<dim3(p.n / (2 * kBlock), 1, count), kBlock, 0, s>>>(c, p.n, p.R, p.t, p.gamma, p.mu_gamma, p.gamma_bits, p.gamma_div_2,
                                                                              p.neg_inv_q_mod_t, p.neg_inv_q_mod_gamma, d.d_base_change, hs, p.m64_gamma, p.lazy_gamma);
    else
        k_decrypt_round<1><<<dim3(p.n / kBlock, 1, count), kBlock, 0, s>>>(c, p.n, p.R, p.t, p.gamma, p.mu_gamma, p.gamma_bits, p.gamma_div_2,
                                                                          p.neg_inv_q_mod_t, p.neg_inv_q_mod_gamma, d.d_base_change, hs, p.m64_gamma, p.lazy_gamma);
    return hipGetLastError();
}

}  // namespace mi355ntt
