// bfv_host.cpp -- parameter bootstrap and drivers of the BFV launch layer (C ABI section "BFV" of include/mi355ntt.h).
// The drivers are the reference's launch sequences (bfv_keygen.cuh:95-151, bfv_encryption.cuh:223-290,
// bfv_decryption.cuh:76-138) after their samplers, issued on the caller's stream: NTT sections through the public
// transforms of the context, element-wise steps through kernels_bfv.hip.
#include "../../include/mi355ntt.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <new>

#include "bfv.hpp"
#include "device_scope.hpp"

using namespace mi355ntt;

namespace mi355ntt {

int bfv_bootstrap(BfvParams* out, unsigned n, unsigned R, const u64* q, u64 t, u64 gamma)
{
    if (R < 2 || R > kMaxPrimes) return MI355NTT_EUNSUPPORTED;
    if (t < 2 || (t & (t - 1)) != 0 || t > (1ull << 31)) return MI355NTT_EUNSUPPORTED;   // masks: `unsigned mask = t - 1`
    const unsigned gbits = bit_length(gamma);
    if (gbits < 3 || gbits > 62 || (gamma & 1) == 0) return MI355NTT_EUNSUPPORTED;
    BfvParams p;
    p.n = n;
    p.R = R;
    p.r = R - 1;
    p.t = t;
    p.gamma = gamma;
    p.gamma_bits = gbits;                          // output_base_bit_lengths[1] (61 for the reference's gamma, demo.cu:100)
    p.mu_gamma = barrett_mu(gamma, gbits);         // demo.cu:218-226
    p.gamma_div_2 = gamma >> 1;                    // demo.cu:94
    p.m64_gamma = ~0ULL / gamma;
    // Lazy summation mod gamma in k_decrypt_round: each Barrett product is below 2 gamma ONLY while its operand val < q_i stays
    // below 2^gamma_bits (Algorithm 7's quotient estimate is then at most two short).  With a q_i wider than gamma the
    // remainder can reach about val + 3 gamma and several terms overflow the 64-bit accumulator where the reference's per-term
    // `% gamma` (poly_arithmetic.cuh:252) never does: such parameter sets keep the per-term reduction (lazy = 1).
    p.lazy_gamma = (unsigned)((~0ULL - gamma) / (2 * (u128)gamma));      // (>= 1 for gamma < 2^62)
    if (p.lazy_gamma < 1) p.lazy_gamma = 1;
    for (unsigned i = 0; i + 1 < R; i++)
        if (bit_length(q[i]) > gbits) p.lazy_gamma = 1;
    p.q_last = q[R - 1];
    p.half_q_last = p.q_last >> 1;
    const unsigned r = p.r;
    u64 mult_t = 1, mult_g = 1;                    // demo.cu:103-117
    for (unsigned i = 0; i < r; i++) {
        if (q[i] % t != 1) return MI355NTT_EPARAM;   // "q mod t is assumed 1", bfv_encryption.cuh:189
        if (q[i] % gamma == 0) return MI355NTT_EPARAM;
        mult_t = mulmod(mult_t, q[i], t);
        mult_g = mulmod(mult_g, q[i], gamma);
    }
    p.neg_inv_q_mod_t = t - modpow(mult_t, t - 2, t);          // the reference's modinv128 is a^(m-2) for every modulus
    p.neg_inv_q_mod_gamma = gamma - modinv(mult_g, gamma);
    const u128 prod_t_gamma = (u128)t * gamma;                 // demo.cu:119-125
    for (unsigned i = 0; i < R; i++) {
        BfvPrime& bp = p.prime[i];
        std::memset(&bp, 0, sizeof(bp));
        bp.q = q[i];
        bp.k = bit_length(q[i]);
        bp.mu = barrett_mu(q[i], bp.k);
        bp.q_div_t = q[i] / t;                                 // demo.cu:84-88
        bp.m64 = ~0ULL / q[i];
        if (i < r) {
            bp.prod_t_gamma_mod_q = (u64)(prod_t_gamma % q[i]);
            u64 punct = 1;                                     // demo.cu:262-276
            for (unsigned j = 0; j < r; j++)
                if (j != i) punct = mulmod(punct, q[j], q[i]);
            bp.inv_punctured_q = modinv(punct, q[i]);
            bp.inv_q_last_mod_q = modinv(q[R - 1] % q[i], q[i]);   // demo.cu:73-79
            bp.half_last_mod_q = p.half_q_last % q[i];
        }
    }
    const u64 base[2] = {t, gamma};                            // demo.cu:281-301
    for (unsigned b = 0; b < 2; b++)
        for (unsigned j = 0; j < r; j++) {
            u64 temp = 1;
            for (unsigned kk = 0; kk < r; kk++)
                if (kk != j) temp = mulmod(temp, q[kk], base[b]);
            p.base_change[b * r + j] = temp;
        }
    *out = p;
    return MI355NTT_OK;
}

}  // namespace mi355ntt

namespace mi355ntt {
void record_hip_error(int e);      // capi.cpp: what mi355ntt_last_hip_error() reports
}

struct mi355ntt_bfv {
    mi355ntt_ctx* ntt = nullptr;
    BfvParams p;
    BfvDevice d;
    void* d_prime = nullptr;
    void* d_bcm = nullptr;
    void* d_epi = nullptr;        // [R] BfvEpiPrime: the epilogue constants of the batched decryption
    bool epi_ok = false;          // the one-product form of the scaling is the reference's words for these moduli
};

#define BFV_HIP(expr)                         \
    do {                                      \
        hipError_t e__ = (expr);              \
        if (e__ != hipSuccess) {              \
            record_hip_error((int)e__);       \
            return MI355NTT_EHIP;             \
        }                                     \
    } while (0)
#define BFV_ON_DEVICE(b)                                   \
    DeviceScope scope__(mi355ntt_ctx_device((b)->ntt));    \
    BFV_HIP(scope__.err)
#define BFV_RC(expr)            \
    do {                        \
        int rc__ = (expr);      \
        if (rc__) return rc__;  \
    } while (0)

extern "C" {

int mi355ntt_bfv_create(mi355ntt_bfv** out, unsigned n, unsigned num_primes, const mi355ntt_u64* q, const mi355ntt_u64* psi,
                        mi355ntt_u64 t, mi355ntt_u64 gamma, int device, unsigned ctx_flags)
{
    if (!out || !q || !psi) return MI355NTT_EINVAL;
    *out = nullptr;
    mi355ntt_bfv* b = new (std::nothrow) mi355ntt_bfv();
    if (!b) return MI355NTT_ENOMEM;
    int rc = bfv_bootstrap(&b->p, n, num_primes, q, t, gamma);
    if (!rc) rc = mi355ntt_ctx_create_ex(&b->ntt, n, num_primes, q, psi, device, ctx_flags);
    if (rc) {
        mi355ntt_bfv_destroy(b);
        return rc;
    }
    DeviceScope scope(device);
    hipError_t e = scope.err;
    if (e != hipSuccess ||
        (e = hipMalloc(&b->d_prime, sizeof(BfvPrime) * num_primes)) != hipSuccess ||
        (e = hipMalloc(&b->d_bcm, sizeof(u64) * 2 * b->p.r)) != hipSuccess ||
        (e = hipMemcpy(b->d_prime, b->p.prime, sizeof(BfvPrime) * num_primes, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(b->d_bcm, b->p.base_change, sizeof(u64) * 2 * b->p.r, hipMemcpyHostToDevice)) != hipSuccess) {
        record_hip_error((int)e);
        mi355ntt_bfv_destroy(b);
        return e == hipErrorOutOfMemory ? MI355NTT_ENOMEM : MI355NTT_EHIP;
    }
    b->d.d_prime = static_cast<const BfvPrime*>(b->d_prime);
    b->d.d_base_change = static_cast<const u64*>(b->d_bcm);
    {
        // the element-wise step the fused product's epilogue takes over in the batched decryption (kernels_epi.cuh): the r primes of
        // the ciphertext modulus are scaled by k_i = (t gamma) q~_i^-1 mod q_i, the dropped prime's slot is left alone.  ONE exact
        // product stands for the reference's two Barrett products only where those are exact -- for every operand the sum can be,
        // q itself included (`>`): the bound of barrett_single_subtraction_exact with q (q - 1) in place of (q - 1)^2.
        BfvEpiPrime epi[kMaxPrimes];
        std::memset(epi, 0, sizeof(epi));
        b->epi_ok = true;
        for (unsigned i = 0; i < num_primes; i++) {
            const u64 qi = b->p.prime[i].q;
            epi[i].k1 = mulmod(b->p.prime[i].prod_t_gamma_mod_q, b->p.prime[i].inv_punctured_q, qi);
            epi[i].k2 = shoup(epi[i].k1, qi);
            epi[i].on = i < b->p.r ? 1u : 0u;
            if (i < b->p.r && !barrett_exact_for_operand_q(qi, b->p.prime[i].k, b->p.prime[i].mu)) b->epi_ok = false;
        }
        if ((e = hipMalloc(&b->d_epi, sizeof(BfvEpiPrime) * num_primes)) != hipSuccess ||
            (e = hipMemcpy(b->d_epi, epi, sizeof(BfvEpiPrime) * num_primes, hipMemcpyHostToDevice)) != hipSuccess) {
            record_hip_error((int)e);
            mi355ntt_bfv_destroy(b);
            return e == hipErrorOutOfMemory ? MI355NTT_ENOMEM : MI355NTT_EHIP;
        }
    }
    *out = b;
    return MI355NTT_OK;
}

int mi355ntt_bfv_destroy(mi355ntt_bfv* b)
{
    if (!b) return MI355NTT_OK;
    DeviceScope scope(b->ntt ? mi355ntt_ctx_device(b->ntt) : 0);
    if (b->d_prime) (void)hipFree(b->d_prime);
    if (b->d_bcm) (void)hipFree(b->d_bcm);
    if (b->d_epi) (void)hipFree(b->d_epi);
    if (b->ntt) mi355ntt_ctx_destroy(b->ntt);
    delete b;
    return MI355NTT_OK;
}

const mi355ntt_ctx* mi355ntt_bfv_ntt(const mi355ntt_bfv* b) { return b ? b->ntt : nullptr; }

int mi355ntt_bfv_constants(const mi355ntt_bfv* b, mi355ntt_u64* inv_punctured_q, mi355ntt_u64* neg_inv_q_mod_t_gamma,
                           mi355ntt_u64* prod_t_gamma_mod_q, mi355ntt_u64* inv_q_last_mod_q, mi355ntt_u64* q_div_t,
                           mi355ntt_u64* base_change_matrix, mi355ntt_u64* mu_gamma)
{
    if (!b) return MI355NTT_EINVAL;
    const BfvParams& p = b->p;
    for (unsigned i = 0; i < p.r; i++) {
        if (inv_punctured_q) inv_punctured_q[i] = p.prime[i].inv_punctured_q;
        if (prod_t_gamma_mod_q) prod_t_gamma_mod_q[i] = p.prime[i].prod_t_gamma_mod_q;
        if (inv_q_last_mod_q) inv_q_last_mod_q[i] = p.prime[i].inv_q_last_mod_q;
    }
    for (unsigned i = 0; i < p.R; i++)
        if (q_div_t) q_div_t[i] = p.prime[i].q_div_t;
    if (neg_inv_q_mod_t_gamma) {
        neg_inv_q_mod_t_gamma[0] = p.neg_inv_q_mod_t;
        neg_inv_q_mod_t_gamma[1] = p.neg_inv_q_mod_gamma;
    }
    if (base_change_matrix)
        for (unsigned i = 0; i < 2 * p.r; i++) base_change_matrix[i] = p.base_change[i];
    if (mu_gamma) *mu_gamma = p.mu_gamma;
    return MI355NTT_OK;
}

/* keygen_rns after its samplers, bfv_keygen.cuh:129-145 */
int mi355ntt_bfv_keygen(const mi355ntt_bfv* b, mi355ntt_u64* d_secret_key, mi355ntt_u64* d_public_key, const mi355ntt_u64* d_e,
                        mi355ntt_stream stream)
{
    if (!b || !d_secret_key || !d_public_key || !d_e) return MI355NTT_EINVAL;
    const unsigned R = b->p.R;
    const size_t half = (size_t)R * b->p.n;
    BFV_ON_DEVICE(b);
    BFV_RC(mi355ntt_forward_batch(b->ntt, d_secret_key, R, R, stream));                                    /* :129 */
    if (!mi355ntt_ctx_uses_literal_kernels(b->ntt)) {
        /* exact arithmetic (every modulus Barrett-exact, or the caller asked for exact results): NTT(-(INTT(a s) + e)) =
         * -(a s + NTT(e)), word for word -- one forward transform of e instead of an inverse and a forward of the product */
        BFV_HIP(hipMemcpyAsync(d_public_key, d_e, half * sizeof(mi355ntt_u64), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        BFV_RC(mi355ntt_forward_batch(b->ntt, d_public_key, R, R, stream));
        BFV_HIP(bfv_keygen_pk0(b->p, b->d, d_public_key, d_public_key + half, d_secret_key, (hipStream_t)stream));
        return MI355NTT_OK;
    }
    /* literal contexts follow the reference's own sequence (its words, including where its Barrett is inexact) */
    BFV_RC(mi355ntt_pointwise_mul(b->ntt, d_public_key, d_public_key + half, d_secret_key, R, R, stream)); /* :131-132 */
    BFV_RC(mi355ntt_inverse_batch(b->ntt, d_public_key, R, R, stream));                                    /* :133 */
    BFV_HIP(bfv_add_negate(b->p, b->d, d_public_key, d_e, (hipStream_t)stream));                           /* :144 */
    return mi355ntt_forward_batch(b->ntt, d_public_key, R, R, stream);                                     /* :145 */
}

/* encryption_rns after its samplers, bfv_encryption.cuh:268-289 */
int mi355ntt_bfv_encrypt(const mi355ntt_bfv* b, mi355ntt_u64* d_c, const mi355ntt_u64* d_public_key, const mi355ntt_u64* d_e,
                         const mi355ntt_u64* d_m, mi355ntt_stream stream)
{
    if (!b || !d_c || !d_public_key || !d_e || !d_m) return MI355NTT_EINVAL;
    const unsigned R = b->p.R;
    BFV_ON_DEVICE(b);
    BFV_RC(mi355ntt_polymul_batch(b->ntt, d_c, d_public_key, 2 * R, R, stream));                           /* :268-271 */
    BFV_HIP(bfv_encrypt_tail(b->p, b->d, d_c, d_e, d_m, (hipStream_t)stream));                             /* :278-289 */
    return MI355NTT_OK;
}

/* decryption_rns, bfv_decryption.cuh:98-137; the plaintext lands at d_c + n (r - 1), r = num_primes - 1 */
int mi355ntt_bfv_decrypt(const mi355ntt_bfv* b, mi355ntt_u64* d_c, const mi355ntt_u64* d_secret_key, mi355ntt_stream stream)
{
    if (!b || !d_c || !d_secret_key) return MI355NTT_EINVAL;
    const unsigned R = b->p.R, r = b->p.r;
    BFV_ON_DEVICE(b);
    BFV_RC(mi355ntt_polymul_batch(b->ntt, d_c + (size_t)R * b->p.n, d_secret_key, r, R, stream));          /* :98-101 */
    BFV_HIP(bfv_decrypt_scale(b->p, b->d, d_c, (hipStream_t)stream));                                      /* :103-121 */
    BFV_HIP(bfv_decrypt_round(b->p, b->d, d_c, (hipStream_t)stream));                                      /* :126-137 */
    return MI355NTT_OK;
}

/* Limits of one batched call, checked up front so that a refused call leaves the ciphertexts untouched: the element-wise
 * kernels put the ciphertext index in gridDim.z (<= 65535), the fused product's key-group size count * R rides in 23 bits of
 * the division word (kernels.hpp, kSharedB), and 2 * count * R polynomials must not overflow the 32-bit polynomial count. */
static bool bfv_batch_count_ok(unsigned count, unsigned R)
{
    if (count > 65535u) return false;
    const unsigned long long group = (unsigned long long)count * R;
    return group < (1ull << 23) && 2 * group <= 0xffffffffull;
}

/* ---- batched drivers: `count` ciphertexts per call, laid out [2][count][num_primes][n] (component-major, so that each
 * component of the whole batch is one contiguous run of polynomials for the fused product) ---- */
int mi355ntt_bfv_encrypt_batch(const mi355ntt_bfv* b, mi355ntt_u64* d_c, const mi355ntt_u64* d_public_key, const mi355ntt_u64* d_e,
                               const mi355ntt_u64* d_m, unsigned count, mi355ntt_stream stream)
{
    if (!b || !d_c || !d_public_key || !d_e || !d_m) return MI355NTT_EINVAL;
    if (count == 0) return MI355NTT_OK;
    const unsigned R = b->p.R;
    if (!bfv_batch_count_ok(count, R)) return MI355NTT_EUNSUPPORTED;      /* before anything touches d_c */
    BFV_ON_DEVICE(b);
    /* :268-271 for the whole batch in one launch: the first count R polynomials with pk0, the rest with pk1 */
    BFV_RC(mi355ntt_polymul_batch_shared(b->ntt, d_c, d_public_key, 2 * count * R, R, count * R, stream));
    BFV_HIP(bfv_encrypt_tail(b->p, b->d, d_c, d_e, d_m, (hipStream_t)stream, count));                            /* :278-289 */
    return MI355NTT_OK;
}

int mi355ntt_bfv_decrypt_batch(const mi355ntt_bfv* b, mi355ntt_u64* d_c, const mi355ntt_u64* d_secret_key, unsigned count,
                               mi355ntt_stream stream)
{
    if (!b || !d_c || !d_secret_key) return MI355NTT_EINVAL;
    if (count == 0) return MI355NTT_OK;
    const unsigned R = b->p.R;
    if (!bfv_batch_count_ok(count, R)) return MI355NTT_EUNSUPPORTED;      /* before anything touches d_c */
    const size_t half = (size_t)count * R * b->p.n;
    BFV_ON_DEVICE(b);
    /* bfv_decryption.cuh:98-101 on the c1 run of the batch; the slot of the dropped last prime is carried along unused.  Where the
     * batch runs the persistent n = 2^15 kernel the scaling (:103-121) rides in the product's store path */
    const int fused = b->epi_ok ? ctx_polymul_epi(b->ntt, 1, d_c + half, d_secret_key, count * R, R, 0, d_c, b->d_epi, (hipStream_t)stream) : 1;
    if (fused < 0) return fused;
    if (fused != 0) {
        BFV_RC(mi355ntt_polymul_batch_shared(b->ntt, d_c + half, d_secret_key, count * R, R, 0, stream));
        BFV_HIP(bfv_decrypt_scale(b->p, b->d, d_c, (hipStream_t)stream, count));                                 /* :103-121 */
    }
    BFV_HIP(bfv_decrypt_round(b->p, b->d, d_c, (hipStream_t)stream, count));                                     /* :126-137 */
    return MI355NTT_OK;
}

/* ---------------- samplers and the complete drivers ---------------- */

static BfvSalsaKey salsa_key(const unsigned char* key32)
{
    BfvSalsaKey k;
    for (int i = 0; i < 8; i++)
        k.k[i] = (unsigned)key32[4 * i] | ((unsigned)key32[4 * i + 1] << 8) | ((unsigned)key32[4 * i + 2] << 16) | ((unsigned)key32[4 * i + 3] << 24);
    return k;
}

int mi355ntt_salsa20_keystream(void* d_out, size_t nbytes, const unsigned char* key32, mi355ntt_u64 nonce, mi355ntt_stream stream)
{
    if (!d_out || !key32) return MI355NTT_EINVAL;
    if (((uintptr_t)d_out & 15) != 0) return MI355NTT_EINVAL;
    BFV_HIP(bfv_salsa20_keystream(d_out, nbytes, salsa_key(key32), nonce, (hipStream_t)stream));
    return MI355NTT_OK;
}

size_t mi355ntt_bfv_keygen_random_bytes(const mi355ntt_bfv* b)
{
    return b ? (size_t)9 * b->p.R * b->p.n + (size_t)4 * b->p.n : 0;      /* bfv_keygen.cuh:99 */
}

size_t mi355ntt_bfv_encrypt_random_bytes(const mi355ntt_bfv* b)
{
    return b ? (size_t)b->p.n + (size_t)8 * b->p.n : 0;                   /* bfv_encryption.cuh:228 */
}

int mi355ntt_bfv_sample_keygen(const mi355ntt_bfv* b, const void* d_in, mi355ntt_u64* d_secret_key, mi355ntt_u64* d_public_key,
                               mi355ntt_u64* d_temp, mi355ntt_stream stream)
{
    if (!b || !d_in || !d_secret_key || !d_public_key || !d_temp) return MI355NTT_EINVAL;
    BFV_ON_DEVICE(b);
    BFV_HIP(bfv_sample_keygen(b->p, b->d, static_cast<const unsigned char*>(d_in), d_secret_key, d_public_key, d_temp, (hipStream_t)stream));
    return MI355NTT_OK;
}

int mi355ntt_bfv_sample_encrypt(const mi355ntt_bfv* b, const void* d_in, mi355ntt_u64* d_c, mi355ntt_u64* d_e, mi355ntt_stream stream)
{
    if (!b || !d_in || !d_c || !d_e) return MI355NTT_EINVAL;
    BFV_ON_DEVICE(b);
    BFV_HIP(bfv_sample_encrypt(b->p, b->d, static_cast<const unsigned char*>(d_in), d_c, d_e, (hipStream_t)stream));
    return MI355NTT_OK;
}

static const unsigned char kDefaultKey[32] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};   /* memset(k, 1, 32), distributions.cuh:236 */

int mi355ntt_bfv_keygen_rns(const mi355ntt_bfv* b, void* d_in, mi355ntt_u64* d_secret_key, mi355ntt_u64* d_public_key,
                            mi355ntt_u64* d_temp, mi355ntt_u64 nonce, mi355ntt_stream stream)
{
    if (!b) return MI355NTT_EINVAL;
    BFV_ON_DEVICE(b);
    BFV_RC(mi355ntt_salsa20_keystream(d_in, mi355ntt_bfv_keygen_random_bytes(b), kDefaultKey, nonce, stream));   /* :99  */
    BFV_RC(mi355ntt_bfv_sample_keygen(b, d_in, d_secret_key, d_public_key, d_temp, stream));                      /* :112-114 */
    return mi355ntt_bfv_keygen(b, d_secret_key, d_public_key, d_temp, stream);                                    /* :129-145 */
}

int mi355ntt_bfv_encryption_rns(const mi355ntt_bfv* b, mi355ntt_u64* d_c, const mi355ntt_u64* d_public_key, void* d_in,
                                mi355ntt_u64* d_e, const mi355ntt_u64* d_m, mi355ntt_u64 nonce, mi355ntt_stream stream)
{
    if (!b) return MI355NTT_EINVAL;
    BFV_ON_DEVICE(b);
    BFV_RC(mi355ntt_salsa20_keystream(d_in, mi355ntt_bfv_encrypt_random_bytes(b), kDefaultKey, nonce, stream));   /* :228 */
    BFV_RC(mi355ntt_bfv_sample_encrypt(b, d_in, d_c, d_e, stream));                                               /* :246 */
    return mi355ntt_bfv_encrypt(b, d_c, d_public_key, d_e, d_m, stream);                                          /* :268-289 */
}

}  // extern "C"
