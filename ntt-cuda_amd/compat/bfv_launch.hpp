// bfv_launch.hpp -- the NTT sections of the reference's BFV drivers as host launch sequences on a context.
//
// In ozgunozerk/NTT-Cuda the drivers keygen_rns / encryption_rns / decryption_rns are pure kernel-launch sequences;
// their NTT hot path is the triple forwardNTT_batch -> barrett_batch* -> inverseNTT_batch on the legacy default
// stream.  These functions are those call sites (same buffer layouts, same num/division arguments), issued on the
// caller's stream through the throughput kernels of libmi355ntt.  Everything around them in the drivers
// (samplers, poly_add_*, divide_and_round_*, base conversion) is outside the NTT path; the element-wise part of it is
// available through the drivers at the end of this file, the samplers stay with the caller.
#pragma once
#include "../../include/mi355ntt.h"

namespace mi355 {

// bfv_keygen.cuh:129-133 -- secret_key: [r][n] ternary sample (in place -> NTT domain);
// public_key: [2][r][n], pk1 (second half) uniform; on return pk0 = INTT(pk1 (.) NTT(sk)) in the coefficient domain.
inline int keygen_ntt_a(const mi355ntt_ctx* ctx, unsigned long long* secret_key, unsigned long long* public_key, unsigned n,
                        unsigned q_amount, mi355ntt_stream stream)
{
    int rc = mi355ntt_forward_batch(ctx, secret_key, q_amount, q_amount, stream);                       // :129
    if (rc) return rc;
    rc = mi355ntt_pointwise_mul(ctx, public_key, public_key + (size_t)q_amount * n, secret_key, q_amount, q_amount, stream);  // :131-132
    if (rc) return rc;
    return mi355ntt_inverse_batch(ctx, public_key, q_amount, q_amount, stream);                          // :133
}

// bfv_keygen.cuh:145 -- after poly_add_negate_xq: pk0 back to the NTT domain
inline int keygen_ntt_b(const mi355ntt_ctx* ctx, unsigned long long* public_key, unsigned q_amount, mi355ntt_stream stream)
{
    return mi355ntt_forward_batch(ctx, public_key, q_amount, q_amount, stream);
}

// bfv_encryption.cuh:268-271 -- c: [2][r][n] (u twice), public_key: [2][r][n] in the NTT domain:
// c[j] = INTT(NTT(c[j]) (.) pk[j]) for all 2r polynomials, one fused launch
inline int encryption_ntt(const mi355ntt_ctx* ctx, unsigned long long* c, const unsigned long long* public_key, unsigned q_amount,
                          mi355ntt_stream stream)
{
    return mi355ntt_polymul_batch(ctx, c, public_key, 2 * q_amount, q_amount, stream);
}

// bfv_decryption.cuh:98-101 -- q_amount = r (the special prime is already dropped, the context still holds r + 1 primes);
// c: [2][r + 1][n]; c1 = c + (r + 1) n; secret_key: [r][n] in the NTT domain: c1[i] = INTT(NTT(c1[i]) (.) sk[i]), i < r
inline int decryption_ntt(const mi355ntt_ctx* ctx, unsigned long long* c, const unsigned long long* secret_key, unsigned n,
                          unsigned q_amount, mi355ntt_stream stream)
{
    return mi355ntt_polymul_batch(ctx, c + (size_t)(q_amount + 1) * n, secret_key, q_amount, q_amount + 1, stream);
}

// ---- the whole drivers after their samplers (C ABI section "BFV"): the element-wise kernels around the NTT sections
// (poly_add_negate_xq, poly_add_xq, divide_and_round_q_last_inplace_*, weird_m_stuff, poly_add_xq_d,
// poly_mul_int_xq_*, fast_convert_array_kernels, poly_mul_int(_t), dec_round) run fused on the same stream ----

// keygen_rns from bfv_keygen.cuh:129 on; temp = the error sample
inline int keygen_rns(const mi355ntt_bfv* bfv, unsigned long long* secret_key, unsigned long long* public_key,
                      const unsigned long long* temp, mi355ntt_stream stream)
{
    return mi355ntt_bfv_keygen(bfv, secret_key, public_key, temp, stream);
}

// encryption_rns from bfv_encryption.cuh:268 on
inline int encryption_rns(const mi355ntt_bfv* bfv, unsigned long long* c, const unsigned long long* public_key,
                          const unsigned long long* e, const unsigned long long* m_poly_device, mi355ntt_stream stream)
{
    return mi355ntt_bfv_encrypt(bfv, c, public_key, e, m_poly_device, stream);
}

// decryption_rns (bfv_decryption.cuh:76-138); plaintext at c + n * (q_amount - 1), q_amount = primes without the special one
inline int decryption_rns(const mi355ntt_bfv* bfv, unsigned long long* c, const unsigned long long* secret_key, mi355ntt_stream stream)
{
    return mi355ntt_bfv_decrypt(bfv, c, secret_key, stream);
}

// ---- many ciphertexts per call (no counterpart in the reference, which encrypts one ciphertext per launch sequence): the
// batch is laid out [2][count][q_amount + 1][n] -- all first components, then all second components; e likewise, m_polys
// [count][n]; the secret key must hold all q_amount + 1 polynomials.  Each ciphertext receives the words the single drivers
// above leave for it; plaintext of ciphertext z at c + (z (q_amount + 1) + q_amount - 1) n. ----
inline int encryption_rns_batch(const mi355ntt_bfv* bfv, unsigned long long* c, const unsigned long long* public_key,
                                const unsigned long long* e, const unsigned long long* m_polys_device, unsigned count, mi355ntt_stream stream)
{
    return mi355ntt_bfv_encrypt_batch(bfv, c, public_key, e, m_polys_device, count, stream);
}

inline int decryption_rns_batch(const mi355ntt_bfv* bfv, unsigned long long* c, const unsigned long long* secret_key, unsigned count,
                                mi355ntt_stream stream)
{
    return mi355ntt_bfv_decrypt_batch(bfv, c, secret_key, count, stream);
}

// ---- the complete drivers, samplers included: `in` is the caller's random-byte buffer as in the reference
// (mi355ntt_bfv_keygen_random_bytes / _encrypt_random_bytes give the sizes the drivers consume); `nonce` = 0 reproduces
// the reference's fixed keystream (generate_random_default) ----
inline int keygen_rns(const mi355ntt_bfv* bfv, unsigned char* in, unsigned long long* secret_key, unsigned long long* public_key,
                      unsigned long long* temp, mi355ntt_stream stream, unsigned long long nonce = 0)
{
    return mi355ntt_bfv_keygen_rns(bfv, in, secret_key, public_key, temp, nonce, stream);
}

inline int encryption_rns(const mi355ntt_bfv* bfv, unsigned long long* c, const unsigned long long* public_key, unsigned char* in,
                          unsigned long long* e, const unsigned long long* m_poly_device, mi355ntt_stream stream,
                          unsigned long long nonce = 0)
{
    return mi355ntt_bfv_encryption_rns(bfv, c, public_key, in, e, m_poly_device, nonce, stream);
}

}  // namespace mi355
