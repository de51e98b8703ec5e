// bfv.hpp -- the BFV launch layer around the NTT path: parameter bootstrap on the host and the
// element-wise kernels of keygen_rns / encryption_rns / decryption_rns.  Shared between bfv_host.cpp and kernels_bfv.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mi355ntt.h"
#include "hostparams.hpp"
#include "kernels.hpp"

namespace mi355ntt {

// Per-prime constants of the element-wise kernels: the reference keeps them in __constant__ arrays
// (q_cons, mu_cons, q_bit_cons: ntt_60bit.cuh:8-10; inv_q_last_mod_q_cons, inv_punctured_q_cons,
// prod_t_gamma_mod_q_cons: bfv_encryption.cuh / bfv_decryption.cuh); here one record per prime in device memory.
struct BfvPrime {
    u64 q, mu;
    unsigned k, pad;
    u64 prod_t_gamma_mod_q;   // (t * gamma) mod q                      demo.cu:119-125
    u64 inv_punctured_q;      // (prod_{j != i, j < r} q_j)^-1 mod q_i   demo.cu:262-276
    u64 inv_q_last_mod_q;     // (q_last mod q_i)^-1 mod q_i             demo.cu:73-79
    u64 q_div_t;              // floor(q_i / t)                          demo.cu:84-88
    u64 half_last_mod_q;      // (q_last >> 1) mod q_i                   bfv_encryption.cuh:138
    u64 m64;                  // floor((2^64 - 1) / q_i): exact x mod q_i for any 64-bit x (reduce64, kernels_bfv.hip) in place of the reference's `%`
};

struct BfvParams {
    unsigned n = 0, R = 0, r = 0;      // R = number of primes including the special last one, r = R - 1
    u64 t = 0, gamma = 0, mu_gamma = 0, gamma_div_2 = 0, m64_gamma = 0;    // m64_gamma: floor((2^64 - 1) / gamma), see BfvPrime::m64
    unsigned gamma_bits = 0, lazy_gamma = 1;               // lazy_gamma: Barrett products mod gamma (each below 2 gamma) that fit on a reduced accumulator in 64 bits
    u64 neg_inv_q_mod_t = 0, neg_inv_q_mod_gamma = 0;     // demo.cu:103-117
    u64 q_last = 0, half_q_last = 0;
    BfvPrime prime[kMaxPrimes];
    u64 base_change[2 * kMaxPrimes];   // [2][r]: punctured products mod t, then mod gamma   demo.cu:281-301
};

// demo.cu:62-272 / decryption_test.cu:60-345 as a host function; returns 0 or a negative MI355NTT_E* code
int bfv_bootstrap(BfvParams* out, unsigned n, unsigned R, const u64* q, u64 t, u64 gamma);

struct BfvDevice {
    const BfvPrime* d_prime = nullptr;   // [R]
    const u64* d_base_change = nullptr;  // [2][r]
};

// ---- samplers ----
struct BfvSalsaKey {
    unsigned k[8];            // the 32 key bytes as little-endian words
};
// generate_random / generate_random_default (distributions.cuh:192-276): Salsa20/20 keystream into d_out, floor(nbytes / 64) blocks
hipError_t bfv_salsa20_keystream(void* d_out, size_t nbytes, const BfvSalsaKey& key, u64 nonce, hipStream_t s);
// ternary_dist_xq, uniform_dist_xq, gaussian_dist_xq (bfv_keygen.cuh:14-79, call sites :112-114) in one pass
hipError_t bfv_sample_keygen(const BfvParams& p, const BfvDevice& d, const unsigned char* in, u64* secret_key, u64* public_key,
                             u64* temp, hipStream_t s);
// convert_ternary_gaussian_x2 (bfv_encryption.cuh:17-109)
hipError_t bfv_sample_encrypt(const BfvParams& p, const BfvDevice& d, const unsigned char* in, u64* c, u64* e, hipStream_t s);

// poly_add_negate_xq (bfv_keygen.cuh:80-93) on [R][n]
hipError_t bfv_add_negate(const BfvParams& p, const BfvDevice& d, u64* pk0, const u64* e, hipStream_t s);
// pk0 (holding NTT(e)) <- -(a_hat (.) s_hat + pk0): the key generation's product, sum and negation in the NTT domain
hipError_t bfv_keygen_pk0(const BfvParams& p, const BfvDevice& d, u64* pk0, const u64* a_hat, const u64* s_hat, hipStream_t s);
// the fused product with the per-prime element-wise step in its store path (capi.cpp, kernels_epi.cuh): 0 done, 1 not available here
// (the caller runs the two steps one after the other), < 0 an MI355NTT_E* code
int ctx_polymul_epi(const mi355ntt_ctx* c, int kind, u64* d_a, const u64* d_bhat, unsigned num, unsigned division, unsigned group,
                    const u64* d_other, const void* d_consts, hipStream_t s);
struct BfvEpiPrime {          // = EpiPrime (kernels_epi.cuh)
    u64 k1, k2;
    unsigned on, pad;
};

// poly_add_xq + divide_and_round_q_last_inplace_add_x2 + divide_and_round_q_last_inplace_loop_xq + weird_m_stuff
// (bfv_encryption.cuh:110-208) on c [2][R][n], e [2][R][n], m [n]: one pass, same words as the four launches
hipError_t bfv_encrypt_tail(const BfvParams& p, const BfvDevice& d, u64* c, const u64* e, const u64* m, hipStream_t s, unsigned count = 1);
// poly_add_xq_d + poly_mul_int_xq_prodtgamma + poly_mul_int_xq_invpq (bfv_decryption.cuh:13-57) on c [2][R][n]
hipError_t bfv_decrypt_scale(const BfvParams& p, const BfvDevice& d, u64* c, hipStream_t s, unsigned count = 1);
// fast_convert_array_kernel_t/_gamma, mod_t, barrett_int (gamma), dec_round_kernel (poly_arithmetic.cuh:128-142,
// 221-268): leaves the same words as the reference in c[0, n), c[n, 2n) and the plaintext at c + n (r - 1)
hipError_t bfv_decrypt_round(const BfvParams& p, const BfvDevice& d, u64* c, hipStream_t s, unsigned count = 1);

}  // namespace mi355ntt
