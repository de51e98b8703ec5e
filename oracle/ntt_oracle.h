/*
 * ntt_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the arithmetic the reference (ozgunozerk/NTT-Cuda,
 * BFV_Scheme/) performs on its NTT hot path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the shipped HIP path
 * never calls into it.
 *
 * Pinning: checked against the reference's only known-answer test
 * (decryption_test.cu:348,355 -> m[i] = i % 10, decryption_test.cu:230-232) and the
 * known-answer constants in parameter.h:31-79 / old/decryption.cu / old/encryption.cu
 * (see tests/test_oracle_golden.py).  The reference itself cannot be built here
 * (CUDA + PTX, needs cuda_runtime.h), so there is no oracle/_ref.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/BFV_Scheme/).
 */
#ifndef NTT_ORACLE_H
#define NTT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned long long u64;

/* ---- parameter / helper layer ------------------------------------------- */
unsigned orc_bit_length(u64 q);                       /* demo.cu:69, decryption_test.cu:60 */
u64 orc_mu(u64 q, unsigned k);                        /* 60bit_ntt_test.cu:47-49 */
u64 orc_mulmod(u64 a, u64 b, u64 m);                  /* host64x2(a,b) % m, uint128.h:278-341 */
u64 orc_modpow(u64 a, u64 b, u64 m);                  /* helper.h:8-28 */
u64 orc_modinv(u64 a, u64 q);                         /* helper.h:52-56 */
u64 orc_bitrev(u64 a, int bits);                      /* helper.h:58-70 */
void orc_fill_table(u64 psi, u64 q, u64* tab, unsigned n);   /* parameter.h:5-20 */
int orc_get_params(u64 n, u64* q, u64* psi, u64* psiinv, u64* ninv, unsigned* qbit); /* parameter.h:31-79 */

/* ---- modular arithmetic --------------------------------------------------- */
u64 orc_barrett(u64 a, u64 b, u64 q, u64 mu, unsigned k);    /* mul64 + singleBarrett, ntt_60bit.cuh:44-61 */

/* ---- single-stage and whole transforms (in place) ------------------------- */
void orc_ct_stage(u64* a, unsigned n, unsigned length, u64 q, u64 mu, unsigned k, const u64* psi_tab);    /* ntt_60bit.cuh:192-223 */
void orc_gs_stage(u64* a, unsigned n, unsigned length, u64 q, u64 mu, unsigned k, const u64* psiinv_tab); /* ntt_60bit.cuh:225-265 */
void orc_forward(u64* a, unsigned n, u64 q, u64 mu, unsigned k, const u64* psi_tab);       /* forwardNTT  :314-348 */
void orc_inverse(u64* a, unsigned n, u64 q, u64 mu, unsigned k, const u64* psiinv_tab);    /* inverseNTT  :350-386 */

/* batch: polynomial y uses modulus/table (y % division); data at a + y*n; tables at tab + (y%division)*n
 * (ntt_60bit.cuh:391-422, :608-697).  threads > 1 uses OpenMP over polynomials (CPU baseline only). */
void orc_forward_batch(u64* a, unsigned n, const u64* psi_tabs, unsigned num, unsigned division,
                       const u64* q, const u64* mu, const unsigned* k, int threads);
void orc_inverse_batch(u64* a, unsigned n, const u64* psiinv_tabs, unsigned num, unsigned division,
                       const u64* q, const u64* mu, const unsigned* k, int threads);

/* ---- pointwise products ---------------------------------------------------- */
void orc_pointwise(u64* a, const u64* b, unsigned n, u64 q, u64 mu, unsigned k);            /* barrett,  poly_arithmetic.cuh:9-34 */
void orc_pointwise_batch(u64* c, const u64* a, const u64* b, unsigned n, unsigned num, unsigned division,
                         const u64* q, const u64* mu, const unsigned* k);                   /* barrett_batch(_3param) :36-98 */
void orc_pointwise_scalar(u64* a, u64 b, unsigned n, u64 q, u64 mu, unsigned k);            /* barrett_int :100-126 */

/* ---- the reference's own CPU check ---------------------------------------- */
void orc_ref_polymul(const u64* a, const u64* b, u64* d, u64 m, unsigned n);               /* refPolyMul128, helper.h:95-126 */

/* ---- BFV decryption (only to replay the reference's known-answer test) ------ */
/* c: 2*(r+1)*n words laid out as decryption_rns expects (bfv_decryption.cuh:61-75), sk: r*n words in
 * NTT domain, qs/psis: r+1 primes (the last one is the dropped special prime), out: n words.
 * Follows decryption_test.cu:47-345 (parameter bootstrap) and bfv_decryption.cuh:76-138.
 * stage_out (may be NULL): 3*r*n words receiving c1 after forward batch, after barrett_batch,
 * after inverse batch. */
int orc_bfv_decrypt(u64* c, const u64* sk, const u64* qs, const u64* psis, unsigned r_plus_1,
                    unsigned n, u64 t, u64 gamma, u64* out, u64* stage_out);
/* keygen_rns / encryption_rns after their samplers (bfv_keygen.cuh:95-151, bfv_encryption.cuh:223-290); r_plus_1 counts
 * the special last prime.  Pinned through the round trip with orc_bfv_decrypt (demo.cu:302-311). */
int orc_bfv_keygen_core(u64* secret_key, u64* public_key, const u64* e, const u64* qs, const u64* psis,
                        unsigned r_plus_1, unsigned n);
int orc_bfv_encrypt_core(u64* c, const u64* public_key, const u64* e, const u64* m, const u64* qs, const u64* psis,
                         unsigned r_plus_1, unsigned n, u64 t);

/* ---- samplers: Salsa20/20 keystream as generate_random_default produces it (distributions.cuh:48-155,
 * 249-276) and the merged conversion kernels (bfv_keygen.cuh:14-79, bfv_encryption.cuh:17-109).  The integer paths are
 * bit-exact restatements; the Gaussian one uses its own inverse normal CDF (CUDA's normcdfinvf is not specified to the ulp). */
void orc_salsa20_keystream(unsigned char* out, unsigned long nblocks, const unsigned char* key, u64 nonce);
void orc_sample_ternary_xq(const unsigned char* in, u64* out, unsigned n, unsigned q_amount, const u64* qs);
void orc_sample_uniform_xq(const unsigned char* in, u64* out, unsigned n, unsigned q_amount, const u64* qs);
void orc_sample_gaussian_xq(const unsigned char* in, u64* out, unsigned n, unsigned q_amount, const u64* qs);

/* ---- the 30-bit path (old/ntt_30bit.cuh): 32-bit words, single prime.  Pinned by the parameter
 * tuples of getParams30 (old/NTT/old_design/final/parameter.h:73-115) and the schoolbook product. */
void orc30_forward(uint32_t* a, unsigned n, uint32_t q, uint32_t mu, int qbit, const uint32_t* psi_tab);
void orc30_inverse(uint32_t* a, unsigned n, uint32_t q, uint32_t mu, int qbit, const uint32_t* psiinv_tab);
void orc30_pointwise(uint32_t* a, const uint32_t* b, unsigned long count, uint32_t q, uint32_t mu, int qbit);

/* constants the bootstrap derives, exposed for the known-answer checks
 * (old/decryption.cu:46,97,103,113; old/encryption.cu:98,101) */
void orc_bfv_constants(const u64* qs, const u64* psis, unsigned r_plus_1, u64 t, u64 gamma,
                       u64* psiinv, u64* inv_punctured_q, u64* neg_inv_q_mod_t_gamma /*2*/,
                       u64* prod_t_gamma_mod_q, u64* inv_q_last_mod_q, u64* qi_div_t);

/* ---- synthetic inputs: splitmix64 stream reduced mod q ---- */
void orc_splitmix_fill(u64* a, unsigned long count, u64 seed, u64 q);

#ifdef __cplusplus
}
#endif
#endif
