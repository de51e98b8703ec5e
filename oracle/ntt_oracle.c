/*
 * ntt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See ntt_oracle.h.
 *
 * Restates, stage by stage and in the reference's own evaluation order, what the CUDA kernels
 * of ozgunozerk/NTT-Cuda compute.  Citations are relative to /root/reference/BFV_Scheme/.
 * The reference's uint128_t (uint128.h:10-129) is replaced by the compiler's unsigned __int128;
 * SURVEY.md 8(a) lists the host-side quirks of uint128_t that are deliberately NOT reproduced
 * (they never fire for values the reference feeds them).
 */
#include "ntt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ helpers */

/* demo.cu:69 / decryption_test.cu:60 : q_bit_lengths.push_back(log2((double)q) + 1) */
unsigned orc_bit_length(u64 q) { return (unsigned)(log2((double)q) + 1); }

/* 60bit_ntt_test.cu:47-49 : mu = (uint128_t::exp2(2*bit_length) / q).low */
u64 orc_mu(u64 q, unsigned k) { return (u64)((((u128)1) << (2 * k)) / q); }

/* host64x2(a,b) % m  (uint128.h:314-341 schoolbook multiply, :278-312 shift-subtract remainder) */
u64 orc_mulmod(u64 a, u64 b, u64 m) { return (u64)(((u128)a * b) % m); }

/* helper.h:8-28 modpow128: note res starts as a (unreduced) when b is odd */
u64 orc_modpow(u64 a, u64 b, u64 m)
{
    u64 res = 1;
    if (1 & b) res = a;
    while (b != 0) {
        b = b >> 1;
        a = orc_mulmod(a, a, m);
        if (b & 1) res = orc_mulmod(res, a, m);
    }
    return res;
}

/* helper.h:52-56 modinv128: Fermat, a^(q-2) */
u64 orc_modinv(u64 a, u64 q) { return orc_modpow(a, q - 2, q); }

/* helper.h:58-70 */
u64 orc_bitrev(u64 a, int bits)
{
    u64 res = 0;
    for (int i = 0; i < bits; i++) {
        res <<= 1;
        res = (a & 1) | res;
        a >>= 1;
    }
    return res;
}

/* parameter.h:5-20 fillTablePsi128: tab[i] = psi^bitReverse(i, log2(n)) mod q */
void orc_fill_table(u64 psi, u64 q, u64* tab, unsigned n)
{
    int lg = (int)log2((double)n);
    for (unsigned i = 0; i < n; i++) tab[i] = orc_modpow(psi, orc_bitrev(i, lg), q);
}

/* parameter.h:31-79 getParams (the active, un-commented sets) + the commented 58-bit n=4096 set
 * (:43-47) under the pseudo-size n = 4096 + 1 so tests can reach it. */
int orc_get_params(u64 n, u64* q, u64* psi, u64* psiinv, u64* ninv, unsigned* qbit)
{
    switch (n) {
    case 2048:  *q = 137438691329ULL;      *psi = 22157790ULL;       *psiinv = 88431458764ULL;        *ninv = 137371582593ULL;       *qbit = 37; return 0;
    case 4096:  *q = 33538049ULL;          *psi = 2386ULL;           *psiinv = 26102329ULL;           *ninv = 33529861ULL;           *qbit = 25; return 0;
    case 4097:  *q = 288230376135196673ULL; *psi = 60193018759093ULL; *psiinv = 236271020333049746ULL; *ninv = 288160007391023041ULL; *qbit = 58; return 0;
    case 8192:  *q = 8796092858369ULL;     *psi = 1734247217ULL;     *psiinv = 5727406356888ULL;      *ninv = 8795019116565ULL;      *qbit = 43; return 0;
    case 16384: *q = 281474976546817ULL;   *psi = 23720796222ULL;    *psiinv = 129310633907832ULL;    *ninv = 281457796677643ULL;    *qbit = 48; return 0;
    case 32768: *q = 36028797017456641ULL; *psi = 1155186985540ULL;  *psiinv = 31335194304461613ULL;  *ninv = 36027697505828911ULL;  *qbit = 55; return 0;
    default: return -1;
    }
}

/* ------------------------------------------------------------------ Barrett */

/* mul64 (uint128.h:353-373) followed by singleBarrett (ntt_60bit.cuh:44-61); the pointwise kernels
 * inline the same sequence (poly_arithmetic.cuh:18-33).  Every intermediate the reference keeps only
 * the .low limb of is truncated to 64 bits here as well. */
static inline u64 barrett128(u128 a, u64 q, u64 mu, unsigned k)
{
    u128 rx;
    rx = (u64)(a >> (k - 2));            /* rx = a >> (qbit - 2); only rx.low is used next           */
    rx = (u128)(u64)rx * mu;             /* mul64(rx.low, mu, rx)                                     */
    rx = rx >> (k + 2);                  /* uint128_t::shiftr(rx, qbit + 2)                           */
    rx = (u128)(u64)rx * q;              /* mul64(rx.low, q, rx)                                      */
    a = a - rx;                          /* sub128(a, rx)                                             */
    u64 lo = (u64)a;                     /* only a.low is inspected                                   */
    if (lo >= q) lo -= q;                /* ONE conditional subtraction                               */
    return lo;
}

u64 orc_barrett(u64 a, u64 b, u64 q, u64 mu, unsigned k) { return barrett128((u128)a * b, q, mu, k); }

/* ------------------------------------------------------------------- stages */

/* One Cooley-Tukey stage, CTBasedNTTInner<l,n> (ntt_60bit.cuh:192-223); the shared-memory kernel
 * CTBasedNTTInnerSingle (:63-123) runs the same butterflies with g = local + blockIdx.x*(n/2l). */
void orc_ct_stage(u64* a, unsigned n, unsigned length, u64 q, u64 mu, unsigned k, const u64* psi_tab)
{
    unsigned step = (n / length) / 2;
    for (unsigned g = 0; g < n / 2; g++) {
        unsigned psi_step = g / step;
        unsigned j = psi_step * step * 2 + g % step;
        u64 psi = psi_tab[length + psi_step];
        u64 U = a[j];
        u64 V = barrett128((u128)a[j + step] * psi, q, mu, k);
        u64 t = U + V;
        t -= q * (t >= q);
        a[j] = t;
        U += q * (U < V);
        a[j + step] = U - V;
    }
}

/* One Gentleman-Sande stage with the n^-1 halving folded in, GSBasedINTTInner<l,n> (:225-265);
 * GSBasedINTTInnerSingle (:125-190) is the same arithmetic in shared memory. */
void orc_gs_stage(u64* a, unsigned n, unsigned length, u64 q, u64 mu, unsigned k, const u64* psiinv_tab)
{
    unsigned step = (n / length) / 2;
    u64 q2 = (q + 1) >> 1;
    for (unsigned g = 0; g < n / 2; g++) {
        unsigned psi_step = g / step;
        unsigned j = psi_step * step * 2 + g % step;
        u64 psiinv = psiinv_tab[length + psi_step];
        u64 U = a[j];
        u64 V = a[j + step];
        u64 t = U + V;
        t -= q * (t >= q);
        a[j] = (t >> 1) + q2 * (t & 1);
        U += q * (U < V);
        u64 d = barrett128((u128)(U - V) * psiinv, q, mu, k);
        a[j + step] = (d >> 1) + q2 * (d & 1);
    }
}

/* forwardNTT (:314-348): stages length = 1,2,...,n/2 in that order (global-memory stages first,
 * then the single-block kernel loops length = l ... n/2, :78-80). */
void orc_forward(u64* a, unsigned n, u64 q, u64 mu, unsigned k, const u64* psi_tab)
{
    for (unsigned length = 1; length < n; length *= 2) orc_ct_stage(a, n, length, q, mu, k, psi_tab);
}

/* inverseNTT (:350-386): stages length = n/2, n/4, ..., 1 (:144-146, then the global stages). */
void orc_inverse(u64* a, unsigned n, u64 q, u64 mu, unsigned k, const u64* psiinv_tab)
{
    for (unsigned length = n / 2; length >= 1; length /= 2) orc_gs_stage(a, n, length, q, mu, k, psiinv_tab);
}

/* forwardNTT_batch / inverseNTT_batch (:608-697) with the *_batch kernels' index rules (:391-422):
 * index = blockIdx.y % division selects q/mu/qbit and the twiddle table; data offset blockIdx.y*n. */
void orc_forward_batch(u64* a, unsigned n, const u64* psi_tabs, unsigned num, unsigned division,
                       const u64* q, const u64* mu, const unsigned* k, int threads)
{
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (long y = 0; y < (long)num; y++) {
        unsigned idx = (unsigned)y % division;
        orc_forward(a + (size_t)y * n, n, q[idx], mu[idx], k[idx], psi_tabs + (size_t)idx * n);
    }
}

void orc_inverse_batch(u64* a, unsigned n, const u64* psiinv_tabs, unsigned num, unsigned division,
                       const u64* q, const u64* mu, const unsigned* k, int threads)
{
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (long y = 0; y < (long)num; y++) {
        unsigned idx = (unsigned)y % division;
        orc_inverse(a + (size_t)y * n, n, q[idx], mu[idx], k[idx], psiinv_tabs + (size_t)idx * n);
    }
}

/* ---------------------------------------------------------------- pointwise */

/* barrett (poly_arithmetic.cuh:9-34): a[i] = a[i]*b[i] mod q; final select is `rc.low < q ? rc.low : rc.low-q` */
void orc_pointwise(u64* a, const u64* b, unsigned n, u64 q, u64 mu, unsigned k)
{
    for (unsigned i = 0; i < n; i++) a[i] = barrett128((u128)a[i] * b[i], q, mu, k);
}

/* barrett_batch (:36-66) when c == a, barrett_batch_3param (:68-98) otherwise.  b is indexed with the
 * same i = x + y*n as a (the reference passes b = public_key / secret_key laid out per polynomial). */
void orc_pointwise_batch(u64* c, const u64* a, const u64* b, unsigned n, unsigned num, unsigned division,
                         const u64* q, const u64* mu, const unsigned* k)
{
    for (unsigned y = 0; y < num; y++) {
        unsigned idx = y % division;
        for (unsigned x = 0; x < n; x++) {
            size_t i = (size_t)y * n + x;
            c[i] = barrett128((u128)a[i] * b[i], q[idx], mu[idx], k[idx]);
        }
    }
}

/* barrett_int (:100-126) */
void orc_pointwise_scalar(u64* a, u64 b, unsigned n, u64 q, u64 mu, unsigned k)
{
    for (unsigned i = 0; i < n; i++) a[i] = barrett128((u128)a[i] * b, q, mu, k);
}

/* ----------------------------------------------------- reference CPU check */

/* refPolyMul128 (helper.h:95-126): O(n^2) schoolbook product modulo (x^n + 1, m) */
void orc_ref_polymul(const u64* a, const u64* b, u64* d, u64 m, unsigned n)
{
    u64* c = (u64*)calloc((size_t)2 * n, sizeof(u64));
    for (unsigned i = 0; i < n; i++)
        for (unsigned j = 0; j < n; j++) {
            c[i + j] = orc_mulmod(a[i], b[j], m) + c[i + j] % m;
            c[i + j] %= m;
        }
    for (unsigned i = 0; i < n; i++) {
        u64 ci = c[i];
        if (ci < c[i + n]) ci += m;
        d[i] = (ci - c[i + n]) % m;
    }
    free(c);
}

/* ------------------------------------------------------ BFV decryption KAT */

/* decryption_test.cu:60-345 bootstrap constants.  Output arrays sized r (= r_plus_1 - 1) unless noted:
 * psiinv[r+1], inv_punctured_q[r], neg_inv_q_mod_t_gamma[2], prod_t_gamma_mod_q[r],
 * inv_q_last_mod_q[r] (old/encryption.cu:98 / decryption_test.cu:66-72), qi_div_t[r+1]. */
void orc_bfv_constants(const u64* qs, const u64* psis, unsigned r_plus_1, u64 t, u64 gamma,
                       u64* psiinv, u64* inv_punctured_q, u64* neg_inv_q_mod_t_gamma,
                       u64* prod_t_gamma_mod_q, u64* inv_q_last_mod_q, u64* qi_div_t)
{
    unsigned r = r_plus_1 - 1;
    for (unsigned i = 0; i < r_plus_1; i++) psiinv[i] = orc_modinv(psis[i], qs[i]);      /* :93-94 */
    u64 mult_t = 1, mult_g = 1;                                                          /* :103-111 */
    for (unsigned i = 0; i < r; i++) {
        mult_t = orc_mulmod(mult_t, qs[i], t);
        mult_g = orc_mulmod(mult_g, qs[i], gamma);
    }
    neg_inv_q_mod_t_gamma[0] = t - orc_modinv(mult_t, t);
    neg_inv_q_mod_t_gamma[1] = gamma - orc_modinv(mult_g, gamma);
    u128 prod_t_gamma = (u128)t * gamma;                                                 /* :119-125 */
    for (unsigned i = 0; i < r; i++) prod_t_gamma_mod_q[i] = (u64)(prod_t_gamma % qs[i]);
    for (unsigned i = 0; i < r; i++) {                                                   /* :262-276 */
        u64 temp = 1;
        for (unsigned j = 0; j < r; j++) {
            if (i == j) continue;
            temp = orc_mulmod(temp, qs[j], qs[i]);
        }
        inv_punctured_q[i] = orc_modinv(temp, qs[i]);
    }
    for (unsigned i = 0; i < r; i++) inv_q_last_mod_q[i] = orc_modinv(qs[r] % qs[i], qs[i]);   /* :66-72 */
    for (unsigned i = 0; i < r_plus_1; i++) qi_div_t[i] = qs[i] / t;                              /* :82-86 */
}

int orc_bfv_decrypt(u64* c, const u64* sk, const u64* qs, const u64* psis, unsigned r_plus_1,
                    unsigned n, u64 t, u64 gamma, u64* out, u64* stage_out)
{
    unsigned r = r_plus_1 - 1;
    if (r_plus_1 > 16 || r < 1) return -1;
    unsigned kbits[16]; u64 mus[16];
    u64 psiinv[16], inv_punct[16], neg_inv[2], ptg[16], iql[16], qdt[16];
    for (unsigned i = 0; i < r_plus_1; i++) {                      /* :57-61, :160-168 */
        kbits[i] = orc_bit_length(qs[i]);
        mus[i] = orc_mu(qs[i], kbits[i]);
    }
    orc_bfv_constants(qs, psis, r_plus_1, t, gamma, psiinv, inv_punct, neg_inv, ptg, iql, qdt);
    /* output_base_bit_lengths = {10, 61}, decryption_test.cu:111 / demo.cu:100: data the caller writes next to gamma = 2305843009213683713
     * (61 bits), used only as the Barrett bit length of gamma (:244-247).  For another gamma a caller writes that gamma's bit length, which
     * is what this computes (61 for the reference's gamma: the KAT is unaffected). */
    const unsigned gamma_bits = orc_bit_length(gamma);
    u64 mu_gamma = orc_mu(gamma, gamma_bits);                      /* :252-258 */
    u64 gamma_div_2 = gamma >> 1;                                  /* :91 */

    u64* psi_tabs = (u64*)malloc(sizeof(u64) * (size_t)n * r_plus_1);      /* :178-199 */
    u64* psiinv_tabs = (u64*)malloc(sizeof(u64) * (size_t)n * r_plus_1);
    for (unsigned i = 0; i < r_plus_1; i++) {
        orc_fill_table(psis[i], qs[i], psi_tabs + (size_t)i * n, n);
        orc_fill_table(psiinv[i], qs[i], psiinv_tabs + (size_t)i * n, n);
    }
    u64 bcm[32];                                                   /* base_change_matrix, :281-301 */
    u64 output_base[2] = { t, gamma };
    for (unsigned i = 0; i < 2; i++)
        for (unsigned j = 0; j < r; j++) {
            u64 temp = 1;
            for (unsigned kk = 0; kk < r; kk++) {
                if (j == kk) continue;
                temp = orc_mulmod(temp, qs[kk], output_base[i]);
            }
            bcm[i * r + j] = temp;
        }

    /* ---- decryption_rns, bfv_decryption.cuh:76-138 (q_amount == r here) ---- */
    u64* c1 = c + (size_t)(r + 1) * n;
    orc_forward_batch(c1, n, psi_tabs, r, r + 1, qs, mus, kbits, 1);                 /* :98  */
    if (stage_out) memcpy(stage_out, c1, sizeof(u64) * (size_t)r * n);
    orc_pointwise_batch(c1, c1, sk, n, r, r, qs, mus, kbits);                         /* :99-100 */
    if (stage_out) memcpy(stage_out + (size_t)r * n, c1, sizeof(u64) * (size_t)r * n);
    orc_inverse_batch(c1, n, psiinv_tabs, r, r + 1, qs, mus, kbits, 1);              /* :101 */
    if (stage_out) memcpy(stage_out + (size_t)2 * r * n, c1, sizeof(u64) * (size_t)r * n);

    for (size_t i = 0; i < (size_t)n * r; i++) {                   /* poly_add_xq_d, :13-23 (note `>`) */
        u64 ra = c[i + (size_t)n * (r + 1)] + c[i];
        if (ra > qs[i / n]) ra -= qs[i / n];
        c[i + (size_t)n * (r + 1)] = ra;
    }
    for (size_t i = 0; i < (size_t)n * r; i++)                     /* poly_mul_int_xq_prodtgamma, :25-40 */
        c1[i] = barrett128((u128)c1[i] * ptg[i / n], qs[i / n], mus[i / n], kbits[i / n]);
    for (size_t i = 0; i < (size_t)n * r; i++)                     /* poly_mul_int_xq_invpq, :42-57 */
        c1[i] = barrett128((u128)c1[i] * inv_punct[i / n], qs[i / n], mus[i / n], kbits[i / n]);

    unsigned mask32 = (unsigned)(t - 1);                           /* fast_convert_array_kernel_t, poly_arithmetic.cuh:221-239 */
    for (unsigned kk = 0; kk < n; kk++) {
        u64 acc = 0;
        for (unsigned i = 0; i < r; i++) {
            u64 tmp = c1[kk + (size_t)i * n] * bcm[i];
            tmp = tmp & mask32;
            acc += tmp;
        }
        c[kk] = acc & mask32;
    }
    for (unsigned kk = 0; kk < n; kk++) {                          /* fast_convert_array_kernel_gamma, :241-256 */
        u64 acc = 0;
        for (unsigned i = 0; i < r; i++) {
            u64 tmp = barrett128((u128)c1[kk + (size_t)i * n] * bcm[i + r], gamma, mu_gamma, gamma_bits);
            acc = (acc + tmp) % gamma;
        }
        c[kk + n] = acc % gamma;
    }
    u64 mask = t - 1;
    for (unsigned i = 0; i < n; i++)                               /* poly_mul_int_t -> mod_t, :128-142 */
        c[i] = (c[i] * neg_inv[0]) & (unsigned)mask;
    for (unsigned i = 0; i < n; i++)                               /* poly_mul_int -> barrett_int on gamma */
        c[n + i] = barrett128((u128)c[n + i] * neg_inv[1], gamma, mu_gamma, gamma_bits);
    u64* result = c + (size_t)n * (r - 1);                         /* dec_round_kernel, :258-268 */
    for (unsigned i = 0; i < n; i++) {
        u64 x0 = c[i], x1 = c[i + n];
        if (x1 > gamma_div_2) result[i] = (x0 + (gamma - x1)) & mask;
        else                  result[i] = (x0 - x1) & mask;
    }
    memcpy(out, result, sizeof(u64) * n);                          /* decryption_test.cu:376-377 */
    free(psi_tabs); free(psiinv_tabs);
    return 0;
}

/* ------------------------------------------- BFV key generation / encryption */

/* Everything in keygen_rns / encryption_rns after the samplers (bfv_keygen.cuh:95-151, bfv_encryption.cuh:223-290).
 * The sampled polynomials are inputs here (the samplers themselves are SURVEY.md 8f row 3).  r_plus_1 = number of
 * primes INCLUDING the special last one ("q_amount" of these two drivers).  Parity of these two is pinned only through
 * the round trip with orc_bfv_decrypt (itself pinned by KAT-1), as in demo.cu:302-311. */
static void bfv_tables(const u64* qs, const u64* psis, unsigned R, unsigned n, u64** psi_tabs, u64** psiinv_tabs,
                       u64* mus, unsigned* kbits)
{
    *psi_tabs = (u64*)malloc(sizeof(u64) * (size_t)n * R);
    *psiinv_tabs = (u64*)malloc(sizeof(u64) * (size_t)n * R);
    for (unsigned i = 0; i < R; i++) {
        kbits[i] = orc_bit_length(qs[i]);
        mus[i] = orc_mu(qs[i], kbits[i]);
        orc_fill_table(psis[i], qs[i], *psi_tabs + (size_t)i * n, n);
        orc_fill_table(orc_modinv(psis[i], qs[i]), qs[i], *psiinv_tabs + (size_t)i * n, n);
    }
}

/* secret_key [R][n]: ternary sample in, NTT domain out.  public_key [2][R][n]: second half = uniform sample (never
 * transformed: it is used as if already in the NTT domain), first half out = NTT(-(a*s + e)).  e [R][n]. */
int orc_bfv_keygen_core(u64* secret_key, u64* public_key, const u64* e, const u64* qs, const u64* psis,
                        unsigned r_plus_1, unsigned n)
{
    unsigned R = r_plus_1;
    if (R > 16 || R < 2) return -1;
    unsigned kbits[16]; u64 mus[16];
    u64 *psi_tabs, *psiinv_tabs;
    bfv_tables(qs, psis, R, n, &psi_tabs, &psiinv_tabs, mus, kbits);
    orc_forward_batch(secret_key, n, psi_tabs, R, R, qs, mus, kbits, 1);                          /* bfv_keygen.cuh:129 */
    orc_pointwise_batch(public_key, public_key + (size_t)R * n, secret_key, n, R, R, qs, mus, kbits);   /* :131-132 */
    orc_inverse_batch(public_key, n, psiinv_tabs, R, R, qs, mus, kbits, 1);                       /* :133 */
    for (size_t i = 0; i < (size_t)R * n; i++) {                                                  /* poly_add_negate_xq, :80-93 */
        u64 q = qs[i / n];
        u64 ra = public_key[i] + e[i];
        if (ra >= q) ra -= q;
        ra = q - ra;
        public_key[i] = ra * (ra != q);
    }
    orc_forward_batch(public_key, n, psi_tabs, R, R, qs, mus, kbits, 1);                          /* :145 */
    free(psi_tabs); free(psiinv_tabs);
    return 0;
}

/* c [2][R][n]: the ternary sample u in both halves on entry, the ciphertext (c0 | c1, last prime's slots dropped but
 * still present) on return.  public_key [2][R][n] (NTT domain), e [2][R][n], m [n] (message, values below t). */
int orc_bfv_encrypt_core(u64* c, const u64* public_key, const u64* e, const u64* m, const u64* qs, const u64* psis,
                         unsigned r_plus_1, unsigned n, u64 t)
{
    unsigned R = r_plus_1, r = R - 1;
    if (R > 16 || R < 2) return -1;
    unsigned kbits[16]; u64 mus[16];
    u64 *psi_tabs, *psiinv_tabs;
    bfv_tables(qs, psis, R, n, &psi_tabs, &psiinv_tabs, mus, kbits);
    orc_forward_batch(c, n, psi_tabs, 2 * R, R, qs, mus, kbits, 1);                               /* bfv_encryption.cuh:268 */
    orc_pointwise_batch(c, c, public_key, n, 2 * R, R, qs, mus, kbits);                           /* :269-270 */
    orc_inverse_batch(c, n, psiinv_tabs, 2 * R, R, qs, mus, kbits, 1);                            /* :271 */
    for (unsigned h = 0; h < 2; h++)                                                              /* poly_add_xq, :173-184 (`>`) */
        for (size_t i = 0; i < (size_t)R * n; i++) {
            size_t x = i + (size_t)n * R * h;
            u64 ra = c[x] + e[x];
            if (ra > qs[i / n]) ra -= qs[i / n];
            c[x] = ra;
        }
    u64 last_modulus = qs[R - 1], half_last = last_modulus >> 1;
    for (size_t i = 0; i < (size_t)2 * n; i++) {                                                  /* divide_and_round_q_last_inplace_add_x2, :110-124 */
        size_t x = (size_t)n * (R - 1) + i % n + ((size_t)n * R) * (i >= n);
        u64 ra = c[x] + half_last;
        if (ra >= last_modulus) ra -= last_modulus;
        c[x] = ra;
    }
    for (size_t i = 0; i < (size_t)2 * n * r; i++) {                                              /* divide_and_round_q_last_inplace_loop_xq, :126-171 */
        size_t i_i = i % n;
        unsigned index = (unsigned)((i % ((size_t)n * r)) / n);
        u64 q = qs[index];
        u64 half_mod = half_last % q;
        u64 iql = orc_modinv(qs[R - 1] % q, q);                                                   /* demo.cu:77 */
        unsigned second_half = i >= (size_t)n * r;
        size_t division = (i - (size_t)n * second_half * r) / n;
        u64* rns_poly_minus1 = c + (size_t)second_half * ((size_t)n * R) + (size_t)n * r;
        u64* input_poly = c + (size_t)second_half * ((size_t)n * R) + (size_t)n * division;
        u64 temp_poly_i = rns_poly_minus1[i_i] % q;
        if (temp_poly_i < half_mod) temp_poly_i += q;
        temp_poly_i -= half_mod;
        if (input_poly[i_i] < temp_poly_i) input_poly[i_i] += q;
        input_poly[i_i] -= temp_poly_i;
        input_poly[i_i] = barrett128((u128)input_poly[i_i] * iql, q, mus[index], kbits[index]);
    }
    for (unsigned j = 0; j < n; j++) {                                                            /* weird_m_stuff, :186-208 */
        u64 numerator = m[j] + ((t + 1) >> 1);
        u64 fix = numerator / t;
        for (unsigned i = 0; i < r; i++)
            c[j + (size_t)i * n] = (c[j + (size_t)i * n] + ((m[j] * (qs[i] / t)) + fix)) % qs[i];  /* qi_div_t: demo.cu:84-88 */
    }
    free(psi_tabs); free(psiinv_tabs);
    return 0;
}

/* ------------------------------------------------------------------ samplers */

/* VecCrypt with blks_per_chunk = 1 over an all-zero buffer (distributions.cuh:48-155, generate_random_default :249-276):
 * Salsa20/20 keystream, "expand 32-byte k", 64-bit nonce, block counter = block index.  Pinned by the ECRYPT
 * Salsa20/20 256-bit known-answer vector (tests/test_samplers.py). */
static uint32_t rotl32(uint32_t u, int c) { return (u << c) | (u >> (32 - c)); }
static uint32_t load_le32(const unsigned char* x)
{
    return (uint32_t)x[0] | ((uint32_t)x[1] << 8) | ((uint32_t)x[2] << 16) | ((uint32_t)x[3] << 24);
}
void orc_salsa20_keystream(unsigned char* out, unsigned long nblocks, const unsigned char* key, u64 nonce)
{
    static const unsigned char sigma[17] = "expand 32-byte k";
    for (unsigned long blockno = 0; blockno < nblocks; blockno++) {
        uint32_t j[16], x[16];
        j[0] = load_le32(sigma + 0);  j[1] = load_le32(key + 0);   j[2] = load_le32(key + 4);   j[3] = load_le32(key + 8);
        j[4] = load_le32(key + 12);   j[5] = load_le32(sigma + 4); j[6] = (uint32_t)nonce;      j[7] = (uint32_t)(nonce >> 32);
        j[8] = (uint32_t)blockno;     j[9] = (uint32_t)((u64)blockno >> 32);
        j[10] = load_le32(sigma + 8); j[11] = load_le32(key + 16); j[12] = load_le32(key + 20); j[13] = load_le32(key + 24);
        j[14] = load_le32(key + 28);  j[15] = load_le32(sigma + 12);
        for (int i = 0; i < 16; i++) x[i] = j[i];
        for (int i = 20; i > 0; i -= 2) {                                          /* ROUNDS = 20, salsa_common.h:14 */
            x[4] ^= rotl32(x[0] + x[12], 7);   x[8] ^= rotl32(x[4] + x[0], 9);    x[12] ^= rotl32(x[8] + x[4], 13);   x[0] ^= rotl32(x[12] + x[8], 18);
            x[9] ^= rotl32(x[5] + x[1], 7);    x[13] ^= rotl32(x[9] + x[5], 9);   x[1] ^= rotl32(x[13] + x[9], 13);   x[5] ^= rotl32(x[1] + x[13], 18);
            x[14] ^= rotl32(x[10] + x[6], 7);  x[2] ^= rotl32(x[14] + x[10], 9);  x[6] ^= rotl32(x[2] + x[14], 13);   x[10] ^= rotl32(x[6] + x[2], 18);
            x[3] ^= rotl32(x[15] + x[11], 7);  x[7] ^= rotl32(x[3] + x[15], 9);   x[11] ^= rotl32(x[7] + x[3], 13);   x[15] ^= rotl32(x[11] + x[7], 18);
            x[1] ^= rotl32(x[0] + x[3], 7);    x[2] ^= rotl32(x[1] + x[0], 9);    x[3] ^= rotl32(x[2] + x[1], 13);    x[0] ^= rotl32(x[3] + x[2], 18);
            x[6] ^= rotl32(x[5] + x[4], 7);    x[7] ^= rotl32(x[6] + x[5], 9);    x[4] ^= rotl32(x[7] + x[6], 13);    x[5] ^= rotl32(x[4] + x[7], 18);
            x[11] ^= rotl32(x[10] + x[9], 7);  x[8] ^= rotl32(x[11] + x[10], 9);  x[9] ^= rotl32(x[8] + x[11], 13);   x[10] ^= rotl32(x[9] + x[8], 18);
            x[12] ^= rotl32(x[15] + x[14], 7); x[13] ^= rotl32(x[12] + x[15], 9); x[14] ^= rotl32(x[13] + x[12], 13); x[15] ^= rotl32(x[14] + x[13], 18);
        }
        for (int i = 0; i < 16; i++) {
            uint32_t v = x[i] + j[i];
            unsigned char* o = out + blockno * 64 + 4 * i;
            o[0] = (unsigned char)v; o[1] = (unsigned char)(v >> 8); o[2] = (unsigned char)(v >> 16); o[3] = (unsigned char)(v >> 24);
        }
    }
}

/* ternary_dist_xq (bfv_keygen.cuh:14-31) / first half of convert_ternary_gaussian_x2 (bfv_encryption.cuh:17-45):
 * the SAME n bytes for every prime; note byte 255 yields the value 2 */
void orc_sample_ternary_xq(const unsigned char* in, u64* out, unsigned n, unsigned q_amount, const u64* qs)
{
    for (size_t i = 0; i < (size_t)n * q_amount; i++) {
        float d = (float)in[i % n];
        d /= (255.0f / 3);
        int b = (int)d - 1;
        out[i] = (u64)(b < 0) * qs[i / n] + (u64)(long long)b;
    }
}

/* uniform_dist_xq (bfv_keygen.cuh:33-45): one 64-bit word per coefficient per prime */
void orc_sample_uniform_xq(const unsigned char* in, u64* out, unsigned n, unsigned q_amount, const u64* qs)
{
    for (size_t i = 0; i < (size_t)n * q_amount; i++) {
        u64 w;
        memcpy(&w, in + 8 * i, 8);
        double d = (double)w;
        d /= (double)18446744073709551615ULL;          /* UINT64_MAX converts to 2^64 */
        d *= (double)(qs[i / n] - 1);
        out[i] = (u64)d;
    }
}

/* Inverse normal CDF for the Gaussian sampler's restatement.  The reference calls CUDA's normcdfinvf, whose last-ulp
 * behaviour is not specified: this (Wichura AS241, double precision, rounded to float) agrees with any faithful
 * implementation except where d * 3.2 lands within an ulp of an integer -- the Gaussian path is compared
 * statistically and by mismatch rate, not bit for bit. */
static double as241(double p)
{
    static const double a[8] = {3.3871328727963666080e0, 1.3314166789178437745e2, 1.9715909503065514427e3, 1.3731693765509461125e4,
                                4.5921953931549871457e4, 6.7265770927008700853e4, 3.3430575583588128105e4, 2.5090809287301226727e3};
    static const double b[8] = {1.0, 4.2313330701600911252e1, 6.8718700749205790830e2, 5.3941960214247511077e3, 2.1213794301586595867e4,
                                3.9307895800092710610e4, 2.8729085735721942674e4, 5.2264952788528545610e3};
    static const double c[8] = {1.42343711074968357734e0, 4.63033784615654529590e0, 5.76949722146069140550e0, 3.64784832476320460504e0,
                                1.27045825245236838258e0, 2.41780725177450611770e-1, 2.27238449892691845833e-2, 7.74545014278341407640e-4};
    static const double d[8] = {1.0, 2.05319162663775882187e0, 1.67638483018380384940e0, 6.89767334985100004550e-1, 1.48103976427480074590e-1,
                                1.51986665636164571966e-2, 5.47593808499534494600e-4, 1.05075007164441684324e-9};
    static const double e[8] = {6.65790464350110377720e0, 5.46378491116411436990e0, 1.78482653991729133580e0, 2.96560571828504891230e-1,
                                2.65321895265761230930e-2, 1.24266094738807843860e-3, 2.71155556874348757815e-5, 2.01033439929228813265e-7};
    static const double f[8] = {1.0, 5.99832206555887937690e-1, 1.36929880922735805310e-1, 1.48753612908506148525e-2, 7.86869131145613259100e-4,
                                1.84631831751005468180e-5, 1.42151175831644588870e-7, 2.04426310338993978564e-15};
    double q = p - 0.5, r, val;
    if (fabs(q) <= 0.425) {
        r = 0.180625 - q * q;
        return q * (((((((a[7] * r + a[6]) * r + a[5]) * r + a[4]) * r + a[3]) * r + a[2]) * r + a[1]) * r + a[0]) /
               (((((((b[7] * r + b[6]) * r + b[5]) * r + b[4]) * r + b[3]) * r + b[2]) * r + b[1]) * r + b[0]);
    }
    r = q < 0 ? p : 1.0 - p;
    r = sqrt(-log(r));
    if (r <= 5.0) {
        r -= 1.6;
        val = (((((((c[7] * r + c[6]) * r + c[5]) * r + c[4]) * r + c[3]) * r + c[2]) * r + c[1]) * r + c[0]) /
              (((((((d[7] * r + d[6]) * r + d[5]) * r + d[4]) * r + d[3]) * r + d[2]) * r + d[1]) * r + d[0]);
    } else {
        r -= 5.0;
        val = (((((((e[7] * r + e[6]) * r + e[5]) * r + e[4]) * r + e[3]) * r + e[2]) * r + e[1]) * r + e[0]) /
              (((((((f[7] * r + f[6]) * r + f[5]) * r + f[4]) * r + f[3]) * r + f[2]) * r + f[1]) * r + f[0]);
    }
    return q < 0 ? -val : val;
}

/* gaussian_dist_xq (bfv_keygen.cuh:47-79) / the two Gaussian halves of convert_ternary_gaussian_x2: the SAME n 32-bit
 * words for every prime; stdev 3.2, mean 0 (salsa_common.h:31-32), clamp +-19.2, truncation toward zero */
void orc_sample_gaussian_xq(const unsigned char* in, u64* out, unsigned n, unsigned q_amount, const u64* qs)
{
    for (size_t i = 0; i < (size_t)n * q_amount; i++) {
        uint32_t w;
        memcpy(&w, in + 4 * (i % n), 4);
        float d = (float)w;
        d /= 4294967295.0f;
        if (d == 0) d += 1.192092896e-07F;
        else if (d == 1) d -= 1.192092896e-07F;
        d = (float)as241((double)d);
        d = d * (float)3.2 + 0;
        if (d > 19.2) d = 19.2;
        else if (d < -19.2) d = -19.2;
        int dd = (int)d;
        out[i] = dd < 0 ? qs[i / n] + (u64)(long long)dd : (u64)dd;
    }
}

/* ------------------------------------------------------------- 30-bit path */

/* old/ntt_30bit.cuh: 32-bit words, 64-bit products; singleBarrett :52-68 on one machine word */
static inline uint32_t barrett30(u64 a, uint32_t q, uint32_t mu, int qbit)
{
    u64 rx = a >> (qbit - 2);
    rx *= mu;
    rx >>= qbit + 2;
    rx *= q;
    a -= rx;
    if (a >= q) a -= q;
    return (uint32_t)a;
}

/* forwardNTT (:321-359): every launch is the stage butterfly of CTBasedNTTInner (:199-227), length = 1 .. n/2 */
void orc30_forward(uint32_t* a, unsigned n, uint32_t q, uint32_t mu, int qbit, const uint32_t* psi_tab)
{
    for (unsigned length = 1; length < n; length *= 2) {
        unsigned step = (n / length) / 2;
        for (unsigned g = 0; g < n / 2; g++) {
            unsigned psi_step = g / step, j = psi_step * step * 2 + g % step;
            uint32_t psi = psi_tab[length + psi_step];
            uint32_t U = a[j];
            uint32_t V = barrett30((u64)a[j + step] * psi, q, mu, qbit);
            uint32_t r = U + V;
            r -= q * (r >= q);
            a[j] = r;
            U += q * (U < V);
            a[j + step] = U - V;
        }
    }
}

/* inverseNTT (:361-405): GSBasedINTTInner (:229-267) with the halving, length = n/2 .. 1 */
void orc30_inverse(uint32_t* a, unsigned n, uint32_t q, uint32_t mu, int qbit, const uint32_t* psiinv_tab)
{
    uint32_t q2 = (q + 1) >> 1;
    for (unsigned length = n / 2; length >= 1; length /= 2) {
        unsigned step = (n / length) / 2;
        for (unsigned g = 0; g < n / 2; g++) {
            unsigned psi_step = g / step, j = psi_step * step * 2 + g % step;
            uint32_t psiinv = psiinv_tab[length + psi_step];
            uint32_t U = a[j], V = a[j + step];
            uint32_t r = U + V;
            r -= q * (r >= q);
            a[j] = (r >> 1) + q2 * (r & 1);
            U += q * (U < V);
            uint32_t d = barrett30((u64)(U - V) * psiinv, q, mu, qbit);
            a[j + step] = (d >> 1) + q2 * (d & 1);
        }
    }
}

/* barrett_30bit (:10-35) */
void orc30_pointwise(uint32_t* a, const uint32_t* b, unsigned long count, uint32_t q, uint32_t mu, int qbit)
{
    for (unsigned long i = 0; i < count; i++) {
        u64 rc = (u64)a[i] * b[i];
        u64 rx = rc >> (qbit - 2);
        rx *= mu;
        rx >>= qbit + 2;
        rx *= q;
        rc -= rx;
        a[i] = rc < q ? (uint32_t)rc : (uint32_t)(rc - q);
    }
}

/* -------------------------------------------------------- synthetic inputs */

/* splitmix64, state x0 = seed, value = z mod q */
void orc_splitmix_fill(u64* a, unsigned long count, u64 seed, u64 q)
{
    u64 x = seed;
    for (unsigned long i = 0; i < count; i++) {
        x += 0x9E3779B97F4A7C15ULL;
        u64 z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        a[i] = z % q;
    }
}
