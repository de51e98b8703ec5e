/*
 * mi355ntt.h -- C ABI of the MI355X-native 60-bit NTT engine (libmi355ntt.so).
 *
 * This is the drop-in boundary for the NTT hot path of ozgunozerk/NTT-Cuda.  The reference has no
 * FFI layer: its "operator API" is a set of free functions/kernels in headers that every program
 * textually includes (BFV_Scheme/ntt_60bit.cuh, poly_arithmetic.cuh).  Each entry point below names
 * the reference interface it replaces (paths relative to /root/reference/BFV_Scheme/).
 *
 * Conventions:
 *   - All data pointers are DEVICE pointers to unsigned 64-bit words; plain pointers and sizes only.
 *   - Transforms are in place.  Inputs/outputs are canonical residues in [0, q).
 *   - Forward output is in bit-reversed order; inverse input is bit-reversed, its output is natural
 *     order and already scaled by n^-1 (as the reference's halving butterflies produce).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All calls are
 *     asynchronous on that stream; nothing here synchronises or allocates after context creation.
 *   - Every function returns MI355NTT_OK (0) or a negative MI355NTT_E* code.  (The reference reports
 *     no errors: unsupported n silently launches nothing, ntt_60bit.cuh:344-347.)
 *   - No global mutable state: the reference's __constant__ q_cons/q_bit_cons/mu_cons
 *     (ntt_60bit.cuh:8-10) live in an immutable context object, usable from any host thread.
 *   - Limits: n a power of two, 2^11 <= n <= 2^16 for the transforms (the reference dispatches
 *     2^11..2^15, ntt_60bit.cuh:316-347); <= 16 primes per context; q < 2^62 odd, q = 1 (mod 2n).
 */
#ifndef MI355NTT_H
#define MI355NTT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned long long mi355ntt_u64;
typedef unsigned int mi355ntt_u32;
typedef struct mi355ntt_ctx mi355ntt_ctx;
typedef void* mi355ntt_stream; /* hipStream_t */

enum {
    MI355NTT_OK = 0,
    MI355NTT_EINVAL = -1,      /* null pointer, bad count, num/division inconsistent                */
    MI355NTT_EUNSUPPORTED = -2, /* n not a supported power of two, too many primes, q out of range */
    MI355NTT_EHIP = -3,        /* a HIP runtime call failed (see mi355ntt_last_hip_error)           */
    MI355NTT_ENOMEM = -4,
    MI355NTT_EPARAM = -5       /* psi is not a primitive 2n-th root of unity mod q, q even, ...     */
};

#define MI355NTT_MAX_PRIMES 16 /* __constant__ arrays are [16], ntt_60bit.cuh:8-10 */

const char* mi355ntt_strerror(int code);
int mi355ntt_last_hip_error(void);         /* hipError_t of the last failing HIP call on this thread */
const char* mi355ntt_version(void);

/* ------------------------------------------------------------------------------------------------
 * Host-only parameter helpers (no GPU needed).  Replace helper.h:8-70, parameter.h:5-20 and the
 * mu / bit-length bootstrap at 60bit_ntt_test.cu:47-49, demo.cu:69,157-165.
 * ---------------------------------------------------------------------------------------------- */
unsigned     mi355ntt_bit_length(mi355ntt_u64 q);                                   /* demo.cu:69 */
mi355ntt_u64 mi355ntt_barrett_mu(mi355ntt_u64 q, unsigned bit_length);              /* 60bit_ntt_test.cu:47-49 */
/* 1 when the reference's single-subtraction Barrett (singleBarrett, ntt_60bit.cuh:44-61) returns the canonical residue
 * for EVERY product of two canonical operands mod q, 0 when it can come out q too large (see mi355ntt_ctx_create). */
int          mi355ntt_barrett_is_exact(mi355ntt_u64 q);
mi355ntt_u64 mi355ntt_mulmod(mi355ntt_u64 a, mi355ntt_u64 b, mi355ntt_u64 m);       /* host64x2 + operator%, uint128.h:278-341 */
mi355ntt_u64 mi355ntt_modpow(mi355ntt_u64 a, mi355ntt_u64 e, mi355ntt_u64 m);       /* modpow128, helper.h:8-28 */
mi355ntt_u64 mi355ntt_modinv(mi355ntt_u64 a, mi355ntt_u64 q);                       /* modinv128, helper.h:52-56 */
mi355ntt_u64 mi355ntt_bit_reverse(mi355ntt_u64 a, int bits);                        /* bitReverse, helper.h:58-70 */
/* fillTablePsi128 (parameter.h:5-12): psi_table[i] = psi^bitrev(i), psiinv_table[i] = psiinv^bitrev(i);
 * either output may be NULL. */
int mi355ntt_fill_tables(mi355ntt_u64 psi, mi355ntt_u64 psiinv, mi355ntt_u64 q, unsigned n,
                         mi355ntt_u64* host_psi_table, mi355ntt_u64* host_psiinv_table);
/* getParams (parameter.h:31-79): the reference's hard-coded single-prime sets, n in {2048..32768}. */
int mi355ntt_get_params(unsigned n, mi355ntt_u64* q, mi355ntt_u64* psi, mi355ntt_u64* psiinv,
                        mi355ntt_u64* ninv, unsigned* bit_length);

/* ------------------------------------------------------------------------------------------------
 * Context: replaces the caller-side bootstrap (demo.cu:62-196: bit lengths, mu, psi^-1, psi tables,
 * cudaMemcpyToSymbol of q_cons/q_bit_cons/mu_cons, table upload).  Immutable after creation.
 *
 * Arithmetic contract.  The throughput kernels compute the exact transform and return canonical residues.  That equals
 * what the reference's kernels print whenever its single-subtraction Barrett is exact for the modulus
 * (mi355ntt_barrett_is_exact: every modulus the reference ships, every q = 2^k - d with d^2 << 2^k).  For the rare
 * other primes (e.g. 68719230977, the second prime of decryption_test.cu) the reference occasionally returns q + r or,
 * one butterfly later, a wrong residue.  A context that holds such a prime therefore returns the reference's own words: its
 * transforms run the single-pass kernels in arithmetic class 0 -- singleBarrett with its one conditional subtraction written
 * out literally (ntt_60bit.cuh:44-61), the value carried from stage to stage exactly as the reference's memory carries it
 * (:199-222, :232-264), a halving in every inverse stage -- one read and one write of memory per transform, in place, for every
 * prime of the context (for the exact ones the literal words ARE the exact transform's), at every batch size and under stream
 * capture like any other call (round 6; until round 5 these polynomials went through stage-per-launch kernels and a gather
 * buffer).  MI355NTT_CTX_EXACT_ON_INEXACT_PRIMES passed to mi355ntt_ctx_create_ex selects the exact (lazy-arithmetic) kernels
 * regardless.  n = 65536 (beyond the reference's dispatch, ntt_60bit.cuh:316-347) runs the reference's coupling stage in memory
 * around two half-size class-0 transforms (up to 8 primes; more, and inexact primes narrower than 34 or wider than 61 bits, run the
 * literal stage-per-launch kernels).  Small batches run kernels of their own that spread a polynomial over many waves.
 * ---------------------------------------------------------------------------------------------- */
#define MI355NTT_CTX_EXACT_ON_INEXACT_PRIMES 1u
int mi355ntt_ctx_create(mi355ntt_ctx** out, unsigned n, unsigned num_primes,
                        const mi355ntt_u64* q, const mi355ntt_u64* psi, int device);
int mi355ntt_ctx_create_ex(mi355ntt_ctx** out, unsigned n, unsigned num_primes,
                           const mi355ntt_u64* q, const mi355ntt_u64* psi, int device, unsigned flags);
int mi355ntt_ctx_destroy(mi355ntt_ctx* ctx);
/* 0: exact (lazy-arithmetic) kernels; 1: every prime is Barrett-inexact -- literal (reference-arithmetic) kernels; 2: some primes
 * are -- the same literal kernels for the whole context, see above */
int mi355ntt_ctx_uses_literal_kernels(const mi355ntt_ctx* ctx);
/* diagnostic: the arithmetic class the throughput kernels of this context were selected by -- bits 0-3: headroom = min over the
 * primes of (64 - bit length), capped at 6 (values stay below 2^headroom q between partial reductions); bit 4: every prime is
 * "near 2^k" (q = 2^k - d with d < 2^24, 2^(64-k) d + 2 d < 2^k and 2 d^2 + 3 d < 2^k: the 3-instruction fold replaces the general
 * partial reduction).  0: the literal-arithmetic class of a context with a Barrett-inexact prime.  tests/test_gpu_fuzz_moduli.py
 * recomputes it for moduli drawn on both sides of every threshold. */
int mi355ntt_ctx_kernel_class(const mi355ntt_ctx* ctx);
unsigned mi355ntt_ctx_n(const mi355ntt_ctx* ctx);
unsigned mi355ntt_ctx_num_primes(const mi355ntt_ctx* ctx);
/* the device the context lives on.  Every launching call on a context (or on a BFV object) runs on THAT device: the
 * library switches to it for the duration of the call when the caller's current device differs and restores the
 * caller's device afterwards; `stream` must be a stream of the context's device (NULL = its default stream). */
int mi355ntt_ctx_device(const mi355ntt_ctx* ctx);
/* per-prime derived parameters; any out pointer may be NULL */
int mi355ntt_ctx_prime(const mi355ntt_ctx* ctx, unsigned prime_idx, mi355ntt_u64* q, mi355ntt_u64* mu,
                       unsigned* bit_length, mi355ntt_u64* psi, mi355ntt_u64* psiinv);
/* device pointers to the reference-format tables, [num_primes][n] contiguous (demo.cu:188-196) */
const mi355ntt_u64* mi355ntt_ctx_psi_tables(const mi355ntt_ctx* ctx);
const mi355ntt_u64* mi355ntt_ctx_psiinv_tables(const mi355ntt_ctx* ctx);

/* ------------------------------------------------------------------------------------------------
 * Transforms on a context
 *
 * Stream capture.  Every launching entry point on a context or a BFV object (transforms, pointwise and fused products, the
 * BFV drivers single and batched, the samplers) only enqueues work on `stream`: no allocation, no synchronisation, no
 * host read, no host-side memcpy after mi355ntt_ctx_create / mi355ntt_bfv_create (mi355ntt_bfv_keygen on a context created
 * with MI355NTT_CTX_EXACT_ON_INEXACT_PRIMES enqueues one device-to-device hipMemcpyAsync, which a capture records as a memcpy node).  They may therefore be called between
 * hipStreamBeginCapture and hipStreamEndCapture and the resulting hipGraph replayed (tools/lat_bench.cpp does, and checks
 * the results).  Not capture-safe: context / BFV creation and destruction, mi355ntt_ctx_probed_clock_mhz, and the
 * FIRST call of a raw-parameter entry point with a given table (it derives and caches a context: see below).
 *
 * n = 65536, batches from about one half-size transform per CU up: the forward transform runs two cooperating workgroups per
 * polynomial (they read each other's half of the input and exchange one "have read it" flag before storing in place; grid <=
 * one workgroup per CU, so it is resident as a whole).  The flag buffer is per device, shared with the 30-bit path's n = 65536
 * pair launches, and belongs to one stream at a time (at most one pair kernel of either word size is in flight per device): a
 * call on another stream while such a launch is still in flight, any call on a capturing stream and any call on a stream created
 * with a CU mask runs the single-workgroup launch instead (same words, ~10 % slower); the hand-over costs one hipEventRecord /
 * hipEventQuery on the host.  A workgroup whose partner has not become resident after 30 s of wall clock (the GPU's constant
 * 100 MHz counter, independent of the shader clock) GIVES UP rather than hang: it stores nothing, marks the launch dead (other
 * workgroups left without a partner, in this launch or in cooperating launches queued behind it, then end within 100 us) and raises a
 * host-visible error word; the next call
 * that wants a cooperating launch on that device returns MI355NTT_EHIP (mi355ntt_last_hip_error() = hipErrorLaunchFailure) without
 * launching and clears the condition, so the call after it runs normally.  The data of the launch that gave up is invalid, as after any
 * asynchronous device fault -- and so is the data of every cooperating launch that was ALREADY ENQUEUED behind it on that device: their
 * waiting workgroups find the dead mark and end without storing, although the calls that enqueued them have long returned
 * MI355NTT_OK.  Launches of the other kernels, and calls on other streams that never ask for the cooperating form, neither see nor
 * report the condition.  mi355ntt_pair_fault_count(device) counts these events and is never reset: a caller that compares it before
 * and after a stretch of n = 65536 work knows whether all of it is valid.  The process keeps its device context (no trap).  This can
 * only happen when another workload holds the device's CUs indefinitely.
 * ---------------------------------------------------------------------------------------------- */
/* cooperating launches on `device` that gave up on a partner since the library was loaded (sticky; see above) */
unsigned long long mi355ntt_pair_fault_count(int device);
/* forwardNTT (ntt_60bit.cuh:314-348): one polynomial, prime prime_idx */
int mi355ntt_forward(const mi355ntt_ctx* ctx, mi355ntt_u64* d_a, unsigned prime_idx, mi355ntt_stream stream);
/* inverseNTT (ntt_60bit.cuh:350-386) */
int mi355ntt_inverse(const mi355ntt_ctx* ctx, mi355ntt_u64* d_a, unsigned prime_idx, mi355ntt_stream stream);
/* forwardNTTdouble (ntt_60bit.cuh:267-312): two polynomials, same prime, on two streams */
int mi355ntt_forward_double(const mi355ntt_ctx* ctx, mi355ntt_u64* d_a, mi355ntt_u64* d_b, unsigned prime_idx,
                            mi355ntt_stream stream1, mi355ntt_stream stream2);
/* forwardNTT_batch / inverseNTT_batch (ntt_60bit.cuh:608-697): num polynomials of n words at
 * d_a + y*n; polynomial y uses prime (y % division) (ntt_60bit.cuh:391,404,422). 1 <= division <= num_primes. */
int mi355ntt_forward_batch(const mi355ntt_ctx* ctx, mi355ntt_u64* d_a, unsigned num, unsigned division,
                           mi355ntt_stream stream);
int mi355ntt_inverse_batch(const mi355ntt_ctx* ctx, mi355ntt_u64* d_a, unsigned num, unsigned division,
                           mi355ntt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Pointwise products on a context
 * ---------------------------------------------------------------------------------------------- */
/* barrett (poly_arithmetic.cuh:9), barrett_batch (:36) when d_c == d_a, barrett_batch_3param (:68)
 * otherwise: d_c[y*n+x] = d_a[y*n+x] * d_b[y*n+x] mod q[y % division]. */
int mi355ntt_pointwise_mul(const mi355ntt_ctx* ctx, mi355ntt_u64* d_c, const mi355ntt_u64* d_a,
                           const mi355ntt_u64* d_b, unsigned num, unsigned division, mi355ntt_stream stream);
/* barrett_int (poly_arithmetic.cuh:100) / poly_mul_int (:317): d_a[i] = d_a[i] * b mod q[prime_idx], one polynomial */
int mi355ntt_pointwise_mul_scalar(const mi355ntt_ctx* ctx, mi355ntt_u64* d_a, mi355ntt_u64 b,
                                  unsigned prime_idx, mi355ntt_stream stream);
/* The composition the BFV drivers run: forwardNTT_batch -> barrett_batch -> inverseNTT_batch
 * (bfv_encryption.cuh:268-271, bfv_decryption.cuh:98-101, half_poly_mul_device poly_arithmetic.cuh:303-310),
 * fused into one pass over d_a: d_a[y] = INTT( NTT(d_a[y]) (.) d_bhat[y] ).  d_bhat is in the NTT domain. */
int mi355ntt_polymul_batch(const mi355ntt_ctx* ctx, mi355ntt_u64* d_a, const mi355ntt_u64* d_bhat,
                           unsigned num, unsigned division, mi355ntt_stream stream);

/* The same with second operands shared by the batch: `division` of them per key group of `group` consecutive
 * polynomials (group a multiple of division; 0 = the whole batch is one group): polynomial y multiplies with
 * d_bhat[(y / group) division + y % division].  The batched BFV drivers multiply every ciphertext with the same key
 * this way (encryption: two groups, the two components of the public key). */
int mi355ntt_polymul_batch_shared(const mi355ntt_ctx* ctx, mi355ntt_u64* d_a, const mi355ntt_u64* d_bhat,
                                  unsigned num, unsigned division, unsigned group, mi355ntt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU: shards of a batch.  The reference is single-GPU; its batch kernels let no two polynomials interact
 * (blockIdx.y is the data offset y*n and the modulus y % division, nothing else: ntt_60bit.cuh:391,404,422), so a batch shards
 * into contiguous ranges of WHOLE polynomials whose starts are multiples of `division`, one replicated context per device, no
 * collective on the data path.  Two forms: one process per GPU over torch.distributed / RCCL (ntt_cuda_amd/shard.py), and ONE
 * process driving several devices through the object below (peer copies over xGMI for a root-resident batch).
 * ---------------------------------------------------------------------------------------------- */
/* the partitioning rule: shard `rank` of `world` = polynomials [*first, *first + *count); groups of `division` polynomials dealt as
 * evenly as possible, a ragged tail (num % division) with the last rank */
int mi355ntt_shard_range(unsigned num, unsigned division, unsigned rank, unsigned world, unsigned* first, unsigned* count);

typedef struct mi355ntt_shards mi355ntt_shards;
enum { MI355NTT_OP_FORWARD = 0, MI355NTT_OP_INVERSE = 1, MI355NTT_OP_FORWARD_INVERSE = 2, MI355NTT_OP_POLYMUL = 3 };
/* ctxs[r]: the context of shard r (same n and primes everywhere; normally one per device, several on one device are allowed --
 * "logical shards").  ctxs[0]'s device is the ROOT.  max_polys_per_piece > 0 allocates, on every non-root lane, three staging
 * buffers of that many polynomials for mi355ntt_shards_scatter_transform_gather (0: device-resident shards only).  Creates three
 * streams per lane; enables peer access towards the root where the hardware offers it. */
int mi355ntt_shards_create(mi355ntt_shards** out, const mi355ntt_ctx* const* ctxs, unsigned world, unsigned max_polys_per_piece);
int mi355ntt_shards_destroy(mi355ntt_shards* shards);
unsigned mi355ntt_shards_world(const mi355ntt_shards* shards);
/* Device-resident shards: d_shard[r] (on ctxs[r]'s device) holds the polynomials mi355ntt_shard_range names for r; every shard is
 * transformed in place, all devices concurrently.  d_bhat_shard: per-shard second operands for MI355NTT_OP_POLYMUL, else NULL.
 * `stream` (of the root device) is the fork and join point: the launches see what was enqueued on it before the call, what is
 * enqueued on it afterwards sees every shard done.  No host synchronisation. */
int mi355ntt_shards_transform(mi355ntt_shards* shards, int op, mi355ntt_u64* const* d_shard, const mi355ntt_u64* const* d_bhat_shard,
                              unsigned num, unsigned division, mi355ntt_stream stream);
/* Root-resident batch d_full [num][n] on the root device: every other lane receives its shard in up to `chunks` pieces by peer
 * copies, transforms piece k while piece k + 1 arrives and piece k - 1 returns into d_full; the root transforms its own shard in
 * place.  MI355NTT_OP_FORWARD / _INVERSE / _FORWARD_INVERSE.  Same fork / join contract on `stream`. */
int mi355ntt_shards_scatter_transform_gather(mi355ntt_shards* shards, int op, mi355ntt_u64* d_full, unsigned num, unsigned division,
                                             unsigned chunks, mi355ntt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement helpers (no reference counterpart: the reference's programs draw inputs on the host, 60bit_ntt_test.cu:52-60)
 * ---------------------------------------------------------------------------------------------- */
/* Synthetic inputs of the benchmark recipe, generated on the context's device: polynomial y of d_a [num][n] receives the
 * splitmix64 stream of seed seed_base + y reduced mod q[y % division] (state x0 = seed,
 * x += 0x9E3779B97F4A7C15 per value, value = mix(x) mod q). */
int mi355ntt_synth_splitmix(const mi355ntt_ctx* ctx, mi355ntt_u64* d_a, unsigned num, unsigned division,
                            mi355ntt_u64 seed_base, mi355ntt_stream stream);
/* Clock probe.  mi355ntt_ctx_clock_probe enqueues a one-wave kernel on `stream` that counts shader cycles over 20 us of the
 * 100 MHz constant clock; mi355ntt_ctx_probed_clock_mhz returns the last probe's result in MHz: the shader clock the launches
 * enqueued in front of the probe left the chip at (bench.py prices its VALU ceiling with it; the power management moves the clock
 * over milliseconds).  The second call synchronises the device.  0 until a probe has run. */
int mi355ntt_ctx_clock_probe(const mi355ntt_ctx* ctx, mi355ntt_stream stream);
/* Foreign load for tests and measurements: enqueues on `stream` a kernel of `workgroups` workgroups that each hold one whole CU (1024
 * threads, 144 KiB of LDS -- the footprint of the n = 2^15 kernels) for `microseconds` and touch no memory.  tests/test_gpu_stress.py
 * uses it to take half of the CUs away from under the cooperating-workgroup launches (they must come out slow, not wrong). */
int mi355ntt_ctx_occupy(const mi355ntt_ctx* ctx, unsigned workgroups, unsigned microseconds, mi355ntt_stream stream);
int mi355ntt_ctx_probed_clock_mhz(const mi355ntt_ctx* ctx, double* mhz);

/* ------------------------------------------------------------------------------------------------
 * Raw-parameter entry points: signature-compatible with the reference (the caller supplies q, mu, bit_length and
 * reference-format device tables on every call, ntt_60bit.cuh:314,350,608,652 + the __constant__ moduli of :8-10).
 *
 * Routing.  A caller that passes what the reference's own bootstrap computes -- mu = floor(2^(2k)/q), bit_length =
 * bitlen(q), tables = fillTablePsi128 of a primitive 2n-th root (60bit_ntt_test.cu:47-49, demo.cu:69,188-196) -- for
 * moduli on which the reference's single-subtraction Barrett is exact (mi355ntt_barrett_is_exact) gets, word for word,
 * the exact transform, so those calls run the THROUGHPUT kernels: on first sight of a (device, n, moduli, mu,
 * bit_length, table address) the library reads the root out of the table, derives a context, compares the caller's
 * table with the derived one, and keeps the context (<= 32, least recently used evicted; first sight synchronises
 * the call's stream -- the table may have been filled asynchronously on it -- and reads the table from the host).  A set with a
 * Barrett-inexact modulus derives a class-0 context (the reference's own butterflies in the single-pass kernels: the reference's
 * words, see the arithmetic contract above).  Any other call (hand-made mu, a table that is not root^bitrev(i), n outside
 * 2^11..2^16, a table that is not 16-byte aligned) follows Algorithm 7 (singleBarrett, ntt_60bit.cuh:44-61) literally with the caller's
 * numbers: the literal stage kernels (the stages inside 2^14 coefficients out of LDS + one stage launch at n = 2^15 = 2 passes
 * over memory; the reference makes 4 and 5).
 *   The reference reads the table on every call, so a cached context must never outlive the table's contents (a
 * freed table whose address is handed out again, a table rewritten in place).  Default, CHECKED: in front of every
 * transform a small kernel compares the caller's table with the context's on the device (stream-ordered, no host
 * synchronisation), the throughput kernel runs only if they are equal and the literal kernels only if they are not:
 * always the caller's table's result, for a few microseconds per call.  The words the comparison writes are per (table, stream): calls on
 * different streams share nothing and a call never touches a stream other than its own (up to 16 streams per table; calls on further
 * streams, and checked calls on a CAPTURING stream, run the literal stage kernels).  A table found changed makes its next call derive a
 * new context.  mi355ntt_raw_trust_tables() is the caller's promise that a table stays
 * as it is: calls on it skip the check (and n = 2^16, whose split path cannot be guarded, runs the throughput kernels
 * only then).  mi355ntt_raw_cache_clear() forgets everything; MI355NTT_RAW_LITERAL=1 in the environment disables the
 * routing.  Calls run on the CURRENT device, as the reference's do.
 * ---------------------------------------------------------------------------------------------- */
int mi355ntt_raw_cache_clear(void);
/* 1 when calls with these arguments run the throughput kernels, 0 for the literal ones (derives + caches on first use) */
int mi355ntt_raw_uses_fast_kernels(unsigned n, const mi355ntt_u64* d_table, int inverse, unsigned division,
                                   const mi355ntt_u64* q, const mi355ntt_u64* mu, const unsigned* bit_length);
/* promise: the table at d_table keeps its contents until mi355ntt_raw_cache_clear().  1: calls with these arguments now
 * run the throughput kernels without the per-call comparison; 0: they run the literal kernels (nothing to trust) */
int mi355ntt_raw_trust_tables(unsigned n, const mi355ntt_u64* d_table, int inverse, unsigned division,
                              const mi355ntt_u64* q, const mi355ntt_u64* mu, const unsigned* bit_length);
int mi355ntt_forward_raw(mi355ntt_u64* d_a, unsigned n, mi355ntt_stream stream, mi355ntt_u64 q, mi355ntt_u64 mu,
                         int bit_length, const mi355ntt_u64* d_psi_table);                        /* forwardNTT :314 */
int mi355ntt_inverse_raw(mi355ntt_u64* d_a, unsigned n, mi355ntt_stream stream, mi355ntt_u64 q, mi355ntt_u64 mu,
                         int bit_length, const mi355ntt_u64* d_psiinv_table);                     /* inverseNTT :350 */
/* q/mu/bit_length: HOST arrays of `division` entries standing in for q_cons/mu_cons/q_bit_cons */
int mi355ntt_forward_batch_raw(mi355ntt_u64* d_a, unsigned n, const mi355ntt_u64* d_psi_tables, unsigned num,
                               unsigned division, const mi355ntt_u64* q, const mi355ntt_u64* mu,
                               const unsigned* bit_length, mi355ntt_stream stream);               /* :608 */
int mi355ntt_inverse_batch_raw(mi355ntt_u64* d_a, unsigned n, const mi355ntt_u64* d_psiinv_tables, unsigned num,
                               unsigned division, const mi355ntt_u64* q, const mi355ntt_u64* mu,
                               const unsigned* bit_length, mi355ntt_stream stream);               /* :652 */
int mi355ntt_barrett_raw(mi355ntt_u64* d_c, const mi355ntt_u64* d_a, const mi355ntt_u64* d_b, unsigned n,
                         unsigned num, unsigned division, const mi355ntt_u64* q, const mi355ntt_u64* mu,
                         const unsigned* bit_length, mi355ntt_stream stream);     /* barrett*, poly_arithmetic.cuh:9-98 */
int mi355ntt_barrett_int_raw(mi355ntt_u64* d_a, mi355ntt_u64 b, unsigned n, mi355ntt_u64 q, mi355ntt_u64 mu,
                             int bit_length, mi355ntt_stream stream);                             /* barrett_int :100 */

/* The stand-alone element-wise host wrappers of poly_arithmetic.cuh:312-352, same argument order (one polynomial of n words, in
 * place in d_a; the BFV drivers above carry the same arithmetic fused into their own kernels).  The reference's words, quirks
 * included: poly_add / poly_add_integer reduce with `>` (a sum equal to q stays q, :144-166); poly_sub (:168-179) only adds q where
 * a[i] < b[i] and never subtracts b -- mirrored literally, it is what the reference computes; poly_negate maps 0 to 0 (:334-338);
 * poly_mul_int_t masks the low 64 bits of a[i] b with t - 1 held in a 32-bit register (:128-142).  Any pointer to 64-bit words is
 * accepted, as by the reference (16-byte aligned ones run 16 bytes per lane).  One difference, deliberate: all n words are processed --
 * the reference launches n / 256 blocks of 256 threads and leaves a tail of n % 256 words untouched (:314,324,329,342,347); its callers
 * only pass ring degrees, which are multiples of 256. */
int mi355ntt_poly_add_raw(mi355ntt_u64* d_a, const mi355ntt_u64* d_b, unsigned n, mi355ntt_stream stream, mi355ntt_u64 q);          /* poly_add_device :312 */
int mi355ntt_poly_mul_int_t_raw(mi355ntt_u64* d_a, mi355ntt_u64 b, unsigned n, mi355ntt_stream stream, mi355ntt_u64 t);             /* poly_mul_int_t :322 */
int mi355ntt_poly_sub_raw(mi355ntt_u64* d_a, const mi355ntt_u64* d_b, unsigned n, mi355ntt_stream stream, mi355ntt_u64 q);          /* poly_sub_device :327 */
int mi355ntt_poly_negate_raw(mi355ntt_u64* d_a, unsigned n, mi355ntt_stream stream, mi355ntt_u64 q);                                /* poly_negate_device :340 */
int mi355ntt_poly_add_integer_raw(mi355ntt_u64* d_a, mi355ntt_u64 b, unsigned n, mi355ntt_stream stream, mi355ntt_u64 q);           /* poly_add_integer_device(_default) :345-352 */

/* ------------------------------------------------------------------------------------------------
 * The reference's 30-bit path (old/ntt_30bit.cuh): 32-bit words, q < 2^30, the caller's mu
 * (floor(2^(2 bits) / q), old/30bit_ntt_test.cu:47-48) and 32-bit psi tables.  Same argument lists as the reference's
 * launchers; the _batch_ forms take `num` polynomials of the same prime.  n in {2048 .. 65536} (the reference's forwardNTT /
 * inverseNTT dispatch N = 2048 ... 65536, old/ntt_30bit.cuh:271-283,321-405).
 * Routing: with the canonical mu / bit_length and a modulus on which the single-subtraction Barrett is exact the calls run
 * the native kernels (32 coefficients per thread in registers, lazy 32-bit Shoup butterflies; the companions of the CALLER's
 * table are recomputed on the device in front of every call, so the transform always follows the table passed; a table
 * entry >= q sends the call to the literal kernels on the device side); otherwise the literal kernels (the reference's
 * butterflies with the caller's mu).  n = 2^16: one stage in memory + two 2^15 transforms.
 * ---------------------------------------------------------------------------------------------- */
int mi355ntt_forward30_raw(mi355ntt_u32* d_a, unsigned n, mi355ntt_stream stream, mi355ntt_u32 q, mi355ntt_u32 mu, int bit_length,
                           const mi355ntt_u32* d_psi_table);                                       /* forwardNTT, :321-359 */
int mi355ntt_inverse30_raw(mi355ntt_u32* d_a, unsigned n, mi355ntt_stream stream, mi355ntt_u32 q, mi355ntt_u32 mu, int bit_length,
                           const mi355ntt_u32* d_psiinv_table);                                    /* inverseNTT, :361-405 */
int mi355ntt_forward30_batch_raw(mi355ntt_u32* d_a, unsigned n, const mi355ntt_u32* d_psi_table, unsigned num, mi355ntt_u32 q,
                                 mi355ntt_u32 mu, int bit_length, mi355ntt_stream stream);
int mi355ntt_inverse30_batch_raw(mi355ntt_u32* d_a, unsigned n, const mi355ntt_u32* d_psiinv_table, unsigned num, mi355ntt_u32 q,
                                 mi355ntt_u32 mu, int bit_length, mi355ntt_stream stream);
/* barrett_30bit (:10-35): a[i] = a[i] * b[i] mod q over `count` words */
int mi355ntt_barrett30_raw(mi355ntt_u32* d_a, const mi355ntt_u32* d_b, size_t count, mi355ntt_u32 q, mi355ntt_u32 mu, int bit_length,
                           mi355ntt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * BFV launch layer around the NTT path: the parameter bootstrap of demo.cu:62-272 and the
 * drivers keygen_rns / encryption_rns / decryption_rns AFTER their samplers -- the sampled polynomials are inputs (the
 * Salsa20 / Gaussian samplers of distributions.cuh stay with the caller).  num_primes counts the special last prime
 * that encryption drops ("q_amount" of keygen_rns / encryption_rns; decryption_rns is called with q_amount - 1).
 * Layouts are the reference's: secret key [num_primes][n], public key and ciphertext [2][num_primes][n].
 * Requires t a power of two with q_i = 1 (mod t) ("q mod t is assumed 1", bfv_encryption.cuh:189) and gamma odd < 2^62.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mi355ntt_bfv mi355ntt_bfv;
/* ctx_flags: as mi355ntt_ctx_create_ex (the object owns an NTT context over all num_primes primes) */
int mi355ntt_bfv_create(mi355ntt_bfv** out, unsigned n, unsigned num_primes, const mi355ntt_u64* q, const mi355ntt_u64* psi,
                        mi355ntt_u64 t, mi355ntt_u64 gamma, int device, unsigned ctx_flags);
int mi355ntt_bfv_destroy(mi355ntt_bfv* bfv);
const mi355ntt_ctx* mi355ntt_bfv_ntt(const mi355ntt_bfv* bfv);
/* the bootstrap constants (host arrays; any pointer may be NULL): inv_punctured_q[r], neg_inv_q_mod_t_gamma[2],
 * prod_t_gamma_mod_q[r], inv_q_last_mod_q[r], q_div_t[r + 1], base_change_matrix[2][r], mu_gamma; r = num_primes - 1
 * (demo.cu:73-79, 84-88, 103-125, 218-226, 262-301) */
int mi355ntt_bfv_constants(const mi355ntt_bfv* bfv, mi355ntt_u64* inv_punctured_q, mi355ntt_u64* neg_inv_q_mod_t_gamma,
                           mi355ntt_u64* prod_t_gamma_mod_q, mi355ntt_u64* inv_q_last_mod_q, mi355ntt_u64* q_div_t,
                           mi355ntt_u64* base_change_matrix, mi355ntt_u64* mu_gamma);
/* keygen_rns (bfv_keygen.cuh:95-151) from :129 on.  In: d_secret_key = the ternary sample as residues, second half of
 * d_public_key = the uniform sample, d_e [num_primes][n] = the error sample.  Out: secret key and both halves of the
 * public key in the NTT domain. */
int mi355ntt_bfv_keygen(const mi355ntt_bfv* bfv, mi355ntt_u64* d_secret_key, mi355ntt_u64* d_public_key, const mi355ntt_u64* d_e,
                        mi355ntt_stream stream);
/* encryption_rns (bfv_encryption.cuh:223-290) from :268 on.  In: d_c = the ternary sample u in both halves,
 * d_e [2][num_primes][n], d_m [n] (message, values < t).  Out: the ciphertext in d_c (slots of the last prime unused). */
int mi355ntt_bfv_encrypt(const mi355ntt_bfv* bfv, mi355ntt_u64* d_c, const mi355ntt_u64* d_public_key, const mi355ntt_u64* d_e,
                         const mi355ntt_u64* d_m, mi355ntt_stream stream);
/* decryption_rns (bfv_decryption.cuh:76-138).  d_secret_key: first num_primes - 1 polynomials of the NTT-domain key.
 * d_c is overwritten exactly as the reference overwrites it; the plaintext is at d_c + n * (num_primes - 2). */
int mi355ntt_bfv_decrypt(const mi355ntt_bfv* bfv, mi355ntt_u64* d_c, const mi355ntt_u64* d_secret_key, mi355ntt_stream stream);

/* Batched drivers: `count` ciphertexts per call in the layout [2][count][num_primes][n] (all first components, then all
 * second components; d_e the same, d_m [count][n]); each ciphertext gets exactly the words mi355ntt_bfv_encrypt /
 * _decrypt leave for it (except that decryption also overwrites the second component's slot of the dropped last
 * prime, which the single driver leaves alone and nothing reads).  The secret key must hold all num_primes polynomials (keygen's output); plaintext of
 * ciphertext z: d_c + (z num_primes + num_primes - 2) n.  One fused product per component for the whole batch and one
 * launch per element-wise step, instead of the per-ciphertext launch sequence (which is bound by launch overhead).
 * Limits: count <= 65535 and count * num_primes < 2^23; beyond them MI355NTT_EUNSUPPORTED, returned before d_c is touched. */
int mi355ntt_bfv_encrypt_batch(const mi355ntt_bfv* bfv, mi355ntt_u64* d_c, const mi355ntt_u64* d_public_key, const mi355ntt_u64* d_e,
                               const mi355ntt_u64* d_m, unsigned count, mi355ntt_stream stream);
int mi355ntt_bfv_decrypt_batch(const mi355ntt_bfv* bfv, mi355ntt_u64* d_c, const mi355ntt_u64* d_secret_key, unsigned count,
                               mi355ntt_stream stream);

/* ---- samplers and the complete drivers --------------------------------------------------
 * generate_random / generate_random_default (distributions.cuh:192-276): Salsa20/20 keystream, floor(nbytes / 64)
 * blocks written to d_out (16-byte aligned), 64-bit nonce, block counter from 0.  The reference uses key = 32 x 0x01
 * (_default) or 32 x 77, nonce 0 -- the same stream on every call; pass a fresh nonce per call for anything real. */
int mi355ntt_salsa20_keystream(void* d_out, size_t nbytes, const unsigned char* key32, mi355ntt_u64 nonce, mi355ntt_stream stream);
/* bytes of keystream keygen_rns / encryption_rns consume (bfv_keygen.cuh:99, bfv_encryption.cuh:228) */
size_t mi355ntt_bfv_keygen_random_bytes(const mi355ntt_bfv* bfv);
size_t mi355ntt_bfv_encrypt_random_bytes(const mi355ntt_bfv* bfv);
/* ternary_dist_xq + uniform_dist_xq + gaussian_dist_xq (bfv_keygen.cuh:14-79): random bytes -> ternary secret key,
 * uniform second half of the public key, error polynomial d_temp.  The integer conversions are the reference's words;
 * the Gaussian one calls the device normcdfinvf (CUDA's is not specified to the ulp): same distribution. */
int mi355ntt_bfv_sample_keygen(const mi355ntt_bfv* bfv, const void* d_in, mi355ntt_u64* d_secret_key, mi355ntt_u64* d_public_key,
                               mi355ntt_u64* d_temp, mi355ntt_stream stream);
/* convert_ternary_gaussian_x2 (bfv_encryption.cuh:17-109): random bytes -> u in both halves of d_c, e0 | e1 in d_e */
int mi355ntt_bfv_sample_encrypt(const mi355ntt_bfv* bfv, const void* d_in, mi355ntt_u64* d_c, mi355ntt_u64* d_e, mi355ntt_stream stream);
/* keygen_rns (bfv_keygen.cuh:95-151) and encryption_rns (bfv_encryption.cuh:223-290) complete: keystream with the
 * reference's default key into the caller's d_in (sizes above), samplers, then the drivers declared earlier */
int mi355ntt_bfv_keygen_rns(const mi355ntt_bfv* bfv, void* d_in, mi355ntt_u64* d_secret_key, mi355ntt_u64* d_public_key,
                            mi355ntt_u64* d_temp, mi355ntt_u64 nonce, mi355ntt_stream stream);
int mi355ntt_bfv_encryption_rns(const mi355ntt_bfv* bfv, mi355ntt_u64* d_c, const mi355ntt_u64* d_public_key, void* d_in,
                                mi355ntt_u64* d_e, const mi355ntt_u64* d_m, mi355ntt_u64 nonce, mi355ntt_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355NTT_H */
