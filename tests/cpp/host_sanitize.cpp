// host_sanitize.cpp -- the HOST side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only; GPU sanitizers are
// not available on this pool).  capi.cpp, hostparams.cpp, bfv_host.cpp and shard.cpp are compiled with g++ -fsanitize=address,undefined
// and linked in front of libmi355ntt.so (which supplies the kernel launchers they call); this program then walks the host-only
// helpers, every argument check, and the creation / failure / destruction paths that run before a device is needed.  Without a GPU
// mi355ntt_ctx_create fails with MI355NTT_EHIP after allocating its host state: the unwinding of that failure is the interesting part.
// With a GPU (the GPU box) creation succeeds and the objects are destroyed again.  Exit status 0 and no sanitizer report = pass.
//   tests/test_abi_host.py::test_host_objects_under_asan_ubsan builds and runs it.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mi355ntt.h"

static int failures = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::printf("FAILED line %d: %s\n", __LINE__, #cond);            \
            failures++;                                                      \
        }                                                                    \
    } while (0)

typedef mi355ntt_u64 u64;

int main()
{
    // ---- host-only helpers ----
    const u64 q60 = 1152921504606584833ULL, psi60 = 4443670208963ULL;
    CHECK(mi355ntt_bit_length(q60) == 60 && mi355ntt_bit_length(1) == 1 && mi355ntt_bit_length(0) == 0);
    CHECK(mi355ntt_barrett_mu(q60, 60) == (u64)((((unsigned __int128)1) << 120) / q60));
    CHECK(mi355ntt_barrett_mu(q60, 0) == 0 && mi355ntt_barrett_mu(0, 60) == 0 && mi355ntt_barrett_mu(q60, 64) == 0);
    CHECK(mi355ntt_mulmod(q60 - 1, q60 - 1, q60) == 1 && mi355ntt_mulmod(5, 7, 0) == 0);
    CHECK(mi355ntt_modpow(psi60, 32768, q60) == q60 - 1 && mi355ntt_modpow(3, 0, q60) == 1 && mi355ntt_modpow(3, 5, 0) == 0);
    const u64 inv = mi355ntt_modinv(psi60, q60);
    CHECK(mi355ntt_mulmod(inv, psi60, q60) == 1 && mi355ntt_modinv(3, 2) == 0);
    CHECK(mi355ntt_bit_reverse(1, 15) == 16384 && mi355ntt_bit_reverse(0x7fff, 15) == 0x7fff && mi355ntt_bit_reverse(5, 0) == 0);
    CHECK(mi355ntt_barrett_is_exact(q60) == 1 && mi355ntt_barrett_is_exact(68719230977ULL) == 0 && mi355ntt_barrett_is_exact(3) == 0 &&
          mi355ntt_barrett_is_exact(~0ULL) == 0);
    for (unsigned n : {2048u, 4096u, 8192u, 16384u, 32768u}) {
        u64 q, psi, psiinv, ninv;
        unsigned bits;
        CHECK(mi355ntt_get_params(n, &q, &psi, &psiinv, &ninv, &bits) == MI355NTT_OK);
        CHECK(mi355ntt_mulmod(psi, psiinv, q) == 1 && mi355ntt_mulmod(ninv, n, q) == 1 && mi355ntt_modpow(psi, n, q) == q - 1);
        std::vector<u64> tp(n), ti(n);
        CHECK(mi355ntt_fill_tables(psi, psiinv, q, n, tp.data(), ti.data()) == MI355NTT_OK);
        CHECK(tp[0] == 1 && ti[0] == 1 && tp[n / 2] == psi && mi355ntt_mulmod(tp[n - 1], ti[n - 1], q) == 1);
        CHECK(mi355ntt_fill_tables(psi, psiinv, q, n, nullptr, nullptr) == MI355NTT_OK);
    }
    CHECK(mi355ntt_get_params(1000, nullptr, nullptr, nullptr, nullptr, nullptr) != MI355NTT_OK);
    CHECK(mi355ntt_fill_tables(psi60, inv, q60, 1000, nullptr, nullptr) == MI355NTT_EINVAL);
    CHECK(mi355ntt_strerror(MI355NTT_EPARAM) != nullptr && mi355ntt_strerror(12345) != nullptr && mi355ntt_version() != nullptr);

    // ---- the shard partition (host only) ----
    for (unsigned num : {0u, 1u, 7u, 1023u, 8192u})
        for (unsigned div : {1u, 3u, 4u, 16u})
            for (unsigned world : {1u, 2u, 8u}) {
                unsigned seen = 0;
                for (unsigned r = 0; r < world; r++) {
                    unsigned first = 99, count = 99;
                    CHECK(mi355ntt_shard_range(num, div, r, world, &first, &count) == MI355NTT_OK);
                    CHECK(first == seen && first % div == 0);
                    seen += count;
                }
                CHECK(seen == num);
            }
    CHECK(mi355ntt_shard_range(8, 0, 0, 1, nullptr, nullptr) == MI355NTT_EINVAL && mi355ntt_shard_range(8, 4, 2, 2, nullptr, nullptr) == MI355NTT_EINVAL);
    CHECK(mi355ntt_shard_range(8, 4, 0, 2, nullptr, nullptr) == MI355NTT_OK);

    // ---- argument checks that return before a device is touched ----
    mi355ntt_ctx* ctx = nullptr;
    u64 qs[17], psis[17];
    for (int i = 0; i < 17; i++) { qs[i] = q60; psis[i] = psi60; }
    CHECK(mi355ntt_ctx_create(nullptr, 32768, 1, qs, psis, 0) == MI355NTT_EINVAL);
    CHECK(mi355ntt_ctx_create(&ctx, 32768, 1, nullptr, psis, 0) == MI355NTT_EINVAL && ctx == nullptr);
    CHECK(mi355ntt_ctx_create(&ctx, 1000, 1, qs, psis, 0) == MI355NTT_EUNSUPPORTED);
    CHECK(mi355ntt_ctx_create(&ctx, 1024, 1, qs, psis, 0) == MI355NTT_EUNSUPPORTED);
    CHECK(mi355ntt_ctx_create(&ctx, 131072, 1, qs, psis, 0) == MI355NTT_EUNSUPPORTED);
    CHECK(mi355ntt_ctx_create(&ctx, 32768, 0, qs, psis, 0) == MI355NTT_EUNSUPPORTED);
    CHECK(mi355ntt_ctx_create(&ctx, 32768, 17, qs, psis, 0) == MI355NTT_EUNSUPPORTED);
    u64 bad_psi = psi60 + 1, even_q = q60 + 1, wide_q = (1ULL << 63) + 1;
    CHECK(mi355ntt_ctx_create(&ctx, 32768, 1, qs, &bad_psi, 0) == MI355NTT_EPARAM);
    CHECK(mi355ntt_ctx_create(&ctx, 32768, 1, &even_q, psis, 0) == MI355NTT_EUNSUPPORTED);
    CHECK(mi355ntt_ctx_create(&ctx, 32768, 1, &wide_q, psis, 0) == MI355NTT_EUNSUPPORTED);
    CHECK(ctx == nullptr);
    CHECK(mi355ntt_ctx_destroy(nullptr) == MI355NTT_OK);
    CHECK(mi355ntt_ctx_n(nullptr) == 0 && mi355ntt_ctx_num_primes(nullptr) == 0 && mi355ntt_ctx_device(nullptr) == -1);
    CHECK(mi355ntt_ctx_uses_literal_kernels(nullptr) == 0 && mi355ntt_ctx_psi_tables(nullptr) == nullptr);
    CHECK(mi355ntt_forward_batch(nullptr, nullptr, 1, 1, nullptr) == MI355NTT_EINVAL);
    CHECK(mi355ntt_polymul_batch(nullptr, nullptr, nullptr, 1, 1, nullptr) == MI355NTT_EINVAL);
    CHECK(mi355ntt_forward_raw(nullptr, 4096, nullptr, q60, 1, 60, nullptr) == MI355NTT_EINVAL);
    unsigned bits60 = 60;
    u64 mu60 = mi355ntt_barrett_mu(q60, 60);
    CHECK(mi355ntt_forward_batch_raw(nullptr, 32768, nullptr, 1, 1, qs, &mu60, &bits60, nullptr) == MI355NTT_EINVAL);
    CHECK(mi355ntt_barrett_raw(nullptr, nullptr, nullptr, 32768, 1, 1, qs, &mu60, &bits60, nullptr) == MI355NTT_EINVAL);
    CHECK(mi355ntt_poly_add_raw(nullptr, nullptr, 16, nullptr, 17) == MI355NTT_EINVAL);
    CHECK(mi355ntt_poly_negate_raw((u64*)4, 16, nullptr, 17) == MI355NTT_EINVAL);      // (not a word boundary; word-aligned pointers are accepted)
    CHECK(mi355ntt_ctx_kernel_class(nullptr) == MI355NTT_EINVAL);
    CHECK(mi355ntt_forward30_raw(nullptr, 2048, nullptr, 12931073, 21767333, 24, nullptr) == MI355NTT_EINVAL);
    CHECK(mi355ntt_raw_uses_fast_kernels(32768, nullptr, 0, 1, qs, &mu60, &bits60) == 0);
    CHECK(mi355ntt_raw_cache_clear() == MI355NTT_OK);
    mi355ntt_shards* sh = nullptr;
    CHECK(mi355ntt_shards_create(&sh, nullptr, 2, 0) == MI355NTT_EINVAL && sh == nullptr);
    CHECK(mi355ntt_shards_create(nullptr, nullptr, 0, 0) == MI355NTT_EINVAL && mi355ntt_shards_destroy(nullptr) == MI355NTT_OK);
    CHECK(mi355ntt_shards_world(nullptr) == 0 && mi355ntt_shards_transform(nullptr, 0, nullptr, nullptr, 1, 1, nullptr) == MI355NTT_EINVAL);
    mi355ntt_bfv* bfv = nullptr;
    CHECK(mi355ntt_bfv_create(nullptr, 4096, 2, qs, psis, 1024, 2305843009213683713ULL, 0, 0) != MI355NTT_OK);
    CHECK(mi355ntt_bfv_destroy(nullptr) == MI355NTT_OK && mi355ntt_bfv_ntt(nullptr) == nullptr);

    // ---- creation: succeeds with a GPU, fails with MI355NTT_EHIP (and unwinds) without one; a mixed context allocates more ----
    const u64 kat_q[3] = {68719403009ULL, 68719230977ULL, 137438822401ULL};
    int rc = mi355ntt_ctx_create(&ctx, 32768, 4, qs, psis, 0);
    if (rc == MI355NTT_OK) {
        CHECK(ctx != nullptr && mi355ntt_ctx_n(ctx) == 32768 && mi355ntt_ctx_num_primes(ctx) == 4 && mi355ntt_ctx_uses_literal_kernels(ctx) == 0);
        u64 q = 0;
        CHECK(mi355ntt_ctx_prime(ctx, 3, &q, nullptr, nullptr, nullptr, nullptr) == MI355NTT_OK && q == q60);
        CHECK(mi355ntt_ctx_prime(ctx, 4, &q, nullptr, nullptr, nullptr, nullptr) == MI355NTT_EINVAL);
        const mi355ntt_ctx* two[2] = {ctx, ctx};
        CHECK(mi355ntt_shards_create(&sh, two, 2, 8) == MI355NTT_OK && mi355ntt_shards_world(sh) == 2);
        CHECK(mi355ntt_shards_scatter_transform_gather(sh, MI355NTT_OP_POLYMUL, (u64*)16, 8, 4, 2, nullptr) == MI355NTT_EINVAL);
        CHECK(mi355ntt_shards_destroy(sh) == MI355NTT_OK);
        CHECK(mi355ntt_ctx_destroy(ctx) == MI355NTT_OK);
        std::printf("device present: contexts created and destroyed\n");
    } else {
        CHECK(rc == MI355NTT_EHIP && ctx == nullptr && mi355ntt_last_hip_error() != 0);
        std::printf("no device: creation failed with MI355NTT_EHIP (hipError_t %d) and unwound\n", mi355ntt_last_hip_error());
    }
    (void)kat_q;
    rc = mi355ntt_bfv_create(&bfv, 4096, 2, qs, psis, 1024, 2305843009213683713ULL, 0, 0);      // (psi of n = 32768 at n = 4096: EPARAM either way)
    CHECK(rc != MI355NTT_OK && bfv == nullptr);
    std::printf(failures ? "%d check(s) failed\n" : "host sanitize: all checks passed\n", failures);
    return failures ? 1 : 0;
}
