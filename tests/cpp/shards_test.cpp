// shards_test.cpp -- the single-process multi-device driver (mi355ntt_shards_*) and the element-wise wrappers of
// poly_arithmetic.cuh:312-352 from compiled C++, as a program written against the reference's C++ surface would use them.
// One GPU here: `world` logical shards on device 0.  tests/test_cpp_compat.py builds and runs it.
//   usage: shards_test [world = 4] [num = 600]      (n = 4096, the reference's 58-bit getParams prime set x 1)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../ntt-cuda_amd/compat/ntt_60bit.hpp"
#include "../../ntt-cuda_amd/compat/poly_arithmetic.hpp"

using namespace mi355;
typedef unsigned long long u64;

#define HIPCK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); return 2; } } while (0)
#define RCCK(x) do { int rc_ = (x); if (rc_) { printf("mi355ntt error %d (%s) line %d\n", rc_, mi355ntt_strerror(rc_), __LINE__); return 2; } } while (0)

int main(int argc, char** argv)
{
    const unsigned world = argc > 1 ? (unsigned)atoi(argv[1]) : 4, num = argc > 2 ? (unsigned)atoi(argv[2]) : 600, n = 4096;
    u64 q, psi, psiinv, ninv;
    unsigned bits;
    RCCK(mi355ntt_get_params(n, &q, &psi, &psiinv, &ninv, &bits));
    std::vector<mi355ntt_ctx*> ctxs(world);
    for (unsigned r = 0; r < world; r++) RCCK(mi355ntt_ctx_create(&ctxs[r], n, 1, &q, &psi, 0));
    mi355ntt_shards* sh = nullptr;
    RCCK(mi355ntt_shards_create(&sh, ctxs.data(), world, 32));

    std::vector<u64> a((size_t)num * n), back((size_t)num * n), fwd((size_t)num * n);
    u64 x = 88172645463325252ULL;
    for (auto& v : a) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x % q; }
    u64 *d_full, *d_ref;
    HIPCK(hipMalloc(&d_full, a.size() * 8));
    HIPCK(hipMalloc(&d_ref, a.size() * 8));
    HIPCK(hipMemcpy(d_full, a.data(), a.size() * 8, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(d_ref, a.data(), a.size() * 8, hipMemcpyHostToDevice));
    hipStream_t s;
    HIPCK(hipStreamCreate(&s));

    // the whole batch in one call against the scatter -> transform -> gather of the same batch through `world` lanes
    RCCK(mi355ntt_forward_batch(ctxs[0], d_ref, num, 1, s));
    RCCK(mi355ntt_shards_scatter_transform_gather(sh, MI355NTT_OP_FORWARD, d_full, num, 1, 3, s));
    HIPCK(hipMemcpyAsync(fwd.data(), d_ref, a.size() * 8, hipMemcpyDeviceToHost, s));
    HIPCK(hipMemcpyAsync(back.data(), d_full, a.size() * 8, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    size_t errors = 0;
    for (size_t i = 0; i < a.size(); i++) errors += fwd[i] != back[i];
    // device-resident shards: pointers into the batch at the ranges mi355ntt_shard_range names
    std::vector<u64*> parts(world);
    unsigned covered = 0;
    for (unsigned r = 0; r < world; r++) {
        unsigned first, count;
        RCCK(mi355ntt_shard_range(num, 1, r, world, &first, &count));
        parts[r] = count ? d_full + (size_t)first * n : nullptr;
        errors += first != covered;
        covered += count;
    }
    errors += covered != num;
    RCCK(mi355ntt_shards_transform(sh, MI355NTT_OP_INVERSE, parts.data(), nullptr, num, 1, s));
    HIPCK(hipMemcpyAsync(back.data(), d_full, a.size() * 8, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    for (size_t i = 0; i < a.size(); i++) errors += a[i] != back[i];

    // element-wise wrappers, reference names and argument order: (a + b) then the reference's poly_sub (adds q where a < b), negate, + 5
    u64* d_b = d_ref;                     // (holds NTT values: any residues below q do)
    RCCK(poly_add_device(d_full, d_b, n, s, q));
    RCCK(poly_sub_device(d_full, d_b, n, s, q));
    RCCK(poly_negate_device(d_full, n, s, q));
    RCCK(poly_add_integer_device(d_full, 5, n, s, q));
    RCCK(poly_mul_int_t(d_full, 3, n, s, 1024));
    HIPCK(hipMemcpyAsync(back.data(), d_full, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    for (unsigned i = 0; i < n; i++) {
        u64 v = a[i] + fwd[i];
        if (v > q) v -= q;                                   // poly_add        (poly_arithmetic.cuh:144-154)
        if (v < fwd[i]) v += q;                              // poly_sub        (:168-179: never subtracts)
        v = q - v;
        if (v == q) v = 0;                                   // poly_negate     (:334-338)
        v += 5;
        if (v > q) v -= q;                                   // poly_add_integer (:156-166)
        v = (v * 3) & (u64)(unsigned)(1024 - 1);             // mod_t           (:128-142)
        errors += v != back[i];
    }
    RCCK(mi355ntt_shards_destroy(sh));
    for (auto c : ctxs) RCCK(mi355ntt_ctx_destroy(c));
    printf("shards = %u, polynomials = %u, errors = %zu\n", world, num, errors);
    return errors ? 1 : 0;
}
