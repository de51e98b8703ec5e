// ntt_test_60bit.cpp -- the reference's NTT/polymul smoke test (BFV_Scheme/60bit_ntt_test.cu) rebuilt on the
// compat headers, with its CPU check enabled (`#define check 1`, 60bit_ntt_test.cu:14,65-66,85-98).
// Build (tests/test_cpp_compat.py does this): hipcc -std=c++17 tests/cpp/ntt_test_60bit.cpp -L ntt-cuda_amd -lmi355ntt
//                                             -L oracle -loracle -o ...
// The schoolbook reference product comes from the oracle (refPolyMul128, helper.h:95-126): checker only.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../ntt-cuda_amd/compat/ntt_60bit.hpp"
#include "../../ntt-cuda_amd/compat/poly_arithmetic.hpp"
#include "../../ntt-cuda_amd/compat/bfv_launch.hpp"   // compile check of the BFV launch layer
#include "../../ntt-cuda_amd/compat/ntt_30bit.hpp"    // ... and of the 30-bit launchers (overloads on unsigned*)
#include "../../oracle/ntt_oracle.h"

using namespace mi355;

#define HIPCK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); return 2; } } while (0)

int main(int argc, char** argv)
{
    unsigned N = argc > 1 ? (unsigned)atoi(argv[1]) : 1024 * 2;            // 60bit_ntt_test.cu:18
    size_t size_array = sizeof(unsigned long long) * N;
    unsigned long long q, psi, psiinv, ninv;
    unsigned q_bit;
    if (mi355ntt_get_params(N, &q, &psi, &psiinv, &ninv, &q_bit)) { printf("unsupported N\n"); return 2; }   // :26

    std::vector<unsigned long long> psiTable(N), psiinvTable(N);
    mi355ntt_fill_tables(psi, psiinv, q, N, psiTable.data(), psiinvTable.data());                             // :30
    unsigned long long *psi_powers, *psiinv_powers;
    HIPCK(hipMalloc(&psi_powers, size_array));
    HIPCK(hipMalloc(&psiinv_powers, size_array));
    HIPCK(hipMemcpy(psi_powers, psiTable.data(), size_array, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(psiinv_powers, psiinvTable.data(), size_array, hipMemcpyHostToDevice));

    unsigned bit_length = q_bit;
    unsigned long long mu = mi355ntt_barrett_mu(q, bit_length);                                               // :47-49

    std::vector<unsigned long long> a(N), b(N), refc(N);
    orc_splitmix_fill(a.data(), N, 21, q);       // the reference draws from an unseeded mt19937_64 (helper.h:72-83)
    orc_splitmix_fill(b.data(), N, 22, q);
    if (N <= 8192) {
        orc_ref_polymul(a.data(), b.data(), refc.data(), q, N);                                               // :65-66
    } else {
        // refPolyMul128 is O(n^2) (~9 min at n = 32768): above 8192 the expected product comes from the oracle's own
        // forward -> pointwise -> inverse (ntt_60bit.cuh:314-386, poly_arithmetic.cuh:9-34 restated; itself pinned on the
        // schoolbook product at the small sizes, tests/test_oracle_golden.py)
        std::vector<unsigned long long> fb(b);
        refc = a;
        orc_forward(refc.data(), N, q, mu, bit_length, psiTable.data());
        orc_forward(fb.data(), N, q, mu, bit_length, psiTable.data());
        orc_pointwise(refc.data(), fb.data(), N, q, mu, bit_length);
        orc_inverse(refc.data(), N, q, mu, bit_length, psiinvTable.data());
    }

    unsigned long long *d_a, *d_b;
    HIPCK(hipMalloc(&d_a, size_array));
    HIPCK(hipMalloc(&d_b, size_array));
    hipStream_t ntt1, ntt2;
    HIPCK(hipStreamCreate(&ntt1));
    HIPCK(hipStreamCreate(&ntt2));
    HIPCK(hipMemcpyAsync(d_a, a.data(), size_array, hipMemcpyHostToDevice, ntt1));
    HIPCK(hipMemcpyAsync(d_b, b.data(), size_array, hipMemcpyHostToDevice, ntt2));

    if (forwardNTTdouble(d_a, d_b, N, ntt1, ntt2, q, mu, bit_length, psi_powers)) return 3;                   // :75
    HIPCK(hipStreamSynchronize(ntt1));
    HIPCK(hipStreamSynchronize(ntt2));
    if (barrett(d_a, d_b, N, q, mu, bit_length)) return 3;                                                    // :76
    HIPCK(hipDeviceSynchronize());
    if (inverseNTT(d_a, N, ntt1, q, mu, bit_length, psiinv_powers)) return 3;                                 // :77
    HIPCK(hipStreamSynchronize(ntt1));
    HIPCK(hipMemcpy(a.data(), d_a, size_array, hipMemcpyDeviceToHost));

    int errors = 0;
    for (unsigned i = 0; i < N; i++)
        if (a[i] != refc[i]) {
            if (errors < 5) printf("error %u   %llu   %llu\n", i, a[i], refc[i]);
            errors++;
        }
    printf("n = %u q = %llu errors = %d\n", N, q, errors);
    return errors ? 1 : 0;
}
