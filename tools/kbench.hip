// kbench.hip -- standalone timing of the n = 2^15 single-pass kernels, launched directly (no library, no context): the lazy class
// <4, near-2^k> the headline runs (k_forward15 / k_inverse15) and the literal class 0 (k_forward15_lit / k_inverse15_lit).
//   tools/build_kbench.sh <tag> [-DMI355NTT_TUNE_HEADER='"my_tune.hpp"']      -> tools/kbench_<tag>
//   tools/kbench_<tag> [polynomials = 1024] [samples = 20] [kind: 0 lazy, 1 literal] [warm launches = 3]
// KB_B2B=L: each sample times L back-to-back launches; KB_PAIR=1: also forward -> inverse pairs over the same buffer.
// The kernels carry no experiment switches any more (round 6): a variant build substitutes csrc/tune.hpp's struct through
// MI355NTT_TUNE_HEADER; the ablation / stamp builds of rounds 1-5 are reproducible from commit 3b06c3b.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels_fast_impl.cuh"

using namespace mi355ntt;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const int LOGN = 15;
    const unsigned n = 1u << LOGN;
    const unsigned num = argc > 1 ? atoi(argv[1]) : 1024;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int kind = argc > 3 ? atoi(argv[3]) : 0;
    const int warm = argc > 4 ? atoi(argv[4]) : 3;
    // lazy: BASELINE's first 60-bit prime; literal: a Barrett-inexact 60-bit prime (tests/params.py INEXACT_PRIMES[60])
    const u64 q = kind ? 1137833256315125761ULL : 1152921504606584833ULL, psi = kind ? 448230823712243253ULL : 4443670208963ULL;
    PrimeParams pp;
    if (derive_prime(n, q, psi, &pp)) { printf("bad prime\n"); return 1; }
    std::vector<u64> tab(n);
    fill_table(psi, q, n, tab.data());
    std::vector<TwPair> tw(n);
    for (unsigned i = 0; i < n; i++) tw[i] = TwPair{tab[i], shoup(tab[i], q)};   // (reference order: the layout is irrelevant for timing)
    PrimeDev d{};
    d.q = q; d.nq = 0ULL - q;
    d.mu = pp.mu; d.k = pp.k; d.red_sh1 = pp.k - 17; d.red_sh2 = 16; d.red_c = (u32)((((u128)1) << (31 + pp.k)) / q);
    for (int j = 0; j < 32; j++) d.twn[j] = TwPair{tab[j] % q, shoup(tab[j] % q, q)};
    d.delta = kind ? 0u : (u32)((1ull << pp.k) - q); d.near_sh = pp.k - 32; d.near_mask = (u32)((1ull << (pp.k - 32)) - 1);
    d.lit = kind ? 1u : 0u;
    u64* a; TwPair* dtw; PrimeDev* dp;
    CK(hipMalloc(&a, (size_t)num * n * 8));
    CK(hipMalloc(&dtw, n * sizeof(TwPair)));
    CK(hipMalloc(&dp, 2 * sizeof(PrimeDev)));      // [0]: the guard / clock record in front of the array
    CK(hipMemset(dp, 0, sizeof(PrimeDev)));
    dp += 1;
    std::vector<u64> h((size_t)num * n);
    u64 x = 88172645463325252ULL;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x % q; }
    CK(hipMemcpy(a, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtw, tw.data(), n * sizeof(TwPair), hipMemcpyHostToDevice));
    CK(hipMemcpy(dp, &d, sizeof(d), hipMemcpyHostToDevice));
    // KB_GRID: another persistent grid than the library's min(num, CUs) (e.g. a balanced one: every workgroup the same number of polynomials)
    const dim3 g(getenv("KB_GRID") ? (unsigned)atoi(getenv("KB_GRID")) : persistent_grid<LOGN>(num)), b(1024);
    auto fwd = [&]() {
        if (kind) k_forward15_lit<LOGN><<<g, b>>>(a, dtw, dp, 1, 0, num);
        else k_forward15<4, true><<<g, b>>>(a, dtw, dp, 1, 0, num);
    };
    auto inv = [&]() {
        if (kind) k_inverse15_lit<LOGN><<<g, b>>>(a, dtw, dp, 1, 0, num);
        else k_inverse15<4, true><<<g, b>>>(a, dtw, dp, inv15_division_word(1, num), 0, num);
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nwhich = getenv("KB_PAIR") ? 3 : 2;
    const int b2b = getenv("KB_B2B") ? atoi(getenv("KB_B2B")) : 1;
    const char* names[] = {"forward", "inverse", "forward+inverse"};
    for (int which = 0; which < nwhich; which++) {
        for (int i = 0; i < warm; i++) {
            if (which != 1) fwd();
            if (which != 0) inv();
        }
        CK(hipDeviceSynchronize());
        std::vector<float> ts;
        for (int i = 0; i < reps; i++) {
            CK(hipEventRecord(e0));
            for (int l = 0; l < b2b; l++) {
                if (which != 1) fwd();
                if (which != 0) inv();
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / b2b);
        }
        std::sort(ts.begin(), ts.end());
        const double med = ts[ts.size() / 2];
        printf("%-16s %s  num %5u  median %.4f ms  min %.4f ms  -> %.3f M transforms/s, %.2f TB/s algorithmic\n", names[which],
               kind ? "literal (class 0)" : "lazy <4, near>", num, med, ts[0], (which == 2 ? 2.0 : 1.0) * num / (med * 1e-3) / 1e6,
               (which == 2 ? 2.0 : 1.0) * num * 524288.0 / (med * 1e-3) / 1e12);
    }
    CK(hipGetLastError());
    return 0;
}
