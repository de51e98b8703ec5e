// Stand-alone timing of the 30-bit native kernels (k_ntt30x) with compile-time ablations; timing only (tables are random
// words below q).  Build: tools/build_kbench30.sh <tag> [-DNTT30_...]; run: tools/kbench30_<tag> [logn] [num] [reps]
#include "../ntt-cuda_amd/csrc/kernels_ntt30.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv)
{
    const int logn = argc > 1 ? atoi(argv[1]) : 15;
    const unsigned n = 1u << logn;
    const unsigned num = argc > 2 ? atoi(argv[2]) : (1u << 27) / n;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    const unsigned q = 19070977, bits = 25, mu = (unsigned)((1ull << 50) / q);
    std::vector<unsigned> h((size_t)num * n), tab(n);
    unsigned long long x = 88172645463325252ULL;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (unsigned)(x % q); }
    for (auto& v : tab) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (unsigned)(x % q); }
    unsigned *a, *dt;
    CK(hipMalloc(&a, h.size() * 4)); CK(hipMalloc(&dt, n * 4));
    CK(hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, tab.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; which++) {
        for (int i = 0; i < 10; i++) CK(which ? mi355ntt::ntt30_inverse(a, n, dt, num, q, mu, bits, 12345, 0) : mi355ntt::ntt30_forward(a, n, dt, num, q, mu, bits, 12345, 0));
        CK(hipDeviceSynchronize());
        std::vector<float> ts;
        for (int i = 0; i < reps; i++) {
            CK(hipEventRecord(e0));
            for (int l = 0; l < 4; l++) CK(which ? mi355ntt::ntt30_inverse(a, n, dt, num, q, mu, bits, 12345, 0) : mi355ntt::ntt30_forward(a, n, dt, num, q, mu, bits, 12345, 0));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / 4);
        }
        std::sort(ts.begin(), ts.end());
        const double gb = (double)num * n * 8 / 1e9;
        printf("%s n=2^%d num=%u  median %.4f ms  => %.0f GB/s algorithmic, %.2f M transforms/s\n", which ? "inverse30" : "forward30", logn, num,
               ts[ts.size() / 2], gb / ts[ts.size() / 2] * 1e3, num / ts[ts.size() / 2] / 1e3);
    }
    return 0;
}
