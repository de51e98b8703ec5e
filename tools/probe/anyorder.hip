// probe: does a kernel launched with hipExtAnyOrderLaunch (AQL barrier bit clear) start on CUs as the workgroups of the
// previous kernel of the SAME stream exit?  Are the workgroups of consecutive any-order kernels placed in launch order?
// And what does the same look like with the second kernel on another stream?  (gfx950, one workgroup per CU.)
//   hipcc --offload-arch=gfx950 -O2 tools/probe/anyorder.hip -o tools/probe/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long realtime()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

// one workgroup per CU (144 KiB of LDS, 1024 threads); busy for base + jitter(blockIdx) ticks of the 100 MHz clock
__global__ void __launch_bounds__(1024) k_busy(unsigned long long* log, unsigned base_ticks, unsigned jitter_ticks)
{
    __shared__ unsigned long long pad[18432];
    const unsigned long long t0 = realtime();
    pad[threadIdx.x] = t0;
    const unsigned long long until = t0 + base_ticks + (blockIdx.x * 37u % 64u) * jitter_ticks / 64u;
    while (realtime() < until) __builtin_amdgcn_s_sleep(16);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hwid));
        log[blockIdx.x * 4 + 0] = t0;
        log[blockIdx.x * 4 + 1] = realtime();
        log[blockIdx.x * 4 + 2] = hwid;
        log[blockIdx.x * 4 + 3] = pad[5];
    }
}

struct Span { double e0, e1, x0, x1; };
static Span span(const std::vector<unsigned long long>& l, unsigned grid, unsigned long long t0)
{
    Span s{1e30, -1e30, 1e30, -1e30};
    for (unsigned b = 0; b < grid; b++) {
        const double e = (double)(l[b * 4] - t0) * 0.01, x = (double)(l[b * 4 + 1] - t0) * 0.01;
        s.e0 = std::min(s.e0, e); s.e1 = std::max(s.e1, e); s.x0 = std::min(s.x0, x); s.x1 = std::max(s.x1, x);
    }
    return s;
}

int main(int argc, char** argv)
{
    const unsigned grid = argc > 1 ? atoi(argv[1]) : 256, NK = 3;
    unsigned long long* log[NK];
    for (unsigned k = 0; k < NK; k++) CK(hipMalloc(&log[k], grid * 32));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    const unsigned base = 10000, jitter = 3000;      // 100 us + up to 30 us
    // mode 0: plain launches on one stream; 1: kernels 2, 3 any-order on the same stream; 2: kernel 2 on another stream (plain), kernel 3 back on
    // the first; 3: as 1 with a full grid of twice the CUs (second wave of workgroups)
    for (int mode = 0; mode < 4; mode++) {
        const unsigned g = mode == 3 ? 2 * grid : grid;
        if (mode == 3) for (unsigned k = 0; k < NK; k++) { CK(hipFree(log[k])); CK(hipMalloc(&log[k], g * 32)); }
        for (int rep = 0; rep < 3; rep++) {
            for (unsigned k = 0; k < NK; k++) CK(hipMemset(log[k], 0, g * 32));
            CK(hipDeviceSynchronize());
            for (unsigned k = 0; k < NK; k++) {
                hipStream_t s = (mode == 2 && k == 1) ? s2 : s1;
                const int flags = ((mode == 1 || mode == 3) && k > 0) ? hipExtAnyOrderLaunch : 0;
                hipExtLaunchKernelGGL(k_busy, dim3(g), dim3(1024), 0, s, nullptr, nullptr, flags, log[k], base, jitter);
                CK(hipGetLastError());
            }
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h[NK];
            unsigned long long t0 = ~0ull;
            for (unsigned k = 0; k < NK; k++) {
                h[k].resize(g * 4);
                CK(hipMemcpy(h[k].data(), log[k], g * 32, hipMemcpyDeviceToHost));
                for (unsigned b = 0; b < g; b++) t0 = std::min(t0, h[k][b * 4]);
            }
            printf("mode %d rep %d (grid %u):\n", mode, rep, g);
            for (unsigned k = 0; k < NK; k++) {
                const Span s = span(h[k], g, t0);
                printf("   kernel %u: entries %8.2f .. %8.2f us   exits %8.2f .. %8.2f us\n", k, s.e0, s.e1, s.x0, s.x1);
            }
            // placement order: how many workgroups of kernel k + 1 entered before the LAST entry of kernel k?
            for (unsigned k = 0; k + 1 < NK; k++) {
                unsigned long long last = 0;
                for (unsigned b = 0; b < g; b++) last = std::max(last, h[k][b * 4]);
                unsigned early = 0;
                for (unsigned b = 0; b < g; b++) early += h[k + 1][b * 4] < last;
                printf("   workgroups of kernel %u entered before the last entry of kernel %u: %u\n", k + 1, k, early);
            }
        }
    }
    return 0;
}
