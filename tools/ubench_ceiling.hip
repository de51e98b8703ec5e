// ubench_ceiling.hip -- steady-state VALU ceiling of the lazy Shoup butterfly of ntt_core.cuh on gfx950.
//
// Every wave runs the butterfly stream until a common deadline (s_memrealtime) and reports how many butterflies it
// completed: all waves are busy for the whole window, so the sum / window is the throughput under full contention with
// the SIMD's own (oldest-first) arbitration -- no start skew, no tail of late waves as in a fixed-work kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ntt-cuda_amd/csrc tools/ubench_ceiling.hip -o tools/ubench_ceiling
//   ./tools/ubench_ceiling [window_us = 3000]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ntt_core.cuh"

using namespace mi355ntt;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int CH = 16;   // butterflies in flight per thread (32 coefficients in registers, as the kernels)

struct Out { unsigned long long bfly, t_first, t_last, clk; };

// VARIANT 0: ct butterfly exactly as ct_round's NEAR form (mul_shoup4 + add + add/sub), twiddles in VGPRs
// VARIANT 1: the same with twiddles in SGPRs (round 1 of the forward kernel)
// VARIANT 2: gs butterfly (add, add/sub, mul_shoup4)
// VARIANT 3: ct_bfly4 (running value folded into the multiply-add accumulator)
template <int VARIANT, int PRIO_BY_SLOT>
__global__ void __launch_bounds__(1024) k_ceiling(Out* out, unsigned long long window_ticks, u64 q, u64 wseed)
{
    u64 x[CH], y[CH], w[4], wp[4];   // 64 data + 16 twiddle VGPRs (the kernels keep a ring of 2 x 4 twiddle pairs)
    const u64 nq = 0 - q, cq = 4 * q;
#pragma unroll
    for (int u = 0; u < CH; u++) {
        x[u] = (threadIdx.x * 1315423911ULL + u * 7919ULL) % q;
        y[u] = (x[u] * 2654435761ULL + wseed) % q;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        w[u] = VARIANT == 1 ? wseed + 12345 + u : (wseed * (u + 3) + threadIdx.x) % q;
        wp[u] = VARIANT == 1 ? wseed * 977 + u : w[u] * 31 + 7;
    }
    if (PRIO_BY_SLOT == 1) {      // static distinct priorities: youngest highest
        const unsigned slot = (threadIdx.x >> 8) & 3u;
        if (slot == 0) __builtin_amdgcn_s_setprio(0); else if (slot == 1) __builtin_amdgcn_s_setprio(1);
        else if (slot == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long t_end = t0 + window_ticks;
    unsigned long long n = 0, tl = t0;
    for (;;) {
#pragma unroll 1
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int u = 0; u < CH; u++) {
                if constexpr (VARIANT == 0 || VARIANT == 1) {
                    const u64 U = x[u];
                    const u64 T = mul_shoup4(y[u], w[u & 3], wp[u & 3], nq);
                    x[u] = U + T;
                    y[u] = U + cq - T;
                } else if constexpr (VARIANT == 2) {
                    const u64 X = x[u], Y = y[u];
                    x[u] = X + Y;
                    y[u] = mul_shoup4(X + cq - Y, w[u & 3], wp[u & 3], nq);
                } else if constexpr (VARIANT == 3) {
                    ct_bfly4(x[u], y[u], w[u & 3], wp[u & 3], nq, cq);
                } else if constexpr (VARIANT == 4) {          // ct: mad-chain cross terms, separate adds
                    const u64 U = x[u];
                    const u64 T = mul_shoup4m<false>(y[u], w[u & 3], wp[u & 3], nq);
                    x[u] = U + T;
                    y[u] = U + cq - T;
                } else if constexpr (VARIANT == 5) {          // ct: mad chain + U folded into the accumulator
                    const u64 U = x[u];
                    u64 D = (U << 1) + cq;
                    asm("" : "+v"(D));
                    const u64 A = mul_shoup4m_acc<false>(y[u], w[u & 3], wp[u & 3], nq, U);
                    x[u] = A;
                    y[u] = D - A;
                } else if constexpr (VARIANT == 6) {          // gs: mad chain
                    const u64 X = x[u], Y = y[u];
                    x[u] = X + Y;
                    y[u] = mul_shoup4m<false>(X + cq - Y, w[u & 3], wp[u & 3], nq);
                } else if constexpr (VARIANT == 7) {          // canonicalisation as the kernels do it today: 3-instruction fold + compare/select
                    PrimeDev pd{}; pd.q = q; pd.nq = nq; pd.delta = (u32)((1ull << 60) - q); pd.near_sh = 28; pd.near_mask = (1u << 28) - 1;
                    x[u] = canon_2q(reduce_2q_near(x[u] + wseed, pd), q);
                    y[u] = canon_2q(reduce_2q_near(y[u] + wseed, pd), q);
                } else if constexpr (VARIANT == 8) {          // the same with an arithmetic mask instead of compare/select
                    PrimeDev pd{}; pd.q = q; pd.nq = nq; pd.delta = (u32)((1ull << 60) - q); pd.near_sh = 28; pd.near_mask = (1u << 28) - 1;
                    auto canon_mask = [&](u64 v) {
                        const u64 r = reduce_2q_near(v + wseed, pd) - q;
                        u32 m;
                        asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m) : "v"(hi32(r)));
                        return r + (q & (((u64)m << 32) | m));
                    };
                    x[u] = canon_mask(x[u]);
                    y[u] = canon_mask(y[u]);
                }
                if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        n += 8 * CH;
        tl = __builtin_amdgcn_s_memrealtime();
        if (tl >= t_end) break;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    u64 s = 0;
#pragma unroll
    for (int u = 0; u < CH; u++) s ^= x[u] ^ y[u];
    if (s == 0x1234567) out[0].bfly = s;
    if ((threadIdx.x & 63) == 0) {
        const unsigned wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        out[1 + wv] = Out{n, t0, tl, c1 - c0};
    }
}

static int g_reps = 2;      // launches per measurement (the last one is reported); "sustain" mode: many long windows back to back
template <int VARIANT, int PRIO>
static void run(const char* name, int waves_per_simd, unsigned long long window_us)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount, block = 256 * waves_per_simd;
    const size_t nw = (size_t)grid * block / 64;
    Out* d;
    CK(hipMalloc(&d, (nw + 1) * sizeof(Out)));
    for (int rep = 0; rep < g_reps; rep++) {
        hipLaunchKernelGGL((k_ceiling<VARIANT, PRIO>), dim3(grid), dim3(block), 0, 0, d, window_us * 100ull, 1152921504606584833ULL, 4443670208963ULL);
        CK(hipDeviceSynchronize());
    }
    std::vector<Out> h(nw + 1);
    CK(hipMemcpy(h.data(), d, (nw + 1) * sizeof(Out), hipMemcpyDeviceToHost));
    double total = 0, clk = 0;
    unsigned long long tmin = ~0ull, tmax = 0, smax = 0;
    double per_slot[4] = {0, 0, 0, 0};
    for (size_t i = 1; i <= nw; i++) {
        total += (double)h[i].bfly * 64.0;
        tmin = std::min(tmin, h[i].t_first); smax = std::max(smax, h[i].t_first); tmax = std::max(tmax, h[i].t_last);
        clk += (double)h[i].clk / (double)(h[i].t_last - h[i].t_first);
        per_slot[((i - 1) % (block / 64)) / 4 % 4] += (double)h[i].bfly;
    }
    const double win = (double)(tmax - tmin) * 1e-8;               // seconds
    const double rate = total / win;                               // lane-butterflies per second
    const double ghz = clk / nw * 0.1;
    const double cyc = ghz * 1e9 / (rate / 64.0 / (grid * 4.0));   // shader cycles per wave-butterfly per SIMD
    printf("%-34s waves/SIMD=%d  %.3e bfly/s  => %6.2f M NTT(2^15)/s  cyc/wave-bfly/SIMD=%6.2f  clk %.2f GHz  start spread %.1f us  share by age:",
           name, waves_per_simd, rate, rate / 245760.0 / 1e6, cyc, ghz, (double)(smax - tmin) * 0.01);
    double ps = per_slot[0] + per_slot[1] + per_slot[2] + per_slot[3];
    for (int k = 0; k < waves_per_simd && k < 4; k++) printf(" %.2f", per_slot[k] / ps);
    printf("\n");
    CK(hipFree(d));
}

int main(int argc, char** argv)
{
    unsigned long long win = argc > 1 ? strtoull(argv[1], nullptr, 10) : 3000;
    if (argc > 2) {
        // sustained mode: `reps` windows back to back (e.g. 20000 us x 100 = 2 s of full VALU issue), so that the figure is the one
        // the chip holds under its power cap, not the first milliseconds at the boost clock; each line reports the LAST window
        g_reps = atoi(argv[2]);
        for (int i = 0; i < 3; i++) {
            run<5, 0>("sustained: ct bfly, mad chain + fold U", 4, win);
            run<6, 0>("sustained: gs bfly, mad chain", 4, win);
        }
        return 0;
    }
    for (int w = 1; w <= 4; w++) run<0, 0>("ct bfly, VGPR twiddles", w, win);
    for (int w = 1; w <= 4; w++) run<1, 0>("ct bfly, SGPR twiddles", w, win);
    for (int w = 1; w <= 4; w++) run<2, 0>("gs bfly, VGPR twiddles", w, win);
    for (int w = 1; w <= 4; w++) run<3, 0>("ct_bfly4 (fold U)", w, win);
    for (int w = 2; w <= 4; w += 2) run<4, 0>("ct bfly, mad-chain cross terms", w, win);
    for (int w = 2; w <= 4; w += 2) run<5, 0>("ct bfly, mad chain + fold U", w, win);
    for (int w = 2; w <= 4; w += 2) run<6, 0>("gs bfly, mad chain", w, win);
    for (int w = 2; w <= 4; w += 2) run<7, 0>("2 x (fold + canon), cmp/cndmask", w, win);
    for (int w = 2; w <= 4; w += 2) run<8, 0>("2 x (fold + canon), arithmetic mask", w, win);
    return 0;
}
