// ubench_bfly.hip -- steady-state throughput of 64-bit modular butterfly variants on gfx950.
// Each kernel runs long enough (ms) for clocks to settle; reports wave-cycles per butterfly per SIMD
// and butterflies/s chip-wide.  Build:
//   hipcc --offload-arch=gfx950 -O3 -I ntt-cuda_amd/csrc tools/ubench_bfly.hip -o tools/ubench_bfly
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "modarith.cuh"

using namespace mi355ntt;

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

constexpr int CH = 16;  // independent butterflies in flight per thread (like 32 coefficients in registers)

struct Consts {
    u64 q, nq, twoq, fourq, mu, w, wp;
};

template <int V>
__device__ __forceinline__ void bfly(u64& x, u64& y, const Consts& c);

// 0: compiler Shoup (exact mul_hi), fully lazy (no conditional subtraction)
template <>
__device__ __forceinline__ void bfly<0>(u64& x, u64& y, const Consts& c)
{
    u64 T = shoup_mul_lazy(y, c.w, c.wp, c.q);
    u64 U = x;
    x = U + T;
    y = U - T + c.twoq;
}

// 1: approximate quotient (two mul_hi + one mad), r = y*w + h*(-q); T in [0,4q)
template <>
__device__ __forceinline__ void bfly<1>(u64& x, u64& y, const Consts& c)
{
    u32 y0 = lo32(y), y1 = hi32(y), p0 = lo32(c.wp), p1 = hi32(c.wp);
    u64 h = mad32(y1, p1, (u64)__umulhi(y0, p1)) + (u64)__umulhi(y1, p0);
    u64 T = y * c.w + h * c.nq;
    u64 U = x;
    x = U + T;
    y = U + c.fourq - T;
}

// 2: as 1 but the low product built from explicit 32-bit pieces with mad chains on the high word
template <>
__device__ __forceinline__ void bfly<2>(u64& x, u64& y, const Consts& c)
{
    u32 y0 = lo32(y), y1 = hi32(y), p0 = lo32(c.wp), p1 = hi32(c.wp);
    u32 w0 = lo32(c.w), w1 = hi32(c.w), n0 = lo32(c.nq), n1 = hi32(c.nq);
    u64 h = mad32(y1, p1, (u64)__umulhi(y0, p1)) + (u64)__umulhi(y1, p0);
    u32 h0 = lo32(h), h1 = hi32(h);
    u64 lo = mad32(h0, n0, mad32(y0, w0, 0));
    u32 hi = hi32(lo) + y0 * w1 + y1 * w0 + h0 * n1 + h1 * n0;
    u64 T = ((u64)hi << 32) | lo32(lo);
    u64 U = x;
    x = U + T;
    y = U + c.fourq - T;
}

// 3: exact Shoup + Harvey conditional subtraction on U (the classical lazy butterfly, q < 2^62)
template <>
__device__ __forceinline__ void bfly<3>(u64& x, u64& y, const Consts& c)
{
    u64 T = shoup_mul_lazy(y, c.w, c.wp, c.q);
    u64 U = x >= c.twoq ? x - c.twoq : x;
    x = U + T;
    y = U - T + c.twoq;
}

// 4: reference-literal (Barrett Algorithm 7 + canonical add/sub)
template <>
__device__ __forceinline__ void bfly<4>(u64& x, u64& y, const Consts& c)
{
    u64 V = barrett_mul(y, c.w, c.q, c.mu, 60);
    u64 U = x;
    x = add_mod(U, V, c.q);
    y = sub_mod(U, V, c.q);
}

// 5: inline-asm minimal sequence: 2 mul_hi + mad + add64 (quotient), 2 mad + 4 mad-on-high (remainder),
//    add64, add64 + sub pair (butterfly).  zero = a VGPR pair partner that stays 0.
template <>
__device__ __forceinline__ void bfly<5>(u64& x, u64& y, const Consts& c)
{
    u32 y0 = lo32(y), y1 = hi32(y), p0 = lo32(c.wp), p1 = hi32(c.wp);
    u32 w0 = lo32(c.w), w1 = hi32(c.w), n0 = lo32(c.nq), n1 = hi32(c.nq);
    u64 a = __umulhi(y0, p1);   // {A, 0}
    u64 b = __umulhi(y1, p0);   // {B, 0}
    u64 h, lo, hiw, T;
    asm volatile(
        "v_mad_u64_u32 %0, vcc, %4, %5, %2\n\t"      // h = y1*p1 + A
        "v_lshl_add_u64 %0, %0, 0, %3\n\t"           // h += B
        "v_mad_u64_u32 %1, vcc, %6, %7, 0\n\t"       // lo = y0*w0
        : "=&v"(h), "=&v"(lo)
        : "v"(a), "v"(b), "v"(y1), "v"(p1), "v"(y0), "v"(w0)
        : "vcc");
    u32 h0 = lo32(h), h1 = hi32(h);
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(lo) : "v"(h0), "v"(n0) : "vcc");   // lo += h0*n0
    hiw = hi32(lo);   // {hi, junk}: only the low dword of the chain below matters
    asm volatile(
        "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
        : "+v"(hiw)
        : "v"(y0), "v"(w1), "v"(y1), "v"(w0), "v"(h0), "v"(n1), "v"(h1), "v"(n0)
        : "vcc");
    T = ((u64)lo32(hiw) << 32) | lo32(lo);
    u64 U = x;
    x = U + T;
    y = U + c.fourq - T;
}

// 6: modmul only, variant 1 (to separate the multiply from the add/sub cost)
template <>
__device__ __forceinline__ void bfly<6>(u64& x, u64& y, const Consts& c)
{
    u32 y0 = lo32(y), y1 = hi32(y), p0 = lo32(c.wp), p1 = hi32(c.wp);
    u64 h = mad32(y1, p1, (u64)__umulhi(y0, p1)) + (u64)__umulhi(y1, p0);
    y = y * c.w + h * c.nq + (x & 1);
}

// 7: GS (inverse) butterfly, Harvey form: x' = x + y (one conditional subtraction), y' = shoup(x - y + 2q)
template <>
__device__ __forceinline__ void bfly<7>(u64& x, u64& y, const Consts& c)
{
    u64 s = x + y;
    u64 d = x - y + c.twoq;
    x = s >= c.twoq ? s - c.twoq : s;
    y = shoup_mul_lazy(d, c.w, c.wp, c.q);
}

// 8: GS butterfly, approximate quotient, no conditional subtraction (bounds tracked by the caller)
template <>
__device__ __forceinline__ void bfly<8>(u64& x, u64& y, const Consts& c)
{
    u64 s = x + y;
    u64 d = x - y + c.fourq;
    u32 y0 = lo32(d), y1 = hi32(d), p0 = lo32(c.wp), p1 = hi32(c.wp);
    u64 h = mad32(y1, p1, (u64)__umulhi(y0, p1)) + (u64)__umulhi(y1, p0);
    x = s;
    y = d * c.w + h * c.nq;
}

// opaque pass-through: the compiler must keep all 64 bits of a mad result it would otherwise narrow to mul_lo + add
#define KEEP64(v) asm("" : "+v"(v))

// 9: CT butterfly, every multiply a v_mad_u64_u32: U rides in the accumulator of the lo*lo products, the four cross
//    products accumulate in the LOW word of a second chain (its high word is junk), one v_add_u32 joins them
template <>
__device__ __forceinline__ void bfly<9>(u64& x, u64& y, const Consts& c)
{
    u32 y0 = lo32(y), y1 = hi32(y), p0 = lo32(c.wp), p1 = hi32(c.wp);
    u32 w0 = lo32(c.w), w1 = hi32(c.w), n0 = lo32(c.nq), n1 = hi32(c.nq);
    u64 U = x;
    u64 h = mad32(y1, p1, (u64)__umulhi(y0, p1)) + (u64)__umulhi(y1, p0);
    u32 h0 = lo32(h), h1 = hi32(h);
    u64 cr = mad32(y0, w1, 0);
    KEEP64(cr);
    cr = mad32(y1, w0, cr);
    KEEP64(cr);
    u64 acc = mad32(y0, w0, U);
    cr = mad32(h0, n1, cr);
    KEEP64(cr);
    acc = mad32(h0, n0, acc);
    cr = mad32(h1, n0, cr);
    KEEP64(cr);
    u32 xh = hi32(acc) + lo32(cr);
    asm("" : "+v"(xh));
    u64 X = ((u64)xh << 32) | lo32(acc);   // U + T
    x = X;
    y = (U << 1) + c.fourq - X;                                  // U + 4q - T
}

// 10: GS butterfly with the same multiply structure
template <>
__device__ __forceinline__ void bfly<10>(u64& x, u64& y, const Consts& c)
{
    u64 s = x + y;
    u64 d = x - y + c.fourq;
    u32 y0 = lo32(d), y1 = hi32(d), p0 = lo32(c.wp), p1 = hi32(c.wp);
    u32 w0 = lo32(c.w), w1 = hi32(c.w), n0 = lo32(c.nq), n1 = hi32(c.nq);
    u64 h = mad32(y1, p1, (u64)__umulhi(y0, p1)) + (u64)__umulhi(y1, p0);
    u32 h0 = lo32(h), h1 = hi32(h);
    u64 cr = mad32(y0, w1, 0);
    KEEP64(cr);
    cr = mad32(y1, w0, cr);
    KEEP64(cr);
    u64 acc = mad32(y0, w0, 0);
    cr = mad32(h0, n1, cr);
    KEEP64(cr);
    acc = mad32(h0, n0, acc);
    cr = mad32(h1, n0, cr);
    KEEP64(cr);
    x = s;
    u32 yh = hi32(acc) + lo32(cr);
    asm("" : "+v"(yh));
    y = ((u64)yh << 32) | lo32(acc);
}

// 11: variant 2 with U riding in the accumulator of the lo*lo products: x' = U + T needs no add of its own,
//     y' = 2U + 4q - x'
template <>
__device__ __forceinline__ void bfly<11>(u64& x, u64& y, const Consts& c)
{
    u32 y0 = lo32(y), y1 = hi32(y), p0 = lo32(c.wp), p1 = hi32(c.wp);
    u32 w0 = lo32(c.w), w1 = hi32(c.w), n0 = lo32(c.nq), n1 = hi32(c.nq);
    u64 U = x;
    u64 D = (U << 1) + c.fourq;
    asm("" : "+v"(D));
    u64 h = mad32(y1, p1, (u64)__umulhi(y0, p1)) + (u64)__umulhi(y1, p0);
    u32 h0 = lo32(h), h1 = hi32(h);
    u64 acc = mad32(h0, n0, mad32(y0, w0, U));
    u32 xh = hi32(acc) + y0 * w1 + y1 * w0 + h0 * n1 + h1 * n0;
    asm("" : "+v"(xh));
    u64 X = ((u64)xh << 32) | lo32(acc);
    x = X;
    y = D - X;
}

// 12/13: variant 1 on 8 chains with the twiddle pair in SGPRs (as round 1 of the kernels has it) vs in per-chain VGPRs
// (rounds 2 and 3): same instruction count, different operand sources
template <bool VTW>
__global__ void __launch_bounds__(1024) k_bfly_tw(unsigned long long* out, const Consts* cp, int iters)
{
    __shared__ unsigned lds_pin[24576];
    if (iters < 0) lds_pin[threadIdx.x] = iters;
    Consts c = *cp;
    constexpr int C8 = 8;
    u64 x[C8], y[C8], w[C8], wp[C8];
    for (int u = 0; u < C8; u++) {
        x[u] = (threadIdx.x * 1315423911ULL + u * 977ULL) & ((1ULL << 59) - 1);
        y[u] = (x[u] * 2654435761ULL + blockIdx.x) & ((1ULL << 59) - 1);
        w[u] = c.w + (VTW ? (u64)threadIdx.x * 7919ULL + u : 0);
        wp[u] = c.wp + (VTW ? (u64)threadIdx.x * 104729ULL + u : 0);
        if (VTW) asm volatile("" : "+v"(w[u]), "+v"(wp[u]));
    }
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters * 2; it++) {
#pragma unroll
        for (int u = 0; u < C8; u++) {
            u32 y0 = lo32(y[u]), y1 = hi32(y[u]), p0 = lo32(wp[u]), p1 = hi32(wp[u]);
            u64 h = mad32(y1, p1, (u64)__umulhi(y0, p1)) + (u64)__umulhi(y1, p0);
            u64 T = y[u] * w[u] + h * c.nq;
            u64 U = x[u];
            x[u] = U + T;
            y[u] = U + c.fourq - T;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    u64 s = 0;
    for (int u = 0; u < C8; u++) s ^= x[u] ^ y[u];
    if (s == 0x12345678) out[4000000] = s + lds_pin[threadIdx.x];
    if ((threadIdx.x & 63) == 0) {
        size_t w_ = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
        out[3 * w_] = t1 - t0;
        out[3 * w_ + 1] = r0;
        out[3 * w_ + 2] = r1;
    }
}

template <int V>
__global__ void __launch_bounds__(1024) k_bfly(unsigned long long* out, const Consts* cp, int iters)
{
    // 96 KiB of LDS per block: at most one block per CU, so the 256-block grid is spread evenly
    __shared__ unsigned lds_pin[24576];
    if (iters < 0) lds_pin[threadIdx.x] = iters;
    Consts c = *cp;
    // give every chain its own twiddle so nothing is hoisted; twiddles live in VGPRs like rounds 2/3
    u64 x[CH], y[CH];
    for (int u = 0; u < CH; u++) {
        x[u] = (threadIdx.x * 1315423911ULL + u * 977ULL) & ((1ULL << 59) - 1);
        y[u] = (x[u] * 2654435761ULL + blockIdx.x) & ((1ULL << 59) - 1);
    }
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < CH; u++) bfly<V>(x[u], y[u], c);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    u64 s = 0;
    for (int u = 0; u < CH; u++) s ^= x[u] ^ y[u];
    if (s == 0x12345678) out[4000000] = s + lds_pin[threadIdx.x];
    // per wave: shader cycles in the loop, and the loop's start/end on the 100 MHz constant clock
    if ((threadIdx.x & 63) == 0) {
        size_t w_ = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
        out[3 * w_] = t1 - t0;
        out[3 * w_ + 1] = r0;
        out[3 * w_ + 2] = r1;
    }
}

typedef void (*kern_t)(unsigned long long*, const Consts*, int);

static void run(const char* name, kern_t k, int block, int blocks_per_cu, int iters, const Consts* dc)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int grid = prop.multiProcessorCount * blocks_per_cu;
    size_t nw = (size_t)grid * block / 64;
    unsigned long long* d;
    CK(hipMalloc(&d, (3 * nw + 8) * sizeof(unsigned long long) + 32000008));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d, dc, iters);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d, dc, iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> h(3 * nw);
    CK(hipMemcpy(h.data(), d, 3 * nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double avg = 0, avg_rt = 0;
    unsigned long long first = ~0ULL, last = 0, last_start = 0;
    for (size_t i = 0; i < nw; i++) {
        avg += (double)h[3 * i];
        avg_rt += (double)(h[3 * i + 2] - h[3 * i + 1]);
        first = std::min(first, h[3 * i + 1]);
        last_start = std::max(last_start, h[3 * i + 1]);
        last = std::max(last, h[3 * i + 2]);
    }
    avg /= nw;
    avg_rt /= nw;
    double waves_per_simd = (double)block * blocks_per_cu / 256.0;
    double bf_per_wave = (double)iters * CH;
    double cyc = avg / (bf_per_wave * waves_per_simd);
    double total_bf = (double)grid * block * bf_per_wave;
    double rate = total_bf / (best * 1e-3);
    // sclk = shader cycles per 100 MHz tick inside the loop; span = first wave's loop start .. last wave's loop end
    printf("%-28s waves/SIMD=%2.0f  cyc/wave-bfly/SIMD=%6.2f  wall=%7.3f ms  loop(avg wave)=%7.3f ms  span=%7.3f ms  start skew=%6.3f ms  sclk=%.2f GHz  %.3e bfly/s  => %.2f M NTT(2^15)/s  (%4.1f%% of 15.26M)\n",
           name, waves_per_simd, cyc, best, avg_rt / 1e5, (double)(last - first) / 1e5, (double)(last_start - first) / 1e5, avg / avg_rt * 0.1, rate,
           rate / 245760 / 1e6, rate / 245760 / 15.26e6 * 100);
    CK(hipFree(d));
}

int main(int argc, char** argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 4000;
    Consts c;
    c.q = 1152921504606584833ULL;
    c.nq = 0ULL - c.q;
    c.twoq = 2 * c.q;
    c.fourq = 4 * c.q;
    c.mu = 1152921504607109119ULL;
    c.w = 4443670208963ULL;
    c.wp = (u64)((((unsigned __int128)c.w) << 64) / c.q);
    Consts* dc;
    CK(hipMalloc(&dc, sizeof(Consts)));
    CK(hipMemcpy(dc, &c, sizeof(Consts), hipMemcpyHostToDevice));
    struct { const char* n; kern_t k; } ks[] = {
        {"0 shoup exact lazy", k_bfly<0>}, {"1 shoup approx (C)", k_bfly<1>}, {"2 shoup approx 32-bit (C)", k_bfly<2>},
        {"3 shoup exact harvey", k_bfly<3>}, {"4 barrett literal", k_bfly<4>}, {"5 shoup approx asm", k_bfly<5>},
        {"6 modmul approx only", k_bfly<6>}, {"7 GS harvey exact", k_bfly<7>}, {"8 GS approx lazy", k_bfly<8>},
        {"9 CT all-mad fold", k_bfly<9>}, {"10 GS all-mad", k_bfly<10>}, {"11 CT fold U", k_bfly<11>},
        {"12 v1, 8 chains, SGPR twiddles", k_bfly_tw<false>}, {"13 v1, 8 chains, VGPR twiddles", k_bfly_tw<true>},
    };
    for (auto& s : ks) {
        run(s.n, s.k, 256, 1, iters, dc);
        run(s.n, s.k, 512, 1, iters, dc);
        run(s.n, s.k, 1024, 1, iters, dc);
    }
    return 0;
}
