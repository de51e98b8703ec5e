// ubench_valu.hip -- gfx950 VALU issue-rate microbenchmark for the instructions a 64-bit modular
// butterfly is made of (v_mad_u64_u32, v_mul_lo/hi_u32, 64-bit add/sub/compare/select, ...), plus
// whole-butterfly variants built from ntt-cuda_amd/csrc/modarith.cuh.
//
// Build: hipcc --offload-arch=gfx950 -O3 -I ntt-cuda_amd/csrc tools/ubench_valu.hip -o tools/ubench_valu
// Run  : ./tools/ubench_valu            (prints cycles per wave-instruction per SIMD)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "modarith.cuh"

using namespace mi355ntt;

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

// iteration count is a kernel argument so runs can be made long enough (ms) for clocks to settle
constexpr int UNROLL = 16;  // independent chains per iteration

// ---- single-instruction kernels: UNROLL independent dependency chains --------------------------
#define INSTR_KERNEL(NAME, DECL, BODY, SINK)                                                     \
    __global__ void NAME(unsigned long long* out, unsigned seed, int ITERS)                      \
    {                                                                                            \
        DECL;                                                                                    \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                    \
        for (int it = 0; it < ITERS; it++) {                                                     \
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) { BODY; }                         \
        }                                                                                        \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                    \
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                \
        SINK;                                                                                    \
        if ((threadIdx.x & 63) == 0) { unsigned w_ = (blockIdx.x * blockDim.x + threadIdx.x) / 64; out[2 * w_] = t1 - t0; out[2 * w_ + 1] = r1 - r0; } \
    }

#define DECL32                                                        \
    unsigned a[UNROLL], b = seed | 1, c = seed * 3 + 7;               \
    for (int u = 0; u < UNROLL; u++) a[u] = threadIdx.x * 17 + u + seed
#define SINK32                                   \
    unsigned s = 0;                              \
    for (int u = 0; u < UNROLL; u++) s ^= a[u];  \
    if (s == 0x12345678) out[2000000] = s

#define DECL64                                                              \
    unsigned long long a[UNROLL];                                           \
    unsigned b = seed | 1, c = seed * 3 + 7;                                \
    unsigned long long b64 = ((unsigned long long)seed << 33) | 12345;      \
    for (int u = 0; u < UNROLL; u++) a[u] = ((unsigned long long)threadIdx.x << 32) * 17 + u + seed
#define SINK64                                        \
    unsigned long long s = 0;                         \
    for (int u = 0; u < UNROLL; u++) s ^= a[u];       \
    if (s == 0x12345678) out[2000000] = s

INSTR_KERNEL(k_mul_lo_u32, DECL32, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b)), SINK32)
INSTR_KERNEL(k_mul_hi_u32, DECL32, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b)), SINK32)
INSTR_KERNEL(k_mul_u32_u24, DECL32, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[u]) : "v"(b)), SINK32)
INSTR_KERNEL(k_mad_u32_u24, DECL32, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c)), SINK32)
INSTR_KERNEL(k_add_u32, DECL32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b)), SINK32)
INSTR_KERNEL(k_add3_u32, DECL32, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c)), SINK32)
INSTR_KERNEL(k_mov_dpp, DECL32, asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[u])), SINK32)
INSTR_KERNEL(k_mad_u64_u32, DECL64,
             asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[u]) : "v"(b), "v"(c) : "vcc"), SINK64)
INSTR_KERNEL(k_lshl_add_u64, DECL64, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[u]) : "v"(b64)), SINK64)
// 64-bit add as a carry pair, and 64-bit compare + 2 selects, on split halves
#define DECL2x32                                                        \
    unsigned al[UNROLL], ah[UNROLL], bl = seed | 1, bh = seed * 3 + 7;  \
    for (int u = 0; u < UNROLL; u++) { al[u] = threadIdx.x * 17 + u + seed; ah[u] = al[u] * 3; }
#define SINK2x32                                          \
    unsigned s = 0;                                       \
    for (int u = 0; u < UNROLL; u++) s ^= al[u] ^ ah[u];  \
    if (s == 0x12345678) out[2000000] = s
INSTR_KERNEL(k_add_co_pair, DECL2x32,
             asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(al[u]), "+v"(ah[u]) : "v"(bl), "v"(bh) : "vcc"),
             SINK2x32)
INSTR_KERNEL(k_cmp_u64, DECL64,
             asm volatile("v_cmp_ge_u64 vcc, %0, %1" : : "v"(a[u]), "v"(b64) : "vcc"), SINK64)
INSTR_KERNEL(k_cndmask, DECL32, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[u]) : "v"(b) : ), SINK32)
INSTR_KERNEL(k_fma_f64, DECL64, asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[u]) : "v"(b64)), SINK64)
INSTR_KERNEL(k_mul_u64_c, DECL64, a[u] = a[u] * b64 + u, SINK64)           // compiler's 64x64->64
INSTR_KERNEL(k_mulhi_u64_c, DECL64, a[u] = mul_hi(a[u], b64) + u, SINK64)  // 64x64 -> high 64

// ---- whole-butterfly kernels --------------------------------------------------------------------
// Each chain carries an (x, y) pair through repeated butterflies with constant twiddle.
#define DECLBF                                                                      \
    u64 x[UNROLL], y[UNROLL];                                                       \
    u64 q = 1152921504606584833ULL + (u64)(seed >> 31);                                   \
    u64 w = 4443670208963ULL + seed, wp = 0x123456789abcdefULL + seed;              \
    u64 mu = 1152921504607109119ULL + (u64)(seed >> 30);                                  \
    u64 twoq = 2 * q;                                                               \
    (void)mu; (void)wp; (void)twoq;                                                 \
    for (int u = 0; u < UNROLL; u++) { x[u] = threadIdx.x * 1315423911ULL + u; y[u] = x[u] * 2654435761ULL + seed; }
#define SINKBF                                            \
    u64 s = 0;                                            \
    for (int u = 0; u < UNROLL; u++) s ^= x[u] ^ y[u];    \
    if (s == 0x12345678) out[2000000] = s

#define BF_KERNEL(NAME, BODY)                                                                    \
    __global__ void NAME(unsigned long long* out, unsigned seed, int ITERS)                      \
    {                                                                                            \
        DECLBF;                                                                                  \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                    \
        for (int it = 0; it < ITERS / 4; it++) {                                                 \
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) { BODY; }                         \
        }                                                                                        \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                    \
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                \
        SINKBF;                                                                                  \
        if ((threadIdx.x & 63) == 0) { unsigned w_ = (blockIdx.x * blockDim.x + threadIdx.x) / 64; out[2 * w_] = t1 - t0; out[2 * w_ + 1] = r1 - r0; } \
    }

// reference-literal CT butterfly: Barrett (Algorithm 7) + canonical add/sub
BF_KERNEL(k_bf_barrett, {
    u64 V = barrett_mul(y[u], w, q, mu, 60);
    u64 U = x[u];
    x[u] = add_mod(U, V, q);
    y[u] = sub_mod(U, V, q);
})
// Shoup + fully lazy add/sub (no conditional subtraction)
BF_KERNEL(k_bf_shoup_lazy, {
    u64 T = shoup_mul_lazy(y[u], w, wp, q);
    u64 U = x[u];
    x[u] = U + T;
    y[u] = U - T + twoq;
})
// Shoup + Harvey (one conditional subtraction on U)
BF_KERNEL(k_bf_shoup_harvey, {
    u64 T = shoup_mul_lazy(y[u], w, wp, q);
    u64 U = x[u];
    U = U >= twoq ? U - twoq : U;
    x[u] = U + T;
    y[u] = U - T + twoq;
})
// Shoup + canonical add/sub every stage
BF_KERNEL(k_bf_shoup_canon, {
    u64 T = csub(shoup_mul_lazy(y[u], w, wp, q), q);
    u64 U = x[u];
    x[u] = add_mod(U, T, q);
    y[u] = sub_mod(U, T, q);
})
// modmul only (Shoup)
BF_KERNEL(k_mul_shoup, { y[u] = shoup_mul_lazy(y[u], w, wp, q) + x[u]; })
// modmul only (Barrett literal)
BF_KERNEL(k_mul_barrett, { y[u] = barrett_mul(y[u], w, q, mu, 60) + (x[u] & 1); })

typedef void (*kern_t)(unsigned long long*, unsigned, int);

static int g_iters = 20000;

static void run(const char* name, kern_t k, int block, int per_iter_ops, int iters)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int grid = prop.multiProcessorCount;  // one block per CU
    unsigned long long* d;
    size_t nw = (size_t)grid * block / 64;
    CK(hipMalloc(&d, 16000064));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d, 1u, iters);  // warm-up
    CK(hipDeviceSynchronize());
    float ms = 1e30f;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d, 1u, iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float m;
        CK(hipEventElapsedTime(&m, e0, e1));
        if (m < ms) ms = m;
    }
    std::vector<unsigned long long> h(2 * nw);
    CK(hipMemcpy(h.data(), d, 2 * nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0;
    for (size_t i = 0; i < nw; i++) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
    cyc /= nw; rt /= nw;
    int waves_per_simd = block / 256;
    double ops = (double)iters * per_iter_ops;  // per wave
    printf("%-20s waves/SIMD=%d  cyc/wave-op/SIMD=%6.2f  wall=%7.3f ms  clk(memtime/realtime)=%.3f GHz  Gops/s(lane)=%.1f\n", name,
           waves_per_simd, cyc / (ops * waves_per_simd), ms, cyc / rt * 0.1, (double)grid * block * ops / (ms * 1e-3) / 1e9);
    CK(hipFree(d));
}

int main(int argc, char** argv)
{
    if (argc > 1) g_iters = atoi(argv[1]);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    int blocks[] = {256, 512, 1024};
    struct { const char* n; kern_t k; } singles[] = {
        {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mul_u32_u24", k_mul_u32_u24},
        {"v_mad_u32_u24", k_mad_u32_u24}, {"v_add_u32", k_add_u32}, {"v_add3_u32", k_add3_u32},
        {"v_mov_b32_dpp", k_mov_dpp}, {"v_mad_u64_u32", k_mad_u64_u32}, {"v_lshl_add_u64", k_lshl_add_u64},
        {"add_co+addc (2)", k_add_co_pair}, {"v_fma_f64", k_fma_f64},
        {"u64 mul (compiler)", k_mul_u64_c}, {"u64 mulhi (4 mad)", k_mulhi_u64_c},
    };
    for (auto& s : singles)
        for (int b : blocks) run(s.n, s.k, b, UNROLL, g_iters);
    return 0;
}
