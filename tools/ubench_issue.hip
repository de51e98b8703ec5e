// ubench_issue.hip -- steady-state issue cost of single VALU instructions on gfx950, 1..4 waves per SIMD.
// Deadline method (see ubench_ceiling.hip): every wave issues the instruction stream until a common s_memrealtime
// deadline; sum of completed instructions / window = throughput under the SIMD's own arbitration, no tails.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_issue.hip -o tools/ubench_issue
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Out { unsigned long long n, t_first, t_last, clk; };
constexpr int CH = 16;

template <int OP>
__global__ void __launch_bounds__(1024) k_issue(Out* out, unsigned long long window_ticks, unsigned seed)
{
    unsigned a[CH], a2[CH];
    unsigned long long l[CH];
    unsigned b = seed | 1u, c = seed * 3u + 7u;
    unsigned long long b64 = ((unsigned long long)seed << 33) | 12345u;
#pragma unroll
    for (int u = 0; u < CH; u++) { a[u] = threadIdx.x * 17u + u + seed; a2[u] = a[u] * 3u; l[u] = ((unsigned long long)a[u] << 32) | a2[u]; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long t_end = t0 + window_ticks;
    unsigned long long n = 0, tl = t0;
    for (;;) {
#pragma unroll 1
        for (int rep = 0; rep < 16; rep++) {
#pragma unroll
            for (int u = 0; u < CH; u++) {
                if constexpr (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b));
                if constexpr (OP == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b));
                if constexpr (OP == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(l[u]) : "v"(b), "v"(c) : "vcc");
                if constexpr (OP == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(l[u]) : "v"(b64));
                if constexpr (OP == 4) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
                if constexpr (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b));
                if constexpr (OP == 6) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a[u]), "+v"(a2[u]) : "v"(b), "v"(c) : "vcc");
                if constexpr (OP == 7) asm volatile("v_sub_co_u32 %0, vcc, %0, %2\n\tv_subb_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a[u]), "+v"(a2[u]) : "v"(b), "v"(c) : "vcc");
                if constexpr (OP == 8) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[u]));
                if constexpr (OP == 9) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[u]) : "v"(b));
                if constexpr (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[u]) : "v"(b) : );
                if constexpr (OP == 11) asm volatile("v_cmp_ge_u64 vcc, %0, %1" : : "v"(l[u]), "v"(b64) : "vcc");
                if constexpr (OP == 12) asm volatile("v_mov_b32 %0, %1" : "+v"(a[u]) : "v"(b));
                if constexpr (OP == 13) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
                if constexpr (OP == 14) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[u]) : "v"(b));
                if constexpr (OP == 15) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[u]) : "v"(b));
                if constexpr (OP == 16) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(l[u]));
                if constexpr (OP == 17) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[u]) : "v"(b));
                if constexpr (OP == 18) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[u]) : "v"(b));
                if constexpr (OP == 19) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(l[u]) : "v"(b64));
                if constexpr (OP == 20) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[u]) : "s"(b));
                if constexpr (OP == 21) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(l[u]) : "v"(a[u]), "v"(c) : "vcc");
                if constexpr (OP == 22) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b));
                if constexpr (OP == 23) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[u]) : "v"(b));
                if constexpr (OP == 24) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(a[u]));
                if constexpr (OP == 25) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[u]) : "v"(b) : "vcc");
                if constexpr (OP == 26) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[u]));
                if constexpr (OP == 27) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
                if constexpr (OP == 28) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(l[u]) : "v"(b), "s"(c) : "vcc");
                if constexpr (OP == 29) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[u]) : "v"(b));
                if constexpr (OP == 30) asm volatile("v_mad_u32_u16 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
                if constexpr (OP == 31) asm volatile("v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
                if constexpr (OP == 32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[u]), "+v"(a2[u]));
                if constexpr (OP == 33) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[u]));
                if constexpr (OP == 34) asm volatile("v_not_b32 %0, %0" : "+v"(a[u]));
                if constexpr (OP == 35) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[u]));
                if constexpr (OP == 36) asm volatile("v_cmp_ge_u64 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %4, %4, %3, vcc" : "+v"(l[u]), "+v"(b64), "+v"(a[u]), "+v"(b), "+v"(a2[u]) : : "vcc");
            }
        }
        n += 16 * CH;
        tl = __builtin_amdgcn_s_memrealtime();
        if (tl >= t_end) break;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
#pragma unroll
    for (int u = 0; u < CH; u++) s ^= a[u] ^ a2[u] ^ (unsigned)l[u] ^ (unsigned)(l[u] >> 32);
    if (s == 0x1234567) out[0].n = s;
    if ((threadIdx.x & 63) == 0) {
        const unsigned wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        out[1 + wv] = Out{n, t0, tl, c1 - c0};
    }
}

typedef void (*kern_t)(Out*, unsigned long long, unsigned);

static int g_reps = 2, g_wps_lo = 1;      // sustained mode: many long windows back to back at 4 waves per SIMD only
static void run(const char* name, kern_t k, int per_op, unsigned long long window_us)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("%-34s", name);
    for (int wps = g_wps_lo; wps <= 4; wps++) {
        const int grid = prop.multiProcessorCount, block = 256 * wps;
        const size_t nw = (size_t)grid * block / 64;
        Out* d;
        CK(hipMalloc(&d, (nw + 1) * sizeof(Out)));
        for (int rep = 0; rep < g_reps; rep++) {
            hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d, window_us * 100ull, 1u);
            CK(hipDeviceSynchronize());
        }
        std::vector<Out> h(nw + 1);
        CK(hipMemcpy(h.data(), d, (nw + 1) * sizeof(Out), hipMemcpyDeviceToHost));
        double total = 0, clk = 0;
        unsigned long long tmin = ~0ull, tmax = 0;
        double per_slot[4] = {0, 0, 0, 0};
        for (size_t i = 1; i <= nw; i++) {
            total += (double)h[i].n * per_op;
            tmin = std::min(tmin, h[i].t_first); tmax = std::max(tmax, h[i].t_last);
            clk += (double)h[i].clk / (double)(h[i].t_last - h[i].t_first);
            per_slot[((i - 1) % (block / 64)) / 4 % 4] += (double)h[i].n;
        }
        const double win = (double)(tmax - tmin) * 1e-8, ghz = clk / nw * 0.1;
        const double cyc = ghz * 1e9 * win * (grid * 4.0) / total;       // shader cycles per wave-instruction per SIMD
        printf("  w%d: %5.2f cyc", wps, cyc);
        if (wps == 4) { double ps = per_slot[0] + per_slot[1] + per_slot[2] + per_slot[3]; printf("  (%.2f GHz; share by age %.2f %.2f %.2f %.2f)", ghz, per_slot[0] / ps, per_slot[1] / ps, per_slot[2] / ps, per_slot[3] / ps); }
        CK(hipFree(d));
    }
    printf("\n");
}

int main(int argc, char** argv)
{
    unsigned long long win = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1500;
    struct { const char* n; kern_t k; int per; } ops[] = {
        {"v_mul_lo_u32", k_issue<0>, 1}, {"v_mul_hi_u32", k_issue<1>, 1}, {"v_mad_u64_u32 (acc)", k_issue<2>, 1}, {"v_mad_u64_u32 (+0)", k_issue<21>, 1},
        {"v_mad_u64_u32 (sgpr src)", k_issue<28>, 1}, {"v_mul_lo_u32 (sgpr src)", k_issue<20>, 1},
        {"v_lshl_add_u64", k_issue<3>, 1}, {"v_add3_u32", k_issue<4>, 1}, {"v_add_u32", k_issue<5>, 1}, {"v_sub_u32", k_issue<22>, 1},
        {"v_add_co_u32", k_issue<25>, 1}, {"v_add_co + v_addc (pair)", k_issue<6>, 1}, {"v_sub_co + v_subb (pair)", k_issue<7>, 1},
        {"v_lshlrev_b32", k_issue<8>, 1}, {"v_lshrrev_b32", k_issue<26>, 1}, {"v_and_b32", k_issue<9>, 1}, {"v_xor_b32", k_issue<23>, 1}, {"v_bfe_u32", k_issue<24>, 1},
        {"v_and_or_b32", k_issue<27>, 1}, {"v_lshl_add_u32", k_issue<18>, 1}, {"v_cndmask_b32", k_issue<10>, 1}, {"v_cmp_ge_u64", k_issue<11>, 1},
        {"v_mov_b32", k_issue<12>, 1}, {"v_mad_u32_u24", k_issue<13>, 1}, {"v_mul_u32_u24", k_issue<14>, 1}, {"v_mul_hi_u32_u24", k_issue<17>, 1},
        {"v_alignbit_b32", k_issue<15>, 1}, {"v_lshlrev_b64", k_issue<16>, 1}, {"v_fma_f64", k_issue<19>, 1},
        {"v_pk_add_u16", k_issue<29>, 1}, {"v_mad_u32_u16", k_issue<30>, 1}, {"v_dot2_u32_u16", k_issue<31>, 1},
        {"v_permlane32_swap_b32", k_issue<32>, 1}, {"v_mov_b32_dpp quad_perm", k_issue<33>, 1}, {"v_not_b32", k_issue<34>, 1},
        {"v_ashrrev_i32", k_issue<35>, 1}, {"v_cmp_ge_u64 + 2 v_cndmask (triple)", k_issue<36>, 1},
    };
    printf("# shader cycles per wave-instruction per SIMD (pairs count as one), 1..4 waves per SIMD, %llu us windows\n", win);
    if (argc > 2) {
        // sustained mode: ubench_issue <window_us> <reps> <op index> ...: the named ops only, 4 waves per SIMD, `reps` windows back to back
        // (e.g. 20000 us x 150 = 3 s per op) -- run it next to a rocm-smi sampler to read the package power a full-rate stream of ONE
        // instruction draws (tools/exp_r04e.sh)
        g_reps = atoi(argv[2]);
        g_wps_lo = 4;
        for (int i = 3; i < argc; i++) {
            const int k = atoi(argv[i]);
            if (k >= 0 && k < (int)(sizeof(ops) / sizeof(ops[0]))) { printf("[%d] ", k); run(ops[k].n, ops[k].k, ops[k].per, win); fflush(stdout); }
        }
        return 0;
    }
    for (auto& o : ops) run(o.n, o.k, o.per, win);
    return 0;
}
