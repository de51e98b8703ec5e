// probe: which store / load cache policies make data written by a workgroup on one XCD visible to a workgroup on ANOTHER XCD of
// the same device inside one kernel (no kernel boundary in between), when the reader's L2 already holds the old lines?  (gfx950)
// Workgroup 0 = producer, workgroup 1 = consumer (dealt to different XCDs; the XCC ids are printed).
//   consumer reads the buffer (old values: its L2 now holds the lines) -> ready flag
//   producer writes new values with store policy SP, publishes the flag
//   consumer re-reads with load policy LP and counts words that are still old
//   hipcc --offload-arch=gfx950 -O2 tools/probe/xcd_visibility.hip -o tools/probe/xcd_visibility
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __amdgpu_buffer_rsrc_t BufRsrc;
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));

constexpr unsigned WORDS_PER_THREAD = 8;             // x 16 bytes x 1024 threads = 128 KiB per test buffer

template <int AUX>
__device__ __forceinline__ v4u32 ld(BufRsrc r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX); }
template <int AUX>
__device__ __forceinline__ void st(BufRsrc r, unsigned off, v4u32 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX); }

// SP: 0 plain store, 1 plain store + buffer_wbl2 sc1 (agent-scope release), 2 store sc0 sc1 (written through), 3 store sc1, 4 store nt
// LP: 0 plain load, 1 load sc1, 2 load sc0 sc1, 3 load sc0, 4 plain load behind buffer_inv sc1 (agent-scope acquire), 5 load nt
template <int SP, int LP>
__global__ void __launch_bounds__(1024) k_vis(unsigned* buf, unsigned* flags, unsigned* out, unsigned gen)
{
    const BufRsrc r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1024u * WORDS_PER_THREAD * 16u, 0x00020000);
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (blockIdx.x == 1) {
        // consumer: pull the old lines into this XCD's L2 (twice, so they are certainly resident)
        unsigned acc = 0;
        for (int rep = 0; rep < 2; rep++)
            for (unsigned i = 0; i < WORDS_PER_THREAD; i++) acc += ld<0>(r, (i * 1024u + threadIdx.x) * 16u).x;
        __syncthreads();
        if (threadIdx.x == 0) {
            out[2] = xcc; out[3] = acc;
            __hip_atomic_store(flags + 0, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) __builtin_amdgcn_s_sleep(4);
        }
        __syncthreads();
        asm volatile("" ::: "memory");
        if (LP == 4) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        unsigned stale = 0;
        for (unsigned i = 0; i < WORDS_PER_THREAD; i++) {
            v4u32 v;
            if (LP == 1) v = ld<16>(r, (i * 1024u + threadIdx.x) * 16u);
            else if (LP == 2) v = ld<17>(r, (i * 1024u + threadIdx.x) * 16u);
            else if (LP == 3) v = ld<1>(r, (i * 1024u + threadIdx.x) * 16u);
            else if (LP == 5) v = ld<2>(r, (i * 1024u + threadIdx.x) * 16u);
            else v = ld<0>(r, (i * 1024u + threadIdx.x) * 16u);
            stale += (v.x != gen) + (v.y != gen) + (v.z != gen) + (v.w != gen);
        }
        atomicAdd(out + 0, stale);
    } else if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            out[1] = xcc;
            while (__hip_atomic_load(flags + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) __builtin_amdgcn_s_sleep(4);
        }
        __syncthreads();
        v4u32 v; v.x = v.y = v.z = v.w = gen;
        for (unsigned i = 0; i < WORDS_PER_THREAD; i++) {
            if (SP == 2) st<17>(r, (i * 1024u + threadIdx.x) * 16u, v);
            else if (SP == 3) st<16>(r, (i * 1024u + threadIdx.x) * 16u, v);
            else if (SP == 4) st<2>(r, (i * 1024u + threadIdx.x) * 16u, v);
            else st<0>(r, (i * 1024u + threadIdx.x) * 16u, v);
        }
        if (SP == 1) asm volatile("buffer_wbl2 sc1" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flags + 1, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int SP, int LP>
void run(unsigned* buf, unsigned* flags, unsigned* out, unsigned& gen, const char* sp, const char* lp)
{
    unsigned worst = 0, xp = 0, xc = 0;
    for (int rep = 0; rep < 5; rep++) {
        gen++;
        CK(hipMemset(out, 0, 16));
        k_vis<SP, LP><<<2, 1024>>>(buf, flags, out, gen);
        CK(hipDeviceSynchronize());
        unsigned h[4];
        CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        worst = h[0] > worst ? h[0] : worst; xp = h[1]; xc = h[2];
    }
    printf("store %-28s load %-32s producer xcc %u consumer xcc %u: stale words (max of 5 runs) %6u of %u%s\n", sp, lp, xp & 15, xc & 15, worst,
           1024u * WORDS_PER_THREAD * 4u, worst ? "" : "   <- coherent");
}

int main()
{
    unsigned *buf, *flags, *out;
    CK(hipMalloc(&buf, 1024u * WORDS_PER_THREAD * 16u));
    CK(hipMalloc(&flags, 64)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, 1024u * WORDS_PER_THREAD * 16u)); CK(hipMemset(flags, 0, 64));
    unsigned gen = 0;
#define ROW(SP, SPN) \
    run<SP, 0>(buf, flags, out, gen, SPN, "plain"); run<SP, 1>(buf, flags, out, gen, SPN, "sc1"); run<SP, 2>(buf, flags, out, gen, SPN, "sc0 sc1"); \
    run<SP, 3>(buf, flags, out, gen, SPN, "sc0"); run<SP, 4>(buf, flags, out, gen, SPN, "plain behind buffer_inv sc1"); run<SP, 5>(buf, flags, out, gen, SPN, "nt");
    ROW(0, "plain")
    ROW(1, "plain + buffer_wbl2 sc1")
    ROW(2, "sc0 sc1 (write-through)")
    ROW(3, "sc1")
    ROW(4, "nt")
    return 0;
}
