// probe: where do the bytes of `buffer_load_dwordx4 ... lds` land?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__global__ void k(const unsigned* a, unsigned* out, unsigned swap)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdeadbeef;
    __syncthreads();
    BufRsrc r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, 1 << 20, 0x00020000);
    // lane L loads 16 bytes from global byte offset voff(L); the LDS base is lds + 64 dwords
    unsigned lane = threadIdx.x;
    unsigned voff = swap ? ((lane ^ 1) * 16u) : (lane < 32 ? lane * 16u : 4096u + (lane - 32) * 16u);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 64), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main()
{
    std::vector<unsigned> h(1 << 18);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned)i;      // dword index as value
    unsigned *a, *o;
    hipMalloc(&a, h.size() * 4); hipMalloc(&o, 2048 * 4);
    hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (unsigned swap = 0; swap < 2; swap++) {
        k<<<1, 64>>>(a, o, swap);
        std::vector<unsigned> r(2048);
        hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
        printf("swap=%u\n", swap);
        for (int i = 56; i < 64 + 64 * 4 + 8; i++) { if (r[i] != 0xdeadbeef) printf("lds[%d]=%u ", i, r[i]); if (i % 8 == 7) printf("\n"); }
        printf("\n");
    }
    return 0;
}
