// calib_copy.hip -- known-byte-count streaming copies in the access widths the NTT kernels use, to calibrate
// rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide reads).
//   copy8 : buffer_load_dwordx2 / buffer_store_dwordx2  (8 B per lane, 512 B per wave instruction)
//   copy16: buffer_load_dwordx4 / buffer_store_dwordx4  (16 B per lane)
// Each launch moves exactly BYTES in and BYTES out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__global__ void copy8(const unsigned long long* a, unsigned long long* b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void copy16(const ulonglong2* a, ulonglong2* b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main()
{
    const size_t BYTES = 512ull << 20;   // 512 MiB each way: past the 256 MiB Infinity Cache
    unsigned long long *a, *b;
    if (hipMalloc(&a, BYTES) != hipSuccess || hipMalloc(&b, BYTES) != hipSuccess) return 1;
    hipMemset(a, 1, BYTES);
    hipMemset(b, 2, BYTES);
    for (int i = 0; i < 3; i++) {
        copy8<<<2048, 256>>>(a, b, BYTES / 8);
        copy16<<<2048, 256>>>((const ulonglong2*)a, (ulonglong2*)b, BYTES / 16);
    }
    hipDeviceSynchronize();
    printf("each launch: %zu bytes read, %zu bytes written\n", BYTES, BYTES);
    return 0;
}
