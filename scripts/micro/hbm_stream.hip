// Achievable HBM bandwidth of plain streaming kernels on this GPU: read-only, write-only, copy and a
// 3-read/2-write mix shaped like the edge update (read e, gather-free; write e_out), 3 GiB per array.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_read(const f4* a, f4* out, size_t n) {
    f4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i];
    if (acc[0] == 1234.5f) out[0] = acc;
}
__global__ void k_write(f4* a, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = f4{1, 2, 3, 4};
}
__global__ void k_copy(const f4* a, f4* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_mix(const f4* a, const f4* b, f4* c, size_t n) {           // 2 reads, 1 write
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c[i] = a[i] + b[i];
}
int main() {
    const size_t n = (size_t)3 << 26;          // f4 elements: 3 GiB per array
    f4 *a, *b, *c;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16);
    hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16); hipMemset(c, 0, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1024, 4096, 16384}) {
        for (int which = 0; which < 4; ++which) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0, 0);
                if (which == 0) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, c, n);
                if (which == 1) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, c, n);
                if (which == 2) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, c, n);
                if (which == 3) hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, a, b, c, n);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double bytes = (double)n * 16 * (which == 0 || which == 1 ? 1 : which == 2 ? 2 : 3);
            const char* names[] = {"read", "write", "copy (1R+1W)", "add (2R+1W)"};
            printf("%6d blocks  %-14s %7.3f ms  %6.2f TB/s\n", blocks, names[which], best, bytes / best * 1e-9);
        }
    }
    return 0;
}
