// Achievable HBM bandwidth of plain streaming kernels on this GPU: read-only, write-only, copy and a
// 2-read/1-write mix, 3 GiB per array.  Every thread keeps U 16-byte requests in flight (U = 1: the plain
// grid-stride loop; U = 4, 8: unrolled, all loads of a round issued before the first store), with default and
// non-temporal cache policy, over several grid sizes; the best of five runs per point.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const f4* __restrict__ a, f4* out, size_t n) {
    f4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride * U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = i + u * stride;
            v[u] = k < n ? (NT ? __builtin_nontemporal_load(a + k) : a[k]) : f4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc[0] == 1234.5f) out[0] = acc;
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_write(f4* __restrict__ a, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = i + u * stride;
            if (k < n) { if (NT) __builtin_nontemporal_store(f4{1, 2, 3, 4}, a + k); else a[k] = f4{1, 2, 3, 4}; }
        }
    }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride * U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = i + u * stride;
            v[u] = k < n ? (NT ? __builtin_nontemporal_load(a + k) : a[k]) : f4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = i + u * stride;
            if (k < n) { if (NT) __builtin_nontemporal_store(v[u], b + k); else b[k] = v[u]; }
        }
    }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_mix(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride * U) {
        f4 v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = i + u * stride;
            v[u] = k < n ? (NT ? __builtin_nontemporal_load(a + k) : a[k]) : f4{0, 0, 0, 0};
            w[u] = k < n ? (NT ? __builtin_nontemporal_load(b + k) : b[k]) : f4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = i + u * stride;
            if (k < n) { if (NT) __builtin_nontemporal_store(v[u] + w[u], c + k); else c[k] = v[u] + w[u]; }
        }
    }
}

template <int U, bool NT>
void run(f4* a, f4* b, f4* c, size_t n, hipEvent_t e0, hipEvent_t e1) {
    const char* names[] = {"read", "write", "copy (1R+1W)", "add (2R+1W)"};
    for (int which = 0; which < 4; ++which) {
        float best = 1e9f; int best_blocks = 0;
        for (int blocks : {512, 1024, 2048, 4096, 8192, 16384, 65536}) {
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0, 0);
                if (which == 0) hipLaunchKernelGGL((k_read<U, NT>), dim3(blocks), dim3(256), 0, 0, a, c, n);
                if (which == 1) hipLaunchKernelGGL((k_write<U, NT>), dim3(blocks), dim3(256), 0, 0, c, n);
                if (which == 2) hipLaunchKernelGGL((k_copy<U, NT>), dim3(blocks), dim3(256), 0, 0, a, c, n);
                if (which == 3) hipLaunchKernelGGL((k_mix<U, NT>), dim3(blocks), dim3(256), 0, 0, a, b, c, n);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) { best = ms; best_blocks = blocks; }
            }
        }
        const double bytes = (double)n * 16 * (which == 0 || which == 1 ? 1 : which == 2 ? 2 : 3);
        printf("U=%d %-3s %-14s %7.3f ms  %6.2f TB/s  (best at %d blocks)\n", U, NT ? "nt" : "", names[which], best,
               bytes / best * 1e-9, best_blocks);
    }
}

int main() {
    const size_t n = (size_t)3 << 26;          // f4 elements: 3 GiB per array
    f4 *a, *b, *c;
    if (hipMalloc(&a, n * 16) != hipSuccess || hipMalloc(&b, n * 16) != hipSuccess || hipMalloc(&c, n * 16) != hipSuccess) return 1;
    hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16); hipMemset(c, 0, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    run<1, false>(a, b, c, n, e0, e1);
    run<4, false>(a, b, c, n, e0, e1);
    run<8, false>(a, b, c, n, e0, e1);
    run<4, true>(a, b, c, n, e0, e1);
    run<8, true>(a, b, c, n, e0, e1);
    return 0;
}
