// Cost of ISSUING vector-memory instructions (cycles from the first to after the last issue, no wait
// in between) and of their completion, for 1 KiB-per-instruction patterns, W waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define NOW(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")

__global__ void k(const float* src, float* dst, const int* rows, unsigned long long* out, int pattern, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * (blockDim.x >> 6) + wave;
    unsigned long long t0, t1, t2, issue = 0, total = 0;
    f4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        f4 v[16];
        const size_t base = ((size_t)(gw * reps + r) * 16) * 256;          // 16 KiB per wave and repetition
        __syncthreads();
        NOW(t0);
        if (pattern == 0) {            // coalesced rows: lane -> 16 B of a 1 KiB run
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = *(const f4*)(src + base + i * 256 + lane * 4);
        } else if (pattern == 1) {     // gathers: 32 rows x 32 B per instruction (lane pair shares a row)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = rows[(gw * 31 + r * 7 + (lane & 31) + 32 * (i >> 2)) & 0xFFFFF];
                v[i] = *(const f4*)(src + (size_t)row * 128 + (i & 3) * 8 + (lane >> 5) * 4);
            }
        } else {                       // coalesced stores
#pragma unroll
            for (int i = 0; i < 16; ++i) *(f4*)(dst + base + i * 256 + lane * 4) = f4{(float)i, 1.f, 2.f, 3.f};
        }
        NOW(t1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NOW(t2);
        if (pattern != 2) for (int i = 0; i < 16; ++i) acc += v[i];
        issue += t1 - t0; total += t2 - t0;
    }
    if (acc[0] == 12345.678f) dst[0] = acc[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = issue / reps; out[1] = total / reps; }
}
int main() {
    const size_t n = (size_t)1 << 28;   // 1 GiB of floats
    float *src, *dst; int* rows; unsigned long long* out;
    hipMalloc(&src, n * 4); hipMalloc(&dst, n * 4); hipMalloc(&rows, (1 << 20) * 4); hipMalloc(&out, 16);
    hipMemset(src, 0, n * 4);
    int* hr = (int*)malloc((1 << 20) * 4);
    unsigned s = 12345; for (int i = 0; i < (1 << 20); ++i) { s = s * 1664525u + 1013904223u; hr[i] = (s >> 8) % 1000000; }
    hipMemcpy(rows, hr, (1 << 20) * 4, hipMemcpyHostToDevice);
    const char* names[] = {"16 coalesced 1 KiB loads", "16 gathers (32 rows x 32 B)", "16 coalesced 1 KiB stores"};
    for (int waves : {1, 4, 8}) {
        for (int p = 0; p < 3; ++p) {
            const int reps = 64;
            hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, src, dst, rows, out, p, reps);
            hipDeviceSynchronize();
            unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            printf("%d waves/CU  %-30s issue %6llu cycles (%5.1f per instr)   issue+complete %6llu\n", waves, names[p], h[0], h[0] / 16.0, h[1]);
        }
    }
    return 0;
}
