// Micro-benchmarks (one workgroup of W waves per CU, s_memtime around each body, lane 0 of wave 0 reports):
// cycles per instruction for a few instruction streams, to calibrate kernel cost models.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define T0 unsigned long long t0_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_) :: "memory");
#define T1(slot) { unsigned long long t1_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory"); if (threadIdx.x == 0 && blockIdx.x == 0) out[slot] = t1_ - t0_; }

__global__ void k(unsigned long long* out, float* sink, int reps) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 0.001f;
    __syncthreads();
    h8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * lane + i); b[i] = (_Float16)(0.5f + i); }
    f16v c = {0};
    float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
    // 0: 24 dependent MFMAs
    { T0 for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 24; ++i) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
      } T1(0) }
    // 1: 96 independent VALU fma (4 chains)
    { T0 for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 24; ++i) { x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f); }
      } T1(1) }
    // 2: 96 dependent VALU fma (1 chain)
    { T0 for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 96; ++i) x0 = fmaf(x0, 1.0001f, 0.5f);
      } T1(2) }
    // 3: 16 ds_read_b128, all issued then one wait
    { f4 acc = {0, 0, 0, 0};
      T0 for (int r = 0; r < reps; ++r) {
        f4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = *(const f4*)(lds + ((lane * 4 + 264 * i + r) & 8188));
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += v[i];
      } T1(3) x1 += acc[0] + acc[1] + acc[2] + acc[3]; }
    // 4: 16 dependent ds_read_b32 round trips (address from the previous value)
    { int idx = lane;
      T0 for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) idx = ((int)lds[idx & 8191] + idx + 1) & 8191;
      } T1(4) x2 += idx; }
    // 5: 24 MFMAs with 4 independent VALU each (same wave)
    { T0 for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
            x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
        }
      } T1(5) }
    // 6: __syncthreads only
    { T0 for (int r = 0; r < reps; ++r) { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); } T1(6) }
    // 7: 16 cvt_pk_f16 + cvt back + sub (split2) on 32 values
    { float y[32]; for (int i = 0; i < 32; ++i) y[i] = x0 + i;
      unsigned accu = 0;
      T0 for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 32; i += 2) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2))); typedef float f2 __attribute__((ext_vector_type(2)));
            f2 v = {y[i] + r, y[i + 1] + r}; h2 hh = __builtin_convertvector(v, h2); f2 rr = v - __builtin_convertvector(hh, f2);
            h2 ll = __builtin_convertvector(rr, h2); accu += __builtin_bit_cast(unsigned, hh) ^ __builtin_bit_cast(unsigned, ll);
        }
      } T1(7) x3 += accu; }
    if (sink) sink[threadIdx.x] = c[0] + c[5] + x0 + x1 + x2 + x3;
}
int main() {
    unsigned long long* d; float* s; hipMalloc(&d, 64 * 8); hipMalloc(&s, 4096 * 4);
    const char* names[] = {"24 dependent MFMA 32x32x16 f16", "96 VALU fma, 4 chains", "96 VALU fma, 1 chain", "16 ds_read_b128 + 16x4 adds", "16 dependent ds_read_b32 round trips", "24 x (MFMA + 4 VALU)", "4 x __syncthreads", "split2 of 32 values"};
    for (int waves : {1, 4, 8}) {
        const int reps = 50;
        hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, d, s, reps);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, d, s, reps);
        unsigned long long h[8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("--- %d waves per workgroup (1 WG per CU)\n", waves);
        for (int i = 0; i < 8; ++i) printf("  %-40s %8.1f cycles per body\n", names[i], (double)h[i] / reps);
    }
    return 0;
}
