// Does v_mfma_f32_32x32x16_f16 honour subnormal fp16 inputs?  Prints D[0][0] for A = 2^-20 (fp16 subnormal), B = 1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float aval) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)1.0f; }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    for (float v : {9.5367431640625e-07f /*2^-20*/, 6.103515625e-05f /*2^-14 min normal*/, 5.9604644775390625e-08f /*2^-24 min subnormal*/}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, v);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a = %.10g  D = %.10g  expected %.10g\n", v, h, 16.0f * v);
    }
    return 0;
}
