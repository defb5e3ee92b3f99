// Micro-test: where do the 16 bytes of each lane land in LDS for global_load_lds_dwordx4 (gfx950), and what does the
// instruction offset move?  One wave; lane l reads src[100 * l .. +3] (+ the instruction offset).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* __restrict__ src, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float buf[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) buf[i] = -1.f;
    __syncthreads();
    const float* p = src + 100 * lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)&buf[0], 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)&buf[256], 16, 32, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = buf[i];
}
int main() {
    std::vector<float> h(100 * 64 + 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    float *src, *out; (void)hipMalloc(&src, h.size() * 4); (void)hipMalloc(&out, 4096);
    (void)hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out);
    std::vector<float> o(1024); (void)hipMemcpy(o.data(), out, 4096, hipMemcpyDeviceToHost);
    printf("first request (offset 0), lanes 0..3: ");
    for (int i = 0; i < 16; ++i) printf("%g ", o[i]);
    printf("\nsecond request (LDS base +1024 B, instruction offset 32), floats 248..279: ");
    for (int i = 248; i < 280; ++i) printf("%g ", o[i]);
    int first = -1; for (int i = 256; i < 1024; ++i) if (o[i] != -1.f) { first = i; break; }
    printf("\nfirst written float at or after 256: %d (value %g)\n", first, first >= 0 ? o[first] : 0.f);
    return 0;
}
