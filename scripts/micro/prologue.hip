// Micro-benchmark: what a node-kernel workgroup spends before its first tile.  One 4-wave workgroup per block,
// `blocks` blocks; variants of loading five 128 x 128 weight slices (fp32, row-major) into fp16 planes:
//   0  nothing (launch + exit)          1  raw loads only (row-strided, 16 B per lane)
//   2  raw loads + split (the product)  3  coalesced loads of a pre-packed buffer (what a prepared weight block would cost)
// Lane 0 of block 0 reports cycles (s_memtime) from kernel entry to the end of the variant; the host times the launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../gnn_amd/csrc/mgn_device.h"

__global__ __launch_bounds__(256, 1) void k(const float* W, const u32x4* packed, int variant, float* sink, unsigned long long* cyc) {
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
    WPlanesH<16> wa; WPlanesH<8> wb, ws, wr;
    float m = 0.f;
    if (variant == 1 || variant == 2) {
        load_wraw_h<16>(wa, W, 256, 0, 32 * w + j, h);
        load_wraw_h<8>(wb, W + 128 * 256, 128, 0, 32 * w + j, h);
        load_wraw_h<8>(ws, W + 128 * 384, 384, 0, 32 * w + j, h);
        load_wraw_h<8>(wr, W + 128 * 384, 384, 128, 32 * w + j, h);
        __builtin_amdgcn_sched_barrier(0);
        if (variant == 2) { split_wraw_h<16>(wa, m); split_wraw_h<8>(wb, m); split_wraw_h<8>(ws, m); split_wraw_h<8>(wr, m); }
    } else if (variant == 3) {
        const u32x4* p = packed + (size_t)w * 80 * 64 + lane;          // [wave][reg quad][lane]: 1 KiB per instruction
#pragma unroll
        for (int c = 0; c < 16; ++c) { wa.p[0][c] = p[(2 * c) * 64]; wa.p[1][c] = p[(2 * c + 1) * 64]; }
#pragma unroll
        for (int c = 0; c < 8; ++c) { wb.p[0][c] = p[(32 + 2 * c) * 64]; wb.p[1][c] = p[(33 + 2 * c) * 64]; }
#pragma unroll
        for (int c = 0; c < 8; ++c) { ws.p[0][c] = p[(48 + 2 * c) * 64]; ws.p[1][c] = p[(49 + 2 * c) * 64]; }
#pragma unroll
        for (int c = 0; c < 8; ++c) { wr.p[0][c] = p[(64 + 2 * c) * 64]; wr.p[1][c] = p[(65 + 2 * c) * 64]; }
    }
    unsigned acc = 0;
    if (variant) {
#pragma unroll
        for (int c = 0; c < 16; ++c) acc ^= wa.p[0][c][0] ^ wa.p[1][c][3] ^ wa.p[0][c][2] ^ wa.p[1][c][1];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc ^= wb.p[0][c][0] ^ wb.p[1][c][3] ^ ws.p[0][c][1] ^ ws.p[1][c][2] ^ wr.p[0][c][3] ^ wr.p[1][c][0];
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (acc == 0x12345678u || m == 1234.5f) sink[t] = 1.f;
    if (t == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 18;
    const int layers = 10;                                          // distinct weight sets, as ten processor blocks have
    const size_t wfloats = 128 * 640;
    std::vector<float> hw(wfloats * layers);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.001f * (float)((i * 2654435761u) % 2001) - 1.f;
    float *W, *sink; u32x4* packed; unsigned long long* cyc;
    (void)hipMalloc(&W, hw.size() * 4); (void)hipMalloc(&sink, 1024); (void)hipMalloc(&cyc, 64);
    (void)hipMalloc(&packed, (size_t)layers * 4 * 80 * 64 * 16);
    (void)hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemset(packed, 1, (size_t)layers * 4 * 80 * 64 * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int variant = 0; variant < 4; ++variant) {
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            for (int it = 0; it < 200; ++it) {
                const int l = it % layers;
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, W + l * wfloats, packed + (size_t)l * 4 * 80 * 64, variant, sink, cyc);
            }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            if (rep == 2) printf("blocks %d variant %d: %.2f us per launch, %llu s_memtime ticks in-kernel (100 MHz)\n", blocks, variant, ms * 5.f, c);
        }
    }
    return 0;
}
