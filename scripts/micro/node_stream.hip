// Memory floor of the node kernel's access pattern: one 256-thread workgroup per CU walks 32-row tiles of 512-byte rows,
// reads a tile of x and of aggr (eight 16-byte loads per thread in flight, `depth` tiles ahead), writes three tiles
// (x_out, ps, pr).  No arithmetic beyond an add: the time of 1M rows is what the memory system allows a kernel that
// keeps that little in flight per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(256, 1) void k_node(const f4* __restrict__ x, const f4* __restrict__ ag, f4* __restrict__ xo,
                                                  f4* __restrict__ ps, f4* __restrict__ pr, int ntiles) {
    const int t = threadIdx.x, srow = t >> 5, sq = t & 31;
    const int t0 = (int)((long long)blockIdx.x * ntiles / gridDim.x), t1 = (int)((long long)(blockIdx.x + 1) * ntiles / gridDim.x);
    if (t0 >= t1) return;
    f4 px[DEPTH][4], pa[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const int tile = min(t0 + d, t1 - 1);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const size_t o = ((size_t)tile * 32 + srow + 8 * it) * 32 + sq;
            px[d][it] = x[o]; pa[d][it] = ag[o];
        }
    }
    for (int tile = t0; tile < t1; tile += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int cur = tile + d;
            if (cur >= t1) break;
            f4 v[4], u[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) { v[it] = px[d][it] + pa[d][it]; u[it] = px[d][it] - pa[d][it]; }
            const int nxt = min(cur + DEPTH, t1 - 1);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const size_t o = ((size_t)nxt * 32 + srow + 8 * it) * 32 + sq;
                px[d][it] = x[o]; pa[d][it] = ag[o];
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const size_t o = ((size_t)cur * 32 + srow + 8 * it) * 32 + sq;
                xo[o] = v[it]; ps[o] = u[it]; pr[o] = v[it] * 2.0f;
            }
        }
    }
}

int main() {
    const int rows = 1000000 / 32 * 32, ntiles = rows / 32;       // whole tiles only
    const size_t bytes = (size_t)rows * 512;
    f4 *x, *ag, *xo, *ps, *pr;
    hipMalloc(&x, bytes); hipMalloc(&ag, bytes); hipMalloc(&xo, bytes); hipMalloc(&ps, bytes); hipMalloc(&pr, bytes);
    hipMemset(x, 0, bytes); hipMemset(ag, 0, bytes); hipMemset(xo, 0, bytes); hipMemset(ps, 0, bytes); hipMemset(pr, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256, 512, 1024, 4096}) {
        for (int depth = 1; depth <= 3; ++depth) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0, 0);
                if (depth == 1) hipLaunchKernelGGL(k_node<1>, dim3(grid), dim3(256), 0, 0, x, ag, xo, ps, pr, ntiles);
                if (depth == 2) hipLaunchKernelGGL(k_node<2>, dim3(grid), dim3(256), 0, 0, x, ag, xo, ps, pr, ntiles);
                if (depth == 3) hipLaunchKernelGGL(k_node<3>, dim3(grid), dim3(256), 0, 0, x, ag, xo, ps, pr, ntiles);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("grid %5d  prefetch depth %d tiles: %7.3f ms  %5.2f TB/s\n", grid, depth, best, 5.0 * bytes / best * 1e-9);
        }
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("error\n"); return 1; }
    return 0;
}
