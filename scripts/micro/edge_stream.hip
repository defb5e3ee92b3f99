// Memory floor of the edge kernel's access pattern on the 1000 x 1000-node mesh (5 992 002 edges in receiver-major order):
// 256-thread workgroups, `PER_CU` per compute unit, walk contiguous receiver-aligned ranges of 32-edge tiles exactly as
// mp_edge_h4_kernel does -- per tile 4 x 16 bytes per thread of streamed edge rows (non-temporal, one tile ahead), the
// sender / receiver ids, the gathers of ps[sender] and pr[receiver] in the accumulator layout (8 x 16 bytes per thread,
// lane (row, half) of wave w reads columns 32w + 8g + 4h), the updated rows written back (non-temporal) and one 512-byte
// row of sums per receiver -- but no matrix work, no LDS tile, no LayerNorm, no scan: a handful of adds keeps the loads
// alive.  Its time is what the memory system allows a kernel with this traffic shape and this much in flight.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include scripts/micro/edge_stream.hip -L gnn_amd -lmgn_hip \
//         -Wl,-rpath,'$ORIGIN/../../gnn_amd' -o scripts/micro/edge_stream
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "mgn_hip.h"

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int H = 128, TM = 32, NT = 256;

template <bool GATHER, bool STORE_E>
__global__ __launch_bounds__(NT, 2) void k_edge(const float* e_in, const float* __restrict__ ps,
                                                const float* __restrict__ pr, const int* __restrict__ rowptr,
                                                const int* __restrict__ sender, const int* __restrict__ recv,
                                                float* e_out, float* __restrict__ aggr, int n_nodes, int n_edges) {
    __shared__ int srange[2];
    __shared__ float part[4][TM];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
    if (t == 0) {                                    // receiver-aligned edge range of this workgroup
        const long long lo = (long long)blockIdx.x * n_edges / gridDim.x, hi = (long long)(blockIdx.x + 1) * n_edges / gridDim.x;
        int a = 0, b = n_nodes;
        while (a < b) { const int m = (a + b) >> 1; if (rowptr[m] < lo) a = m + 1; else b = m; }
        srange[0] = rowptr[a];
        a = 0; b = n_nodes;
        while (a < b) { const int m = (a + b) >> 1; if (rowptr[m] < hi) a = m + 1; else b = m; }
        srange[1] = rowptr[a];
    }
    __syncthreads();
    const int e0 = srange[0], e1 = srange[1];
    if (e0 >= e1) return;
    const int last = e1 - e0 - 1;
    const float* ein_g = e_in + (size_t)e0 * H;
    float* eout_g = e_out + (size_t)e0 * H;
    const int srow = t >> 5, sq = t & 31, sc = t & (H - 1), sg = t >> 7;
    f4 cur[4], pre[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) cur[it] = *(const f4*)(ein_g + (unsigned)(min(srow + 8 * it, last) * H + 4 * sq));
    int s_cur = sender[e0 + min(j, last)], r_cur = recv[e0 + min(j, last)];
    for (int rel = 0; rel <= last; rel += TM) {
        const int nvalid = min(TM, last + 1 - rel);
        f4 gs[4], gr[4];
        if (GATHER) {
            const unsigned so = (unsigned)s_cur * H + 32 * w + 4 * h, ro = (unsigned)r_cur * H + 32 * w + 4 * h;
#pragma unroll
            for (int c = 0; c < 4; ++c) { gs[c] = *(const f4*)(ps + so + 8 * c); gr[c] = *(const f4*)(pr + ro + 8 * c); }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
            pre[it] = __builtin_nontemporal_load((const f4*)(ein_g + (unsigned)(min(rel + TM + srow + 8 * it, last) * H + 4 * sq)));
        const int enx = min(rel + TM + j, last);
        const int s_nxt = sender[e0 + enx], r_nxt = recv[e0 + enx];
        const int rb = (rel + j + 1 <= last) ? recv[e0 + min(rel + j + 1, last)] : -1;
        float acc = 0.f;
        if (GATHER) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc += (gs[c][0] + gs[c][1]) + (gs[c][2] + gs[c][3]) + (gr[c][0] + gr[c][1]) + (gr[c][2] + gr[c][3]);
        }
        part[w][j] = acc;                            // one value per tile row and wave: stands for the hidden tile
        const unsigned mask = (unsigned)__ballot(j < nvalid && r_cur != rb);    // bit r: row r ends its receiver's segment
        __syncthreads();
        const float add = part[0][srow] + part[1][srow] + part[2][srow] + part[3][srow];
        if (STORE_E) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = srow + 8 * it;
                if (row < nvalid) __builtin_nontemporal_store(cur[it] + add, (f4*)(eout_g + (unsigned)((rel + row) * H + 4 * sq)));
            }
        }
        const unsigned gmask = (mask >> (16 * sg)) & 0xFFFFu;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if ((gmask >> r) & 1u) {
                const int node = __builtin_amdgcn_readlane(r_cur, 16 * sg + r);
                aggr[(unsigned)(node * H + sc)] = add + (float)r;
            }
        }
        __syncthreads();
        s_cur = s_nxt; r_cur = r_nxt;
#pragma unroll
        for (int it = 0; it < 4; ++it) cur[it] = pre[it];
    }
}

#define CHECK(x) do { if ((x) != hipSuccess) { std::printf("HIP error at %s\n", #x); return 1; } } while (0)

int main() {
    const int side = 1000;
    std::vector<int64_t> cells;
    for (int y = 0; y + 1 < side; ++y)
        for (int x = 0; x + 1 < side; ++x) {
            const int64_t v0 = (int64_t)y * side + x, v1 = v0 + 1, v2 = v0 + side, v3 = v2 + 1;
            const int64_t tri[6] = {v0, v1, v3, v0, v3, v2};
            cells.insert(cells.end(), tri, tri + 6);
        }
    const int64_t n_cells = (int64_t)cells.size() / 3;
    std::vector<int64_t> snd(6 * n_cells), rcv(6 * n_cells);
    int64_t E = 0, N = 0;
    if (mgn_cells_to_edge_index(cells.data(), n_cells, snd.data(), rcv.data(), 6 * n_cells, &E, &N)) return 1;
    std::vector<int32_t> rowptr(N + 1), cs(E), cr(E);
    std::vector<int64_t> perm(E);
    if (mgn_receiver_csr(snd.data(), rcv.data(), E, N, rowptr.data(), cs.data(), cr.data(), perm.data())) return 1;
    std::printf("mesh %d x %d: %lld nodes, %lld edges\n", side, side, (long long)N, (long long)E);
    float *e_in, *e_out, *ps, *pr, *aggr; int *d_rowptr, *d_s, *d_r;
    CHECK(hipMalloc(&e_in, (size_t)E * H * 4)); CHECK(hipMalloc(&e_out, (size_t)E * H * 4));
    CHECK(hipMalloc(&ps, (size_t)N * H * 4)); CHECK(hipMalloc(&pr, (size_t)N * H * 4)); CHECK(hipMalloc(&aggr, (size_t)N * H * 4));
    CHECK(hipMalloc(&d_rowptr, (N + 1) * 4)); CHECK(hipMalloc(&d_s, E * 4)); CHECK(hipMalloc(&d_r, E * 4));
    CHECK(hipMemset(e_in, 0, (size_t)E * H * 4)); CHECK(hipMemset(ps, 0, (size_t)N * H * 4)); CHECK(hipMemset(pr, 0, (size_t)N * H * 4));
    CHECK(hipMemcpy(d_rowptr, rowptr.data(), (N + 1) * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_s, cs.data(), E * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_r, cr.data(), E * 4, hipMemcpyHostToDevice));
    hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    const double alg = 1032.0 * E, moved = 2.0 * E * 512 + 3.0 * N * 512 + 8.0 * E;
    for (int variant = 0; variant < 4; ++variant) {
        for (int grid : {512, 1024, 2048}) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CHECK(hipEventRecord(t0, 0));
                if (variant == 0) hipLaunchKernelGGL((k_edge<true, true>), dim3(grid), dim3(NT), 0, 0, e_in, ps, pr, d_rowptr, d_s, d_r, e_out, aggr, (int)N, (int)E);
                if (variant == 1) hipLaunchKernelGGL((k_edge<false, true>), dim3(grid), dim3(NT), 0, 0, e_in, ps, pr, d_rowptr, d_s, d_r, e_out, aggr, (int)N, (int)E);
                if (variant == 2) hipLaunchKernelGGL((k_edge<true, false>), dim3(grid), dim3(NT), 0, 0, e_in, ps, pr, d_rowptr, d_s, d_r, e_out, aggr, (int)N, (int)E);
                if (variant == 3) hipLaunchKernelGGL((k_edge<true, true>), dim3(grid), dim3(NT), 0, 0, e_in, ps, pr, d_rowptr, d_s, d_r, e_in, aggr, (int)N, (int)E);
                CHECK(hipEventRecord(t1, 0)); CHECK(hipEventSynchronize(t1));
                float ms; CHECK(hipEventElapsedTime(&ms, t0, t1));
                if (ms < best) best = ms;
            }
            const char* names[] = {"full pattern", "without the gathers", "without the edge-row stores", "full pattern, rows in place"};
            std::printf("%-28s grid %5d: %7.3f ms  (algorithmic 1032 B/edge: %5.2f TB/s; compulsory bytes of the full pattern: %5.2f TB/s)\n",
                        names[variant], grid, best, alg / best * 1e-9, moved / best * 1e-9);
        }
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}
