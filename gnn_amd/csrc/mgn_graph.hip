// Device-side edge build of libmgn_hip.so: triangle cells -> sorted unique directed edges, and any
// directed edge list -> receiver-major CSR, for meshes that already live in HBM.  Integer work only,
// results identical to the host routines in mgn_host.cpp.  Key generation, splitting and the CSR
// scatter are kernels of this file; the 64-bit / 32-bit key sorts and the duplicate removal are
// hipCUB device primitives running in caller-provided workspace.
//
// Unlike the compute entry points these two return sizes to the host, so they synchronise the stream
// (twice and once); they run once per mesh.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "mgn_device.h"

namespace {

__global__ void cell_range_kernel(const long long* __restrict__ cells, long long n, long long* __restrict__ minmax) {
    long long lo = LLONG_MAX, hi = LLONG_MIN;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long v = cells[i];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const long long ol = __shfl_xor(lo, m), oh = __shfl_xor(hi, m);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&minmax[0], lo);
        atomicMax(&minmax[1], hi);
    }
}

// six directed edges per triangle as keys sender * N + receiver (core/datasetclass.py:12-19)
__global__ void cell_keys_kernel(const long long* __restrict__ cells, long long n_cells, unsigned long long N,
                                 unsigned long long* __restrict__ keys) {
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < n_cells; c += (long long)gridDim.x * blockDim.x) {
        const unsigned long long v0 = cells[3 * c], v1 = cells[3 * c + 1], v2 = cells[3 * c + 2];
        unsigned long long* k = keys + 6 * c;
        k[0] = v0 * N + v1; k[1] = v1 * N + v0;
        k[2] = v1 * N + v2; k[3] = v2 * N + v1;
        k[4] = v2 * N + v0; k[5] = v0 * N + v2;
    }
}

__global__ void split_keys_kernel(const unsigned long long* __restrict__ keys, long long n, unsigned long long N,
                                  long long* __restrict__ sender, long long* __restrict__ recv) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        sender[i] = (long long)(k / N);
        recv[i] = (long long)(k % N);
    }
}

// receiver ids as 32-bit sort keys, edge positions as values; flags ids outside [0, N)
__global__ void csr_keys_kernel(const long long* __restrict__ sender, const long long* __restrict__ recv, long long n,
                                long long N, unsigned* __restrict__ keys, unsigned* __restrict__ vals,
                                int* __restrict__ bad) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long s = sender[i], r = recv[i];
        if (s < 0 || s >= N || r < 0 || r >= N) { atomicMax(bad, 1); keys[i] = 0; }
        else keys[i] = (unsigned)r;
        vals[i] = (unsigned)i;
    }
}

__global__ void csr_fill_kernel(const unsigned* __restrict__ keys, const unsigned* __restrict__ vals,
                                const long long* __restrict__ sender, long long n, int* __restrict__ csr_sender,
                                int* __restrict__ csr_recv, long long* __restrict__ perm) {
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (long long)gridDim.x * blockDim.x) {
        const unsigned p = vals[j];
        csr_recv[j] = (int)keys[j];
        csr_sender[j] = (int)sender[p];
        if (perm) perm[j] = (long long)p;
    }
}

// rowptr[v] = number of edges whose receiver is < v = first position of v in the sorted receivers
__global__ void csr_rowptr_kernel(const unsigned* __restrict__ keys, long long n_edges, long long n_nodes,
                                  int* __restrict__ rowptr) {
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v <= n_nodes; v += (long long)gridDim.x * blockDim.x) {
        long long lo = 0, hi = n_edges;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if ((long long)keys[mid] < v) lo = mid + 1; else hi = mid;
        }
        rowptr[v] = (int)lo;
    }
}

int bits_for(unsigned long long max_value) {
    int b = 1;
    while (b < 64 && (max_value >> b) != 0) ++b;
    return b;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

int grid_1d(long long n) { return (int)std::min<long long>((n + NT - 1) / NT, 16384); }

#define HIP_TRY(expr, name)                                                                  \
    do {                                                                                     \
        const hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) { mgn_set_error("%s: %s", name, hipGetErrorString(e_)); return (int)e_; } \
    } while (0)

size_t sort64_temp_bytes(long long n) {
    size_t b = 0;
    hipcub::DoubleBuffer<unsigned long long> d(nullptr, nullptr);
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, b, d, (int)n, 0, 64, (hipStream_t)0);
    return b;
}

size_t unique_temp_bytes(long long n) {
    size_t b = 0;
    (void)hipcub::DeviceSelect::Unique(nullptr, b, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                 (long long*)nullptr, (int)n, (hipStream_t)0);
    return b;
}

size_t sort32_temp_bytes(long long n) {
    size_t b = 0;
    hipcub::DoubleBuffer<unsigned> k(nullptr, nullptr), v(nullptr, nullptr);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, k, v, (int)n, 0, 32, (hipStream_t)0);
    return b;
}

}  // namespace

extern "C" int64_t mgn_edge_build_workspace_bytes(int64_t n_cells) {
    if (n_cells < 0 || 6 * n_cells >= (1ll << 31)) return -1;
    const long long n = 6 * std::max<int64_t>(n_cells, 1);
    return (int64_t)(256 + 2 * align256((size_t)n * 8) + align256(std::max(sort64_temp_bytes(n), unique_temp_bytes(n))));
}

extern "C" int mgn_cells_to_edge_index_dev(const int64_t* cells, int64_t n_cells, void* workspace, int64_t ws_bytes,
                                           int64_t* out_sender, int64_t* out_recv, int64_t cap, int64_t* out_n_edges,
                                           int64_t* out_n_nodes, void* stream) {
    const char* name = "mgn_cells_to_edge_index_dev";
    MGN_REQUIRE(n_cells >= 0 && out_n_edges && out_n_nodes && (n_cells == 0 || cells), name);
    *out_n_edges = 0;
    *out_n_nodes = 0;
    if (n_cells == 0) return 0;
    const int64_t need = mgn_edge_build_workspace_bytes(n_cells);
    MGN_REQUIRE(need > 0 && workspace && ws_bytes >= need, name);
    hipStream_t s = (hipStream_t)stream;
    const long long nk = 6 * n_cells;
    char* base = (char*)workspace;
    long long* scal = (long long*)base;                       // [0] min id, [1] max id, [2] number of unique keys
    unsigned long long* k0 = (unsigned long long*)(base + 256);
    unsigned long long* k1 = (unsigned long long*)(base + 256 + align256((size_t)nk * 8));
    void* temp = base + 256 + 2 * align256((size_t)nk * 8);
    size_t temp_bytes = (size_t)ws_bytes - (256 + 2 * align256((size_t)nk * 8));

    const long long init[3] = {LLONG_MAX, LLONG_MIN, 0};
    long long host[3];
    HIP_TRY(hipMemcpyAsync(scal, init, sizeof(init), hipMemcpyHostToDevice, s), name);
    hipLaunchKernelGGL(cell_range_kernel, dim3(grid_1d(3 * n_cells)), dim3(NT), 0, s, (const long long*)cells,
                       (long long)(3 * n_cells), scal);
    HIP_TRY(hipMemcpyAsync(host, scal, 2 * sizeof(long long), hipMemcpyDeviceToHost, s), name);
    HIP_TRY(hipStreamSynchronize(s), name);
    if (host[0] < 0) { mgn_set_error("%s: negative node id %lld", name, host[0]); return MGN_E_RANGE; }
    const unsigned long long N = (unsigned long long)host[1] + 1;
    if (N > (1ull << 31)) { mgn_set_error("%s: node id %lld too large", name, host[1]); return MGN_E_RANGE; }

    hipLaunchKernelGGL(cell_keys_kernel, dim3(grid_1d(n_cells)), dim3(NT), 0, s, (const long long*)cells,
                       (long long)n_cells, N, k0);
    hipcub::DoubleBuffer<unsigned long long> keys(k0, k1);
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, keys, (int)nk, 0, bits_for(N * N - 1), s), name);
    unsigned long long* sorted = keys.Current();
    unsigned long long* uniq = keys.Alternate();
    HIP_TRY(hipcub::DeviceSelect::Unique(temp, temp_bytes, (const unsigned long long*)sorted, uniq, scal + 2, (int)nk, s), name);
    HIP_TRY(hipMemcpyAsync(host + 2, scal + 2, sizeof(long long), hipMemcpyDeviceToHost, s), name);
    HIP_TRY(hipStreamSynchronize(s), name);
    const long long ne = host[2];
    *out_n_edges = ne;
    *out_n_nodes = (int64_t)N;
    if (ne > cap || !out_sender || !out_recv) {
        mgn_set_error("%s: capacity %lld < %lld edges", name, (long long)cap, ne);
        return MGN_E_CAP;
    }
    hipLaunchKernelGGL(split_keys_kernel, dim3(grid_1d(ne)), dim3(NT), 0, s, (const unsigned long long*)uniq, ne, N,
                       (long long*)out_sender, (long long*)out_recv);
    return check_launch(name);
}

extern "C" int64_t mgn_receiver_csr_workspace_bytes(int64_t n_edges) {
    if (n_edges < 0 || n_edges >= (1ll << 31) - 64) return -1;
    const long long n = std::max<int64_t>(n_edges, 1);
    return (int64_t)(256 + 4 * align256((size_t)n * 4) + align256(sort32_temp_bytes(n)));
}

extern "C" int mgn_receiver_csr_dev(const int64_t* sender, const int64_t* recv, int64_t n_edges, int64_t n_nodes,
                                    void* workspace, int64_t ws_bytes, int32_t* rowptr, int32_t* csr_sender,
                                    int32_t* csr_recv, int64_t* perm, void* stream) {
    const char* name = "mgn_receiver_csr_dev";
    MGN_REQUIRE(n_edges >= 0 && n_nodes >= 0 && rowptr, name);
    MGN_REQUIRE(n_edges == 0 || (sender && recv && csr_sender && csr_recv), name);
    if (n_nodes >= (1ll << 31) - 64 || n_edges >= (1ll << 31) - 64) {
        mgn_set_error("%s: sizes exceed the int32 index range", name);
        return MGN_E_RANGE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (n_edges == 0) {
        HIP_TRY(hipMemsetAsync(rowptr, 0, (size_t)(n_nodes + 1) * sizeof(int32_t), s), name);
        return 0;
    }
    const int64_t need = mgn_receiver_csr_workspace_bytes(n_edges);
    MGN_REQUIRE(need > 0 && workspace && ws_bytes >= need, name);
    char* base = (char*)workspace;
    int* bad = (int*)base;
    const size_t arr = align256((size_t)n_edges * 4);
    unsigned* k0 = (unsigned*)(base + 256);
    unsigned* k1 = (unsigned*)(base + 256 + arr);
    unsigned* v0 = (unsigned*)(base + 256 + 2 * arr);
    unsigned* v1 = (unsigned*)(base + 256 + 3 * arr);
    void* temp = base + 256 + 4 * arr;
    size_t temp_bytes = (size_t)ws_bytes - (256 + 4 * arr);

    HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int), s), name);
    hipLaunchKernelGGL(csr_keys_kernel, dim3(grid_1d(n_edges)), dim3(NT), 0, s, (const long long*)sender,
                       (const long long*)recv, (long long)n_edges, (long long)n_nodes, k0, v0, bad);
    hipcub::DoubleBuffer<unsigned> keys(k0, k1), vals(v0, v1);
    // a stable sort by receiver: the edges of one receiver keep their input order
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, vals, (int)n_edges, 0,
                                               bits_for((unsigned long long)std::max<int64_t>(n_nodes - 1, 1)), s), name);
    hipLaunchKernelGGL(csr_fill_kernel, dim3(grid_1d(n_edges)), dim3(NT), 0, s, (const unsigned*)keys.Current(),
                       (const unsigned*)vals.Current(), (const long long*)sender, (long long)n_edges, csr_sender, csr_recv,
                       (long long*)perm);
    hipLaunchKernelGGL(csr_rowptr_kernel, dim3(grid_1d(n_nodes + 1)), dim3(NT), 0, s, (const unsigned*)keys.Current(),
                       (long long)n_edges, (long long)n_nodes, rowptr);
    int host_bad = 0;
    HIP_TRY(hipMemcpyAsync(&host_bad, bad, sizeof(int), hipMemcpyDeviceToHost, s), name);
    HIP_TRY(hipStreamSynchronize(s), name);
    if (host_bad) {
        mgn_set_error("%s: an edge has an endpoint outside [0, %lld)", name, (long long)n_nodes);
        return MGN_E_RANGE;
    }
    return check_launch(name);
}
