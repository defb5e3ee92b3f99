// Coarse graph of the multiscale option on the device (core/meshgraphnet.py:237-283): farthest-point sampling of the
// mesh nodes and the k-nearest-neighbour graph over the sample, for one mesh or for a block-diagonal batch of meshes
// (every block sampled and connected on its own).  Index work: the results are node ids, reproduced exactly.
//
// Sampling follows the reference's torch expressions value for value (:243-254): the distance of a point to the last
// sample is torch.norm(points - last, dim=-1) in float32 -- on the CPU that is a fused multiply-add chain over the
// coordinates, then a correctly rounded square root --, the running distance is their minimum, and the next sample is
// torch.argmax of it, which returns the lowest index among equal values.  Regular meshes are full of equal
// distances, so both the rounding and the tie rule decide which nodes are sampled.
//
// knn_graph comes from torch-cluster in the reference (:268), which is not part of its tree: restated as "the k nearest
// other points by exact float64 squared distance, nearest first, equal distances by lower index, edges
// (neighbour -> point) grouped by point".
#include <hip/hip_runtime.h>

#include "mgn_device.h"

namespace {

constexpr int FPS_T = 1024;                 // one workgroup of 16 waves per block of the batch (4 waves for blocks of up to 1024 points)
constexpr int FPS_MAX_R = 16;               // up to 16 points per thread are kept in registers

__device__ inline void keep_better(float& bv, int& bi, float v, int i) {
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

template <int DIM>
__device__ inline float fps_distance(const float* p, const float* q) {
    const float d0 = p[0] - q[0];
    float acc = d0 * d0;
#pragma unroll
    for (int k = 1; k < DIM; ++k) {
        const float d = p[k] - q[k];
        acc = __builtin_fmaf(d, d, acc);
    }
    return __builtin_sqrtf(acc);            // correctly rounded (the compiler's default for HIP)
}

// R > 0: block sizes up to R * FPS_T, points and running distances in registers.  R == 0: any size, running distances
// in `dist` (one float per point of the batch), points re-read every round (they stay in L2 up to a few hundred
// thousand points).  One barrier per sample: the wave winners go through two alternating LDS slots.
template <int DIM, int R, int T>
__global__ __launch_bounds__(T) void fps_kernel(const float* __restrict__ pts, const int* __restrict__ node_ptr,
                                                     const int* __restrict__ samp_ptr, const int* __restrict__ starts,
                                                     float* __restrict__ dist, long long* __restrict__ out) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n0 = node_ptr[b], n = node_ptr[b + 1] - n0;
    const int m0 = samp_ptr[b], m = samp_ptr[b + 1] - m0;
    if (m <= 0) return;
    int cur = starts[b];
    if (n <= 0 || cur < 0 || cur >= n) {            // nothing to sample from, or a first sample outside the block
        for (int i = tid; i < m; i += T) out[m0 + i] = -1;
        return;
    }
    const float* P = pts + (size_t)n0 * DIM;
    float* D = dist + n0;
    __shared__ __align__(16) float s_v[2][(T / 64)];
    __shared__ __align__(16) int s_i[2][(T / 64)];
    // blocks of up to 4096 points also sit in LDS: the winner's coordinates are then an LDS read away instead of a
    // dependent round trip to L2 (the larger part of a round on a mesh of a few hundred nodes)
    constexpr int LDS_PTS = (R > 0 && R <= 4) ? R * T : 1;
    __shared__ float s_p[LDS_PTS * DIM];
    constexpr int RR = R > 0 ? R : 1;
    float pr[RR][DIM], dr[RR];
    const float inf = __builtin_inff();
    if (R > 0) {
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            const int j = tid + r * T;
            dr[r] = j < n ? inf : -1.f;              // a slot without a point never wins (distances are >= 0)
#pragma unroll
            for (int k = 0; k < DIM; ++k) pr[r][k] = j < n ? P[(size_t)j * DIM + k] : 0.f;
            if (LDS_PTS > 1 && j < n) {
#pragma unroll
                for (int k = 0; k < DIM; ++k) s_p[j * DIM + k] = pr[r][k];
            }
        }
        if (LDS_PTS > 1) __syncthreads();
    } else {
        for (int j = tid; j < n; j += T) D[j] = inf;
    }
    if (tid == 0) out[m0] = (long long)n0 + cur;
    for (int it = 1; it < m; ++it) {
        float q[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) q[k] = LDS_PTS > 1 ? s_p[cur * DIM + k] : P[(size_t)cur * DIM + k];
        float bv = -1.f;
        int bi = 0x7fffffff;
        if (R > 0) {
#pragma unroll
            for (int r = 0; r < RR; ++r) {
                const float d = fps_distance<DIM>(pr[r], q);
                const float nd = dr[r] < 0.f ? -1.f : (d < dr[r] ? d : dr[r]);
                dr[r] = nd;
                if (nd > bv) { bv = nd; bi = tid + r * T; }      // ascending ids per thread: the first maximum stays
            }
        } else {
            constexpr int U = 8;                       // loads of eight points in flight per thread
            for (int j0 = tid; j0 < n; j0 += U * T) {
                float p[U][DIM], old[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = j0 + u * T;
                    old[u] = j < n ? D[j] : -1.f;
#pragma unroll
                    for (int k = 0; k < DIM; ++k) p[u][k] = j < n ? P[(size_t)j * DIM + k] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = j0 + u * T;
                    const float d = fps_distance<DIM>(p[u], q);
                    const float nd = d < old[u] ? d : old[u];
                    if (j < n && d < old[u]) D[j] = d;
                    if (nd > bv) { bv = nd; bi = j; }      // ascending ids per thread; a slot past the end holds -1
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            keep_better(bv, bi, ov, oi);
        }
        const int slot = it & 1;
        if ((tid & 63) == 0) { s_v[slot][tid >> 6] = bv; s_i[slot][tid >> 6] = bi; }
        __syncthreads();
        // every thread folds the 16 wave winners itself: no second barrier, and the slot written two rounds later
        // cannot be overwritten before the barrier in between has been passed by every reader
        const float4* sv4 = reinterpret_cast<const float4*>(s_v[slot]);
        const int4* si4 = reinterpret_cast<const int4*>(s_i[slot]);
        bv = -1.f; bi = 0x7fffffff;
#pragma unroll
        for (int w = 0; w < (T / 64) / 4; ++w) {
            const float4 v = sv4[w];
            const int4 i = si4[w];
            keep_better(bv, bi, v.x, i.x); keep_better(bv, bi, v.y, i.y);
            keep_better(bv, bi, v.z, i.z); keep_better(bv, bi, v.w, i.w);
        }
        cur = bi;
        if (tid == 0) out[m0 + it] = (long long)n0 + cur;
    }
}

template <int DIM, int R, int T>
void launch_fps(int n_blocks, hipStream_t s, const float* pts, const int* node_ptr, const int* samp_ptr, const int* starts,
                float* dist, long long* out) {
    hipLaunchKernelGGL((fps_kernel<DIM, R, T>), dim3(n_blocks), dim3(T), 0, s, pts, node_ptr, samp_ptr, starts, dist, out);
}

template <int DIM>
void dispatch_fps(long long max_block, int n_blocks, hipStream_t s, const float* pts, const int* node_ptr,
                  const int* samp_ptr, const int* starts, float* dist, long long* out) {
    // meshes of a few hundred nodes: four waves (a cheap barrier, one wave per SIMD), four points per thread
    if (max_block <= 4 * 256) launch_fps<DIM, 4, 256>(n_blocks, s, pts, node_ptr, samp_ptr, starts, dist, out);
    else if (max_block <= 4 * FPS_T) launch_fps<DIM, 4, FPS_T>(n_blocks, s, pts, node_ptr, samp_ptr, starts, dist, out);
    else if (max_block <= FPS_MAX_R * FPS_T) launch_fps<DIM, FPS_MAX_R, FPS_T>(n_blocks, s, pts, node_ptr, samp_ptr, starts, dist, out);
    else launch_fps<DIM, 0, FPS_T>(n_blocks, s, pts, node_ptr, samp_ptr, starts, dist, out);
}

// ---- one mesh too large for one compute unit to sweep per sample: two launches per sample over the whole GPU.
// fps_round_kernel: every workgroup updates the running distances of its contiguous slice against the last sample
// (read from *cur) and leaves its slice's best (distance, index) in part[]; fps_pick_kernel: one workgroup folds the
// slices' winners (lowest index among equal distances, as above), writes the sample and *cur for the next round.
// The same values and the same choices as fps_kernel; the launches of a stream run in order, nothing waits in a kernel.
constexpr int FPSR_T = 256;
constexpr int FPSR_MAX_G = 1024;

template <int DIM>
__global__ __launch_bounds__(FPSR_T) void fps_round_kernel(const float* __restrict__ P, int n, const int* __restrict__ cur,
                                                           float* __restrict__ D, float* __restrict__ part_v,
                                                           int* __restrict__ part_i, int first_round) {
    const int tid = threadIdx.x;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int j0 = blockIdx.x * per, j1 = min(n, j0 + per);
    const int c = *cur;
    float q[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) q[k] = P[(size_t)c * DIM + k];
    float bv = -1.f;
    int bi = 0x7fffffff;
    for (int j = j0 + tid; j < j1; j += FPSR_T) {
        float p[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) p[k] = P[(size_t)j * DIM + k];
        const float d = fps_distance<DIM>(p, q), old = first_round ? __builtin_inff() : D[j];
        const float nd = d < old ? d : old;
        if (first_round || d < old) D[j] = nd;
        if (nd > bv) { bv = nd; bi = j; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        keep_better(bv, bi, ov, oi);
    }
    __shared__ float sv[FPSR_T / 64];
    __shared__ int si[FPSR_T / 64];
    if ((tid & 63) == 0) { sv[tid >> 6] = bv; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < FPSR_T / 64; ++w) keep_better(bv, bi, sv[w], si[w]);
        part_v[blockIdx.x] = bv; part_i[blockIdx.x] = bi;
    }
}

__global__ __launch_bounds__(FPSR_T) void fps_pick_kernel(const float* __restrict__ part_v, const int* __restrict__ part_i,
                                                          int g, int* __restrict__ cur, long long* __restrict__ out_slot,
                                                          long long base) {
    const int tid = threadIdx.x;
    float bv = -1.f;
    int bi = 0x7fffffff;
    for (int b = tid; b < g; b += FPSR_T) keep_better(bv, bi, part_v[b], part_i[b]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        keep_better(bv, bi, ov, oi);
    }
    __shared__ float sv[FPSR_T / 64];
    __shared__ int si[FPSR_T / 64];
    if ((tid & 63) == 0) { sv[tid >> 6] = bv; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < FPSR_T / 64; ++w) keep_better(bv, bi, sv[w], si[w]);
        *cur = bi;
        *out_slot = base + bi;
    }
}

__global__ void fps_first_kernel(int start, int* __restrict__ cur, long long* __restrict__ out_slot, long long base) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { *cur = start; *out_slot = base + start; }
}

template <int DIM>
void fps_by_rounds(const float* P, int n, int m, int start, long long base, float* D, float* extra, long long* out, hipStream_t s) {
    float* part_v = extra;
    int* part_i = reinterpret_cast<int*>(extra + FPSR_MAX_G);
    int* cur = reinterpret_cast<int*>(extra + 2 * FPSR_MAX_G);
    int g = (n + 4 * FPSR_T - 1) / (4 * FPSR_T);
    g = g > FPSR_MAX_G ? FPSR_MAX_G : (g < 1 ? 1 : g);
    hipLaunchKernelGGL(fps_first_kernel, dim3(1), dim3(64), 0, s, start, cur, out, base);
    for (int it = 1; it < m; ++it) {
        hipLaunchKernelGGL((fps_round_kernel<DIM>), dim3(g), dim3(FPSR_T), 0, s, P, n, (const int*)cur, D, part_v, part_i,
                           it == 1 ? 1 : 0);
        hipLaunchKernelGGL(fps_pick_kernel, dim3(1), dim3(FPSR_T), 0, s, (const float*)part_v, (const int*)part_i, g, cur,
                           out + it, base);
    }
}

constexpr int KNN_T = 256;

// One thread per query point; the block's points pass through LDS 256 at a time, in ascending index order, so a
// candidate that only ties the current k-th best never displaces it.  Products and sums are rounded one by one
// (no fused multiply-add): the values numpy computes for the same float64 expression.
template <int DIM, int K>
__global__ __launch_bounds__(KNN_T) void knn_kernel(const float* __restrict__ pts, const long long* __restrict__ first,
                                                     const int* __restrict__ cptr, const long long* __restrict__ eptr,
                                                     long long* __restrict__ out_snd, long long* __restrict__ out_rcv) {
#pragma clang fp contract(off)
    const int b = blockIdx.y, tid = threadIdx.x;
    const int c0 = cptr[b], nb = cptr[b + 1] - c0;
    if ((long long)blockIdx.x * KNN_T >= nb) return;
    const int q = blockIdx.x * KNN_T + tid;
    const bool valid = q < nb;
    double qc[DIM];
    {
        const long long row = valid ? (first ? first[c0 + q] : (long long)(c0 + q)) : 0;
#pragma unroll
        for (int k = 0; k < DIM; ++k) qc[k] = valid ? (double)pts[(size_t)row * DIM + k] : 0.0;
    }
    double bd[K];
    int bi[K];
#pragma unroll
    for (int s = 0; s < K; ++s) { bd[s] = __builtin_inf(); bi[s] = -1; }
    __shared__ double tile[KNN_T][DIM];
    for (int t0 = 0; t0 < nb; t0 += KNN_T) {
        __syncthreads();
        const int j = t0 + tid;
        if (j < nb) {
            const long long row = first ? first[c0 + j] : (long long)(c0 + j);
#pragma unroll
            for (int k = 0; k < DIM; ++k) tile[tid][k] = (double)pts[(size_t)row * DIM + k];
        }
        __syncthreads();
        const int cnt = nb - t0 < KNN_T ? nb - t0 : KNN_T;
        if (valid) {
            for (int jj = 0; jj < cnt; ++jj) {
                double d2 = 0.0;
#pragma unroll
                for (int k = 0; k < DIM; ++k) {
                    const double d = tile[jj][k] - qc[k];
                    const double sq = d * d;
                    d2 = k == 0 ? sq : d2 + sq;
                }
                const int idx = t0 + jj;
                if (idx != q && d2 < bd[K - 1]) {
                    bd[K - 1] = d2; bi[K - 1] = idx;
#pragma unroll
                    for (int s = K - 1; s > 0; --s) {
                        if (bd[s] < bd[s - 1]) {
                            const double td = bd[s]; bd[s] = bd[s - 1]; bd[s - 1] = td;
                            const int ti = bi[s]; bi[s] = bi[s - 1]; bi[s - 1] = ti;
                        }
                    }
                }
            }
        }
    }
    if (!valid) return;
    const int kk = K < nb - 1 ? K : nb - 1;
    const long long base = eptr[b] + (long long)q * kk;
#pragma unroll
    for (int s = 0; s < K; ++s) {
        if (s < kk) { out_snd[base + s] = (long long)c0 + bi[s]; out_rcv[base + s] = (long long)c0 + q; }
    }
}

template <int DIM, int K>
void launch_knn(dim3 grid, hipStream_t s, const float* pts, const long long* first, const int* cptr, const long long* eptr,
                long long* snd, long long* rcv) {
    hipLaunchKernelGGL((knn_kernel<DIM, K>), grid, dim3(KNN_T), 0, s, pts, first, cptr, eptr, snd, rcv);
}

template <int DIM>
bool dispatch_knn(int k, dim3 grid, hipStream_t s, const float* pts, const long long* first, const int* cptr,
                  const long long* eptr, long long* snd, long long* rcv) {
    switch (k) {
        case 1: launch_knn<DIM, 1>(grid, s, pts, first, cptr, eptr, snd, rcv); return true;
        case 2: launch_knn<DIM, 2>(grid, s, pts, first, cptr, eptr, snd, rcv); return true;
        case 3: launch_knn<DIM, 3>(grid, s, pts, first, cptr, eptr, snd, rcv); return true;
        case 4: launch_knn<DIM, 4>(grid, s, pts, first, cptr, eptr, snd, rcv); return true;
        case 6: launch_knn<DIM, 6>(grid, s, pts, first, cptr, eptr, snd, rcv); return true;
        case 8: launch_knn<DIM, 8>(grid, s, pts, first, cptr, eptr, snd, rcv); return true;
        default: return false;
    }
}

}  // namespace

extern "C" int64_t mgn_fps_workspace_floats(int64_t n_points) {
    return n_points < 0 ? -1 : n_points + 2 * FPSR_MAX_G + 64;
}

extern "C" int mgn_fps_single(const float* points, int32_t dim, int64_t n_points, int64_t n_samples, int64_t start,
                              float* workspace, int64_t* out_idx, void* stream) {
    const char* name = "mgn_fps_single";
    MGN_REQUIRE((dim == 2 || dim == 3) && n_points > 0 && n_points < (1ll << 31) - 4096 && n_samples >= 0, name);
    if (n_samples == 0) return 0;
    MGN_REQUIRE(points && workspace && out_idx && start >= 0 && start < n_points && n_samples < (1ll << 31), name);
    hipStream_t s = (hipStream_t)stream;
    if (dim == 2) fps_by_rounds<2>(points, (int)n_points, (int)n_samples, (int)start, 0, workspace, workspace + n_points, (long long*)out_idx, s);
    else fps_by_rounds<3>(points, (int)n_points, (int)n_samples, (int)start, 0, workspace, workspace + n_points, (long long*)out_idx, s);
    return check_launch(name);
}

extern "C" int mgn_fps(const float* points, int32_t dim, const int32_t* node_ptr, const int32_t* sample_ptr,
                       const int32_t* starts, int32_t n_blocks, int64_t max_block_nodes, float* dist_ws, int64_t* out_idx,
                       void* stream) {
    const char* name = "mgn_fps";
    MGN_REQUIRE(n_blocks >= 0 && max_block_nodes >= 0 && (dim == 2 || dim == 3), name);
    if (n_blocks == 0) return 0;
    MGN_REQUIRE(points && node_ptr && sample_ptr && starts && out_idx, name);
    MGN_REQUIRE(max_block_nodes < (1ll << 31) - FPS_T, name);
    MGN_REQUIRE(max_block_nodes <= (int64_t)FPS_MAX_R * FPS_T || dist_ws, name);
    hipStream_t s = (hipStream_t)stream;
    if (dim == 2) dispatch_fps<2>(max_block_nodes, n_blocks, s, points, node_ptr, sample_ptr, starts, dist_ws, (long long*)out_idx);
    else dispatch_fps<3>(max_block_nodes, n_blocks, s, points, node_ptr, sample_ptr, starts, dist_ws, (long long*)out_idx);
    return check_launch(name);
}

extern "C" int mgn_knn_graph(const float* points, int32_t dim, const int64_t* first, const int32_t* coarse_ptr,
                             const int64_t* edge_ptr, int32_t n_blocks, int64_t max_block_points, int32_t k,
                             int64_t* out_sender, int64_t* out_recv, void* stream) {
    const char* name = "mgn_knn_graph";
    MGN_REQUIRE(n_blocks >= 0 && max_block_points >= 0 && (dim == 2 || dim == 3) && k >= 1, name);
    if (n_blocks == 0 || max_block_points == 0) return 0;
    MGN_REQUIRE(points && coarse_ptr && edge_ptr && out_sender && out_recv, name);
    MGN_REQUIRE(max_block_points < (1ll << 31) - KNN_T && n_blocks <= 65535, name);
    const dim3 grid((unsigned)((max_block_points + KNN_T - 1) / KNN_T), (unsigned)n_blocks);
    hipStream_t s = (hipStream_t)stream;
    const bool ok = dim == 2
        ? dispatch_knn<2>(k, grid, s, points, (const long long*)first, coarse_ptr, (const long long*)edge_ptr, (long long*)out_sender, (long long*)out_recv)
        : dispatch_knn<3>(k, grid, s, points, (const long long*)first, coarse_ptr, (const long long*)edge_ptr, (long long*)out_sender, (long long*)out_recv);
    if (!ok) { mgn_set_error("%s: k = %d is not one of 1, 2, 3, 4, 6, 8", name, (int)k); return MGN_E_ARG; }
    return check_launch(name);
}
