// Device side of the one-hop halo exchange of a mesh partitioned by contiguous node ranges (not in the
// reference, which is single-device: core/rollout.py:28, run_train_multiscale.py:97).  A rank's rows are
// numbered [owned | halo]; the transport between ranks (RCCL through torch.distributed, or any other) moves
// contiguous buffers, and the kernels here move rows between those buffers and the row-indexed tensors:
//   pack        rows listed in send_idx            -> contiguous send buffer        (forward halo, per MP step)
//   unpack_add  rows returned by the peers         -> added to their owners' rows   (reverse halo, backward pass)
//   stage_state (world_pos, phi) of the owned rows -> local [owned | halo] arrays + 3-float send records
//   unpack_state received 3-float records          -> halo rows of the local arrays (per rollout step)
// No kernel allocates, none uses atomics: the add-back walks a per-row list in peer order, so sums do not
// depend on timing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "mgn_internal.h"

namespace {

constexpr int NT = 256;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define HALO_REQUIRE(cond, name)                                            \
    do { if (!(cond)) { mgn_set_error("%s: bad argument (%s)", name, #cond); return MGN_E_ARG; } } while (0)

inline int launched(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mgn_set_error("%s: %s", what, hipGetErrorString(e)); return (int)e; }
    return 0;
}

inline int grid_of(int64_t threads) { return (int)std::min<int64_t>((threads + NT - 1) / NT, 8192); }

// dst[i][:] = src[idx[i]][:], rows of w4 16-byte pieces
__global__ void halo_pack4_kernel(const f32x4* __restrict__ src, const int32_t* __restrict__ idx, f32x4* __restrict__ dst,
                                  long long total, int w4) {
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const long long r = i / w4;
        const int c = (int)(i - r * w4);
        dst[i] = src[(long long)idx[r] * w4 + c];
    }
}

// the same for row widths that are not a multiple of four floats
__global__ void halo_pack1_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, float* __restrict__ dst,
                                  long long total, int w) {
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const long long r = i / w;
        const int c = (int)(i - r * w);
        dst[i] = src[(long long)idx[r] * w + c];
    }
}

// buf[rows[d]][:] += recv[pos[k]][:] for k in [rowptr[d], rowptr[d+1]), in that order
__global__ void halo_unpack_add_kernel(const f32x4* __restrict__ recv, const int32_t* __restrict__ rowptr,
                                       const int32_t* __restrict__ rows, const int32_t* __restrict__ pos,
                                       f32x4* __restrict__ buf, long long total, int w4) {
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const long long d = i / w4;
        const int c = (int)(i - d * w4);
        const long long o = (long long)rows[d] * w4 + c;
        f32x4 acc = buf[o];
        for (int k = rowptr[d]; k < rowptr[d + 1]; ++k) acc += recv[(long long)pos[k] * w4 + c];
        buf[o] = acc;
    }
}

__global__ void halo_stage_state_kernel(const f32x2* __restrict__ world_in, const float* __restrict__ phi_in, int n_owned,
                                        const int32_t* __restrict__ idx, int n_send, f32x2* __restrict__ world_loc,
                                        float* __restrict__ phi_loc, float* __restrict__ send) {
    const int total = n_owned + n_send;
    for (int i = blockIdx.x * NT + threadIdx.x; i < total; i += gridDim.x * NT) {
        if (i < n_owned) {
            if (world_loc != world_in) world_loc[i] = world_in[i];
            if (phi_loc != phi_in) phi_loc[i] = phi_in[i];
        } else {
            const int j = i - n_owned;
            const int s = idx[j];
            const f32x2 w = world_in[s];
            send[3 * j + 0] = w.x;
            send[3 * j + 1] = w.y;
            send[3 * j + 2] = phi_in[s];
        }
    }
}

__global__ void halo_unpack_state_kernel(const float* __restrict__ recv, int n_halo, f32x2* __restrict__ world_halo,
                                         float* __restrict__ phi_halo) {
    for (int j = blockIdx.x * NT + threadIdx.x; j < n_halo; j += gridDim.x * NT) {
        f32x2 w;
        w.x = recv[3 * j + 0];
        w.y = recv[3 * j + 1];
        world_halo[j] = w;
        phi_halo[j] = recv[3 * j + 2];
    }
}

}  // namespace

extern "C" int mgn_halo_pack(const float* src, const int32_t* send_idx, float* send_buf, int64_t n_send, int32_t width,
                             void* stream) {
    const char* name = "mgn_halo_pack";
    HALO_REQUIRE(n_send >= 0 && width > 0 && n_send * (int64_t)width < ((int64_t)1 << 40), name);
    if (n_send == 0) return 0;
    HALO_REQUIRE(src && send_idx && send_buf, name);
    if (width % 4 == 0) {
        const int64_t total = n_send * (width / 4);
        hipLaunchKernelGGL(halo_pack4_kernel, dim3(grid_of(total)), dim3(NT), 0, (hipStream_t)stream, (const f32x4*)src,
                           send_idx, (f32x4*)send_buf, (long long)total, (int)(width / 4));
    } else {
        const int64_t total = n_send * width;
        hipLaunchKernelGGL(halo_pack1_kernel, dim3(grid_of(total)), dim3(NT), 0, (hipStream_t)stream, src, send_idx,
                           send_buf, (long long)total, (int)width);
    }
    return launched(name);
}

extern "C" int mgn_halo_unpack_add(const float* recv_buf, const int32_t* add_rowptr, const int32_t* add_rows,
                                   const int32_t* add_pos, float* buf, int64_t n_rows, int32_t width, void* stream) {
    const char* name = "mgn_halo_unpack_add";
    HALO_REQUIRE(n_rows >= 0 && width > 0 && width % 4 == 0, name);
    if (n_rows == 0) return 0;
    HALO_REQUIRE(recv_buf && add_rowptr && add_rows && add_pos && buf, name);
    const int64_t total = n_rows * (width / 4);
    hipLaunchKernelGGL(halo_unpack_add_kernel, dim3(grid_of(total)), dim3(NT), 0, (hipStream_t)stream,
                       (const f32x4*)recv_buf, add_rowptr, add_rows, add_pos, (f32x4*)buf, (long long)total, (int)(width / 4));
    return launched(name);
}

extern "C" int mgn_halo_stage_state(const float* world_pos, const float* phi, int64_t n_owned, const int32_t* send_idx,
                                    int64_t n_send, float* world_local, float* phi_local, float* send_buf, void* stream) {
    const char* name = "mgn_halo_stage_state";
    HALO_REQUIRE(n_owned >= 0 && n_send >= 0 && n_owned + n_send < ((int64_t)1 << 31), name);
    if (n_owned + n_send == 0) return 0;
    HALO_REQUIRE(world_pos && phi && world_local && phi_local && (n_send == 0 || (send_idx && send_buf)), name);
    hipLaunchKernelGGL(halo_stage_state_kernel, dim3(grid_of(n_owned + n_send)), dim3(NT), 0, (hipStream_t)stream,
                       (const f32x2*)world_pos, phi, (int)n_owned, send_idx, (int)n_send, (f32x2*)world_local, phi_local,
                       send_buf);
    return launched(name);
}

extern "C" int mgn_halo_unpack_state(const float* recv_buf, int64_t n_halo, float* world_halo, float* phi_halo,
                                     void* stream) {
    const char* name = "mgn_halo_unpack_state";
    HALO_REQUIRE(n_halo >= 0 && n_halo < ((int64_t)1 << 31), name);
    if (n_halo == 0) return 0;
    HALO_REQUIRE(recv_buf && world_halo && phi_halo, name);
    hipLaunchKernelGGL(halo_unpack_state_kernel, dim3(grid_of(n_halo)), dim3(NT), 0, (hipStream_t)stream, recv_buf,
                       (int)n_halo, (f32x2*)world_halo, phi_halo);
    return launched(name);
}
