// Training side of libmgn_hip.so: forward-with-saved-activations and backward building
// blocks of the MeshGraphNets MLPs for gfx950, fp32 throughout (fp32 MFMA for the products).
//
// Every 128-wide Linear of the model (inputs narrower than 128 are zero padded by the caller)
// is evaluated by the same few row-tile kernels:
//   train_linear_fwd   out = act(A Wa^T [+ B Wb^T] [+ bias] [+ P[idxP] + Q[idxQ]])
//   train_ln_fwd       y = LayerNorm(v) [, y + res]
//   train_ln_bwd       grad wrt LayerNorm input (optionally through a preceding ReLU), dgamma, dbeta
//   train_linear_bwd   dW = G^T A, db = colsum(G), GA = (G W) [* (mask > 0)] [+ res]
//   segment_sum        out[n] = sum of rows rowptr[n] .. rowptr[n+1] (optionally through an index list)
// Weight and bias gradients are summed per workgroup in registers, written to a scratch
// buffer and reduced by a second kernel in block order, so they are deterministic.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>

#include "mgn_device.h"

namespace {

constexpr int PARTIAL_W = H * H;            // floats of one workgroup's weight-gradient partial
constexpr int PARTIAL = PARTIAL_W + H;      // + bias-gradient partial
constexpr int TRAIN_BLOCKS_MAX = 1024;

// Transposed register-resident weight slice: for output column `col` (an input feature of the
// Linear), the contraction runs over the Linear's output features f = 8c + 4h + 0..3.
__device__ __forceinline__ void load_wfrag_t(f32x4 (&w)[16], const float* __restrict__ W, int ldw, int col, int h) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float* p = W + (size_t)(8 * c + 4 * h) * ldw + col;
        w[c] = f32x4{p[0], p[ldw], p[2 * (size_t)ldw], p[3 * (size_t)ldw]};
    }
}

// contiguous tile range of this workgroup
__device__ __forceinline__ void tile_range(long long rows, int& t0, int& t1) {
    const int ntiles = (int)((rows + TM - 1) / TM);
    const int nb = gridDim.x, rho = range_of_block(blockIdx.x, nb);
    t0 = (int)((long long)rho * ntiles / nb);
    t1 = (int)((long long)(rho + 1) * ntiles / nb);
}

// ------------------------------------------------------------------------ linear forward
struct LinFwdArgs {
    const float* A; const float* B; const float* W; int ldw; int koff_a; int koff_b;
    const float* bias; const float* P; const int* idxP; const float* Q; const int* idxQ;
    int relu; float* out; long long rows;
};

// HALF: the product runs on the fp16 matrix pipe (two-term split, see mgn_device.h); otherwise fp32 MFMA.
template <bool TWO, bool HALF>
__global__ __launch_bounds__(NT, 2) void train_linear_fwd_kernel(LinFwdArgs a) {
    constexpr int K4 = TWO ? H / 2 : H / 4;
    constexpr int KH = TWO ? 2 * H : H;              // columns of the staged tile
    constexpr int LDH = KH + 8, PLN = TM * LDH;      // fp16 plane geometry
    __shared__ __attribute__((aligned(16))) unsigned short Xs[HALF ? 2 * PLN : 2 * tile_floats(K4)];   // planes, or fp32 tile
    __shared__ __attribute__((aligned(16))) float O[tile_floats(H / 4)];
    __shared__ __attribute__((aligned(16))) float vec[H];
    float* X = (float*)Xs;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
    f32x4 wa[HALF ? 1 : 16], wb[(TWO && !HALF) ? 16 : 1];
    WPlanesH<HALF ? 8 : 1> ha;
    WPlanesH<(HALF && TWO) ? 8 : 1> hb;
    if constexpr (HALF) {
        load_wplanes_h<8>(ha, a.W, a.ldw, a.koff_a, 32 * w + j, h);
        if constexpr (TWO) load_wplanes_h<8>(hb, a.W, a.ldw, a.koff_b, 32 * w + j, h);
    } else {
        load_wfrag<16>(wa, a.W, a.ldw, a.koff_a, 32 * w + j, h);
        if constexpr (TWO) load_wfrag<16>(wb, a.W, a.ldw, a.koff_b, 32 * w + j, h);
    }
    constexpr float BS = HALF ? W_SCALE : 1.0f, US = HALF ? W_UNSCALE : 1.0f;
    if (t < H) vec[t] = a.bias ? a.bias[t] * BS : 0.f;
    int t0, t1;
    tile_range(a.rows, t0, t1);
    const int srow = t >> 5, sq = t & 31;
    float rmax = 0.f;                                // largest magnitude this thread has split (range guard)
    __syncthreads();
    for (int tile = t0; tile < t1; ++tile) {
        const long long base = (long long)tile * TM;
        const int nvalid = (int)min((long long)TM, a.rows - base);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = srow + 8 * it;
            const long long n = min(base + row, a.rows - 1);
            const f32x4 va = *(const f32x4*)(a.A + n * H + 4 * sq);
            if constexpr (HALF) hplanes_store4(Xs, PLN, LDH, row, 4 * sq, va, rmax);
            else *(f32x4*)(X + toff(row, sq, K4)) = va;
            if constexpr (TWO) {
                const f32x4 vb = *(const f32x4*)(a.B + n * H + 4 * sq);
                if constexpr (HALF) hplanes_store4(Xs, PLN, LDH, row, H + 4 * sq, vb, rmax);
                else *(f32x4*)(X + toff(row, 32 + sq, K4)) = vb;
            }
        }
        __syncthreads();
        f32x16 acc = acc_from_vec(vec, w, h);
        if constexpr (HALF) {
            acc = wave_gemm_h<0, 8, 8>(Xs, PLN, LDH, 0, ha, j, h, acc);
            if constexpr (TWO) acc = wave_gemm_h<0, 8, 8>(Xs, PLN, LDH, H, hb, j, h, acc);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] *= US;
        } else {
            acc = wave_gemm<16>(X, K4, 0, wa, j, h, acc);
            if constexpr (TWO) acc = wave_gemm<16>(X, K4, 32, wb, j, h, acc);
        }
        if (a.P) {                                   // per-row gathered addends, accumulator layout
            const long long n = min(base + j, a.rows - 1);
            const size_t ip = (size_t)a.idxP[n] * H, iq = (size_t)a.idxQ[n] * H;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 p = *(const f32x4*)(a.P + ip + 32 * w + 8 * g + 4 * h);
                const f32x4 q = *(const f32x4*)(a.Q + iq + 32 * w + 8 * g + 4 * h);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[4 * g + k] += p[k] + q[k];
            }
        }
        if (a.relu) acc = relu16(acc);
        acc_to_tile(O, acc, w, j, h);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = srow + 8 * it;
            if (row < nvalid) *(f32x4*)(a.out + (base + row) * H + 4 * sq) = *(const f32x4*)(O + toff(row, sq, H / 4));
        }
        // the tile is restaged at the top of the next pass: every wave has finished its products before
        // the barrier above; O is rewritten only after the next pass's first barrier
    }
    if constexpr (HALF) range_report(rmax);
}

// ------------------------------------------------------------------------ LayerNorm forward
// 32 lanes per row, 16 bytes per lane; statistics as torch computes them (mean, then the mean
// of squared deviations).
__device__ __forceinline__ float row_sum32(float v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 32);
    return v;
}

__device__ __forceinline__ void row_stats(const f32x4 v, float& mean, float& rstd) {
    mean = row_sum32(v[0] + v[1] + v[2] + v[3]) * (1.0f / H);
    const f32x4 d = v - mean;
    const float var = row_sum32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.0f / H);
    rstd = 1.0f / sqrtf(var + LN_EPS);
}

__global__ __launch_bounds__(NT) void train_ln_fwd_kernel(const float* __restrict__ v, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* res,
                                                          float* y, float* yres, long long rows) {
    const int sq = threadIdx.x & 31;
    const f32x4 ga = *(const f32x4*)(gamma + 4 * sq), be = *(const f32x4*)(beta + 4 * sq);
    for (long long r = (long long)blockIdx.x * 8 + (threadIdx.x >> 5); r < rows; r += (long long)gridDim.x * 8) {
        const f32x4 x = *(const f32x4*)(v + r * H + 4 * sq);
        float mean, rstd;
        row_stats(x, mean, rstd);
        const f32x4 o = (x - mean) * rstd * ga + be;
        if (y) *(f32x4*)(y + r * H + 4 * sq) = o;
        if (yres) *(f32x4*)(yres + r * H + 4 * sq) = o + *(const f32x4*)(res + r * H + 4 * sq);
    }
}

// ------------------------------------------------------------------------ LayerNorm backward
// gy = G[r] (+ Gg[idx[r]]);  gv = rstd * (gy*gamma - mean(gy*gamma) - vhat * mean(gy*gamma*vhat));
// with relu_mask the gradient continues through the ReLU that produced v (v > 0).
// partial[block][0..127] = sum gy * vhat (dgamma), [128..255] = sum gy (dbeta).
__global__ __launch_bounds__(NT) void train_ln_bwd_kernel(const float* __restrict__ G, const float* __restrict__ Gg,
                                                          const int* __restrict__ idx, const float* __restrict__ v,
                                                          const float* __restrict__ gamma, int relu_mask,
                                                          float* __restrict__ gpre, float* __restrict__ partial,
                                                          long long rows) {
    __shared__ __attribute__((aligned(16))) float sh[8][2 * H];
    const int sq = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const f32x4 ga = *(const f32x4*)(gamma + 4 * sq);
    f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
    // contiguous row range per workgroup, rows dealt to the 8 row groups in turn
    const long long per = (rows + gridDim.x - 1) / gridDim.x;
    const long long r0 = (long long)blockIdx.x * per, r1 = min(rows, r0 + per);
    for (long long r = r0 + grp; r < r1; r += 8) {
        f32x4 gy = {0.f, 0.f, 0.f, 0.f};
        if (G) gy = *(const f32x4*)(G + r * H + 4 * sq);
        if (Gg) gy += *(const f32x4*)(Gg + (size_t)idx[r] * H + 4 * sq);
        const f32x4 x = *(const f32x4*)(v + r * H + 4 * sq);
        float mean, rstd;
        row_stats(x, mean, rstd);
        const f32x4 xh = (x - mean) * rstd;
        dg += gy * xh; db += gy;
        const f32x4 gh = gy * ga;
        const float m1 = row_sum32(gh[0] + gh[1] + gh[2] + gh[3]) * (1.0f / H);
        const float m2 = row_sum32(gh[0] * xh[0] + gh[1] * xh[1] + gh[2] * xh[2] + gh[3] * xh[3]) * (1.0f / H);
        f32x4 gv = (gh - m1 - xh * m2) * rstd;
        if (relu_mask) {
#pragma unroll
            for (int k = 0; k < 4; ++k) gv[k] = x[k] > 0.f ? gv[k] : 0.f;
        }
        *(f32x4*)(gpre + r * H + 4 * sq) = gv;
    }
    *(f32x4*)(&sh[grp][4 * sq]) = dg;
    *(f32x4*)(&sh[grp][H + 4 * sq]) = db;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += sh[g][threadIdx.x];
    partial[(size_t)blockIdx.x * (2 * H) + threadIdx.x] = s;
}

// ------------------------------------------------------------------------ linear backward
struct LinBwdArgs {
    const float* G; const float* A; const float* W; int ldw; int koff;
    const float* mask; const float* res; float* GA; float* partial; long long rows;
};

template <bool DO_GA, bool DO_DW>
__global__ __launch_bounds__(NT, 2) void train_linear_bwd_kernel(LinBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float Gt[tile_floats(H / 4)];
    __shared__ __attribute__((aligned(16))) float At[DO_DW ? tile_floats(H / 4) : 4];
    __shared__ __attribute__((aligned(16))) float Ot[DO_GA ? tile_floats(H / 4) : 4];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
    f32x4 wt[DO_GA ? 16 : 1];
    if constexpr (DO_GA) load_wfrag_t(wt, a.W, a.ldw, a.koff + 32 * w + j, h);
    f32x16 dw[DO_DW ? 4 : 1];
#pragma unroll
    for (int kb = 0; kb < (DO_DW ? 4 : 1); ++kb) dw[kb] = zero16();
    float dbias = 0.f;
    int t0, t1;
    tile_range(a.rows, t0, t1);
    const int srow = t >> 5, sq = t & 31;
    constexpr int LD = H + 4;                      // tile row stride in floats
    for (int tile = t0; tile < t1; ++tile) {
        const long long base = (long long)tile * TM;
        const int nvalid = (int)min((long long)TM, a.rows - base);
        __syncthreads();                           // previous pass done with Gt / At / Ot
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = srow + 8 * it;
            const bool ok = row < nvalid;          // rows past the end contribute zeros to the sums
            const long long n = min(base + row, a.rows - 1);
            const f32x4 g = *(const f32x4*)(a.G + n * H + 4 * sq);
            *(f32x4*)(Gt + toff(row, sq, H / 4)) = ok ? g : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (DO_DW) *(f32x4*)(At + toff(row, sq, H / 4)) = *(const f32x4*)(a.A + n * H + 4 * sq);
        }
        __syncthreads();
        if constexpr (DO_GA) {
            f32x16 acc = zero16();
            acc = wave_gemm<16>(Gt, H / 4, 0, wt, j, h, acc);
            acc_to_tile(Ot, acc, w, j, h);
        }
        if constexpr (DO_DW) {
#pragma unroll
            for (int s = 0; s < TM / 2; ++s) {
                const float gval = Gt[(2 * s + h) * LD + 32 * w + j];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
                    dw[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(gval, At[(2 * s + h) * LD + 32 * kb + j], dw[kb], 0, 0, 0);
            }
            if (t < H) {
#pragma unroll 8
                for (int r = 0; r < TM; ++r) dbias += Gt[r * LD + t];
            }
        }
        if constexpr (DO_GA) {
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = srow + 8 * it;
                if (row < nvalid) {
                    const size_t o = (size_t)(base + row) * H + 4 * sq;
                    f32x4 val = *(const f32x4*)(Ot + toff(row, sq, H / 4));
                    if (a.mask) {
                        const f32x4 m = *(const f32x4*)(a.mask + o);
#pragma unroll
                        for (int k = 0; k < 4; ++k) val[k] = m[k] > 0.f ? val[k] : 0.f;
                    }
                    if (a.res) val += *(const f32x4*)(a.res + o);
                    *(f32x4*)(a.GA + o) = val;
                }
            }
        }
    }
    if constexpr (DO_DW) {
        float* p = a.partial + (size_t)blockIdx.x * PARTIAL;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int f = 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                p[f * H + 32 * kb + j] = dw[kb][reg];
            }
        if (t < H) p[PARTIAL_W + t] = dbias;
    }
}

// ------------------------------------------------------------------------ linear backward, split-bf16 path
// Same contract as train_linear_bwd_kernel with both products on the bf16 matrix pipe (three exact
// bf16 terms per fp32 operand, six partial products; gradients keep their full exponent range).
//   GA = G W      : contraction over the Linear's outputs f; operand = 8 consecutive features of one tile row
//                   (one 16-byte row read), W^T in registers.
//   dW = G^T A    : contraction over the 32 rows of the tile; operand = 8 consecutive rows of one feature,
//                   fetched from the SAME row-major image with the transposing LDS read (ds_read_b64_tr_b16:
//                   per 16 lanes a block of 4 rows x 16 columns, lane i receives column i), two reads per operand.
// One image per plane serves both: 256-byte rows without padding, the 16-byte chunk ch of row r stored at chunk
// ch ^ (4 (r & 3) | ((r >> 2) & 3)), which keeps the row reads, the transposed reads and the 8-byte stores
// free of bank conflicts (scripts/micro/tr_read.hip checks the lane map and the swizzle on the device).
// One workgroup per CU (weight planes and gradient accumulators take 160 registers per lane); the next tile's rows
// are requested as soon as the current tile's have been staged.
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int SPL = TM * H;                 // halves per plane of the image

__device__ __forceinline__ int swz_off(int row, int ch) { return row * H + 8 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// four consecutive features k .. k+3 (k a multiple of 4) of one tile row -> the three planes
__device__ __forceinline__ void splanes_store4(unsigned short* img, int row, int k, const f32x4 v) {
    u32x2 ph, pm, pl;
    split_pack4(v, ph, pm, pl);
    unsigned short* d = img + swz_off(row, k >> 3) + (k & 4);
    *(u32x2*)(d) = ph; *(u32x2*)(d + SPL) = pm; *(u32x2*)(d + 2 * SPL) = pl;
}

// The transposing read needs every lane of the wave active: call from workgroup-uniform control flow only.
__device__ __forceinline__ u32x2 lds_tr4(const unsigned short* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p));
}

// halves offset of this lane's address for the block of rows 16s + 8h + 4hh + 0..3, columns 32 blk + 16 gg + 0..15
// (h = lane >> 5, gg = bit 4 of the lane): lane 4q + p of the 16 supplies row q, columns 4p .. 4p+3
__device__ __forceinline__ int tr_off(int lane, int s, int hh, int blk) {
    const int h = lane >> 5, gg = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    return swz_off(16 * s + 8 * h + 4 * hh + q, 4 * blk + 2 * gg + (p >> 1)) + 4 * (p & 1);
}

// The addresses of the other column blocks and of the second 16 rows follow from the one of block 0, rows 0..15:
// the block index only enters bits 5-6 of the halves offset (through the swizzle: an XOR), s adds 16 rows.
struct TrBase { int o[2]; };                 // hh = 0, 1
__device__ __forceinline__ TrBase tr_base(int lane) { return TrBase{{tr_off(lane, 0, 0, 0), tr_off(lane, 0, 1, 0)}}; }

// v_mfma_f32_32x32x16_bf16 operand of lane (j = lane & 31, h = lane >> 5): rows 16s + 8h + 0..7 of column 32 blk + j
__device__ __forceinline__ void tr_operand3(const unsigned short* img, const TrBase& tb, int s, int blk, u32x4& xh, u32x4& xm, u32x4& xl) {
    const unsigned short* p0 = img + ((tb.o[0] ^ (32 * blk)) + 16 * H * s);
    const unsigned short* p1 = img + ((tb.o[1] ^ (32 * blk)) + 16 * H * s);
    const u32x2 h0 = lds_tr4(p0), h1 = lds_tr4(p1), m0 = lds_tr4(p0 + SPL), m1 = lds_tr4(p1 + SPL);
    const u32x2 l0 = lds_tr4(p0 + 2 * SPL), l1 = lds_tr4(p1 + 2 * SPL);
    xh = u32x4{h0[0], h0[1], h1[0], h1[1]};
    xm = u32x4{m0[0], m0[1], m1[0], m1[1]};
    xl = u32x4{l0[0], l0[1], l1[0], l1[1]};
}

// wave_gemm3_32 on the swizzled image.  rb = swz_off(j, h), the lane's offset for k-step 0; k-step c sits at rb ^ 16c
// (2c + h = 2c ^ h, and the swizzle is an XOR as well), so one register carries all eight addresses.
template <int C0, int C1>
__device__ __forceinline__ f32x16 wave_gemm3_32s(const unsigned short* img, const WPlanes32& w, int rb, f32x16 acc) {
    const unsigned short* r = img + (rb ^ (16 * C0));
    u32x4 xh = *(const u32x4*)(r), xm = *(const u32x4*)(r + SPL), xl = *(const u32x4*)(r + 2 * SPL);
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
    for (int c = C0; c < C1; ++c) {
        u32x4 nh = xh, nm = xm, nl = xl;
        if (c + 1 < C1) {
            const unsigned short* n = img + (rb ^ (16 * (c + 1)));
            nh = *(const u32x4*)(n); nm = *(const u32x4*)(n + SPL); nl = *(const u32x4*)(n + 2 * SPL);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        }
        acc = mfma32_bf16(w.p[2][c], xh, acc);   // lo * hi
        acc = mfma32_bf16(w.p[0][c], xl, acc);   // hi * lo
        acc = mfma32_bf16(w.p[1][c], xm, acc);   // mid * mid
        acc = mfma32_bf16(w.p[1][c], xh, acc);   // mid * hi
        acc = mfma32_bf16(w.p[0][c], xm, acc);   // hi * mid
        acc = mfma32_bf16(w.p[0][c], xh, acc);   // hi * hi
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        xh = nh; xm = nm; xl = nl;
    }
    return acc;
}

// Accumulator layout -> image: lane (j, h) of wave w holds features 32w + 8g + 4h + 0..3 of row j in registers
// 4g .. 4g+3.  sb = swz_off(j, 4w) + 4h is the offset of group 0; group g sits at sb ^ 8g.
__device__ __forceinline__ void splanes_store_acc4(unsigned short* img, int sb, int g, const f32x4 v) {
    u32x2 ph, pm, pl;
    split_pack4(v, ph, pm, pl);
    unsigned short* d = img + (sb ^ (8 * g));
    *(u32x2*)(d) = ph; *(u32x2*)(d + SPL) = pm; *(u32x2*)(d + 2 * SPL) = pl;
}

// dw[kb] += G^T A over the 32 rows of the tile: feature block `w` of G (this wave's 32 weight rows) against the column
// blocks kb = KB0 .. KB0+NKB-1 of A.  Partial products smallest first, per accumulator in the order of wave_gemm3_32.
template <int KB0, int NKB>
__device__ __forceinline__ void wave_outer3(const unsigned short* Gimg, const unsigned short* Aimg, const TrBase& tb, int w, f32x16* dw) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u32x4 gh, gm, gl, ah[NKB], am[NKB], al[NKB];
        tr_operand3(Gimg, tb, s, w, gh, gm, gl);
#pragma unroll
        for (int q = 0; q < NKB; ++q) tr_operand3(Aimg, tb, s, KB0 + q, ah[q], am[q], al[q]);
#pragma unroll
        for (int q = 0; q < NKB; ++q) dw[KB0 + q] = mfma32_bf16(gl, ah[q], dw[KB0 + q]);
#pragma unroll
        for (int q = 0; q < NKB; ++q) dw[KB0 + q] = mfma32_bf16(gh, al[q], dw[KB0 + q]);
#pragma unroll
        for (int q = 0; q < NKB; ++q) dw[KB0 + q] = mfma32_bf16(gm, am[q], dw[KB0 + q]);
#pragma unroll
        for (int q = 0; q < NKB; ++q) dw[KB0 + q] = mfma32_bf16(gm, ah[q], dw[KB0 + q]);
#pragma unroll
        for (int q = 0; q < NKB; ++q) dw[KB0 + q] = mfma32_bf16(gh, am[q], dw[KB0 + q]);
#pragma unroll
        for (int q = 0; q < NKB; ++q) dw[KB0 + q] = mfma32_bf16(gh, ah[q], dw[KB0 + q]);
    }
}

// W^T planes for output column `col` of GA: k-step c covers the Linear's outputs f = 16c + 8h + 0..7
__device__ __forceinline__ void load_wplanes32_t(WPlanes32& w, const float* __restrict__ W, int ldw, int col, int h) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float* p = W + (size_t)(16 * c + 8 * h) * ldw + col;
        const size_t l = (size_t)ldw;
        const f32x4 lo4 = {p[0], p[l], p[2 * l], p[3 * l]}, hi4 = {p[4 * l], p[5 * l], p[6 * l], p[7 * l]};
        u32x2 a[3], b[3];
        split_pack4(lo4, a[0], a[1], a[2]);
        split_pack4(hi4, b[0], b[1], b[2]);
#pragma unroll
        for (int q = 0; q < 3; ++q) w.p[q][c] = u32x4{a[q][0], a[q][1], b[q][0], b[q][1]};
    }
}

struct TileRows { f32x4 g[4], a[4], m[4], r[4]; };     // one thread's share of a tile: rows srow + 8*it, columns 4*sq ..

template <bool DO_GA, bool DO_DW>
__global__ __launch_bounds__(NT, 1) void train_linear_bwd3_kernel(LinBwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short Gs[3 * SPL];                    // G tile, three planes
    __shared__ __attribute__((aligned(16))) unsigned short As[DO_DW ? 3 * SPL : 8];        // A tile, three planes
    __shared__ __attribute__((aligned(16))) float Ot[tile_floats(H / 4)];                  // GA tile; bias sums at the end

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
    WPlanes32 wt;
    if constexpr (DO_GA) load_wplanes32_t(wt, a.W, a.ldw, a.koff + 32 * w + j, h);
    f32x16 dw[DO_DW ? 4 : 1];
#pragma unroll
    for (int kb = 0; kb < (DO_DW ? 4 : 1); ++kb) dw[kb] = zero16();
    f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
    int t0, t1;
    tile_range(a.rows, t0, t1);
    const int srow = t >> 5, sq = t & 31;
    const TrBase tb = tr_base(lane);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // An absent mask / residual is read all the same, every lane at offset 0 of G (a line that stays in the cache), and
    // dropped by a select: the tile loop has no branch around a load.
    // A mask that is the layer input itself (the usual case: the ReLU output that fed this Linear) is not read twice.
    const bool has_mask = DO_GA && a.mask, has_res = DO_GA && a.res, mask_is_a = DO_GA && DO_DW && a.mask == a.A;
    const float* mask_p = (has_mask && !mask_is_a) ? a.mask : a.G;
    const float* res_p = has_res ? a.res : a.G;
    const size_t mmul = (has_mask && !mask_is_a) ? 1 : 0, rmul = has_res ? 1 : 0;
    auto request = [&](TileRows& p, int tile) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const size_t o = (size_t)min((long long)tile * TM + srow + 8 * it, a.rows - 1) * H + 4 * sq;
            p.g[it] = *(const f32x4*)(a.G + o);
            if constexpr (DO_DW) p.a[it] = *(const f32x4*)(a.A + o);
            if constexpr (DO_GA) { p.m[it] = *(const f32x4*)(mask_p + o * mmul); p.r[it] = *(const f32x4*)(res_p + o * rmul); }
        }
    };
    // One tile: `cur` holds its rows; the next tile's are requested into `nxt` before anything else, so they have the
    // whole pass to arrive.  The two register sets swap roles from pass to pass (the loop below is unrolled by two):
    // nothing is copied from one to the other, and no pass ends waiting for its own requests or stores.
    auto pass = [&](TileRows& cur, TileRows& nxt, int tile) {
        const long long base = (long long)tile * TM;
        const int nvalid = (int)min((long long)TM, a.rows - base);
        request(nxt, min(tile + 1, t1 - 1));         // the last pass requests its own rows again (never used)
        __syncthreads();                             // the previous tile's products and output pass are done
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const f32x4 g = srow + 8 * it < nvalid ? cur.g[it] : zero4;     // rows past the end add nothing to the sums
            dbacc += g;
            splanes_store4(Gs, srow + 8 * it, 4 * sq, g);
            if constexpr (DO_DW) splanes_store4(As, srow + 8 * it, 4 * sq, cur.a[it]);
        }
        __syncthreads();
        if constexpr (DO_GA) {
            f32x16 acc = zero16();
            acc = wave_gemm3_32s<0, 8>(Gs, wt, swz_off(j, h), acc);
            acc_to_tile(Ot, acc, w, j, h);
        }
        if constexpr (DO_DW) wave_outer3<0, 4>(Gs, As, tb, w, dw);
        if constexpr (DO_GA) {
            __syncthreads();
            // the next tile's rows are taken as arrived before this tile's stores: loads and stores share one in-order
            // counter, so a wait for them placed after the stores (at their first use, in the next pass) would
            // wait for the stores as well
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                asm volatile("" : "+v"(nxt.g[it]), "+v"(nxt.m[it]), "+v"(nxt.r[it]));
                if constexpr (DO_DW) asm volatile("" : "+v"(nxt.a[it]));
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = srow + 8 * it;
                if (row < nvalid) {
                    f32x4 val = *(const f32x4*)(Ot + toff(row, sq, H / 4));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float mk = (DO_DW && mask_is_a) ? cur.a[it][k] : cur.m[it][k];
                        val[k] = (!has_mask || mk > 0.f) ? val[k] : 0.f;
                        val[k] += has_res ? cur.r[it][k] : 0.f;
                    }
                    *(f32x4*)(a.GA + (size_t)(base + row) * H + 4 * sq) = val;
                }
            }
        }
    };
    TileRows ra, rb;
    if (t0 < t1) request(ra, t0);
    for (int tile = t0; tile < t1; tile += 2) {
        pass(ra, rb, tile);
        if (tile + 1 < t1) pass(rb, ra, tile + 1);
    }
    if constexpr (DO_DW) {
        float* p = a.partial + (size_t)blockIdx.x * PARTIAL;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int f = 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                p[f * H + 32 * kb + j] = dw[kb][reg];
            }
        __syncthreads();
        *(f32x4*)(Ot + srow * H + 4 * sq) = dbacc;
        __syncthreads();
        if (t < H) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) sum += Ot[g * H + t];
            p[PARTIAL_W + t] = sum;
        }
    }
}

// ------------------------------------------------------------------------ edge update backward, fused
// Backward of  e' = LN(ReLU(W2 ReLU(ps[s] + pr[r] + W0e e) + b2))  from the block's INPUTS only: the two
// hidden activations are recomputed per 32-edge tile with exactly the arithmetic of mp_edge_h4_kernel
// (same split, same products, same order), so their ReLU patterns are the forward's.  Per tile:
//   1  h1 = ReLU(ps[s] + pr[r] + W0e e)        fp16-split product; kept in registers (mask), written as
//                                               fp16 planes (product 2) and bf16 planes (dW2, read transposed)
//   2  v = ReLU(W2 h1 + b2), LayerNorm statistics;  gy = ge + gaggr[r];  LayerNorm backward -> g2
//      (two row reductions through LDS); g2 as bf16 planes, one image for the row reads and the transposed reads
//   3  dW2 += g2^T h1;  g1 = (g2 W2) * (h1 > 0)   split-bf16 products
//   4  g1 -> global (coalesced), dpr[r] = sum of g1 over the in-edges of r (same scan as the forward)
// db2, dgamma, dbeta accumulate per lane in accumulator layout and are reduced once at the end.
// Workgroups own whole receivers (contiguous CSR ranges), one workgroup per CU.
struct EdgeBwdArgs {
    const float* el; const float* ge; const float* gaggr; const float* ps; const float* pr;
    const int* sender; const int* recv; const int* rowptr;
    const float* w0; const float* w1; const float* b1; const float* ln_w;
    float* g1; float* dpr; float* partial; int n_nodes; int n_edges;
};

constexpr int PARTIAL_E = H * H + 3 * H;     // dW2 | db2 | dgamma | dbeta
constexpr int PKH = H + 8;                   // halves per row of an fp16 plane
constexpr int PLH = TM * PKH;                // halves per fp16 plane

__global__ __launch_bounds__(NT, 1) void train_edge_bwd_kernel(EdgeBwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short elP[2 * PLH];        // e tile, fp16 planes
    __shared__ __attribute__((aligned(16))) unsigned short h1P[2 * PLH];        // h1 tile, fp16 planes; then Ot
    __shared__ __attribute__((aligned(16))) unsigned short H1S[3 * SPL];        // h1 tile, bf16 planes (transposed reads)
    __shared__ __attribute__((aligned(16))) unsigned short G2S[3 * SPL];        // g2 tile, bf16 planes (row and transposed reads)
    __shared__ __attribute__((aligned(16))) float vec[2][H];                    // W_SCALE * b2, gamma
    __shared__ __attribute__((aligned(16))) float red[LN_RED_FLOATS];
    __shared__ __attribute__((aligned(16))) float red2[LN_RED_FLOATS];
    __shared__ float cout[H], tcarry[H];
    __shared__ int srange[2];
    float* Ot = (float*)h1P;                         // g1 tile (fp32); h1P is dead once product 2 has run
    static_assert(sizeof(h1P) >= sizeof(float) * tile_floats(H / 4), "Ot must fit into h1P");

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
    WPlanesH<8> wa, wb;
    WPlanes32 w2t;
    load_wplanes_h<8>(wa, a.w0, 3 * H, 2 * H, 32 * w + j, h);
    load_wplanes_h<8>(wb, a.w1, H, 0, 32 * w + j, h);
    load_wplanes32_t(w2t, a.w1, H, 32 * w + j, h);
    if (t < H) { vec[0][t] = a.b1[t] * W_SCALE; vec[1][t] = a.ln_w[t]; tcarry[t] = 0.f; }
    if (t < 64) block_edge_range(a.rowptr, a.n_nodes, a.n_edges, srange);   // wave 0, all of its lanes
    f32x16 dw2[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) dw2[kb] = zero16();
    f32x16 db2 = zero16(), dgam = zero16(), dbet = zero16();
    __syncthreads();
    const int e0 = __builtin_amdgcn_readfirstlane(srange[0]), e1 = __builtin_amdgcn_readfirstlane(srange[1]);
    const int last = e1 - e0 - 1;                    // row offsets are relative to e0 and clamped to [0, last]
    if (e0 < e1) {
    const float* el_g = a.el + (size_t)e0 * H;
    const float* ge_g = a.ge ? a.ge + (size_t)e0 * H : nullptr;
    float* g1_g = a.g1 + (size_t)e0 * H;
    const int* snd_g = a.sender + e0;
    const int* rcv_g = a.recv + e0;
    const int srow = t >> 5, sq = t & 31;            // row map: rows srow + 8*it, 16-byte chunk sq
    const int sc = t & (H - 1);                      // scan map: column sc, rows 16*sg .. 16*sg+15
    const int sg = __builtin_amdgcn_readfirstlane(t >> 7);

    f32x4 pre[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) pre[it] = *(const f32x4*)(el_g + (unsigned)(min(srow + 8 * it, last) * H + 4 * sq));
    int s_cur = snd_g[min(j, last)], r_cur = rcv_g[min(j, last)];
    // an absent edge-latent gradient is read all the same (row 0 of e, a line that stays in the cache) and dropped by a select
    const float* ge_p = ge_g ? ge_g : el_g;
    const unsigned ge_mul = ge_g ? 1u : 0u;
    f32x4 gs[4], gr[4];
    {
        const unsigned so = (unsigned)s_cur * H + 32 * w + 4 * h, ro = (unsigned)r_cur * H + 32 * w + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) { gs[g] = *(const f32x4*)(a.ps + so + 8 * g); gr[g] = *(const f32x4*)(a.pr + ro + 8 * g); }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) hplanes_store4(elP, PLH, PKH, srow + 8 * it, 4 * sq, pre[it]);
    __syncthreads();
    STAMP_DECL
    for (int rel = 0; rel <= last; rel += TM) {
        STAMP(0);
        const int nvalid = min(TM, last + 1 - rel);
        const bool valid = j < nvalid;
        // requests: the next tile of e and of the indices (the gathered projections of this tile were requested during
        // the previous one)
        f32x4 ye[4], ya[4];
#pragma unroll
        for (int it = 0; it < 4; ++it)
            pre[it] = *(const f32x4*)(el_g + (unsigned)(min(rel + TM + srow + 8 * it, last) * H + 4 * sq));
        const int s_nxt = snd_g[min(rel + TM + j, last)], r_nxt = rcv_g[min(rel + TM + j, last)];
        const int ra = r_cur;
        const int rb = (rel + j + 1 <= last) ? rcv_g[rel + j + 1] : -1;

        // ---- 1: first hidden activation (the forward's arithmetic)
        STAMP(1);
        f32x16 acc = zero16();
        acc = wave_gemm_h<0, 8, 8>(elP, PLH, PKH, 0, wa, j, h, acc);
        STAMP(2);
        unsigned hmask = 0;                          // bit i: h1 of accumulator register i is positive
        {
            const int sb = swz_off(opaque(j), 4 * w) + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = fmaxf(fmaf(acc[4 * g + k], W_UNSCALE, gs[g][k] + gr[g][k]), 0.f);
                    hmask |= v[k] > 0.f ? 1u << (4 * g + k) : 0u;
                }
                hplanes_store4(h1P, PLH, PKH, j, 32 * w + 8 * g + 4 * h, v);
                splanes_store_acc4(H1S, sb, g, v);
            }
        }
        // upstream gradient rows (accumulator layout), consumed after the second product
        {
            const unsigned ro = (unsigned)r_cur * H + 32 * w + 4 * h;
            const unsigned eo = (unsigned)min(rel + j, last) * H * ge_mul + 32 * w + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) { ya[g] = *(const f32x4*)(a.gaggr + ro + 8 * g); ye[g] = *(const f32x4*)(ge_p + eo + 8 * g); }
        }
        STAMP(3);
        __syncthreads();                                                     // (B) h1 planes complete; elP dead
        STAMP(4);
        // ---- 2: second hidden activation, LayerNorm statistics
        acc = acc_from_vec(vec[0], w, h);
        acc = wave_gemm_h<0, 8, 8>(h1P, PLH, PKH, 0, wb, j, h, acc);
        acc = relu16(acc);                                                   // W_SCALE * v
        ln_local_stats_rows(acc, red, w, j, h);
#pragma unroll
        for (int it = 0; it < 4; ++it) hplanes_store4(elP, PLH, PKH, srow + 8 * it, 4 * sq, pre[it]);   // next tile
        STAMP(5);
        __syncthreads();                                                     // (C) statistics exchanged; h1P dead
        STAMP(6);
        float rstd;
        f32x16 xh, g2;
        {
            const f32x4 ma = *(const f32x4*)(red + j * LN_RED_LD), mb = *(const f32x4*)(red + j * LN_RED_LD + 4);
            const f32x4 qa = *(const f32x4*)(red + j * LN_RED_LD + 8), qb = *(const f32x4*)(red + j * LN_RED_LD + 12);
            const float mean = (((ma[0] + ma[1]) + (ma[2] + ma[3])) + ((mb[0] + mb[1]) + (mb[2] + mb[3]))) * 0.125f;
            const float m2 = ((qa[0] + qa[1]) + (qa[2] + qa[3])) + ((qb[0] + qb[1]) + (qb[2] + qb[3]));
            const f32x4 da = ma - mean, db = mb - mean;
            const float between = ((da[0] * da[0] + da[1] * da[1]) + (da[2] * da[2] + da[3] * da[3])) +
                                  ((db[0] * db[0] + db[1] * db[1]) + (db[2] * db[2] + db[3] * db[3]));
            const float var = (m2 + 16.0f * between) * (1.0f / H);
            rstd = __builtin_amdgcn_rsqf(var + LN_EPS * W_SCALE * W_SCALE);          // of the scaled values
            const float shift = -mean * rstd;
#pragma unroll
            for (int i = 0; i < 16; ++i) xh[i] = fmaf(acc[i], rstd, shift);
        }
        // upstream gradient of e' (edge latent residual branch + receiver aggregate), LayerNorm backward
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 ga = *(const f32x4*)(vec[1] + 32 * w + 8 * g + 4 * h);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = 4 * g + k;
                const float gy = valid ? (ge_g ? ye[g][k] : 0.f) + ya[g][k] : 0.f;   // rows past the end add nothing
                dbet[i] += gy;
                dgam[i] = fmaf(gy, xh[i], dgam[i]);
                const float ghv = gy * ga[k];
                g2[i] = ghv;
                s1 += ghv;
                s2 = fmaf(ghv, xh[i], s2);
            }
        }
        red2[j * LN_RED_LD + 2 * w + h] = s1;
        red2[j * LN_RED_LD + 8 + 2 * w + h] = s2;
        // the next tile's projections, into registers the upstream gradient rows have just left: most of a tile to arrive
        __builtin_amdgcn_sched_barrier(0);
        {
            const unsigned so = (unsigned)s_nxt * H + 32 * w + 4 * h, ro = (unsigned)r_nxt * H + 32 * w + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) { gs[g] = *(const f32x4*)(a.ps + so + 8 * g); gr[g] = *(const f32x4*)(a.pr + ro + 8 * g); }
        }
        STAMP(7);
        __syncthreads();                                                     // (C2) row sums exchanged
        STAMP(8);
        {
            const f32x4 pa = *(const f32x4*)(red2 + j * LN_RED_LD), pb = *(const f32x4*)(red2 + j * LN_RED_LD + 4);
            const f32x4 qa = *(const f32x4*)(red2 + j * LN_RED_LD + 8), qb = *(const f32x4*)(red2 + j * LN_RED_LD + 12);
            const float m1 = (((pa[0] + pa[1]) + (pa[2] + pa[3])) + ((pb[0] + pb[1]) + (pb[2] + pb[3]))) * (1.0f / H);
            const float m2 = (((qa[0] + qa[1]) + (qa[2] + qa[3])) + ((qb[0] + qb[1]) + (qb[2] + qb[3]))) * (1.0f / H);
            const float rs = rstd * W_SCALE;                                 // 1 / sqrt(var(v) + eps)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float gv = rs * (g2[i] - m1 - xh[i] * m2);
                g2[i] = acc[i] > 0.f ? gv : 0.f;                             // through the ReLU that produced v
                db2[i] += g2[i];
            }
        }
        {
            const int sb = swz_off(opaque(j), 4 * w) + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                splanes_store_acc4(G2S, sb, g, f32x4{g2[4 * g], g2[4 * g + 1], g2[4 * g + 2], g2[4 * g + 3]});
        }
        STAMP(9);
        __syncthreads();                                                     // (D) g2 planes complete
        STAMP(10);
        // ---- 3: dW2 += g2^T h1,  g1 = (g2 W2) * (h1 > 0)
        {
            const TrBase tb = tr_base(opaque(lane));
            wave_outer3<0, 2>(G2S, H1S, tb, w, dw2);     // two column blocks at a time: 24 operand registers, not 48
            wave_outer3<2, 2>(G2S, H1S, tb, w, dw2);
        }
        STAMP(11);
        acc = zero16();
        acc = wave_gemm3_32s<0, 8>(G2S, w2t, swz_off(opaque(j), h), acc);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (hmask >> i) & 1u ? acc[i] : 0.f;
        acc_to_tile(Ot, acc, w, j, h);
        STAMP(12);
        __syncthreads();                                                     // (E) g1 tile complete
        STAMP(13);
        // ---- 4: g1 rows to memory, receiver-segmented sums
        // (the next tile's projections are taken as arrived before the stores below: one in-order counter for loads and
        // stores, so a wait for them placed after the stores would wait for the stores as well)
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(gs[g]), "+v"(gr[g]));
        const unsigned mask = (unsigned)__ballot(valid && ra != rb);         // bit r: row r ends its segment
        const unsigned gmask = (mask >> (16 * sg)) & 0xFFFFu;
        float xr[16], part = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) xr[r] = Ot[(16 * sg + r) * (H + 4) + sc];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            part += xr[r];
            xr[r] = part;
            if ((gmask >> r) & 1u) part = 0.f;
        }
        float cin = tcarry[sc];
        if (sg == 0) cout[sc] = gmask ? part : cin + part;
        f32x4 y[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) y[it] = *(const f32x4*)(Ot + toff(srow + 8 * it, sq, H / 4));
        STAMP(14);
        __syncthreads();                                                     // (S) cout visible
        if (sg == 1) cin = cout[sc];
        {
            float* go = g1_g + (unsigned)(rel * H);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = srow + 8 * it;
                if (row < nvalid) *(f32x4*)(go + (unsigned)(row * H + 4 * sq)) = y[it];
            }
        }
        bool open = true;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if ((gmask >> r) & 1u) {
                const int node = __builtin_amdgcn_readlane(ra, 16 * sg + r);
                a.dpr[(unsigned)(node * H + sc)] = open ? cin + xr[r] : xr[r];
                open = false;
            }
        }
        if (sg == 1) tcarry[sc] = gmask ? part : cin + part;
        STAMP(15);
        __syncthreads();                                                     // (F) Ot / tcarry settled
        s_cur = s_nxt; r_cur = r_nxt;
    }
    STAMP_FLUSH(w == 0 ? 0 : (w == 2 ? 1 : -1))
    }
    // ---- partial sums of this workgroup
    float* p = a.partial + (size_t)blockIdx.x * PARTIAL_E;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int f = 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            p[f * H + 32 * kb + j] = dw2[kb][reg];
        }
#pragma unroll
    for (int q = 0; q < 3; ++q) {                    // column sums of the per-lane (per-row) accumulators
        __syncthreads();
        acc_to_tile(Ot, q == 0 ? db2 : (q == 1 ? dgam : dbet), w, j, h);
        __syncthreads();
        if (t < H) {
            float sum = 0.f;
#pragma unroll 8
            for (int r = 0; r < TM; ++r) sum += Ot[r * (H + 4) + t];
            p[H * H + q * H + t] = sum;
        }
    }
}

// Sums one workgroup-partial layout over the blocks, in block order: a matrix of `mcount` floats
// (element i -> out[(i / H) * ld + (i % H)]; mcount may be 0) followed by `nvec` vectors of H floats
// (vector v -> vec[v]).  Eight independent running sums per thread (blocks b = k mod 8) keep eight
// loads in flight; they are combined pairwise in a fixed order, so the result does not depend on the launch.
struct ReduceArgs { const float* partial; int nblocks; int stride; int mcount; int ld; float* out; int nvec; float* vec[3]; };

__global__ __launch_bounds__(64) void reduce_partials_kernel(ReduceArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.mcount + a.nvec * H) return;
    const float* p = a.partial + i;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= a.nblocks; b += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += p[(size_t)(b + k) * a.stride];
    }
    for (int k = 0; b < a.nblocks; ++b, ++k) s[k] += p[(size_t)b * a.stride];
    const float sum = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    if (i < a.mcount) a.out[(size_t)(i / H) * a.ld + (i % H)] = sum;
    else a.vec[(i - a.mcount) / H][(i - a.mcount) % H] = sum;
}

// Many such reductions in one launch (blockIdx.y = record): the records travel by value in the kernel arguments.
constexpr int REDUCE_MANY_MAX = 48;
struct ReduceManyArgs { MgnReduceRec r[REDUCE_MANY_MAX]; };
static_assert(sizeof(MgnReduceRec) == 64 && sizeof(ReduceManyArgs) <= 3072, "records must fit the kernel argument segment");

__global__ __launch_bounds__(64) void reduce_many_kernel(ReduceManyArgs a) {
    const MgnReduceRec& r = a.r[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r.mcount + r.nvec * H) return;
    const float* p = r.partial + i;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= r.nblocks; b += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += p[(size_t)(b + k) * r.stride];
    }
    for (int k = 0; b < r.nblocks; ++b, ++k) s[k] += p[(size_t)b * r.stride];
    const float sum = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));     // as reduce_partials_kernel
    if (i < r.mcount) r.out[(size_t)(i / H) * r.ld + (i % H)] = sum;
    else r.vec[(i - r.mcount) / H][(i - r.mcount) % H] = sum;
}

// ------------------------------------------------------------------------ segmented row sums
__global__ __launch_bounds__(NT) void segment_sum_kernel(const float* __restrict__ src, const int* __restrict__ rowptr,
                                                         const int* __restrict__ order, float* __restrict__ out,
                                                         long long n_seg) {
    const int sq = threadIdx.x & 31;
    for (long long n = (long long)blockIdx.x * 8 + (threadIdx.x >> 5); n < n_seg; n += (long long)gridDim.x * 8) {
        const int k0 = rowptr[n], k1 = rowptr[n + 1];
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = k0; k < k1; ++k) {
            const size_t row = order ? (size_t)order[k] : (size_t)k;
            s += *(const f32x4*)(src + row * H + 4 * sq);
        }
        *(f32x4*)(out + n * H + 4 * sq) = s;
    }
}

// the same with a weight per source row (rows in segment order): out[n] = sum of weight[k] * src[k] (attention option,
// forward of the path that keeps every Linear as its own launch)
__global__ __launch_bounds__(NT) void segment_sum_weighted_kernel(const float* __restrict__ src, const int* __restrict__ rowptr,
                                                                  const float* __restrict__ weight, float* __restrict__ out,
                                                                  long long n_seg) {
    const int sq = threadIdx.x & 31;
    for (long long n = (long long)blockIdx.x * 8 + (threadIdx.x >> 5); n < n_seg; n += (long long)gridDim.x * 8) {
        const int k0 = rowptr[n], k1 = rowptr[n + 1];
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = k0; k < k1; ++k) s += *(const f32x4*)(src + (size_t)k * H + 4 * sq) * weight[k];
        *(f32x4*)(out + n * H + 4 * sq) = s;
    }
}

// ------------------------------------------------------------------------ attention option, backward
// Forward (core/meshgraphnet.py:100-108): alpha_e = leaky_relu(<a[s_e], a[r_e]>), w_e = alpha_e / max(sum of alpha over
// the in-edges of r_e, 1e-6), aggr[r] = sum of w_e * m_e with m_e the block's pre-residual edge output.
//
// Per edge (32 lanes, 16 bytes of each row per lane): gw_e = <g_aggr[r_e], m_e> with m_e = e_out - e_in (e_out alone when
// the caller kept m_e itself and passes no e_in), and the
// gradient reaching m_e from the sum, w_e * g_aggr[r_e], added to the upstream gradient of e_out (g_e_in, or nothing).
__global__ __launch_bounds__(NT) void attn_edge_bwd_kernel(const float* __restrict__ g_aggr, const int* __restrict__ recv,
                                                           const float* __restrict__ e_out, const float* __restrict__ e_in,
                                                           const float* __restrict__ w, const float* __restrict__ g_e_in,
                                                           float* __restrict__ gw, float* __restrict__ g_e_out, long long n_edges) {
    const int l = threadIdx.x & 31;
    for (long long e = ((long long)blockIdx.x * NT + threadIdx.x) >> 5; e < n_edges; e += ((long long)gridDim.x * NT) >> 5) {
        const f32x4 ga = *(const f32x4*)(g_aggr + (size_t)recv[e] * H + 4 * l);
        f32x4 m = *(const f32x4*)(e_out + (size_t)e * H + 4 * l);
        if (e_in) m -= *(const f32x4*)(e_in + (size_t)e * H + 4 * l);
        float d = (ga[0] * m[0] + ga[1] * m[1]) + (ga[2] * m[2] + ga[3] * m[3]);
#pragma unroll
        for (int k = 16; k >= 1; k >>= 1) d += __shfl_xor(d, k, 32);
        if (l == 0) gw[e] = d;
        f32x4 o = ga * w[e];
        if (g_e_in) o += *(const f32x4*)(g_e_in + (size_t)e * H + 4 * l);
        *(f32x4*)(g_e_out + (size_t)e * H + 4 * l) = o;
    }
}

// Per receiver: gradient of the dot products from the gradient of the weights.  With S = sum of alpha, D = max(S, 1e-6) and
// deg the receiver's in-degree:  g_alpha_e = gw_e / D - deg * [S >= 1e-6] * (sum of gw * alpha) / D^2, times the slope of the
// leaky ReLU (1 where alpha > 0, else 0.01).  The factor deg is what autograd computes for the reference's in-place
// ``alpha[col] = alpha[col].clamp(min=1e-6)`` (core/meshgraphnet.py:106): the assignment through an index with repeats hands
// the gradient of a receiver's denominator to each of its deg copies, and all of them flow back into the one sum.
__global__ void attn_alpha_bwd_kernel(const int* __restrict__ rowptr, const float* __restrict__ alpha,
                                      const float* __restrict__ gw, float* __restrict__ gdot, int n_nodes) {
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < n_nodes; n += gridDim.x * blockDim.x) {
        const int b = rowptr[n], e = rowptr[n + 1];
        float s = 0.f, t = 0.f;
        for (int k = b; k < e; ++k) { s += alpha[k]; t += gw[k] * alpha[k]; }
        const float d = fmaxf(s, 1e-6f);
        const float c = (s >= 1e-6f) ? (float)(e - b) * t / (d * d) : 0.f;
        for (int k = b; k < e; ++k) gdot[k] = (gw[k] / d - c) * (alpha[k] > 0.f ? 1.0f : 0.01f);
    }
}

// Per node v: out[v] (+)= sum over its edge list (rowptr, optionally through `order`) of gdot[e] * a[other[e]]:
// with the receiver CSR and other = sender this is the gradient of a[receiver], with the sender view and other = receiver
// that of a[sender].  Fixed summation order.
__global__ __launch_bounds__(NT) void attn_node_grad_kernel(const float* __restrict__ a, const int* __restrict__ rowptr,
                                                            const int* __restrict__ order, const int* __restrict__ other,
                                                            const float* __restrict__ gdot, float* __restrict__ out,
                                                            int accumulate, long long n_nodes) {
    const int sq = threadIdx.x & 31;
    for (long long n = (long long)blockIdx.x * 8 + (threadIdx.x >> 5); n < n_nodes; n += (long long)gridDim.x * 8) {
        const int k0 = rowptr[n], k1 = rowptr[n + 1];
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = k0; k < k1; ++k) {
            const size_t e = order ? (size_t)order[k] : (size_t)k;
            s += *(const f32x4*)(a + (size_t)other[e] * H + 4 * sq) * gdot[e];
        }
        f32x4* o = (f32x4*)(out + n * H + 4 * sq);
        *o = accumulate ? *o + s : s;
    }
}

// ------------------------------------------------------------------------ features, loss
__global__ void pad_features_kernel(const float* __restrict__ feat, const float* __restrict__ mean_std,
                                    float* __restrict__ out, long long rows, int F) {
    const long long total = rows * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / H; const int k = (int)(i - r * H);
        float v = 0.f;
        if (k < F) {
            v = feat[r * F + k];
            if (mean_std) v = (v - mean_std[k]) / mean_std[F + k];
        }
        out[i] = v;
    }
}

// per component c: sum over t and over nodes without the Dirichlet flag of column c + 1 of
// (decoded[n][3t + c] * (t + 1) - target[t][n][c])^2; partial[block][c] in double
__global__ __launch_bounds__(NT) void loss_fwd_kernel(const float* __restrict__ dec, const float* __restrict__ tgt,
                                                      const float* __restrict__ node_type, int T, long long n_nodes,
                                                      double* __restrict__ partial) {
    __shared__ double sh[NT];
    double s[3] = {0.0, 0.0, 0.0};
    const long long per = (n_nodes + gridDim.x - 1) / gridDim.x;
    const long long n0 = (long long)blockIdx.x * per, n1 = min(n_nodes, n0 + per);
    for (long long n = n0 + threadIdx.x; n < n1; n += NT)
        for (int c = 0; c < 3; ++c) {
            if (node_type[n * 5 + c + 1] == 1.0f) continue;
            for (int tt = 0; tt < T; ++tt) {
                const float d = dec[n * H + 3 * tt + c] * (float)(tt + 1) - tgt[((size_t)tt * n_nodes + n) * 3 + c];
                s[c] += (double)(d * d);
            }
        }
    for (int c = 0; c < 3; ++c) {
        sh[threadIdx.x] = s[c];
        __syncthreads();
        for (int m = NT / 2; m >= 1; m >>= 1) {
            if (threadIdx.x < m) sh[threadIdx.x] += sh[threadIdx.x + m];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[(size_t)blockIdx.x * 3 + c] = sh[0];
        __syncthreads();
    }
}

__global__ void loss_final_kernel(const double* __restrict__ partial, int nblocks, int T, float* __restrict__ out3) {
    const int c = threadIdx.x;
    if (c >= 3) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * 3 + c];
    out3[c] = (float)(s / (double)T);
}

// gradient of sum_c gsc[c] * loss_c with respect to the padded decoder output [N,128]
__global__ void loss_bwd_kernel(const float* __restrict__ dec, const float* __restrict__ tgt,
                                const float* __restrict__ node_type, const float* __restrict__ gsc, int T,
                                long long n_nodes, float* __restrict__ gdec) {
    const long long total = n_nodes * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / H; const int k = (int)(i - n * H);
        float g = 0.f;
        if (k < 3 * T) {
            const int tt = k / 3, c = k - 3 * tt;
            if (node_type[n * 5 + c + 1] != 1.0f) {
                const float sc = (float)(tt + 1);
                const float d = dec[i] * sc - tgt[((size_t)tt * n_nodes + n) * 3 + c];
                g = gsc[c] * (2.0f / (float)T) * d * sc;
            }
        }
        gdec[i] = g;
    }
}

// ------------------------------------------------------------------------ Adam over many tensors
// One workgroup per table entry (a chunk of one parameter tensor): p, g, m, v pointers and a count.
// torch.optim.Adam semantics (L2 weight decay added to the gradient, bias-corrected moments):
//   g += wd * p;  m += (1 - b1) * (g - m);  v = b2 * v + (1 - b2) * g * g
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
struct AdamEntry { float* p; const float* g; float* m; float* v; long long n; };

__global__ __launch_bounds__(NT) void adam_kernel(const AdamEntry* __restrict__ table, float step_size, float inv_sqrt_bc2,
                                                  float b1, float b2, float eps, float wd) {
    const AdamEntry e = table[blockIdx.x];
    for (long long i = threadIdx.x; i < e.n; i += NT) {
        const float p = e.p[i];
        const float g = fmaf(wd, p, e.g[i]);
        const float m = fmaf(1.0f - b1, g - e.m[i], e.m[i]);
        const float v = fmaf(1.0f - b2, g * g, b2 * e.v[i]);
        e.m[i] = m; e.v[i] = v;
        e.p[i] = p - step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
    }
}

// The same update with its six scalars in device memory (a captured launch sequence cannot take new kernel
// arguments from step to step): hyper = {lr / (1 - b1^t), 1 / sqrt(1 - b2^t), b1, b2, eps, weight decay}.
__global__ __launch_bounds__(NT) void adam_dev_kernel(const AdamEntry* __restrict__ table, const float* __restrict__ hyper) {
    const AdamEntry e = table[blockIdx.x];
    const float step_size = hyper[0], inv_sqrt_bc2 = hyper[1], b1 = hyper[2], b2 = hyper[3], eps = hyper[4], wd = hyper[5];
    for (long long i = threadIdx.x; i < e.n; i += NT) {
        const float p = e.p[i];
        const float g = fmaf(wd, p, e.g[i]);
        const float m = fmaf(1.0f - b1, g - e.m[i], e.m[i]);
        const float v = fmaf(1.0f - b2, g * g, b2 * e.v[i]);
        e.m[i] = m; e.v[i] = v;
        e.p[i] = p - step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
    }
}

// Deferral of the calling thread (mgn_train_defer_reductions): while a record list is installed, the backward entry
// points append their reduction to it instead of launching it.
struct DeferState { MgnReduceRec* recs; int capacity; int32_t* count; };
thread_local DeferState t_defer = {nullptr, 0, nullptr};

// 0 on success; a full record list is an error of the caller (the reduction is then launched at once, so no result is lost)
int launch_reduce(const float* partial, int nblocks, int stride, int mcount, int ld, float* out, int nvec, float* v0,
                  float* v1, float* v2, hipStream_t s) {
    if (t_defer.recs && *t_defer.count < t_defer.capacity) {
        t_defer.recs[(*t_defer.count)++] = MgnReduceRec{partial, out, {v0, v1, v2}, nblocks, stride, mcount, ld, nvec, 0};
        return 0;
    }
    ReduceArgs r{partial, nblocks, stride, mcount, ld, out, nvec, {v0, v1, v2}};
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((mcount + nvec * H + 63) / 64), dim3(64), 0, s, r);
    return t_defer.recs ? 1 : 0;
}

int train_grid(long long rows, int per_cu) { return grid_for((rows + TM - 1) / TM, per_cu); }

int stream_grid(long long items, int per_block) {
    long long g = (items + per_block - 1) / per_block;
    const long long cap = (long long)num_cus() * 8;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

#ifdef MGN_PHASE_STAMPS
extern "C" int mgn_debug_read_train_stamps(unsigned long long* out32, int reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

int mgn_train_read_range_flag(int* tripped, int reset, void* stream) {
    return read_range_flag(tripped, reset, (hipStream_t)stream, "mgn_range_check");
}

int mgn_train_poll_range_flag(int* host_word, void* stream) {
    return poll_range_flag(host_word, (hipStream_t)stream, "mgn_range_poll");
}

extern "C" int64_t mgn_train_scratch_bytes(void) {
    return (int64_t)TRAIN_BLOCKS_MAX * PARTIAL * (int64_t)sizeof(float);
}

extern "C" int64_t mgn_train_partial_bytes(int32_t kind, int64_t rows) {
    if (rows < 0 || kind < 0 || kind > 2) return -1;
    long long g, stride;
    if (kind == 0) {                                           // mgn_train_linear_bwd (either matrix path: the larger grid)
        g = train_grid(rows, 2); stride = PARTIAL;
        if (g > TRAIN_BLOCKS_MAX) g = TRAIN_BLOCKS_MAX;
    } else if (kind == 1) {                                    // mgn_train_ln_bwd
        g = stream_grid(rows, 64); stride = 2 * H;
        if (g > TRAIN_BLOCKS_MAX) g = TRAIN_BLOCKS_MAX;
    } else {                                                   // mgn_train_edge_bwd (rows = edges)
        g = train_grid(rows, 1); stride = PARTIAL_E;
    }
    return (int64_t)(g * stride * (long long)sizeof(float));
}

extern "C" int mgn_train_defer_reductions(MgnReduceRec* recs, int32_t capacity, int32_t* count) {
    if (recs == nullptr) { t_defer = DeferState{nullptr, 0, nullptr}; return 0; }
    MGN_REQUIRE(capacity > 0 && count && *count >= 0 && *count <= capacity, "mgn_train_defer_reductions");
    t_defer = DeferState{recs, (int)capacity, count};
    return 0;
}

extern "C" int mgn_train_reduce_many(const MgnReduceRec* recs, int32_t n, void* stream) {
    MGN_REQUIRE(n >= 0 && (n == 0 || recs), "mgn_train_reduce_many");
    if (n == 0) return 0;
    for (int i = 0; i < n; ++i) {
        const MgnReduceRec& r = recs[i];
        MGN_REQUIRE(r.partial && r.nblocks > 0 && r.stride > 0 && r.mcount >= 0 && r.nvec >= 0 && r.nvec <= 3, "mgn_train_reduce_many");
        MGN_REQUIRE(r.mcount == 0 || (r.out && r.ld >= H), "mgn_train_reduce_many");
        for (int v = 0; v < r.nvec; ++v) MGN_REQUIRE(r.vec[v], "mgn_train_reduce_many");
    }
    for (int first = 0; first < n; first += REDUCE_MANY_MAX) {
        const int cnt = n - first < REDUCE_MANY_MAX ? n - first : REDUCE_MANY_MAX;
        ReduceManyArgs a;
        int widest = 0;
        for (int i = 0; i < cnt; ++i) {
            a.r[i] = recs[first + i];
            const int items = a.r[i].mcount + a.r[i].nvec * H;
            widest = items > widest ? items : widest;
        }
        for (int i = cnt; i < REDUCE_MANY_MAX; ++i) a.r[i] = MgnReduceRec{nullptr, nullptr, {nullptr, nullptr, nullptr}, 0, 0, 0, 0, 0, 0};
        if (widest == 0) continue;
        hipLaunchKernelGGL(reduce_many_kernel, dim3((widest + 63) / 64, cnt), dim3(64), 0, (hipStream_t)stream, a);
    }
    return check_launch("mgn_train_reduce_many");
}

extern "C" int mgn_train_linear_fwd(const float* A, const float* B, const float* W, int32_t ldw, int32_t koff_a,
                                    int32_t koff_b, const float* bias, const float* P, const int32_t* idxP,
                                    const float* Q, const int32_t* idxQ, int32_t relu, float* out, int64_t rows,
                                    void* stream) {
    MGN_REQUIRE(rows >= 0 && rows < (1ll << 31), "mgn_train_linear_fwd");
    if (rows == 0) return 0;
    MGN_REQUIRE(A && W && out && ldw >= H && koff_a >= 0 && koff_a + H <= ldw, "mgn_train_linear_fwd");
    MGN_REQUIRE(!B || (koff_b >= 0 && koff_b + H <= ldw), "mgn_train_linear_fwd");
    MGN_REQUIRE((!P && !Q) || (P && Q && idxP && idxQ), "mgn_train_linear_fwd");
    MGN_REQUIRE(ldw % 4 == 0 && koff_a % 4 == 0 && koff_b % 4 == 0, "mgn_train_linear_fwd");
    LinFwdArgs a{A, B, W, ldw, koff_a, koff_b, bias, P, idxP, Q, idxQ, relu, out, rows};
    const int g = train_grid(rows, 2);
    const bool half = mgn_matmul_mode() == 2;
    hipStream_t s = (hipStream_t)stream;
    if (B) {
        if (half) hipLaunchKernelGGL((train_linear_fwd_kernel<true, true>), dim3(g), dim3(NT), 0, s, a);
        else hipLaunchKernelGGL((train_linear_fwd_kernel<true, false>), dim3(g), dim3(NT), 0, s, a);
    } else {
        if (half) hipLaunchKernelGGL((train_linear_fwd_kernel<false, true>), dim3(g), dim3(NT), 0, s, a);
        else hipLaunchKernelGGL((train_linear_fwd_kernel<false, false>), dim3(g), dim3(NT), 0, s, a);
    }
    return check_launch("mgn_train_linear_fwd");
}

extern "C" int mgn_train_ln_fwd(const float* v, const float* gamma, const float* beta, const float* res, float* y,
                                float* yres, int64_t rows, void* stream) {
    MGN_REQUIRE(rows >= 0, "mgn_train_ln_fwd");
    if (rows == 0) return 0;
    MGN_REQUIRE(v && gamma && beta && (y || yres) && (!yres || res), "mgn_train_ln_fwd");
    hipLaunchKernelGGL(train_ln_fwd_kernel, dim3(stream_grid(rows, 8)), dim3(NT), 0, (hipStream_t)stream, v, gamma,
                       beta, res, y, yres, (long long)rows);
    return check_launch("mgn_train_ln_fwd");
}

extern "C" int mgn_train_ln_bwd(const float* G, const float* Gg, const int32_t* idx, const float* v,
                                const float* gamma, int32_t relu_mask, float* gpre, float* dgamma, float* dbeta,
                                void* scratch, int64_t rows, void* stream) {
    MGN_REQUIRE(rows > 0, "mgn_train_ln_bwd");
    MGN_REQUIRE((G || Gg) && (!Gg || idx) && v && gamma && gpre && dgamma && dbeta && scratch, "mgn_train_ln_bwd");
    int g = stream_grid(rows, 64);
    if (g > TRAIN_BLOCKS_MAX) g = TRAIN_BLOCKS_MAX;
    hipLaunchKernelGGL(train_ln_bwd_kernel, dim3(g), dim3(NT), 0, (hipStream_t)stream, G, Gg, idx, v, gamma,
                       relu_mask, gpre, (float*)scratch, (long long)rows);
    if (launch_reduce((const float*)scratch, g, 2 * H, 0, 0, nullptr, 2, dgamma, dbeta, nullptr, (hipStream_t)stream))
        { mgn_set_error("mgn_train_ln_bwd: the deferred-reduction list is full"); return MGN_E_CAP; }
    return check_launch("mgn_train_ln_bwd");
}

extern "C" int mgn_train_linear_bwd(const float* G, const float* A, const float* W, int32_t ldw, int32_t koff,
                                    const float* mask, const float* res, float* GA, float* dW, int32_t lddw,
                                    float* db, void* scratch, int64_t rows, void* stream) {
    MGN_REQUIRE(rows > 0 && rows < (1ll << 31), "mgn_train_linear_bwd");
    MGN_REQUIRE(G && (GA || dW), "mgn_train_linear_bwd");
    MGN_REQUIRE(!GA || (W && ldw >= H && koff >= 0 && koff + H <= ldw), "mgn_train_linear_bwd");
    MGN_REQUIRE(!dW || (A && scratch && lddw >= H), "mgn_train_linear_bwd");
    MGN_REQUIRE(!db || dW, "mgn_train_linear_bwd");
    MGN_REQUIRE(ldw % 4 == 0 || !GA, "mgn_train_linear_bwd");
    LinBwdArgs a{G, A, W, ldw, koff, mask, res, GA, (float*)scratch, rows};
    const bool split = mgn_matmul_mode() >= 1;           // bf16 split products; "f32" keeps the fp32 matrix pipe
    int g = train_grid(rows, split ? 1 : 2);
    if (g > TRAIN_BLOCKS_MAX) g = TRAIN_BLOCKS_MAX - TRAIN_BLOCKS_MAX % 8;
    hipStream_t s = (hipStream_t)stream;
    if (split) {
        if (GA && dW) hipLaunchKernelGGL((train_linear_bwd3_kernel<true, true>), dim3(g), dim3(NT), 0, s, a);
        else if (GA) hipLaunchKernelGGL((train_linear_bwd3_kernel<true, false>), dim3(g), dim3(NT), 0, s, a);
        else hipLaunchKernelGGL((train_linear_bwd3_kernel<false, true>), dim3(g), dim3(NT), 0, s, a);
    } else if (GA && dW) hipLaunchKernelGGL((train_linear_bwd_kernel<true, true>), dim3(g), dim3(NT), 0, s, a);
    else if (GA) hipLaunchKernelGGL((train_linear_bwd_kernel<true, false>), dim3(g), dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((train_linear_bwd_kernel<false, true>), dim3(g), dim3(NT), 0, s, a);
    if (dW && launch_reduce((const float*)scratch, g, PARTIAL, PARTIAL_W, lddw, dW, db ? 1 : 0, db, nullptr, nullptr, s))
        { mgn_set_error("mgn_train_linear_bwd: the deferred-reduction list is full"); return MGN_E_CAP; }
    return check_launch("mgn_train_linear_bwd");
}

extern "C" int mgn_train_edge_bwd(const float* e_in, const float* g_e, const float* g_aggr, const float* ps,
                                  const float* pr, const int32_t* sender, const int32_t* recv, const int32_t* rowptr,
                                  const float* w0, const float* w1, const float* b1, const float* ln_w, float* g1,
                                  float* dpr, float* dW1, float* db1, float* dgamma, float* dbeta, void* scratch,
                                  int64_t n_nodes, int64_t n_edges, void* stream) {
    MGN_REQUIRE(n_nodes > 0 && n_edges > 0 && n_nodes < (1ll << 24) && n_edges < (1ll << 24), "mgn_train_edge_bwd");
    MGN_REQUIRE(e_in && g_aggr && ps && pr && sender && recv && rowptr && w0 && w1 && b1 && ln_w, "mgn_train_edge_bwd");
    MGN_REQUIRE(g1 && dpr && dW1 && db1 && dgamma && dbeta && scratch, "mgn_train_edge_bwd");
    hipStream_t s = (hipStream_t)stream;
    // receivers without in-edges are never visited by the scan
    if (hipMemsetAsync(dpr, 0, (size_t)n_nodes * H * sizeof(float), s) != hipSuccess) return check_launch("mgn_train_edge_bwd");
    EdgeBwdArgs a{e_in, g_e, g_aggr, ps, pr, sender, recv, rowptr, w0, w1, b1, ln_w, g1, dpr, (float*)scratch,
                  (int)n_nodes, (int)n_edges};
    const int g = train_grid(n_edges, 1);
    hipLaunchKernelGGL(train_edge_bwd_kernel, dim3(g), dim3(NT), 0, s, a);
    if (launch_reduce((const float*)scratch, g, PARTIAL_E, PARTIAL_W, H, dW1, 3, db1, dgamma, dbeta, s))
        { mgn_set_error("mgn_train_edge_bwd: the deferred-reduction list is full"); return MGN_E_CAP; }
    return check_launch("mgn_train_edge_bwd");
}

extern "C" int mgn_segment_sum(const float* src, const int32_t* rowptr, const int32_t* order, float* out,
                               int64_t n_segments, void* stream) {
    MGN_REQUIRE(n_segments >= 0, "mgn_segment_sum");
    if (n_segments == 0) return 0;
    MGN_REQUIRE(src && rowptr && out, "mgn_segment_sum");
    hipLaunchKernelGGL(segment_sum_kernel, dim3(stream_grid(n_segments, 8)), dim3(NT), 0, (hipStream_t)stream, src,
                       rowptr, order, out, (long long)n_segments);
    return check_launch("mgn_segment_sum");
}

extern "C" int mgn_segment_sum_weighted(const float* src, const int32_t* rowptr, const float* weight, float* out,
                                        int64_t n_segments, void* stream) {
    MGN_REQUIRE(n_segments >= 0 && n_segments < (1ll << 31), "mgn_segment_sum_weighted");
    if (n_segments == 0) return 0;
    MGN_REQUIRE(src && rowptr && weight && out, "mgn_segment_sum_weighted");
    hipLaunchKernelGGL(segment_sum_weighted_kernel, dim3(stream_grid(n_segments, 8)), dim3(NT), 0, (hipStream_t)stream, src,
                       rowptr, weight, out, (long long)n_segments);
    return check_launch("mgn_segment_sum_weighted");
}

extern "C" int mgn_train_attn_edge_bwd(const float* g_aggr, const int32_t* csr_recv, const float* e_out, const float* e_in,
                                       const float* edge_weight, const float* g_e_in, float* g_weight, float* g_e_out,
                                       int64_t n_edges, void* stream) {
    MGN_REQUIRE(n_edges >= 0 && n_edges < (1ll << 31), "mgn_train_attn_edge_bwd");
    if (n_edges == 0) return 0;
    MGN_REQUIRE(g_aggr && csr_recv && e_out && edge_weight && g_weight && g_e_out, "mgn_train_attn_edge_bwd");
    hipLaunchKernelGGL(attn_edge_bwd_kernel, dim3(stream_grid(n_edges * 32, NT)), dim3(NT), 0, (hipStream_t)stream, g_aggr,
                       csr_recv, e_out, e_in, edge_weight, g_e_in, g_weight, g_e_out, (long long)n_edges);
    return check_launch("mgn_train_attn_edge_bwd");
}

extern "C" int mgn_train_attn_alpha_bwd(const int32_t* rowptr, const float* alpha, const float* g_weight, float* g_dot,
                                        int64_t n_nodes, void* stream) {
    MGN_REQUIRE(n_nodes >= 0 && n_nodes < (1ll << 31), "mgn_train_attn_alpha_bwd");
    if (n_nodes == 0) return 0;
    MGN_REQUIRE(rowptr && alpha && g_weight && g_dot, "mgn_train_attn_alpha_bwd");
    hipLaunchKernelGGL(attn_alpha_bwd_kernel, dim3(stream_grid(n_nodes, NT)), dim3(NT), 0, (hipStream_t)stream, rowptr, alpha,
                       g_weight, g_dot, (int)n_nodes);
    return check_launch("mgn_train_attn_alpha_bwd");
}

extern "C" int mgn_train_attn_node_grad(const float* a_proj, const int32_t* rowptr, const int32_t* order,
                                        const int32_t* other, const float* g_dot, float* out, int32_t accumulate,
                                        int64_t n_nodes, void* stream) {
    MGN_REQUIRE(n_nodes >= 0, "mgn_train_attn_node_grad");
    if (n_nodes == 0) return 0;
    MGN_REQUIRE(a_proj && rowptr && other && g_dot && out, "mgn_train_attn_node_grad");
    hipLaunchKernelGGL(attn_node_grad_kernel, dim3(stream_grid(n_nodes, 8)), dim3(NT), 0, (hipStream_t)stream, a_proj, rowptr,
                       order, other, g_dot, out, (int)accumulate, (long long)n_nodes);
    return check_launch("mgn_train_attn_node_grad");
}

extern "C" int mgn_train_pad_features(const float* feat, int32_t n_feat, const float* mean_std, float* out,
                                      int64_t rows, void* stream) {
    MGN_REQUIRE(rows >= 0 && n_feat >= 1 && n_feat <= H, "mgn_train_pad_features");
    if (rows == 0) return 0;
    MGN_REQUIRE(feat && out, "mgn_train_pad_features");
    hipLaunchKernelGGL(pad_features_kernel, dim3(stream_grid(rows * H, NT * 4)), dim3(NT), 0, (hipStream_t)stream, feat,
                       mean_std, out, (long long)rows, n_feat);
    return check_launch("mgn_train_pad_features");
}

extern "C" int mgn_train_loss_fwd(const float* decoded, const float* target_norm, const float* node_type,
                                  int32_t time_dim, int64_t n_nodes, float* loss3, void* scratch, void* stream) {
    MGN_REQUIRE(n_nodes > 0 && time_dim >= 1 && 3 * time_dim <= MGN_MAX_DEC_OUT, "mgn_train_loss_fwd");
    MGN_REQUIRE(decoded && target_norm && node_type && loss3 && scratch, "mgn_train_loss_fwd");
    int g = stream_grid(n_nodes, NT * 4);
    if (g > TRAIN_BLOCKS_MAX) g = TRAIN_BLOCKS_MAX;
    hipLaunchKernelGGL(loss_fwd_kernel, dim3(g), dim3(NT), 0, (hipStream_t)stream, decoded, target_norm, node_type,
                       time_dim, (long long)n_nodes, (double*)scratch);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)scratch, g, time_dim,
                       loss3);
    return check_launch("mgn_train_loss_fwd");
}

extern "C" int mgn_train_loss_bwd(const float* decoded, const float* target_norm, const float* node_type,
                                  const float* grad3, int32_t time_dim, int64_t n_nodes, float* gdecoded, void* stream) {
    MGN_REQUIRE(n_nodes > 0 && time_dim >= 1 && 3 * time_dim <= MGN_MAX_DEC_OUT, "mgn_train_loss_bwd");
    MGN_REQUIRE(decoded && target_norm && node_type && grad3 && gdecoded, "mgn_train_loss_bwd");
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(stream_grid(n_nodes * H, NT * 4)), dim3(NT), 0, (hipStream_t)stream, decoded,
                       target_norm, node_type, grad3, time_dim, (long long)n_nodes, gdecoded);
    return check_launch("mgn_train_loss_bwd");
}

extern "C" int mgn_adam_step(const void* table_dev, int32_t n_entries, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int64_t step, void* stream) {
    MGN_REQUIRE(n_entries >= 0 && step >= 1, "mgn_adam_step");
    if (n_entries == 0) return 0;
    MGN_REQUIRE(table_dev && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "mgn_adam_step");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(n_entries), dim3(NT), 0, (hipStream_t)stream, (const AdamEntry*)table_dev,
                       (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, weight_decay);
    return check_launch("mgn_adam_step");
}

extern "C" int mgn_adam_hyper(float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, float* hyper6) {
    MGN_REQUIRE(hyper6 && step >= 1 && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "mgn_adam_hyper");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hyper6[0] = (float)((double)lr / bc1); hyper6[1] = (float)(1.0 / sqrt(bc2));
    hyper6[2] = beta1; hyper6[3] = beta2; hyper6[4] = eps; hyper6[5] = weight_decay;
    return 0;
}

extern "C" int mgn_adam_step_dev(const void* table_dev, int32_t n_entries, const float* hyper_dev, void* stream) {
    MGN_REQUIRE(n_entries >= 0, "mgn_adam_step_dev");
    if (n_entries == 0) return 0;
    MGN_REQUIRE(table_dev && hyper_dev, "mgn_adam_step_dev");
    hipLaunchKernelGGL(adam_dev_kernel, dim3(n_entries), dim3(NT), 0, (hipStream_t)stream, (const AdamEntry*)table_dev, hyper_dev);
    return check_launch("mgn_adam_step_dev");
}
