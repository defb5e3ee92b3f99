// Device and host helpers shared by the kernel translation units of libmgn_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "mgn_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int H = MGN_H;    // latent width
constexpr int TM = 32;      // rows per tile
constexpr int NT = 256;     // threads per workgroup
constexpr float LN_EPS = 1e-5f;

// float offset of 16-byte chunk q of row `row` in a tile whose rows hold k4 chunks (+1 pad)
__device__ __forceinline__ constexpr int tile_floats(int k4) { return TM * (4 * k4 + 4); }
__device__ __forceinline__ int toff(int row, int q, int k4) { return row * (4 * k4 + 4) + 4 * q; }

// Register-resident weight slice: for output feature `feat`, the k values this lane
// feeds to the MFMAs of chunk c are W[feat][koff + 8c + 4h + 0..3].
template <int KC>
__device__ __forceinline__ void load_wfrag(f32x4 (&w)[KC], const float* __restrict__ W, int ldw, int koff,
                                           int feat, int h, bool valid = true) {
    const float* p = W + (size_t)feat * ldw + koff + 4 * h;
#pragma unroll
    for (int c = 0; c < KC; ++c) w[c] = valid ? *(const f32x4*)(p + 8 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
}

// acc[feature][row] += W[feature][k] * tile[row][k] over k = 8*KC values starting at chunk qoff.
// The LDS reads run two chunks ahead of the MFMAs that consume them.
template <int KC>
__device__ __forceinline__ f32x16 wave_gemm(const float* tile, int k4, int qoff, const f32x4 (&w)[KC],
                                            int j, int h, f32x16 acc) {
    const float* p = tile + toff(j, qoff + h, k4);
    f32x4 a0 = *(const f32x4*)(p), a1 = *(const f32x4*)(p + 8);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);          // 2 LDS reads
#pragma unroll
    for (int c = 0; c < KC; c += 2) {
        f32x4 n0 = a0, n1 = a1;
        if (c + 2 < KC) {
            n0 = *(const f32x4*)(p + 8 * (c + 2)); n1 = *(const f32x4*)(p + 8 * (c + 3));
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // next two chunks first ...
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][0], a0[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][1], a0[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][2], a0[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][3], a0[3], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c + 1][0], a1[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c + 1][1], a1[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c + 1][2], a1[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c + 1][3], a1[3], acc, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);      // ... then the 8 MFMAs of this pair
        a0 = n0; a1 = n1;
    }
    return acc;
}

__device__ __forceinline__ f32x16 acc_from_vec(const float* vec_lds, int w, int h) {
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b = *(const f32x4*)(vec_lds + 32 * w + 8 * g + 4 * h);
        acc[4 * g + 0] = b[0]; acc[4 * g + 1] = b[1]; acc[4 * g + 2] = b[2]; acc[4 * g + 3] = b[3];
    }
    return acc;
}

__device__ __forceinline__ f32x16 zero16() {
    return f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ f32x16 relu16(f32x16 v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
    return v;
}

// LayerNorm over the 128 features of every tile row.  Each lane holds 16 of a row's
// features and forms their mean m and centred sum of squares M2 locally (two passes over
// registers); the 8 (m, M2) pairs of a row (4 waves x 2 lane halves) meet in `red`, LDS
// float[2][8][TM], and combine exactly as a pooled variance:
//   mean = avg(m_k),  M2 = sum(M2_k) + 16 * sum((m_k - mean)^2),  var = M2 / 128 (biased).
// Contains one workgroup barrier; after it every wave has finished reading whatever tile
// fed the accumulators.
__device__ __forceinline__ void ln_local_stats(const f32x16& v, float* red, int w, int j, int h) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    const float m = s * (1.0f / 16.0f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float d = v[i] - m; q += d * d; }
    red[(2 * w + h) * TM + j] = m;
    red[8 * TM + (2 * w + h) * TM + j] = q;
}

__device__ __forceinline__ f32x16 ln_finish(f32x16 v, const float* red, int w, int j, int h,
                                            const float* gamma_lds, const float* beta_lds) {
    float mk[8], mean = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { mk[k] = red[k * TM + j]; mean += mk[k]; m2 += red[8 * TM + k * TM + j]; }
    mean *= 0.125f;
    float between = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float d = mk[k] - mean; between += d * d; }
    const float var = (m2 + 16.0f * between) * (1.0f / H);
    const float rstd = 1.0f / sqrtf(var + LN_EPS);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 ga = *(const f32x4*)(gamma_lds + 32 * w + 8 * g + 4 * h);
        const f32x4 be = *(const f32x4*)(beta_lds + 32 * w + 8 * g + 4 * h);
#pragma unroll
        for (int t = 0; t < 4; ++t) v[4 * g + t] = (v[4 * g + t] - mean) * rstd * ga[t] + be[t];
    }
    return v;
}

__device__ __forceinline__ f32x16 layernorm_rows(f32x16 v, float* red, int w, int j, int h,
                                                 const float* gamma_lds, const float* beta_lds) {
    ln_local_stats(v, red, w, j, h);
    __syncthreads();
    return ln_finish(v, red, w, j, h, gamma_lds, beta_lds);
}

// The same LayerNorm with the exchange buffer laid out per row: `red` is float[TM][LN_RED_LD]
// (row j: m[0..7], M2[0..7], pad -- rows 80 bytes apart so that the 16-byte reads of 16
// consecutive rows fall on distinct banks), read back with four 16-byte loads per lane.  Fewer
// instructions; wants 16 transient registers, so kernels at their register limit use the form above.
constexpr int LN_RED_LD = 20;
constexpr int LN_RED_FLOATS = TM * LN_RED_LD;

__device__ __forceinline__ float sum16_tree(const f32x16& v) {
    const float a = (v[0] + v[1]) + (v[2] + v[3]), b = (v[4] + v[5]) + (v[6] + v[7]);
    const float c = (v[8] + v[9]) + (v[10] + v[11]), d = (v[12] + v[13]) + (v[14] + v[15]);
    return (a + b) + (c + d);
}

// short dependency chains (pairwise sums) on purpose: the calling wave has nothing else to overlap with
__device__ __forceinline__ void ln_local_stats_rows(const f32x16& v, float* red, int w, int j, int h) {
    const float m = sum16_tree(v) * (1.0f / 16.0f);
    f32x16 d2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float d = v[i] - m; d2[i] = d * d; }
    const float q = sum16_tree(d2);
    red[j * LN_RED_LD + 2 * w + h] = m;
    red[j * LN_RED_LD + 8 + 2 * w + h] = q;
}

__device__ __forceinline__ f32x16 ln_finish_rows(f32x16 v, const float* red, int w, int j, int h,
                                                 const float* gamma_lds, const float* beta_lds, float eps = LN_EPS) {
    const f32x4 ma = *(const f32x4*)(red + j * LN_RED_LD), mb = *(const f32x4*)(red + j * LN_RED_LD + 4);
    const f32x4 qa = *(const f32x4*)(red + j * LN_RED_LD + 8), qb = *(const f32x4*)(red + j * LN_RED_LD + 12);
    const float mean = (((ma[0] + ma[1]) + (ma[2] + ma[3])) + ((mb[0] + mb[1]) + (mb[2] + mb[3]))) * 0.125f;
    const float m2 = ((qa[0] + qa[1]) + (qa[2] + qa[3])) + ((qb[0] + qb[1]) + (qb[2] + qb[3]));
    const f32x4 da = ma - mean, db = mb - mean;
    const float between = ((da[0] * da[0] + da[1] * da[1]) + (da[2] * da[2] + da[3] * da[3])) +
                          ((db[0] * db[0] + db[1] * db[1]) + (db[2] * db[2] + db[3] * db[3]));
    const float var = (m2 + 16.0f * between) * (1.0f / H);
    const float rstd = __builtin_amdgcn_rsqf(var + eps);      // v_rsq_f32, 1 ulp
    const float shift = -mean * rstd;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 ga = *(const f32x4*)(gamma_lds + 32 * w + 8 * g + 4 * h);
        const f32x4 be = *(const f32x4*)(beta_lds + 32 * w + 8 * g + 4 * h);
#pragma unroll
        for (int t = 0; t < 4; ++t) v[4 * g + t] = fmaf(fmaf(v[4 * g + t], rstd, shift), ga[t], be[t]);
    }
    return v;
}

// store the accumulator (row j, features 32w..32w+31) into a [TM][H] LDS tile
__device__ __forceinline__ void acc_to_tile(float* tile, f32x16 v, int w, int j, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *(f32x4*)(tile + toff(j, 8 * w + 2 * g + h, H / 4)) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
}

// ---------------------------------------------------------------------------- two-term fp16 split
// fp32 products on the fp16 matrix pipe.  Every fp32 operand is written as x = h1 + h2 + d with
// h1 = fp16(x), h2 = fp16(x - h1) (both round-to-nearest; x - h1 is exact in fp32), so |d| <= 2^-22 |x|
// while h2 is a normal fp16 (below 2^-14 the pipe still takes the subnormal h2, at 2^-25 absolute).
// A product is the three partial products h1*h1 + h1*h2 + h2*h1 accumulated in fp32 by
// v_mfma_f32_32x32x16_f16, smallest first; the dropped h2*h2 is <= 2^-22 of the product.
// Weights are pre-scaled by W_SCALE (a power of two, exact) so that their second term stays
// normal; the accumulator is multiplied back by W_UNSCALE where the bias is added.
// Operands must stay below the fp16 range (65504 / W_SCALE for weights): larger values become inf.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr float W_SCALE = 128.0f, W_UNSCALE = 1.0f / 128.0f;

__device__ __forceinline__ void split2_pair(float a, float b, unsigned& p1, unsigned& p2) {
    const f32x2 v = {a, b};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 r = v - __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(r, f16x2);
    p1 = __builtin_bit_cast(unsigned, h);
    p2 = __builtin_bit_cast(unsigned, l);
}

// four consecutive fp32 -> per plane two dwords (4 fp16)
__device__ __forceinline__ void split2_pack4(const f32x4 v, u32x2& p1, u32x2& p2) {
    unsigned a1, a2, b1, b2;
    split2_pair(v[0], v[1], a1, a2);
    split2_pair(v[2], v[3], b1, b2);
    p1 = u32x2{a1, b1}; p2 = u32x2{a2, b2};
}

// Range guard of the split: a value of 65520 or more rounds to an fp16 infinity, and what follows is silently wrong
// (inf - inf = NaN, and the ReLUs turn NaN into 0).  Every thread keeps the largest magnitude it has split
// (two values per v_max3_f32) and raises a sticky per-GPU flag at the end of the kernel; the host reads it with
// mgn_range_check().  One flag per translation unit of the library.
constexpr float F16_SPLIT_LIMIT = 65520.0f;
static __device__ unsigned g_range_flag = 0;

__device__ __forceinline__ void track4(float& m, const f32x4 v) {
    m = fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1])));
    m = fmaxf(m, fmaxf(fabsf(v[2]), fabsf(v[3])));
}
__device__ __forceinline__ void range_report(float m) {
    if (m >= F16_SPLIT_LIMIT) atomicOr(&g_range_flag, 1u);
}

// A tile of `rows` x K fp16 in two planes: plane p, row r, column k at  p * plane + r * ld + k  (halves),
// ld = K + 8 (16 bytes of padding per row keep the 16-byte reads of 16 consecutive rows on distinct banks).
__device__ __forceinline__ void hplanes_store4(unsigned short* tile, int plane, int ld, int row, int k, const f32x4 v) {
    u32x2 p1, p2;
    split2_pack4(v, p1, p2);
    unsigned short* d = tile + row * ld + k;
    *(u32x2*)(d) = p1; *(u32x2*)(d + plane) = p2;
}

__device__ __forceinline__ void hplanes_store4(unsigned short* tile, int plane, int ld, int row, int k, const f32x4 v, float& m) {
    track4(m, v);
    hplanes_store4(tile, plane, ld, row, k, v);
}

// Register-resident fp16 planes of the weights of 32 output features over 16*KS inputs
// (v_mfma_f32_32x32x16_f16: lane (i = lane & 31, h = lane >> 5) feeds k = 16c + 8h + 0..7).
template <int KS>
struct WPlanesH { u32x4 p[2][KS]; };

// Two phases, so that a kernel with several weight sets can have all their loads in flight before the first
// conversion (the raw fp32 values wait in the registers the planes will occupy: 8 values = 8 registers either way):
// load_wraw_h issues the loads, split_wraw_h converts in place.
template <int KS>
__device__ __forceinline__ void load_wraw_h(WPlanesH<KS>& w, const float* __restrict__ W, int ldw, int koff, int feat, int h) {
    const float* row = W + (size_t)feat * ldw + koff + 8 * h;
#pragma unroll
    for (int c = 0; c < KS; ++c) {
        w.p[0][c] = __builtin_bit_cast(u32x4, *(const f32x4*)(row + 16 * c));
        w.p[1][c] = __builtin_bit_cast(u32x4, *(const f32x4*)(row + 16 * c + 4));
    }
}

template <int KS>
__device__ __forceinline__ void split_wraw_h(WPlanesH<KS>& w, float& m) {
#pragma unroll
    for (int c = 0; c < KS; ++c) {
        const f32x4 lo4 = __builtin_bit_cast(f32x4, w.p[0][c]) * W_SCALE, hi4 = __builtin_bit_cast(f32x4, w.p[1][c]) * W_SCALE;
        track4(m, lo4); track4(m, hi4);
        u32x2 a1, a2, b1, b2;
        split2_pack4(lo4, a1, a2);
        split2_pack4(hi4, b1, b2);
        w.p[0][c] = u32x4{a1[0], a1[1], b1[0], b1[1]};
        w.p[1][c] = u32x4{a2[0], a2[1], b2[0], b2[1]};
    }
}

// The same planes prepared once per weight set (pack_weights_kernel): u32x4 [4 waves][KS][2 planes][64 lanes], i.e.
// every register quad of every lane stored where a fully coalesced 1-KiB load finds it.  Loading a set from the
// row-major fp32 matrix takes 32 bytes out of each 128-byte line per instruction (every lane is its own row) and
// costs a workgroup 1.5 us per 128 x 128 set before its first tile; from the packed block it costs a third of that
// and no conversion (scripts/micro/prologue.hip) -- what matters when a launch has one tile per workgroup.
template <int KS>
__device__ __forceinline__ void load_wpacked_h(WPlanesH<KS>& w, const void* __restrict__ packed, int wave, int lane) {
    const u32x4* p = (const u32x4*)packed + (size_t)wave * KS * 2 * 64 + lane;
#pragma unroll
    for (int c = 0; c < KS; ++c) { w.p[0][c] = p[(2 * c) * 64]; w.p[1][c] = p[(2 * c + 1) * 64]; }
}

template <int KS>
__device__ __forceinline__ void load_wplanes_h(WPlanesH<KS>& w, const float* __restrict__ W, int ldw, int koff, int feat, int h) {
    float m = 0.f;
    load_wraw_h<KS>(w, W, ldw, koff, feat, h);
    split_wraw_h<KS>(w, m);
    range_report(m);                                // a weight of 512 or more (after the pre-scale) is out of range too
}

__device__ __forceinline__ f32x16 mfma32_f16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// acc[32 feats][32 rows] += W_SCALE * W * tile^T over k-steps [C0, C1) of 16 starting at column kbase.
// The two LDS reads of a k-step are issued one step ahead of the three MFMAs that consume them.
template <int C0, int C1, int KS>
__device__ __forceinline__ f32x16 wave_gemm_h(const unsigned short* tile, int plane, int ld, int kbase,
                                              const WPlanesH<KS>& w, int j, int h, f32x16 acc) {
    const unsigned short* r = tile + j * ld + kbase + 8 * h;
    u32x4 x1 = *(const u32x4*)(r + 16 * C0), x2 = *(const u32x4*)(r + plane + 16 * C0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int c = C0; c < C1; ++c) {
        u32x4 n1 = x1, n2 = x2;
        if (c + 1 < C1) {
            n1 = *(const u32x4*)(r + 16 * (c + 1)); n2 = *(const u32x4*)(r + plane + 16 * (c + 1));
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // next step's reads first ...
        }
        acc = mfma32_f16(w.p[1][c], x1, acc);   // h2 * h1
        acc = mfma32_f16(w.p[0][c], x2, acc);   // h1 * h2
        acc = mfma32_f16(w.p[0][c], x1, acc);   // h1 * h1
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);          // ... then this step's products
        x1 = n1; x2 = n2;
    }
    return acc;
}

// Two products over the same activation tile (two weight sets, two accumulators): one pair of LDS
// reads per k-step feeds six MFMAs.  Each accumulator sees its partial products in the order of
// wave_gemm_h, so the results are identical to two separate calls.
template <int C0, int C1, int KS>
__device__ __forceinline__ void wave_gemm_h2(const unsigned short* tile, int plane, int ld, int kbase,
                                             const WPlanesH<KS>& wa, const WPlanesH<KS>& wb, int j, int h,
                                             f32x16& acc_a, f32x16& acc_b) {
    const unsigned short* r = tile + j * ld + kbase + 8 * h;
    u32x4 x1 = *(const u32x4*)(r + 16 * C0), x2 = *(const u32x4*)(r + plane + 16 * C0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int c = C0; c < C1; ++c) {
        u32x4 n1 = x1, n2 = x2;
        if (c + 1 < C1) {
            n1 = *(const u32x4*)(r + 16 * (c + 1)); n2 = *(const u32x4*)(r + plane + 16 * (c + 1));
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        acc_a = mfma32_f16(wa.p[1][c], x1, acc_a); acc_b = mfma32_f16(wb.p[1][c], x1, acc_b);   // h2 * h1
        acc_a = mfma32_f16(wa.p[0][c], x2, acc_a); acc_b = mfma32_f16(wb.p[0][c], x2, acc_b);   // h1 * h2
        acc_a = mfma32_f16(wa.p[0][c], x1, acc_a); acc_b = mfma32_f16(wb.p[0][c], x1, acc_b);   // h1 * h1
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        x1 = n1; x2 = n2;
    }
}

// wave_gemm_h2 with a piece of unrelated vector / LDS-store work per k-step (`side(c)`, in program order after the
// step's matrix instructions): with one wave per SIMD the vector ALU has nothing to do while the matrix pipe works
// through a k-step (six instructions, 32 cycles each), so work placed here costs its issue slots only.
// NV, NW: vector-ALU instructions and LDS stores of one piece (upper bounds; they size the scheduling groups).
template <int C0, int C1, int KS, int NV, int NW, typename Side>
__device__ __forceinline__ void wave_gemm_h2_with(const unsigned short* tile, int plane, int ld, int kbase,
                                                  const WPlanesH<KS>& wa, const WPlanesH<KS>& wb, int j, int h,
                                                  f32x16& acc_a, f32x16& acc_b, Side&& side) {
    const unsigned short* r = tile + j * ld + kbase + 8 * h;
    u32x4 x1 = *(const u32x4*)(r + 16 * C0), x2 = *(const u32x4*)(r + plane + 16 * C0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int c = C0; c < C1; ++c) {
        u32x4 n1 = x1, n2 = x2;
        if (c + 1 < C1) {
            n1 = *(const u32x4*)(r + 16 * (c + 1)); n2 = *(const u32x4*)(r + plane + 16 * (c + 1));
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        acc_a = mfma32_f16(wa.p[1][c], x1, acc_a); acc_b = mfma32_f16(wb.p[1][c], x1, acc_b);   // h2 * h1
        side(c);
        acc_a = mfma32_f16(wa.p[0][c], x2, acc_a); acc_b = mfma32_f16(wb.p[0][c], x2, acc_b);   // h1 * h2
        acc_a = mfma32_f16(wa.p[0][c], x1, acc_a); acc_b = mfma32_f16(wb.p[0][c], x1, acc_b);   // h1 * h1
        // the step's side work spread between its matrix instructions (left alone, the scheduler piles it up at the end)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV / 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV / 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV / 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV / 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, NW, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        x1 = n1; x2 = n2;
    }
}

// ======================================================================== split-bf16 matrix path
// fp32 products on the bf16 matrix pipe.  Every fp32 operand is cut EXACTLY into three bf16
// terms by truncation (8 + 8 + 8 significant bits: x = hi + mid + lo, no rounding anywhere),
// and a product x*w is evaluated as the six partial products whose weight is >= 2^-16 of it:
//   hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi        (dropped: mid*lo, lo*mid, lo*lo < 2^-23)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16, smallest terms first.  Unlike the fp32 MFMA, the
// bf16 MFMA runs beside vector-ALU work of another wave on the same SIMD, and the six products
// cost 6/16 of the fp32 matrix time.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PK = H + 8;                          // bf16 per plane row: 256 B of data + 16 B pad
constexpr int PLANE = TM * PK;                     // bf16 per plane of a 32-row tile

// x = hi + mid + lo (exact); returns the three terms as fp32 bit patterns with 16 low zero bits
__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(hi);
    mid = __float_as_uint(r1) & 0xFFFF0000u;
    lo = __float_as_uint(r1 - __uint_as_float(mid));          // at most 8 significant bits are left
}
// two bf16 (upper halves of a and b) in one dword, a in the low half
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// four consecutive fp32 -> per plane two dwords (4 bf16)
__device__ __forceinline__ void split_pack4(const f32x4 v, u32x2& ph, u32x2& pm, u32x2& pl) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split3(v[i], h[i], m[i], l[i]);
    ph = u32x2{pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3])};
    pm = u32x2{pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3])};
    pl = u32x2{pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3])};
}

// write 4 consecutive features of one tile row as three bf16 planes
__device__ __forceinline__ void planes_store4(unsigned short* tile, int row, int k, const f32x4 v) {
    u32x2 ph, pm, pl;
    split_pack4(v, ph, pm, pl);
    unsigned short* d = tile + row * PK + k;
    *(u32x2*)(d) = ph; *(u32x2*)(d + PLANE) = pm; *(u32x2*)(d + 2 * PLANE) = pl;
}

// Register-resident weight planes of one 128-wide Linear for the 32 output features of a wave
// (v_mfma_f32_32x32x16_bf16: lane (i = lane & 31, h = lane >> 5) feeds k = 16c + 8h + 0..7).
struct WPlanes32 { u32x4 p[3][8]; };

__device__ __forceinline__ void load_wplanes32(WPlanes32& w, const float* __restrict__ W, int ldw, int koff, int feat, int h) {
    const float* row = W + (size_t)feat * ldw + koff + 8 * h;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4 lo4 = *(const f32x4*)(row + 16 * c), hi4 = *(const f32x4*)(row + 16 * c + 4);
        u32x2 a[3], b[3];
        split_pack4(lo4, a[0], a[1], a[2]);
        split_pack4(hi4, b[0], b[1], b[2]);
#pragma unroll
        for (int p = 0; p < 3; ++p) w.p[p][c] = u32x4{a[p][0], a[p][1], b[p][0], b[p][1]};
    }
}

__device__ __forceinline__ f32x16 mfma32_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// acc[32 feats][32 rows] += W * planes^T over k-steps [C0, C1) of 16; `tile` = plane 0 of a
// [3][TM][PK] bf16 image; lane (j = lane & 31, h = lane >> 5).  Smallest partial products first.
template <int C0, int C1>
__device__ __forceinline__ f32x16 wave_gemm3_32(const unsigned short* tile, const WPlanes32& w, int j, int h, f32x16 acc) {
    const unsigned short* r = tile + j * PK + 8 * h;
    // the three LDS reads of a k-step are issued one step ahead of the six MFMAs that consume them
    u32x4 xh = *(const u32x4*)(r + 16 * C0), xm = *(const u32x4*)(r + PLANE + 16 * C0),
          xl = *(const u32x4*)(r + 2 * PLANE + 16 * C0);
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
    for (int c = C0; c < C1; ++c) {
        u32x4 nh = xh, nm = xm, nl = xl;
        if (c + 1 < C1) {
            nh = *(const u32x4*)(r + 16 * (c + 1)); nm = *(const u32x4*)(r + PLANE + 16 * (c + 1));
            nl = *(const u32x4*)(r + 2 * PLANE + 16 * (c + 1));
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

// Diagnostic build only (-DMGN_PHASE_STAMPS, see scripts/phase_stamps.py; one table per translation unit): cycle sums between
// stamps of a kernel's main loop, taken by lane 0 of waves 0 and 2.  Never in the product build.
#ifdef MGN_PHASE_STAMPS
__device__ unsigned long long g_stamps[2][16];
#define STAMP_DECL unsigned long long st_prev = 0, st_sum[16] = {0}; bool st_on = false;
#define STAMP(i)                                                                          \
    do {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        unsigned long long now_;                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");       \
        if (st_on) st_sum[i] += now_ - st_prev;                                           \
        st_prev = now_; st_on = true;                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                \
    } while (0)
#define STAMP_FLUSH(slot)                                                                 \
    if ((threadIdx.x & 63) == 0 && (slot) >= 0)                                           \
        for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_stamps[slot][i_], st_sum[i_]);
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH(slot)
#endif

// The value as the optimiser cannot see through.  Two uses: what is derived from it is formed where it is used instead
// of being carried in registers across a tile; and a loaded value passed through it has arrived -- placed before a loop,
// it keeps the wait for a prologue load out of the loop header, where it would also wait for the previous pass's
// stores (loads and stores share one counter).
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// Blocks b and b + 8 share an XCD (and its L2); hand each XCD a contiguous run of ranges.
__device__ __forceinline__ int range_of_block(int b, int nb) {
    return (nb & 7) ? b : (b & 7) * (nb >> 3) + (b >> 3);
}

// Edge range of this workgroup, cut at receiver boundaries: [rowptr[i0], rowptr[i1]) with i = first index whose
// rowptr reaches the block's share of the edges.  Called by all 64 lanes of ONE wave: lanes 0..31 look for the
// lower end, lanes 32..63 for the upper end, each half with a 33-way search (32 probes per round trip to memory:
// five rounds at a million receivers, where a binary search by one thread makes forty dependent loads -- tens of
// microseconds during which the workgroup only waits).
__device__ __forceinline__ void block_edge_range(const int* __restrict__ rowptr, int n_nodes, long long n_edges, int* srange) {
    const int lane = threadIdx.x & 63, half = lane >> 5, sub = lane & 31;
    const int nb = gridDim.x, rho = range_of_block(blockIdx.x, nb);
    const long long key = (long long)rowptr[0] + (long long)(rho + half) * n_edges / nb;   // rowptr may start inside a larger CSR
    int lo = 0, hi = n_nodes;                 // answer in [lo, hi]; rowptr[hi] >= key holds (rowptr[n_nodes] = first + n_edges)
    while (lo < hi) {
        const int step = (hi - lo + 32) / 33;                     // ceil(span / 33) >= 1
        const int p = lo + (sub + 1) * step - 1;                  // probes lo+step-1, lo+2*step-1, ...
        const bool less = p < hi && (long long)rowptr[p] < key;   // monotone: the true flags form a prefix
        const int cnt = __popc((unsigned)(__ballot(less) >> (32 * half)));
        hi = cnt < 32 ? min(hi, lo + (cnt + 1) * step - 1) : hi;  // the first probe that is not less bounds the answer
        lo += cnt * step;                                         // everything up to the last "less" probe is below it
    }
    if (sub == 0) srange[half] = rowptr[lo];
}

// ------------------------------------------------------------------------ host helpers
inline int num_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cached[dev]) {
        hipDeviceProp_t p;
        cached[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    return cached[dev];
}

// number of workgroups for `ntiles` tiles at `per_cu` resident workgroups per CU;
// a multiple of 8 whenever it exceeds 8 so the XCD mapping applies
inline int grid_for(long long ntiles, int per_cu) {
    long long g = (long long)num_cus() * per_cu;
    if (ntiles < g) g = ntiles;
    if (g > 8) g -= g % 8;
    return (int)(g < 1 ? 1 : g);
}

// reads (and optionally clears) this translation unit's range flag of the current device; synchronises `stream`
inline int read_range_flag(int* tripped, int reset, hipStream_t stream, const char* name) {
    unsigned v = 0;
    hipError_t e = hipMemcpyFromSymbolAsync(&v, HIP_SYMBOL(g_range_flag), sizeof(v), 0, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e == hipSuccess && reset && v) {
        const unsigned z = 0;
        e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_range_flag), &z, sizeof(z), 0, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
    }
    if (e != hipSuccess) { mgn_set_error("%s: %s", name, hipGetErrorString(e)); return (int)e; }
    *tripped |= (int)(v != 0);
    return 0;
}

// enqueues a copy of this translation unit's range flag to `host_word` (pinned host memory) on `stream`; no synchronisation
inline int poll_range_flag(int* host_word, hipStream_t stream, const char* name) {
    const hipError_t e = hipMemcpyFromSymbolAsync(host_word, HIP_SYMBOL(g_range_flag), sizeof(unsigned), 0, hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) { mgn_set_error("%s: %s", name, hipGetErrorString(e)); return (int)e; }
    return 0;
}

inline int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mgn_set_error("%s: %s", what, hipGetErrorString(e)); return (int)e; }
    return 0;
}

}  // namespace

#define MGN_REQUIRE(cond, name)                                             \
    do { if (!(cond)) { mgn_set_error("%s: bad argument (%s)", name, #cond); return MGN_E_ARG; } } while (0)
