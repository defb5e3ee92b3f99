// Sandbox (never part of libmgn_hip.so): variants of the fp16-split edge kernel for timing experiments.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o scripts/micro/libedge_variants.so \
//         scripts/micro/edge_variants.hip gnn_amd/csrc/mgn_host.cpp
// Driver: scripts/exp_edge_variants.py
#include <hip/hip_runtime.h>
#include <cstdio>

#include "../../gnn_amd/csrc/mgn_device.h"

namespace {
struct EdgeArgs {
    const float* e_in; const float* ps; const float* pr;
    const int* rowptr; const int* sender; const int* recv;
    const float* w0; const float* w1; const float* b1; const float* ln_w; const float* ln_b;
    float* e_out; float* aggr;
    int n_edges; int n_nodes; float abl_c;
};
constexpr int PKH = H + 8;
constexpr int PLH = TM * PKH;

template <int ABL>
__global__ __launch_bounds__(NT, 2) void edge_h4_var(EdgeArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short Xp[2][2 * PLH];     // e_in tile, 2 fp16 planes, double buffered
    __shared__ __attribute__((aligned(16))) unsigned short Hp[2 * PLH];        // hidden tile, 2 fp16 planes; then Os
    __shared__ __attribute__((aligned(16))) float vec[3][H];                   // W_SCALE * b1, ln_w, ln_b
    __shared__ __attribute__((aligned(16))) float red[LN_RED_FLOATS];
    __shared__ float cout[H], tcarry[H];             // open sum after row 15 / after the tile, per column
    __shared__ int srange[2];
    __shared__ __attribute__((aligned(16))) float Os2[(ABL & 512) ? tile_floats(H / 4) : 4];
    float* Os = (ABL & 512) ? Os2 : (float*)Hp;      // pre-residual output tile (fp32)
    static_assert(sizeof(Hp) >= sizeof(float) * tile_floats(H / 4), "Os must fit into Hp");

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
    WPlanesH<8> wa, wb;
    load_wplanes_h<8>(wa, a.w0, 3 * H, 2 * H, 32 * w + j, h);
    load_wplanes_h<8>(wb, a.w1, H, 0, 32 * w + j, h);
    if (t < H) { vec[0][t] = a.b1[t] * W_SCALE; vec[1][t] = a.ln_w[t]; vec[2][t] = a.ln_b[t]; tcarry[t] = 0.f; }
    if (t == 0) {
        const int nb = gridDim.x, rho = range_of_block(blockIdx.x, nb);
        const long long first = a.rowptr[0];         // rowptr may start inside a larger CSR (receiver sub-range)
        const long long lo = first + (long long)rho * a.n_edges / nb, hi = first + (long long)(rho + 1) * a.n_edges / nb;
        srange[0] = a.rowptr[lower_bound_i32(a.rowptr, a.n_nodes + 1, lo)];
        srange[1] = a.rowptr[lower_bound_i32(a.rowptr, a.n_nodes + 1, hi)];
    }
    __syncthreads();
    const int e0 = __builtin_amdgcn_readfirstlane(srange[0]), e1 = __builtin_amdgcn_readfirstlane(srange[1]);
    if (e0 >= e1) return;
    const int last = e1 - e0 - 1;                    // row offsets are relative to e0 and clamped to [0, last]
    const float* ein_g = a.e_in + (size_t)e0 * H;
    float* eout_g = a.e_out + (size_t)e0 * H;
    const int* snd_g = a.sender + e0;
    const int* rcv_g = a.recv + e0;
    const __amdgpu_buffer_rsrc_t eo_rs = __builtin_amdgcn_make_buffer_rsrc(
        a.e_out ? (void*)(a.e_out + (size_t)e0 * H) : (void*)a.aggr, 0, a.e_out ? (last + 1) * (H * 4) : 0, 0x00020000);
    const int srow = t >> 5, sq = t & 31;            // row map: rows srow + 8*it, 16-byte chunk sq
    const int sc = t & (H - 1);                      // scan map: column sc, rows 16*sg .. 16*sg+15
    const int sg = __builtin_amdgcn_readfirstlane(t >> 7);

    f32x4 cur[4], pre[4];                            // rows of this tile (residual) and of the next one
#pragma unroll
    for (int it = 0; it < 4; ++it) cur[it] = *(const f32x4*)(ein_g + (unsigned)(min(srow + 8 * it, last) * H + 4 * sq));
    int s_cur = snd_g[min(j, last)], r_cur = rcv_g[min(j, last)];
#pragma unroll
    for (int it = 0; it < 4; ++it) hplanes_store4(Xp[0], PLH, PKH, srow + 8 * it, 4 * sq, cur[it]);
    __syncthreads();
    STAMP_DECL
    int b = 0;
    for (int rel = 0; rel <= last; rel += TM, b ^= 1) {
        STAMP(0);
        const int nvalid = min(TM, last + 1 - rel);
        f32x4 gs[4], gr[4];
        int s_nxt, r_nxt, rb;
        const int ra = r_cur;                                                // receiver of tile row j ...
        f32x16 acc = zero16();
        if constexpr (ABL & 128) {
            // the tile's loads interleaved with the first product: two vector-memory loads per k-step
            const unsigned so = (unsigned)s_cur * H + 32 * w + 4 * h, ro = (unsigned)r_cur * H + 32 * w + 4 * h;
            const int enx = min(rel + TM + j, last);
            if constexpr (ABL & 1024) __builtin_amdgcn_s_setprio(1);
            const unsigned short* r = Xp[b] + j * PKH + 8 * h;
            u32x4 x1 = *(const u32x4*)(r), x2 = *(const u32x4*)(r + PLH);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                u32x4 n1 = x1, n2 = x2;
                if (c + 1 < 8) {
                    n1 = *(const u32x4*)(r + 16 * (c + 1)); n2 = *(const u32x4*)(r + PLH + 16 * (c + 1));
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                if (c < 4) {
                    gs[c] = *(const f32x4*)(a.ps + so + 8 * c); gr[c] = *(const f32x4*)(a.pr + ro + 8 * c);
                    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                } else {
                    if constexpr (ABL & 2048) pre[c - 4] = __builtin_nontemporal_load((const f32x4*)(ein_g + (unsigned)(min(rel + TM + srow + 8 * (c - 4), last) * H + 4 * sq)));
                    else pre[c - 4] = *(const f32x4*)(ein_g + (unsigned)(min(rel + TM + srow + 8 * (c - 4), last) * H + 4 * sq));
                    if (c == 4) s_nxt = snd_g[enx];
                    if (c == 5) r_nxt = rcv_g[enx];
                    if (c == 6) rb = (rel + j + 1 <= last) ? rcv_g[min(rel + j + 1, last)] : -1;
                    if (c == 7) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                }
                acc = mfma32_f16(wa.p[1][c], x1, acc);
                acc = mfma32_f16(wa.p[0][c], x2, acc);
                acc = mfma32_f16(wa.p[0][c], x1, acc);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                x1 = n1; x2 = n2;
            }
            if constexpr (ABL & 1024) __builtin_amdgcn_s_setprio(0);
        } else {
        // requests first: everything below waits only for loads that are at least a phase old
        const unsigned so = (unsigned)((ABL & 1) ? 0 : s_cur) * H + 32 * w + 4 * h, ro = (unsigned)((ABL & 1) ? 0 : r_cur) * H + 32 * w + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (ABL & 2) { gs[g] = f32x4{a.abl_c, a.abl_c, a.abl_c, a.abl_c}; gr[g] = gs[g]; }
            else { gs[g] = *(const f32x4*)(a.ps + so + 8 * g); gr[g] = *(const f32x4*)(a.pr + ro + 8 * g); }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if constexpr (ABL & 64) pre[it] = cur[it] * a.abl_c;
            else pre[it] = *(const f32x4*)(ein_g + (unsigned)(min(rel + TM + srow + 8 * it, last) * H + 4 * sq));
        }
        s_nxt = snd_g[min(rel + TM + j, last)]; r_nxt = rcv_g[min(rel + TM + j, last)];
        rb = (rel + j + 1 <= last) ? rcv_g[rel + j + 1] : -1;      // ... and of the row after it
        STAMP(1);
        acc = wave_gemm_h<0, 8, 8>(Xp[b], PLH, PKH, 0, wa, j, h, acc);
        }
        STAMP(2);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaf(acc[4 * g + k], W_UNSCALE, gs[g][k] + gr[g][k]), 0.f);
            hplanes_store4(Hp, PLH, PKH, j, 32 * w + 8 * g + 4 * h, v);
        }
        STAMP(3);
        __syncthreads();                                                     // (B) hidden tile complete
        STAMP(4);
        acc = acc_from_vec(vec[0], w, h);
        if constexpr (ABL & 1024) __builtin_amdgcn_s_setprio(1);
        acc = wave_gemm_h<0, 8, 8>(Hp, PLH, PKH, 0, wb, j, h, acc);
        if constexpr (ABL & 1024) __builtin_amdgcn_s_setprio(0);
        acc = relu16(acc);
        if constexpr (!(ABL & 32)) ln_local_stats_rows(acc, red, w, j, h);
        STAMP(5);
        __syncthreads();                                                     // (C) statistics exchanged; Hp dead
        STAMP(6);
        if constexpr (!(ABL & 32)) acc = ln_finish_rows(acc, red, w, j, h, vec[1], vec[2], LN_EPS * W_SCALE * W_SCALE);
        acc_to_tile(Os, acc, w, j, h);
        if constexpr (ABL & 512) {
#pragma unroll
            for (int it = 0; it < 4; ++it) hplanes_store4(Xp[b ^ 1], PLH, PKH, srow + 8 * it, 4 * sq, pre[it]);
        }
        STAMP(7);
        __syncthreads();                                                     // (D) output tile complete
        STAMP(8);
        // next tile into the other plane buffer (its rows were requested at the top of this pass)
        if constexpr (!(ABL & 512)) {
#pragma unroll
            for (int it = 0; it < 4; ++it) hplanes_store4(Xp[b ^ 1], PLH, PKH, srow + 8 * it, 4 * sq, pre[it]);
        }
        STAMP(9);
        // receiver-segmented sum: thread (column sc, group sg) scans rows 16sg .. 16sg+15 in order;
        // xr[r] = sum of the rows since the last segment end before r
        const unsigned mask = (unsigned)__ballot(j < nvalid && ra != rb);   // bit r: row r ends its segment
        const unsigned gmask = (mask >> (16 * sg)) & 0xFFFFu;
        float xr[16], part = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) xr[r] = Os[(16 * sg + r) * (H + 4) + sc];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            part += xr[r];
            xr[r] = part;
            if ((gmask >> r) & 1u) part = 0.f;
        }
        float cin = tcarry[sc];                      // open segment of the previous tile (rewritten after S)
        if (sg == 0) cout[sc] = gmask ? part : cin + part;
        // residual for the store below, taken before Os can change
        f32x4 y[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) y[it] = *(const f32x4*)(Os + toff(srow + 8 * it, sq, H / 4)) + cur[it];
        if constexpr (ABL & 256) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y[it]), eo_rs,
                                                       (unsigned)((rel + srow + 8 * it) * (H * 4) + 16 * sq), 0, 0);
        }
        STAMP(10);
        __syncthreads();                                                     // (S) cout visible
        STAMP(11);
        if (sg == 1) cin = cout[sc];
        {
            float* eo = eout_g + (unsigned)(rel * H);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = srow + 8 * it;
                if (!(ABL & (4 | 256)) && row < nvalid && a.e_out) {
                    if constexpr (ABL & 2048) __builtin_nontemporal_store(y[it], (f32x4*)(eo + (unsigned)(row * H + 4 * sq)));
                    else *(f32x4*)(eo + (unsigned)(row * H + 4 * sq)) = y[it];
                }
            }
        }
        bool open = true;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (!(ABL & 8) && ((gmask >> r) & 1u)) {
                const int node = __builtin_amdgcn_readlane(ra, 16 * sg + r);
                a.aggr[(unsigned)(node * H + sc)] = open ? cin + xr[r] : xr[r];
                open = false;
            }
        }
        if (sg == 1) tcarry[sc] = gmask ? part : cin + part;
        STAMP(12);
        if constexpr (!(ABL & 512)) __syncthreads();                         // (E) Os / Xp[b^1] / tcarry settled
        s_cur = s_nxt; r_cur = r_nxt;
#pragma unroll
        for (int it = 0; it < 4; ++it) cur[it] = pre[it];
    }
    STAMP_FLUSH(w == 0 ? 0 : (w == 2 ? 1 : -1))
}


// ======================================================================== 8-wave variant, 16-feature slices
// 512 threads; wave w owns output features [16w, 16w+16) of both Linears (64 weight registers), products on
// v_mfma_f32_16x16x32_f16 (A = weights, B = two 16-row blocks of the tile), two workgroups per CU = 4 waves per SIMD.
constexpr int NT8 = 512;
constexpr int PK8 = H + 8;           // halves per plane row: 256 B of data + 16 B pad
constexpr int PL8 = TM * PK8;

struct WPlanes16 { u32x4 p[2][4]; };

__device__ __forceinline__ void load_wplanes16(WPlanes16& w, const float* __restrict__ W, int ldw, int koff, int feat, int kq) {
    const float* row = W + (size_t)feat * ldw + koff + 8 * kq;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x4 lo4 = *(const f32x4*)(row + 32 * c) * W_SCALE, hi4 = *(const f32x4*)(row + 32 * c + 4) * W_SCALE;
        u32x2 a1, a2, b1, b2;
        split2_pack4(lo4, a1, a2);
        split2_pack4(hi4, b1, b2);
        w.p[0][c] = u32x4{a1[0], a1[1], b1[0], b1[1]};
        w.p[1][c] = u32x4{a2[0], a2[1], b2[0], b2[1]};
    }
}

__device__ __forceinline__ f32x4 mfma16_f16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// sum over the four 16-lane rows of a wave (lanes n, n+16, n+32, n+48), result in every lane
__device__ __forceinline__ float sum_rows4(float v) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const unsigned us = __float_as_uint(s);
    auto r2 = __builtin_amdgcn_permlane32_swap(us, us, false, false);
    return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}

template <int F>
__global__ __launch_bounds__(NT8, 4) void edge_h8(EdgeArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short Xp[2][2 * PL8];
    __shared__ __attribute__((aligned(16))) unsigned short Hp[2 * PL8];
    __shared__ __attribute__((aligned(16))) float Os[tile_floats(H / 4)];
    __shared__ __attribute__((aligned(16))) float vec[3][H];
    __shared__ __attribute__((aligned(16))) float red[LN_RED_FLOATS];
    __shared__ float cout[4][H], tcarry[2][H];
    __shared__ int ids[2][2][TM + 4];                // [buffer][sender | receiver][row], receiver row 32 = first row of the next tile
    __shared__ int srange[2];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, n = lane & 15, q = lane >> 4, j = lane & 31;
    WPlanes16 wa, wb;
    load_wplanes16(wa, a.w0, 3 * H, 2 * H, 16 * w + n, q);
    load_wplanes16(wb, a.w1, H, 0, 16 * w + n, q);
    if (t < H) { vec[0][t] = a.b1[t] * W_SCALE; vec[1][t] = a.ln_w[t]; vec[2][t] = a.ln_b[t]; tcarry[0][t] = 0.f; tcarry[1][t] = 0.f; }
    if (t == 0) {
        const int nb = gridDim.x, rho = range_of_block(blockIdx.x, nb);
        const long long first = a.rowptr[0];
        const long long lo = first + (long long)rho * a.n_edges / nb, hi = first + (long long)(rho + 1) * a.n_edges / nb;
        srange[0] = a.rowptr[lower_bound_i32(a.rowptr, a.n_nodes + 1, lo)];
        srange[1] = a.rowptr[lower_bound_i32(a.rowptr, a.n_nodes + 1, hi)];
    }
    __syncthreads();
    const int e0 = __builtin_amdgcn_readfirstlane(srange[0]), e1 = __builtin_amdgcn_readfirstlane(srange[1]);
    if (e0 >= e1) return;
    const int last = e1 - e0 - 1;
    const float* ein_g = a.e_in + (size_t)e0 * H;
    const int* snd_g = a.sender + e0;
    const int* rcv_g = a.recv + e0;
    // e_out through a buffer descriptor over this workgroup's rows: stores beyond the range are dropped by the hardware,
    // so the stores of a partial last tile (and of a launch without an edge output) need no branch
    const __amdgpu_buffer_rsrc_t eo_rs = __builtin_amdgcn_make_buffer_rsrc(
        a.e_out ? (void*)(a.e_out + (size_t)e0 * H) : (void*)a.aggr, 0, a.e_out ? (last + 1) * (H * 4) : 0, 0x00020000);
    const int srow = t >> 5, sq = t & 31;            // row map: rows srow, srow + 16; 16-byte chunk sq
    const int sc = t & (H - 1);
    const int sg = __builtin_amdgcn_readfirstlane(t >> 7);      // scan group: rows 8sg .. 8sg+7

    f32x4 cur[2], pre[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) cur[it] = *(const f32x4*)(ein_g + (unsigned)(min(srow + 16 * it, last) * H + 4 * sq));
    if (w == 0) {
        if (lane <= TM) {
            ids[0][1][lane] = rcv_g[min(lane, last)];
            if (lane < TM) ids[0][0][lane] = snd_g[min(lane, last)];
        }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) hplanes_store4(Xp[0], PL8, PK8, srow + 16 * it, 4 * sq, cur[it]);
    __syncthreads();
    int b = 0;
    for (int rel = 0; rel <= last; rel += TM, b ^= 1) {
        const int nvalid = min(TM, last + 1 - rel);
        const int s0 = ids[b][0][n], s1 = ids[b][0][16 + n], r0 = ids[b][1][n], r1 = ids[b][1][16 + n];
        f32x4 gs[2], gr[2];
        int id_s = 0, id_r = 0;                      // wave 0: ids of the next tile
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        {
            const unsigned short* rp = Xp[b] + n * PK8 + 8 * q;
            u32x4 x1 = *(const u32x4*)(rp), x2 = *(const u32x4*)(rp + PL8);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int st = 0; st < 8; ++st) {         // st = 4 * row block + k-step
                const int rbk = st >> 2, c = st & 3;
                u32x4 n1 = x1, n2 = x2;
                if (st + 1 < 8) {
                    const int rb2 = (st + 1) >> 2, c2 = (st + 1) & 3;
                    n1 = *(const u32x4*)(rp + 16 * rb2 * PK8 + 32 * c2); n2 = *(const u32x4*)(rp + 16 * rb2 * PK8 + 32 * c2 + PL8);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                if (st == 0) { gs[0] = *(const f32x4*)(a.ps + ((unsigned)s0 * H + 16 * w + 4 * q)); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
                if (st == 1) { gr[0] = *(const f32x4*)(a.pr + ((unsigned)r0 * H + 16 * w + 4 * q)); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
                if (st == 2) { gs[1] = *(const f32x4*)(a.ps + ((unsigned)s1 * H + 16 * w + 4 * q)); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
                if (st == 3) { gr[1] = *(const f32x4*)(a.pr + ((unsigned)r1 * H + 16 * w + 4 * q)); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
                acc[rbk] = mfma16_f16(wa.p[1][c], x1, acc[rbk]);
                acc[rbk] = mfma16_f16(wa.p[0][c], x2, acc[rbk]);
                acc[rbk] = mfma16_f16(wa.p[0][c], x1, acc[rbk]);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                x1 = n1; x2 = n2;
            }
        }
#pragma unroll
        for (int rbk = 0; rbk < 2; ++rbk) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaf(acc[rbk][k], W_UNSCALE, gs[rbk][k] + gr[rbk][k]), 0.f);
            hplanes_store4(Hp, PL8, PK8, 16 * rbk + n, 16 * w + 4 * q, v);
        }
        __syncthreads();                                                     // (B) hidden tile complete
        {
            const f32x4 bias = *(const f32x4*)(vec[0] + 16 * w + 4 * q);
            acc[0] = bias; acc[1] = bias;
            const unsigned short* rp = Hp + n * PK8 + 8 * q;
            u32x4 x1 = *(const u32x4*)(rp), x2 = *(const u32x4*)(rp + PL8);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int rbk = st >> 2, c = st & 3;
                u32x4 n1 = x1, n2 = x2;
                if (st + 1 < 8) {
                    const int rb2 = (st + 1) >> 2, c2 = (st + 1) & 3;
                    n1 = *(const u32x4*)(rp + 16 * rb2 * PK8 + 32 * c2); n2 = *(const u32x4*)(rp + 16 * rb2 * PK8 + 32 * c2 + PL8);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                // next tile's rows and ids: requested here, where the gathered projections are dead
                if (st == 0) { pre[0] = *(const f32x4*)(ein_g + (unsigned)(min(rel + TM + srow, last) * H + 4 * sq)); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
                if (st == 1) { pre[1] = *(const f32x4*)(ein_g + (unsigned)(min(rel + TM + srow + 16, last) * H + 4 * sq)); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
                if (st == 2 && w == 0) {
                    id_s = snd_g[min(rel + TM + lane, last)];
                    id_r = rcv_g[min(rel + TM + lane, last)];
                }
                acc[rbk] = mfma16_f16(wb.p[1][c], x1, acc[rbk]);
                acc[rbk] = mfma16_f16(wb.p[0][c], x2, acc[rbk]);
                acc[rbk] = mfma16_f16(wb.p[0][c], x1, acc[rbk]);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                x1 = n1; x2 = n2;
            }
        }
        // ReLU, LayerNorm statistics: mean and centred sum of squares of the wave's 16 features per row
#pragma unroll
        for (int rbk = 0; rbk < 2; ++rbk) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[rbk][k] = fmaxf(acc[rbk][k], 0.f);
            const float m = sum_rows4((acc[rbk][0] + acc[rbk][1]) + (acc[rbk][2] + acc[rbk][3])) * (1.0f / 16.0f);
            const f32x4 d = acc[rbk] - m;
            const float q16 = sum_rows4((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
            if (q == 0) {
                red[(16 * rbk + n) * LN_RED_LD + w] = m;
                red[(16 * rbk + n) * LN_RED_LD + 8 + w] = q16;
            }
        }
        __syncthreads();                                                     // (C) statistics exchanged
        {
            const f32x4 ga = *(const f32x4*)(vec[1] + 16 * w + 4 * q), be = *(const f32x4*)(vec[2] + 16 * w + 4 * q);
#pragma unroll
            for (int rbk = 0; rbk < 2; ++rbk) {
                const float* rr = red + (16 * rbk + n) * LN_RED_LD;
                const f32x4 ma = *(const f32x4*)(rr), mb = *(const f32x4*)(rr + 4);
                const f32x4 qa = *(const f32x4*)(rr + 8), qb = *(const f32x4*)(rr + 12);
                const float mean = (((ma[0] + ma[1]) + (ma[2] + ma[3])) + ((mb[0] + mb[1]) + (mb[2] + mb[3]))) * 0.125f;
                const float m2 = ((qa[0] + qa[1]) + (qa[2] + qa[3])) + ((qb[0] + qb[1]) + (qb[2] + qb[3]));
                const f32x4 da = ma - mean, db = mb - mean;
                const float between = ((da[0] * da[0] + da[1] * da[1]) + (da[2] * da[2] + da[3] * da[3])) +
                                      ((db[0] * db[0] + db[1] * db[1]) + (db[2] * db[2] + db[3] * db[3]));
                const float var = (m2 + 16.0f * between) * (1.0f / H);
                const float rstd = __builtin_amdgcn_rsqf(var + LN_EPS * W_SCALE * W_SCALE);
                const float shift = -mean * rstd;
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = fmaf(fmaf(acc[rbk][k], rstd, shift), ga[k], be[k]);
                *(f32x4*)(Os + toff(16 * rbk + n, 4 * w + q, H / 4)) = o;
            }
        }
        // next tile -> the other plane buffer, its ids -> the other id buffer (published by barrier D)
#pragma unroll
        for (int it = 0; it < 2; ++it) hplanes_store4(Xp[b ^ 1], PL8, PK8, srow + 16 * it, 4 * sq, pre[it]);
        if (w == 0) {
            if (lane <= TM) { ids[b ^ 1][0][lane] = id_s; ids[b ^ 1][1][lane] = id_r; }     // sender row 32 is padding
        }
        __syncthreads();                                                     // (D) output tile, next tile staged
        const int ra = ids[b][1][j], rb = ids[b][1][j + 1];
        // bit r: row r ends its segment (the last row of the workgroup's range always does; both lane halves agree)
        const unsigned mask = (unsigned)__ballot(j < nvalid && (ra != rb || rel + j == last));
        const unsigned gmask = (mask >> (8 * sg)) & 0xFFu;
        float xr[8], part = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) xr[r] = Os[(8 * sg + r) * (H + 4) + sc];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            part += xr[r];
            xr[r] = part;
            if ((gmask >> r) & 1u) part = 0.f;
        }
        cout[sg][sc] = part;
        // residual + store (rows beyond the workgroup's range are dropped by the buffer bounds check)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const f32x4 y = *(const f32x4*)(Os + toff(srow + 16 * it, sq, H / 4)) + cur[it];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), eo_rs, (unsigned)((rel + srow + 16 * it) * (H * 4) + 16 * sq), 0, 0);
        }
        __syncthreads();                                                     // (S) group sums visible
        float cin = tcarry[b][sc];                   // open segment entering this tile
#pragma unroll
        for (int g2 = 0; g2 < 3; ++g2) {
            if (g2 < sg) {
                const float c2 = cout[g2][sc];
                cin = ((mask >> (8 * g2)) & 0xFFu) ? c2 : cin + c2;
            }
        }
        bool open = true;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if ((gmask >> r) & 1u) {
                const int node = __builtin_amdgcn_readlane(ra, 8 * sg + r);
                a.aggr[(unsigned)(node * H + sc)] = open ? cin + xr[r] : xr[r];
                open = false;
            }
        }
        if (sg == 3) tcarry[b ^ 1][sc] = gmask ? part : cin + part;
#pragma unroll
        for (int it = 0; it < 2; ++it) cur[it] = pre[it];
    }
}

}  // namespace

// variant: 0 = the product kernel's code; bits: 1 gathers from row 0, 2 no gathers, 4 no e_out store, 8 no aggr
// stores, 32 no LayerNorm, 64 no e_in stream (tile reused)
extern "C" int edge_variant(int variant, const float* e_in, const float* ps, const float* pr, const int* rowptr,
                            const int* sender, const int* recv, const float* w0, const float* w1, const float* b1,
                            const float* ln_w, const float* ln_b, float* e_out, float* aggr, int n_edges, int n_nodes,
                            void* stream) {
    EdgeArgs a{e_in, ps, pr, rowptr, sender, recv, w0, w1, b1, ln_w, ln_b, e_out, aggr, n_edges, n_nodes, 0.5f};
    const int g = grid_for((n_edges + TM - 1) / TM, 2);
#define V(N) case N: hipLaunchKernelGGL((edge_h4_var<N>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a); break;
    if (variant == 1000) {
        static bool told = false;
        if (!told) {
            int nblk = -1;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, edge_h8<0>, NT8, 0);
            fprintf(stderr, "edge_h8: %d resident workgroups per CU (occupancy query)\n", nblk);
            told = true;
        }
        hipLaunchKernelGGL((edge_h8<0>), dim3(g), dim3(NT8), 0, (hipStream_t)stream, a);
        return check_launch("edge_variant");
    }
    switch (variant) {
        V(0) V(128) V(2176) V(384) V(896) V(1920) V(1) V(2) V(4) V(8) V(12) V(32) V(64) V(14) V(78) V(110)
        default: return -1;
    }
#undef V
    return check_launch("edge_variant");
}
