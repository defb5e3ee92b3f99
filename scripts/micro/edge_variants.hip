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
    constexpr int AGS = 8;                           // segment sums staged per tile (the rest is stored directly)
    __shared__ __attribute__((aligned(16))) float Ag[(ABL & 4096) ? 2 * AGS * H : 4];
    __shared__ int agn[2][AGS];
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
    int ag_count = 0;                                // staged segment sums of the previous tile
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
        if constexpr (ABL & 4096) {
            // segment sums of the previous tile, staged in LDS: whole 512-byte rows, 16 bytes per lane
            if (srow < ag_count) {
                const int node = agn[b ^ 1][srow];
                *(f32x4*)(a.aggr + ((unsigned)node * H + 4 * sq)) = *(const f32x4*)(Ag + ((b ^ 1) * AGS + srow) * H + 4 * sq);
            }
        }
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
                const float val = open ? cin + xr[r] : xr[r];
                if constexpr (ABL & 4096) {
                    const int k = __builtin_popcount(mask & ((1u << (16 * sg + r)) - 1u));
                    if (k < AGS) {
                        Ag[(b * AGS + k) * H + sc] = val;
                        if (sc == 0) agn[b][k] = node;
                    } else {
                        a.aggr[(unsigned)(node * H + sc)] = val;
                    }
                } else {
                    a.aggr[(unsigned)(node * H + sc)] = val;
                }
                open = false;
            }
        }
        if constexpr (ABL & 4096) ag_count = min(AGS, __builtin_popcount(mask));
        if (sg == 1) tcarry[sc] = gmask ? part : cin + part;
        STAMP(12);
        if constexpr (!(ABL & 512)) __syncthreads();                         // (E) Os / Xp[b^1] / tcarry settled
        s_cur = s_nxt; r_cur = r_nxt;
#pragma unroll
        for (int it = 0; it < 4; ++it) cur[it] = pre[it];
    }
    if constexpr (ABL & 4096) {
        if constexpr (ABL & 512) __syncthreads();    // (the loop's last barrier E is absent in that variant)
        if (srow < ag_count) {
            const int node = agn[b ^ 1][srow];
            *(f32x4*)(a.aggr + ((unsigned)node * H + 4 * sq)) = *(const f32x4*)(Ag + ((b ^ 1) * AGS + srow) * H + 4 * sq);
        }
    }
    STAMP_FLUSH(w == 0 ? 0 : (w == 2 ? 1 : -1))
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
    switch (variant) {
        V(0) V(128) V(4224) V(2176) V(384) V(896) V(1920) V(1) V(2) V(4) V(8) V(12) V(32) V(64) V(14) V(78) V(110)
        default: return -1;
    }
#undef V
    return check_launch("edge_variant");
}
