// Device side of libmgn_hip.so: MeshGraphNets processor, encoder, decoder and normaliser
// kernels for gfx950 (MI355X), fp32 throughout.
//
// Common structure of the MLP kernels ("row-tile" kernels)
//   * a workgroup is 4 waves (256 threads) and walks a contiguous range of 32-row tiles;
//   * wave w owns output features [32w, 32w+32) of every 128-wide Linear and keeps its
//     slice of the weights in registers for the whole kernel (loaded once);
//   * the activation tile lives in LDS as rows of K floats padded by one 16-byte chunk,
//     so that the 16 lanes of a ds_read_b128 group (16 different rows, same chunk) hit
//     16 different bank groups while every chunk stays at a constant offset from the
//     row base (immediate-offset addressing, no per-chunk address registers);
//   * products run on v_mfma_f32_32x32x2_f32 with A = weights (row = feature) and
//     B = activations (column = tile row): lane (j = lane & 31, h = lane >> 5) ends up
//     holding, for tile row j, features 32w + 8g + 4h + {0..3} in accumulator
//     registers 4g .. 4g+3 -- i.e. whole 16-byte chunks of that row;
//   * the k index consumed by MFMA step t of chunk c is 8c + 4h + t for both operands,
//     so each lane fetches its operands with one 16-byte access per four MFMAs.
//
// Sums over edges use no atomics: edges are in receiver-major order, a workgroup owns
// whole receivers, and one thread per feature column adds the rows of a segment in
// order, so results do not depend on the launch geometry.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "mgn_internal.h"

#include "mgn_device.h"

namespace {

// ======================================================================== processor: edges
// Two 4-wave workgroups per CU (one wave per SIMD each), every workgroup walking its own
// contiguous, receiver-aligned edge range in 32-edge tiles:
//   stage   the tile's rows (prefetched into registers during the previous tile) -> X
//   P1      first product (edge latents x W0[:, 2H:3H]) + gathered node projections, ReLU -> Hs
//   P2      second product, ReLU, LayerNorm -> Hs (pre-residual edge output)
//   P3      residual + coalesced store (waves 2,3) / receiver-segmented sum (waves 0,1)
// fp32 MFMA and the vector ALU do not overlap on a SIMD, so the kernel is tuned for a low
// instruction count around the 128 MFMAs of a tile rather than for phase overlap.
struct EdgeArgs {
    const float* e_in; const float* ps; const float* pr;
    const int* rowptr; const int* sender; const int* recv;
    const float* w0; const float* w1; const float* b1; const float* ln_w; const float* ln_b;
    float* e_out; float* aggr;
    int n_edges; int n_nodes;
    const float* ew;                 // ATT kernels: per-edge weight of the message in the receiver's sum (attention)
    const void* pk0; const void* pk1;    // fp16-split path: w0[:, 2H:] and w1 as packed planes (pack_weights_kernel), or null
};

template <bool ATT>
__global__ __launch_bounds__(NT, 2) void mp_edge_kernel(EdgeArgs a) {
    __shared__ __attribute__((aligned(16))) float X[2][tile_floats(H / 4)];   // staged e_in tile, double buffered
    __shared__ __attribute__((aligned(16))) float Hs[tile_floats(H / 4)];     // hidden tile, then the edge output
    __shared__ __attribute__((aligned(16))) float vec[3][H];      // b1, ln_w, ln_b
    __shared__ float red[16 * TM];
    __shared__ int rid[2][TM + 4];
    __shared__ int srange[2];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;

    f32x4 wa[16], wb[16];
    load_wfrag<16>(wa, a.w0, 3 * H, 2 * H, 32 * w + j, h);
    load_wfrag<16>(wb, a.w1, H, 0, 32 * w + j, h);
    if (t < H) { vec[0][t] = a.b1[t]; vec[1][t] = a.ln_w[t]; vec[2][t] = a.ln_b[t]; }
    if (t < 64) block_edge_range(a.rowptr, a.n_nodes, a.n_edges, srange);   // wave 0, all of its lanes
    __syncthreads();
    const int e0 = srange[0], e1 = srange[1];
    if (e0 >= e1) return;

    const int srow = t >> 5, sq = t & 31;       // staging map: rows srow + 8*it, chunk sq
    f32x4 pre[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = min(e0 + srow + 8 * it, e1 - 1);
        pre[it] = *(const f32x4*)(a.e_in + (size_t)r * H + 4 * sq);
    }
    int s_nxt = a.sender[min(e0 + j, e1 - 1)], r_nxt = a.recv[min(e0 + j, e1 - 1)];
    float run = 0.f;                            // running segment sum (threads 0..127: one column each)
    int buf = 0;
    for (int base = e0; base < e1; base += TM, buf ^= 1) {
        const int nvalid = min(TM, e1 - base);
        const bool more = base + TM < e1;
        float* Xb = X[buf];
#pragma unroll
        for (int it = 0; it < 4; ++it) *(f32x4*)(Xb + toff(srow + 8 * it, sq, H / 4)) = pre[it];
        if (t <= TM) rid[buf][t] = (base + t < e1) ? a.recv[base + t] : -1;
        const int s = s_nxt, r = r_nxt;
        __syncthreads();                                                     // (A)
        // per-node projections of this lane's edge (sender part; receiver part + bias), issued
        // ahead of the first product and consumed after it
        f32x4 gs[4], gr[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            gs[g] = *(const f32x4*)(a.ps + (size_t)s * H + 32 * w + 8 * g + 4 * h);
            gr[g] = *(const f32x4*)(a.pr + (size_t)r * H + 32 * w + 8 * g + 4 * h);
        }
        if (more) {
            const int ej = min(base + TM + j, e1 - 1);
            s_nxt = a.sender[ej]; r_nxt = a.recv[ej];
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc = wave_gemm<16>(Xb, H / 4, 0, wa, j, h, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {                                  // next tile's rows: in flight during the second product
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rr = min(base + TM + srow + 8 * it, e1 - 1);
                pre[it] = *(const f32x4*)(a.e_in + (size_t)rr * H + 4 * sq);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[4 * g + k] = fmaxf(acc[4 * g + k] + gs[g][k] + gr[g][k], 0.f);
        acc_to_tile(Hs, acc, w, j, h);
        __syncthreads();                                                     // (B)
        acc = acc_from_vec(vec[0], w, h);
        __builtin_amdgcn_sched_barrier(0);
        acc = wave_gemm<16>(Hs, H / 4, 0, wb, j, h, acc);
        acc = relu16(acc);
        acc = layernorm_rows(acc, red, w, j, h, vec[1], vec[2]);             // (C)
        acc_to_tile(Hs, acc, w, j, h);
        __syncthreads();                                                     // (E)
        if (w >= 2) {
            // residual + coalesced store of the new edge latents
            const int tt = t - 128;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 y[4], x[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int idx = tt + 128 * (4 * half + it), row = idx >> 5, q = idx & 31;
                    y[it] = *(const f32x4*)(Hs + toff(row, q, H / 4));
                    x[it] = *(const f32x4*)(Xb + toff(row, q, H / 4));
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int idx = tt + 128 * (4 * half + it), row = idx >> 5, q = idx & 31;
                    if (row < nvalid && a.e_out) *(f32x4*)(a.e_out + (size_t)(base + row) * H + 4 * q) = y[it] + x[it];
                }
            }
        } else {
            // receiver-segmented sum of the pre-residual outputs, one column per thread, rows
            // added strictly in order; LDS reads batched eight rows at a time
            const int* rr = rid[buf];
            const float* col = Hs + t;
#pragma unroll
            for (int r0 = 0; r0 < TM; r0 += 8) {
                if (r0 < nvalid) {
                    float v[8]; int id[9];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        v[k] = col[(r0 + k) * (H + 4)]; id[k] = rr[r0 + k];
                        if constexpr (ATT) v[k] *= a.ew[min(base + r0 + k, e1 - 1)];
                    }
                    id[8] = rr[r0 + 8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (r0 + k < nvalid) {
                            run += v[k];
                            if (id[k + 1] != id[k]) { a.aggr[(size_t)id[k] * H + t] = run; run = 0.f; }
                        }
                    }
                }
            }
        }
        // the next iteration stages into the other X / rid buffer; Hs and red are rewritten
        // only after its barrier (A), which every thread reaches after finishing this pass
    }
}

constexpr int PKH = H + 8;          // halves per plane row of a 128-column tile
constexpr int NT3 = 512;

// Edge update on the split-bf16 path: one 8-wave workgroup per CU, two roles that share every
// SIMD pairwise (wave w and wave w + 4) and run as a three-deep software pipeline over 32-edge
// tiles, so that one role's matrix phase always runs beside the other role's vector phase:
//
//   iteration i        role 0 (waves 0-3: first Linear)        role 1 (waves 4-7: second Linear)
//   phase 1  ..bm..b1  product 1 of tile i (MFMA)              LayerNorm finish -> Es | store + scan, tile i-2
//   phase 2  ..b2      ReLU, split -> Hp[i&1]; stage tile i+1  scan finish (i-2); product 2 of tile i-1 (MFMA), stats
//
// Wave w of a role owns output features [32w, 32w+32) of its Linear with the three bf16 weight
// planes in registers (96 VGPRs).
template <bool ATT>
__global__ __launch_bounds__(NT3, 2) void mp_edge3_kernel(EdgeArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short Xp[3 * PLANE];      // staged e_in tile, 3 planes
    __shared__ __attribute__((aligned(16))) unsigned short Hp[2][3 * PLANE];   // hidden tile, 3 planes, double buffered
    __shared__ __attribute__((aligned(16))) float Es[tile_floats(H / 4)];      // pre-residual output tile (fp32)
    __shared__ __attribute__((aligned(16))) float Gs[tile_floats(H / 4)];      // ps[sender] + pr[receiver] per tile row
    __shared__ __attribute__((aligned(16))) float vec[3][H];                   // b1, ln_w, ln_b
    __shared__ __attribute__((aligned(16))) float red[LN_RED_FLOATS];
    __shared__ float cout[H], tcarry[H];
    __shared__ int srange[2];

    const int t = threadIdx.x, lane = t & 63, role = t >> 8, w = (t >> 6) & 3, t1 = t & 255;
    const int j = lane & 31, h = lane >> 5;

    if (t < H) { vec[0][t] = a.b1[t]; vec[1][t] = a.ln_w[t]; vec[2][t] = a.ln_b[t]; tcarry[t] = 0.f; }
    if (t < 64) block_edge_range(a.rowptr, a.n_nodes, a.n_edges, srange);   // wave 0, all of its lanes
    __syncthreads();
    const int e0 = __builtin_amdgcn_readfirstlane(srange[0]), e1 = __builtin_amdgcn_readfirstlane(srange[1]);
    if (e0 >= e1) return;
    const int last = e1 - e0 - 1;                    // row offsets below are relative to e0 and clamped to [0, last]
    const int ntiles = last / TM + 1;
    const float* ein_g = a.e_in + (size_t)e0 * H;
    float* eout_g = a.e_out ? a.e_out + (size_t)e0 * H : nullptr;      // no edge output: never dereferenced
    const int* snd_g = a.sender + e0;
    const int* rcv_g = a.recv + e0;
    const int srow = t1 >> 5, sq = t1 & 31;          // row-major map of a role's 256 threads: rows srow + 8*it, chunk sq

    if (role == 0) {
        // ------------------------------------------------------------------ role 0: first Linear
        WPlanes32 wa;
        load_wplanes32(wa, a.w0, 3 * H, 2 * H, 32 * w + j, h);
        f32x4 nx[4];
        int sid[4], rid[4];                          // sender / receiver of the rows this thread gathers; kept raw so
                                                     // that nothing waits for these loads in the iteration issuing them
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = min(srow + 8 * it, last);
            nx[it] = *(const f32x4*)(ein_g + (unsigned)(r * H + 4 * sq));
            sid[it] = snd_g[r]; rid[it] = rcv_g[r];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) planes_store4(Xp, srow + 8 * it, 4 * sq, nx[it]);
        __syncthreads();                                                     // b0: tile 0 staged
        STAMP_DECL
        for (int i = 0; i < ntiles + 2; ++i) {
            STAMP(0);
            const bool act = i < ntiles;
            const int rel = i * TM;
            f32x16 acc;
            f32x4 ga[4], gb[4];
            if (act) {
                // this tile's per-node projections, gathered as whole 512-byte rows (thread = row, chunk);
                // next tile's rows and ids
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    ga[it] = *(const f32x4*)(a.ps + ((unsigned)sid[it] * H + 4 * sq));
                    gb[it] = *(const f32x4*)(a.pr + ((unsigned)rid[it] * H + 4 * sq));
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = min(rel + TM + srow + 8 * it, last);
                    nx[it] = *(const f32x4*)(ein_g + (unsigned)(r * H + 4 * sq));
                    sid[it] = snd_g[r]; rid[it] = rcv_g[r];
                }
                STAMP(6);
                acc = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc = wave_gemm3_32<0, 4>(Xp, wa, j, h, acc);
            }
            STAMP(1);
            __syncthreads();                                                 // bm
            STAMP(2);
            if (act) {
                acc = wave_gemm3_32<4, 8>(Xp, wa, j, h, acc);
                // the gathers have had both halves of the product to arrive
#pragma unroll
                for (int it = 0; it < 4; ++it) *(f32x4*)(Gs + toff(srow + 8 * it, sq, H / 4)) = ga[it] + gb[it];
            }
            STAMP(3);
            __syncthreads();                                                 // b1: every wave is done reading Xp; Gs complete
            STAMP(4);
            if (act) {
                unsigned short* Hb = Hp[i & 1];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 gg = *(const f32x4*)(Gs + toff(j, 8 * w + 2 * g + h, H / 4));
                    f32x4 v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaxf(acc[4 * g + k] + gg[k], 0.f);
                    planes_store4(Hb, j, 32 * w + 8 * g + 4 * h, v);
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) planes_store4(Xp, srow + 8 * it, 4 * sq, nx[it]);   // tile i+1
            }
            STAMP(5);
            __syncthreads();                                                 // b2
        }
        STAMP_FLUSH(w == 0 ? 0 : -1)
    } else {
        // ------------------------------------------------------------------ role 1: second Linear
        WPlanes32 wb;
        load_wplanes32(wb, a.w1, H, 0, 32 * w + j, h);
        const int sc = t1 & (H - 1);
        const int sg = __builtin_amdgcn_readfirstlane(t1 >> 7);     // scan group: rows 16*sg .. 16*sg+15
        f32x16 acc;
        f32x4 er[4];                                 // fp32 rows of the tile stored next (residual)
        int ra = 0, rb = -1;                         // receiver of tile row (lane & 31) and of the row after it
        float part = 0.f, cin = 0.f, xr[16];
        unsigned mask = 0, gmask = 0;
        __syncthreads();                                                     // b0
        STAMP_DECL
        for (int i = 0; i < ntiles + 2; ++i) {
            STAMP(0);
            const int k = i - 2;                     // tile in its LayerNorm / store stage
            const int relk = k * TM, nvalid = min(TM, last + 1 - relk);
            if (k >= 0) {
                acc = ln_finish_rows(acc, red, w, j, h, vec[1], vec[2]);
                acc_to_tile(Es, acc, w, j, h);
            }
            STAMP(1);
            __syncthreads();                                                 // bm
            STAMP(2);
            if (k >= 0) {
                // segment ends first: this consumes the receiver ids loaded one iteration ago, before
                // the conditional stores below make the outstanding-load count unknown to the compiler
                mask = (unsigned)__ballot(j < nvalid && ra != rb);          // bit r: row r ends its segment
                gmask = (mask >> (16 * sg)) & 0xFFFFu;
                float* eo = eout_g + (unsigned)(relk * H);
#pragma unroll
                for (int it = 0; it < 4; ++it) {     // residual + coalesced store
                    const int row = srow + 8 * it;
                    const f32x4 y = *(const f32x4*)(Es + toff(row, sq, H / 4));
                    if (row < nvalid && a.e_out) *(f32x4*)(eo + (unsigned)(row * H + 4 * sq)) = y + er[it];
                }
                // receiver-segmented sum: thread (column sc, group sg) scans rows 16sg .. 16sg+15 in
                // order; xr[r] = sum of the rows since the last segment end before r
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    xr[r] = Es[(16 * sg + r) * (H + 4) + sc];
                    if constexpr (ATT) xr[r] *= a.ew[e0 + min(relk + 16 * sg + r, last)];
                }
                part = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    part += xr[r];
                    xr[r] = part;
                    if ((gmask >> r) & 1u) part = 0.f;
                }
                if (sg == 0) cout[sc] = part;
                cin = tcarry[sc];                    // open segment of the previous tile (rewritten after b1)
            }
            STAMP(3);
            __syncthreads();                                                 // b1
            STAMP(4);
            if (k >= 0) {
                // chain the two groups and the open segment of the previous tile, store the sums
                if (sg == 1) { const float c0 = cout[sc]; cin = (mask & 0xFFFFu) ? c0 : cin + c0; }
                bool open = true;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if ((gmask >> r) & 1u) {
                        const int node = __builtin_amdgcn_readlane(ra, 16 * sg + r);
                        a.aggr[(unsigned)(node * H + sc)] = open ? cin + xr[r] : xr[r];
                        open = false;
                    }
                }
                if (sg == 1) tcarry[sc] = gmask ? part : cin + part;         // read again after the next b1
            }
            STAMP(6);
            if (i >= 1 && i <= ntiles) {             // tile i-1: second product + LayerNorm statistics
                const int relm = (i - 1) * TM;
#pragma unroll
                for (int it = 0; it < 4; ++it)       // consumed one iteration later
                    er[it] = *(const f32x4*)(ein_g + (unsigned)(min(relm + srow + 8 * it, last) * H + 4 * sq));
                ra = rcv_g[min(relm + j, last)];
                rb = (relm + j + 1 <= last) ? rcv_g[min(relm + j + 1, last)] : -1;
                acc = acc_from_vec(vec[0], w, h);
                acc = wave_gemm3_32<0, 8>(Hp[(i - 1) & 1], wb, j, h, acc);
                acc = relu16(acc);
                ln_local_stats_rows(acc, red, w, j, h);
            }
            STAMP(5);
            __syncthreads();                                                 // b2
        }
        STAMP_FLUSH(w == 0 ? 1 : -1)
    }
}

constexpr int PLH = TM * PKH;         // halves per fp16 plane of a 32-row tile

// ======================================================================== processor: edges, fp16 split, 4-wave workgroups
// The edge update on the fp16 matrix pipe with the layout of mp_edge_kernel: two independent 4-wave
// workgroups per CU (one wave of each per SIMD), every wave running both products of its tile.  The
// two workgroups share nothing, so one's matrix phases overlap the other's vector / LDS phases without
// any cross-role hand-over, and a 4-wave workgroup barrier costs tens of cycles where an 8-wave one
// costs hundreds.  Per 32-edge tile i:
//   P1    product 1 from Xp[i&1], with the requests for rows / ids of tile i+1 and the gathers of ps[s], pr[r]
//         of tile i (registers, accumulator layout) issued between its matrix instructions;
//         ReLU, split -> Hp                                                            barrier B
//   P2    product 2 from Hp; ReLU, LayerNorm statistics -> red                         barrier C
//   LN    LayerNorm finish -> Os (aliases Hp, which is dead by now)                    barrier D
//   out   split + stage tile i+1 -> Xp[(i+1)&1]; receiver-segmented scan (16 rows per thread, the two
//         halves chained through LDS: barrier S); residual (rows kept in registers) + coalesced store;
//         segment sums -> aggr                                                         barrier E
// All global stores of a tile come after its last wait for a load, so no wait drains the store queue.
template <bool ATT>
__global__ __launch_bounds__(NT, 2) void mp_edge_h4_kernel(EdgeArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short Xp[2][2 * PLH];     // e_in tile, 2 fp16 planes, double buffered
    __shared__ __attribute__((aligned(16))) unsigned short Hp[2 * PLH];        // hidden tile, 2 fp16 planes; then Os
    __shared__ __attribute__((aligned(16))) float vec[3][H];                   // W_SCALE * b1, ln_w, ln_b
    __shared__ __attribute__((aligned(16))) float red[LN_RED_FLOATS];
    __shared__ float cout[H], tcarry[H];             // open sum after row 15 / after the tile, per column
    __shared__ int srange[2];
    float* Os = (float*)Hp;                          // pre-residual output tile (fp32)
    static_assert(sizeof(Hp) >= sizeof(float) * tile_floats(H / 4), "Os must fit into Hp");

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
    // wave 0 looks up the edge range first: its round trips to memory overlap the other waves' weight loads
    if (t < 64) block_edge_range(a.rowptr, a.n_nodes, a.n_edges, srange);
    WPlanesH<8> wa, wb;
    float rmax = 0.f;                                // largest magnitude this thread has split (range guard)
    const bool packed = a.pk0 != nullptr;            // weight planes prepared by pack_weights_kernel (range checked there)
    if (packed) { load_wpacked_h<8>(wa, a.pk0, w, lane); load_wpacked_h<8>(wb, a.pk1, w, lane); }
    else { load_wraw_h<8>(wa, a.w0, 3 * H, 2 * H, 32 * w + j, h); load_wraw_h<8>(wb, a.w1, H, 0, 32 * w + j, h); }
    if (t < H) { vec[0][t] = a.b1[t] * W_SCALE; vec[1][t] = a.ln_w[t]; vec[2][t] = a.ln_b[t]; tcarry[t] = 0.f; }
    __builtin_amdgcn_sched_barrier(0);               // all loads in flight before the first conversion
    if (!packed) { split_wraw_h<8>(wa, rmax); split_wraw_h<8>(wb, rmax); }
    __syncthreads();
    const int e0 = __builtin_amdgcn_readfirstlane(srange[0]), e1 = __builtin_amdgcn_readfirstlane(srange[1]);
    if (e0 >= e1) { range_report(rmax); return; }
    const int last = e1 - e0 - 1;                    // row offsets are relative to e0 and clamped to [0, last]
    const float* ein_g = a.e_in + (size_t)e0 * H;
    float* eout_g = a.e_out ? a.e_out + (size_t)e0 * H : nullptr;      // no edge output: never dereferenced
    const int* snd_g = a.sender + e0;
    const int* rcv_g = a.recv + e0;
    const int srow = t >> 5, sq = t & 31;            // row map: rows srow + 8*it, 16-byte chunk sq
    const int sc = t & (H - 1);                      // scan map: column sc, rows 16*sg .. 16*sg+15
    const int sg = __builtin_amdgcn_readfirstlane(t >> 7);

    f32x4 cur[4], pre[4];                            // rows of this tile (residual) and of the next one
#pragma unroll
    for (int it = 0; it < 4; ++it) cur[it] = *(const f32x4*)(ein_g + (unsigned)(min(srow + 8 * it, last) * H + 4 * sq));
    int s_cur = snd_g[min(j, last)], r_cur = rcv_g[min(j, last)];
#pragma unroll
    for (int it = 0; it < 4; ++it) hplanes_store4(Xp[0], PLH, PKH, srow + 8 * it, 4 * sq, cur[it], rmax);
    __syncthreads();
    STAMP_DECL
    int b = 0;
    for (int rel = 0; rel <= last; rel += TM, b ^= 1) {
        STAMP(0);
        const int nvalid = min(TM, last + 1 - rel);
        // This tile's gathers and the next tile's rows / ids are issued between the matrix instructions of the
        // first product, two vector-memory loads per k-step: issued in one burst at the top, the loads of the
        // four waves queue up in the CU's address unit and every wave stalls on the issue.  The streamed rows
        // are marked non-temporal (read once; the gathered projections are what should stay in L2).
        f32x4 gs[4], gr[4];
        int s_nxt, r_nxt, rb;
        const int ra = r_cur;                                                // receiver of tile row j ...
        f32x16 acc = zero16();
        {
            const unsigned so = (unsigned)s_cur * H + 32 * w + 4 * h, ro = (unsigned)r_cur * H + 32 * w + 4 * h;
            const int enx = min(rel + TM + j, last);
            const unsigned short* r = Xp[b] + j * PKH + 8 * h;
            u32x4 x1 = *(const u32x4*)(r), x2 = *(const u32x4*)(r + PLH);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            STAMP(1);
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
                    pre[c - 4] = __builtin_nontemporal_load(
                        (const f32x4*)(ein_g + (unsigned)(min(rel + TM + srow + 8 * (c - 4), last) * H + 4 * sq)));
                    if (c == 4) s_nxt = snd_g[enx];
                    if (c == 5) r_nxt = rcv_g[enx];
                    if (c == 6) rb = (rel + j + 1 <= last) ? rcv_g[min(rel + j + 1, last)] : -1;   // ... and of the row after it
                    if (c == 7) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                }
                acc = mfma32_f16(wa.p[1][c], x1, acc);   // h2 * h1
                acc = mfma32_f16(wa.p[0][c], x2, acc);   // h1 * h2
                acc = mfma32_f16(wa.p[0][c], x1, acc);   // h1 * h1 (the order of wave_gemm_h)
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                x1 = n1; x2 = n2;
            }
        }
        STAMP(2);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaf(acc[4 * g + k], W_UNSCALE, gs[g][k] + gr[g][k]), 0.f);
            hplanes_store4(Hp, PLH, PKH, j, 32 * w + 8 * g + 4 * h, v, rmax);
        }
        STAMP(3);
        __syncthreads();                                                     // (B) hidden tile complete
        STAMP(4);
        acc = acc_from_vec(vec[0], w, h);
        acc = wave_gemm_h<0, 8, 8>(Hp, PLH, PKH, 0, wb, j, h, acc);
        acc = relu16(acc);
        ln_local_stats_rows(acc, red, w, j, h);
        STAMP(5);
        __syncthreads();                                                     // (C) statistics exchanged; Hp dead
        STAMP(6);
        acc = ln_finish_rows(acc, red, w, j, h, vec[1], vec[2], LN_EPS * W_SCALE * W_SCALE);
        acc_to_tile(Os, acc, w, j, h);
        STAMP(7);
        __syncthreads();                                                     // (D) output tile complete
        STAMP(8);
        // next tile into the other plane buffer (its rows were requested at the top of this pass)
#pragma unroll
        for (int it = 0; it < 4; ++it) hplanes_store4(Xp[b ^ 1], PLH, PKH, srow + 8 * it, 4 * sq, pre[it], rmax);
        STAMP(9);
        // receiver-segmented sum: thread (column sc, group sg) scans rows 16sg .. 16sg+15 in order;
        // xr[r] = sum of the rows since the last segment end before r
        const unsigned mask = (unsigned)__ballot(j < nvalid && ra != rb);   // bit r: row r ends its segment
        const unsigned gmask = (mask >> (16 * sg)) & 0xFFFFu;
        float xr[16], part = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            xr[r] = Os[(16 * sg + r) * (H + 4) + sc];
            if constexpr (ATT) xr[r] *= a.ew[e0 + min(rel + 16 * sg + r, last)];      // wave-uniform address
        }
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
        STAMP(10);
        __syncthreads();                                                     // (S) cout visible
        STAMP(11);
        if (sg == 1) cin = cout[sc];
        {
            float* eo = eout_g + (unsigned)(rel * H);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = srow + 8 * it;
                if (row < nvalid && a.e_out) __builtin_nontemporal_store(y[it], (f32x4*)(eo + (unsigned)(row * H + 4 * sq)));
            }
        }
        bool open = true;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if ((gmask >> r) & 1u) {
                const int node = __builtin_amdgcn_readlane(ra, 16 * sg + r);
                a.aggr[(unsigned)(node * H + sc)] = open ? cin + xr[r] : xr[r];
                open = false;
            }
        }
        if (sg == 1) tcarry[sc] = gmask ? part : cin + part;
        STAMP(12);
        __syncthreads();                                                     // (E) Os / Xp[b^1] / tcarry settled
        s_cur = s_nxt; r_cur = r_nxt;
#pragma unroll
        for (int it = 0; it < 4; ++it) cur[it] = pre[it];
    }
    range_report(rmax);
    STAMP_FLUSH(w == 0 ? 0 : (w == 2 ? 1 : -1))
}

// ======================================================================== processor: nodes
// Node update, optionally followed by the NEXT layer's per-node projections of the freshly
// updated latents (PROJ): one 4-wave workgroup per CU, all five weight slices in registers
// (the unified 512-entry file at one wave per SIMD), next tile's rows prefetched into
// registers during the products.
struct NodeArgs {
    const float* x; const float* aggr; const int* rowptr;
    const float* w0; const float* b0; const float* w1; const float* b1; const float* ln_w; const float* ln_b;
    const float* nw0; const float* nb0;          // next layer's first edge Linear [H][3H] and bias (PROJ)
    float* x_out; float* ps; float* pr; int n_nodes;
    const void* pk0; const void* pk1; const void* pks; const void* pkr;   // packed planes of w0, w1, nw0[:, :H], nw0[:, H:2H], or null
};

template <bool PROJ>
__global__ __launch_bounds__(NT, 1) void mp_node_kernel(NodeArgs a) {
    __shared__ __attribute__((aligned(16))) float XA[tile_floats(H / 2)];  // [x | aggr] tile
    __shared__ __attribute__((aligned(16))) float Hs[tile_floats(H / 4)];  // hidden tile, then the new x
    __shared__ __attribute__((aligned(16))) float Ps[PROJ ? tile_floats(H / 4) : 4];
    __shared__ __attribute__((aligned(16))) float Pr[PROJ ? tile_floats(H / 4) : 4];
    __shared__ __attribute__((aligned(16))) float vec[5][H];       // b0, b1, ln_w, ln_b, next b0
    __shared__ float red[16 * TM];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
    f32x4 wa[32], wb[16], ws[16], wr[16];
    load_wfrag<32>(wa, a.w0, 2 * H, 0, 32 * w + j, h);
    load_wfrag<16>(wb, a.w1, H, 0, 32 * w + j, h);
    if constexpr (PROJ) {
        load_wfrag<16>(ws, a.nw0, 3 * H, 0, 32 * w + j, h);
        load_wfrag<16>(wr, a.nw0, 3 * H, H, 32 * w + j, h);
    }
    if (t < H) {
        vec[0][t] = a.b0[t]; vec[1][t] = a.b1[t]; vec[2][t] = a.ln_w[t]; vec[3][t] = a.ln_b[t];
        vec[4][t] = PROJ ? a.nb0[t] : 0.f;
    }
    const int ntiles = (a.n_nodes + TM - 1) / TM;
    const int nb = gridDim.x, rho = range_of_block(blockIdx.x, nb);
    const int t0 = (int)((long long)rho * ntiles / nb), t1 = (int)((long long)(rho + 1) * ntiles / nb);
    if (t0 >= t1) return;
    const int srow = t >> 5, sq = t & 31;
    const int nlast = a.n_nodes - 1;

    f32x4 px[4], pa[4];
    auto prefetch = [&](int tile) {
        int deg[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int n = min(tile * TM + srow + 8 * it, nlast);
            px[it] = *(const f32x4*)(a.x + (size_t)n * H + 4 * sq);
            deg[it] = a.rowptr[n + 1] - a.rowptr[n];
            pa[it] = *(const f32x4*)(a.aggr + (size_t)n * H + 4 * sq);
        }
        // aggr rows of nodes without in-edges were never written: read them as zero (all loads first, then the selects)
#pragma unroll
        for (int it = 0; it < 4; ++it) pa[it] = deg[it] > 0 ? pa[it] : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    prefetch(t0);
    __syncthreads();
    for (int tile = t0; tile < t1; ++tile) {
        const int base = tile * TM, nvalid = min(TM, a.n_nodes - base);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            *(f32x4*)(XA + toff(srow + 8 * it, sq, H / 2)) = px[it];
            *(f32x4*)(XA + toff(srow + 8 * it, 32 + sq, H / 2)) = pa[it];
        }
        __syncthreads();
        prefetch(min(tile + 1, t1 - 1));
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc = acc_from_vec(vec[0], w, h);
        acc = wave_gemm<32>(XA, H / 2, 0, wa, j, h, acc);
        acc = relu16(acc);
        acc_to_tile(Hs, acc, w, j, h);
        __syncthreads();
        acc = acc_from_vec(vec[1], w, h);
        acc = wave_gemm<16>(Hs, H / 4, 0, wb, j, h, acc);
        acc = relu16(acc);
        acc = layernorm_rows(acc, red, w, j, h, vec[2], vec[3]);     // barrier inside: Hs reads are done
#pragma unroll
        for (int g = 0; g < 4; ++g) {                                // residual, in accumulator layout
            const f32x4 xr = *(const f32x4*)(XA + toff(j, 8 * w + 2 * g + h, H / 2));
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[4 * g + k] += xr[k];
        }
        acc_to_tile(Hs, acc, w, j, h);
        __syncthreads();
        if constexpr (PROJ) {
            f32x16 as = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 ar = acc_from_vec(vec[4], w, h);
            as = wave_gemm<16>(Hs, H / 4, 0, ws, j, h, as);
            ar = wave_gemm<16>(Hs, H / 4, 0, wr, j, h, ar);
            acc_to_tile(Ps, as, w, j, h);
            acc_to_tile(Pr, ar, w, j, h);
            __syncthreads();
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = srow + 8 * it;
            if (row < nvalid) {
                const size_t o = (size_t)(base + row) * H + 4 * sq;
                *(f32x4*)(a.x_out + o) = *(const f32x4*)(Hs + toff(row, sq, H / 4));
                if constexpr (PROJ) {
                    *(f32x4*)(a.ps + o) = *(const f32x4*)(Ps + toff(row, sq, H / 4));
                    *(f32x4*)(a.pr + o) = *(const f32x4*)(Pr + toff(row, sq, H / 4));
                }
            }
        }
        // XA is rewritten at the top of the next iteration: every thread has passed the last
        // barrier above after its final XA read; Hs / Ps / Pr are rewritten only after the
        // next iteration's first barrier, which follows this store pass in program order
    }
}

// ======================================================================== processor: nodes, fp16-split products
// The node update (and the next block's projections) with every product on the fp16 matrix pipe
// (two-term split, three partial products): one 4-wave workgroup per CU, one wave per SIMD, all five
// weight slices as fp16 planes in registers (320 of the 512 per lane).  With 120 short MFMAs per
// 32-node tile instead of 320 fp32 ones the kernel is bound by its HBM rows, not by the matrix pipe.
constexpr int PKH2 = 2 * H + 8;     // ... of the 256-column [x | aggr] tile

template <bool PROJ, bool PACKED>
__global__ __launch_bounds__(NT, 1) void mp_node_h_kernel(NodeArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short XAp[2 * TM * PKH2];   // [x | aggr] tile, 2 planes
    __shared__ __attribute__((aligned(16))) float Xf[tile_floats(H / 4)];        // x rows in fp32 (residual)
    __shared__ __attribute__((aligned(16))) unsigned short Hp[2 * TM * PKH];     // hidden tile, 2 planes
    __shared__ __attribute__((aligned(16))) unsigned short Np[PROJ ? 2 * TM * PKH : 8];   // new x, 2 planes
    __shared__ __attribute__((aligned(16))) float Os[tile_floats(H / 4)];        // new x in fp32 (store)
    __shared__ __attribute__((aligned(16))) float Ps[PROJ ? tile_floats(H / 4) : 4];
    __shared__ __attribute__((aligned(16))) float Pr[PROJ ? tile_floats(H / 4) : 4];
    __shared__ __attribute__((aligned(16))) float vec[5][H];       // b0, b1, ln_w, ln_b, next b0
    __shared__ float red[16 * TM];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
    const int ntiles = (a.n_nodes + TM - 1) / TM;
    const int nb = gridDim.x, rho = range_of_block(blockIdx.x, nb);
    const int t0 = (int)((long long)rho * ntiles / nb), t1 = (int)((long long)(rho + 1) * ntiles / nb);
    if (t0 >= t1) return;
    const int srow = t >> 5, sq = t & 31;
    const int nlast = a.n_nodes - 1;
    constexpr int PL2 = TM * PKH2, PL = TM * PKH;

    f32x4 px[4], pa[4];
    auto prefetch = [&](int tile) {
        int deg[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int n = min(tile * TM + srow + 8 * it, nlast);
            px[it] = *(const f32x4*)(a.x + (size_t)n * H + 4 * sq);
            deg[it] = a.rowptr[n + 1] - a.rowptr[n];
            pa[it] = *(const f32x4*)(a.aggr + (size_t)n * H + 4 * sq);
        }
        // aggr rows of nodes without in-edges were never written: read them as zero (all loads first, then the selects)
#pragma unroll
        for (int it = 0; it < 4; ++it) pa[it] = deg[it] > 0 ? pa[it] : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    WPlanesH<16> wa;
    WPlanesH<8> wb, ws, wr;
    // all weight loads in flight before the first conversion (converted one set at a time, with this kernel's
    // register file full, the later sets load two values per round trip to memory)
    constexpr bool packed = PACKED;                  // weight planes prepared by pack_weights_kernel (range checked there)
    if constexpr (packed) {
        load_wpacked_h<16>(wa, a.pk0, w, lane);
        load_wpacked_h<8>(wb, a.pk1, w, lane);
        if constexpr (PROJ) { load_wpacked_h<8>(ws, a.pks, w, lane); load_wpacked_h<8>(wr, a.pkr, w, lane); }
    } else {
        load_wraw_h<16>(wa, a.w0, 2 * H, 0, 32 * w + j, h);
        load_wraw_h<8>(wb, a.w1, H, 0, 32 * w + j, h);
        if constexpr (PROJ) {
            load_wraw_h<8>(ws, a.nw0, 3 * H, 0, 32 * w + j, h);
            load_wraw_h<8>(wr, a.nw0, 3 * H, H, 32 * w + j, h);
        }
    }
    float v5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (t < H) { v5[0] = a.b0[t]; v5[1] = a.b1[t]; v5[2] = a.ln_w[t]; v5[3] = a.ln_b[t]; v5[4] = PROJ ? a.nb0[t] : 0.f; }
    prefetch(t0);                                    // first tile's rows: in flight with the weights
    __builtin_amdgcn_sched_barrier(0);
    if (t < H) {
#pragma unroll
        for (int k = 0; k < 5; ++k) vec[k][t] = v5[k];
    }
    if constexpr (!packed) {
        float m = 0.f;
        split_wraw_h<16>(wa, m);
        split_wraw_h<8>(wb, m);
        if constexpr (PROJ) { split_wraw_h<8>(ws, m); split_wraw_h<8>(wr, m); }
        range_report(m);
    }
    // accumulator (W_SCALE * product) -> value: acc * W_UNSCALE + bias, optional ReLU
    auto finish = [&](f32x16 acc, const float* bias, bool relu) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *(const f32x4*)(bias + 32 * w + 8 * g + 4 * h);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float v = fmaf(acc[4 * g + k], W_UNSCALE, b[k]);
                acc[4 * g + k] = relu ? fmaxf(v, 0.f) : v;
            }
        }
        return acc;
    };
    // range guard: this kernel has no register to spare for a running maximum (all 512 are taken), so every
    // group of splits reports its own maximum right away
    auto planes_from_acc = [&](unsigned short* tile, const f32x16& v) {
        float m = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            hplanes_store4(tile, PL, PKH, j, 32 * w + 8 * g + 4 * h, f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]}, m);
        range_report(m);
    };
    // piece c (of 8) of staging the rows held in px / pa: x rows in fp32 (residual) and as fp16 planes, aggr rows as planes
    auto stage_piece = [&](int c, float& m) {
        const int it = c >> 1, row = srow + 8 * it;
        if ((c & 1) == 0) {
            *(f32x4*)(Xf + toff(row, sq, H / 4)) = px[it];
            hplanes_store4(XAp, PL2, PKH2, row, 4 * sq, px[it], m);
        } else {
            hplanes_store4(XAp, PL2, PKH2, row, H + 4 * sq, pa[it], m);
        }
    };
    {
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) stage_piece(c, m);
        range_report(m);
    }
    __syncthreads();                                                 // first tile staged
    STAMP_DECL
    for (int tile = t0; tile < t1; ++tile) {
        const int base = tile * TM, nvalid = min(TM, a.n_nodes - base);
        // The tile is staged already (by the previous pass, under its projections): its rows have left px / pa,
        // which take the next tile's rows now -- most of a pass to arrive.
        STAMP(2);
        prefetch(min(tile + 1, t1 - 1));
        STAMP(3);
        f32x16 acc = zero16();
        acc = wave_gemm_h<0, 16, 16>(XAp, PL2, PKH2, 0, wa, j, h, acc);
        STAMP(4);
        acc = finish(acc, vec[0], true);
        planes_from_acc(Hp, acc);
        STAMP(5);
        __syncthreads();                                             // (2) hidden tile complete
        STAMP(6);
        acc = zero16();
        acc = wave_gemm_h<0, 8, 8>(Hp, PL, PKH, 0, wb, j, h, acc);
        acc = finish(acc, vec[1], true);
        STAMP(7);
        acc = layernorm_rows(acc, red, w, j, h, vec[2], vec[3]);     // (3) barrier inside
        STAMP(8);
#pragma unroll
        for (int g = 0; g < 4; ++g) {                                // residual, in accumulator layout
            const f32x4 xr = *(const f32x4*)(Xf + toff(j, 8 * w + 2 * g + h, H / 4));
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[4 * g + k] += xr[k];
        }
        acc_to_tile(Os, acc, w, j, h);
        if constexpr (PROJ) planes_from_acc(Np, acc);
        STAMP(9);
        __syncthreads();                                             // (4) new x complete
        STAMP(10);
        // The next tile's rows (requested after barrier 1) are taken as arrived here, before this tile's first store:
        // loads and stores share one counter that only counts down in order, so a wait for these rows placed after
        // the stores (where the compiler would put it: at their first use, in the next pass) also waits for the stores.
#pragma unroll
        for (int it = 0; it < 4; ++it) asm volatile("" : "+v"(px[it]), "+v"(pa[it]));
        if constexpr (PROJ) {
            // the new rows leave before the projections: their stores queue up behind the matrix instructions
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = srow + 8 * it;
                if (row < nvalid) *(f32x4*)(a.x_out + (size_t)(base + row) * H + 4 * sq) = *(const f32x4*)(Os + toff(row, sq, H / 4));
            }
            // The next tile is staged between the matrix instructions of the projections, one piece per k-step: every
            // wave has passed barrier 4, so nobody reads XAp (product 1) or Xf (residual) of this tile any more.
            f32x16 as = zero16(), ar = zero16();
            float m = 0.f;
            wave_gemm_h2_with<0, 8, 8, 24, 3>(Np, PL, PKH, 0, ws, wr, j, h, as, ar, [&](int c) { stage_piece(c, m); });
            range_report(m);
#pragma unroll
            for (int i = 0; i < 16; ++i) as[i] *= W_UNSCALE;
            ar = finish(ar, vec[4], false);
            acc_to_tile(Ps, as, w, j, h);
            acc_to_tile(Pr, ar, w, j, h);
            STAMP(11);
            __syncthreads();                                         // (5) projections complete, next tile staged
            STAMP(12);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = srow + 8 * it;
                if (row < nvalid) {
                    const size_t o = (size_t)(base + row) * H + 4 * sq;
                    *(f32x4*)(a.ps + o) = *(const f32x4*)(Ps + toff(row, sq, H / 4));
                    *(f32x4*)(a.pr + o) = *(const f32x4*)(Pr + toff(row, sq, H / 4));
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = srow + 8 * it;
                if (row < nvalid) *(f32x4*)(a.x_out + (size_t)(base + row) * H + 4 * sq) = *(const f32x4*)(Os + toff(row, sq, H / 4));
            }
            float m = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) stage_piece(c, m);
            range_report(m);
            __syncthreads();                                         // (5) next tile staged
        }
        STAMP(13);
        // Os / Ps / Pr / Hp / Np are rewritten only after the next pass's barriers
    }
    if constexpr (PROJ) { STAMP_FLUSH(w == 0 ? 0 : (w == 2 ? 1 : -1)) }
}

// ======================================================================== processor: projections
struct ProjArgs { const float* x; const float* w0; const float* b0; float* ps; float* pr; int n_nodes; };

template <bool HALF>
__global__ __launch_bounds__(NT, 2) void mp_project_kernel(ProjArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short Xp[HALF ? 2 * PLH : 2 * tile_floats(H / 4)];  // x tile: fp16 planes or fp32
    __shared__ __attribute__((aligned(16))) float Ps[tile_floats(H / 4)];
    __shared__ __attribute__((aligned(16))) float Pr[tile_floats(H / 4)];
    __shared__ __attribute__((aligned(16))) float vec[H];
    float* X = (float*)Xp;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
    f32x4 ws[HALF ? 1 : 16], wr[HALF ? 1 : 16];
    WPlanesH<HALF ? 8 : 1> hs, hr;
    if constexpr (HALF) {
        load_wplanes_h<8>(hs, a.w0, 3 * H, 0, 32 * w + j, h);
        load_wplanes_h<8>(hr, a.w0, 3 * H, H, 32 * w + j, h);
    } else {
        load_wfrag<16>(ws, a.w0, 3 * H, 0, 32 * w + j, h);
        load_wfrag<16>(wr, a.w0, 3 * H, H, 32 * w + j, h);
    }
    constexpr float BS = HALF ? W_SCALE : 1.0f, US = HALF ? W_UNSCALE : 1.0f;
    if (t < H) vec[t] = a.b0[t] * BS;

    const int ntiles = (a.n_nodes + TM - 1) / TM;
    const int nb = gridDim.x, rho = range_of_block(blockIdx.x, nb);
    const int t0 = (int)((long long)rho * ntiles / nb), t1 = (int)((long long)(rho + 1) * ntiles / nb);
    const int srow = t >> 5, sq = t & 31;
    float rmax = 0.f;                                // largest magnitude this thread has split (range guard)
    __syncthreads();
    for (int tile = t0; tile < t1; ++tile) {
        const int base = tile * TM, nvalid = min(TM, a.n_nodes - base);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = srow + 8 * it, n = min(base + row, a.n_nodes - 1);
            const f32x4 v = *(const f32x4*)(a.x + (size_t)n * H + 4 * sq);
            if constexpr (HALF) hplanes_store4(Xp, PLH, PKH, row, 4 * sq, v, rmax);
            else *(f32x4*)(X + toff(row, sq, H / 4)) = v;
        }
        __syncthreads();
        f32x16 as = zero16();
        f32x16 ar = acc_from_vec(vec, w, h);
        if constexpr (HALF) {
            wave_gemm_h2<0, 8, 8>(Xp, PLH, PKH, 0, hs, hr, j, h, as, ar);
#pragma unroll
            for (int i = 0; i < 16; ++i) { as[i] *= US; ar[i] *= US; }
        } else {
            as = wave_gemm<16>(X, H / 4, 0, ws, j, h, as);
            ar = wave_gemm<16>(X, H / 4, 0, wr, j, h, ar);
        }
        acc_to_tile(Ps, as, w, j, h);
        acc_to_tile(Pr, ar, w, j, h);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = srow + 8 * it;
            if (row < nvalid) {
                *(f32x4*)(a.ps + (size_t)(base + row) * H + 4 * sq) = *(const f32x4*)(Ps + toff(row, sq, H / 4));
                *(f32x4*)(a.pr + (size_t)(base + row) * H + 4 * sq) = *(const f32x4*)(Pr + toff(row, sq, H / 4));
            }
        }
        __syncthreads();
    }
    if constexpr (HALF) range_report(rmax);
}

// ======================================================================== encoder
struct EncArgs {
    const float* feat; const float* mean_std;
    const float* w0; const float* b0; const float* w1; const float* b1; const float* ln_w; const float* ln_b;
    float* out; int rows; int n_feat;
};

// First Linear on the vector ALU (F <= 16 inputs, weights of 4 output features per thread in registers,
// F4 = ceil(F / 4) input chunks), ReLU; second Linear on the matrix pipe; LayerNorm; coalesced store.
// SPLIT = true: three-term bf16 split (six products).  The encoders are the one place where the
// hidden values are not bounded by a LayerNorm upstream (a normalised input feature can be huge when
// its training variance was tiny), so they never use the range-limited fp16 split.  SPLIT = false: fp32 MFMA.
template <int F4, bool SPLIT>
__global__ __launch_bounds__(NT, 2) void encode_kernel(EncArgs a) {
    constexpr bool HALF = SPLIT;                     // (name kept short below)
    constexpr int FP = 4 * F4;                       // padded input width
    __shared__ __attribute__((aligned(16))) float Fs[TM * FP];
    __shared__ __attribute__((aligned(16))) unsigned short Hp[HALF ? 3 * PLANE : 2 * tile_floats(H / 4)];  // hidden tile: 3 bf16 planes, or fp32
    float* Hs = (float*)Hp;
    __shared__ __attribute__((aligned(16))) float Os[tile_floats(H / 4)];
    __shared__ __attribute__((aligned(16))) float vec[3][H];       // W_SCALE * b1, ln_w, ln_b
    __shared__ float ms[2 * MGN_MAX_FEAT];
    __shared__ __attribute__((aligned(16))) float red[LN_RED_FLOATS];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
    const int F = a.n_feat;
    WPlanes32 wb;
    f32x4 wf[HALF ? 1 : 16];
    if constexpr (HALF) load_wplanes32(wb, a.w1, H, 0, 32 * w + j, h);
    else load_wfrag<16>(wf, a.w1, H, 0, 32 * w + j, h);
    constexpr float BS = 1.0f;
    const int srow = t >> 5, sq = t & 31;
    // first-layer weights of features 4*sq .. 4*sq+3, zero padded to FP inputs
    float w0r[4][FP], b0r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        b0r[i] = a.b0[4 * sq + i];
#pragma unroll
        for (int k = 0; k < FP; ++k) w0r[i][k] = (k < F) ? a.w0[(size_t)(4 * sq + i) * F + k] : 0.f;
    }
    if (t < H) { vec[0][t] = a.b1[t] * BS; vec[1][t] = a.ln_w[t]; vec[2][t] = a.ln_b[t]; }
    if (t < MGN_MAX_FEAT) {
        ms[t] = (a.mean_std && t < F) ? a.mean_std[t] : 0.f;
        ms[MGN_MAX_FEAT + t] = (a.mean_std && t < F) ? a.mean_std[F + t] : 1.f;
    }
    const int ntiles = (a.rows + TM - 1) / TM;
    const int nb = gridDim.x, rho = range_of_block(blockIdx.x, nb);
    const int t0 = (int)((long long)rho * ntiles / nb), t1 = (int)((long long)(rho + 1) * ntiles / nb);
    // raw features of the next tile, requested one pass ahead (TM * FP <= 2 * NT elements)
    constexpr int NPF = (TM * FP + NT - 1) / NT;
    float pf[NPF];
    auto request = [&](int tile) {
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int idx = t + q * NT, row = idx / FP, k = idx - row * FP;
            const int n = min(tile * TM + row, a.rows - 1);
            pf[q] = (idx < TM * FP && k < F) ? a.feat[(size_t)n * F + k] : 0.f;
        }
    };
    if (t0 < t1) request(t0);
    __syncthreads();
    for (int tile = t0; tile < t1; ++tile) {
        const int base = tile * TM, nvalid = min(TM, a.rows - base);
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int idx = t + q * NT, k = idx % FP;
            if (idx < TM * FP) Fs[idx] = (k < F) ? (pf[q] - ms[k]) / ms[MGN_MAX_FEAT + k] : 0.f;
        }
        __syncthreads();
        if (tile + 1 < t1) request(tile + 1);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = srow + 8 * it;
            f32x4 hv = {b0r[0], b0r[1], b0r[2], b0r[3]};
#pragma unroll
            for (int k4 = 0; k4 < F4; ++k4) {
                const f32x4 f = *(const f32x4*)(Fs + row * FP + 4 * k4);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) hv[i] += w0r[i][4 * k4 + k] * f[k];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) hv[i] = fmaxf(hv[i], 0.f);
            if constexpr (HALF) planes_store4(Hp, row, 4 * sq, hv);
            else *(f32x4*)(Hs + toff(row, sq, H / 4)) = hv;
        }
        __syncthreads();
        f32x16 acc = acc_from_vec(vec[0], w, h);
        if constexpr (HALF) acc = wave_gemm3_32<0, 8>(Hp, wb, j, h, acc);
        else acc = wave_gemm<16>(Hs, H / 4, 0, wf, j, h, acc);
        ln_local_stats_rows(acc, red, w, j, h);
        __syncthreads();
        acc = ln_finish_rows(acc, red, w, j, h, vec[1], vec[2], LN_EPS * BS * BS);
        acc_to_tile(Os, acc, w, j, h);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = srow + 8 * it;
            if (row < nvalid)
                *(f32x4*)(a.out + (size_t)(base + row) * H + 4 * sq) = *(const f32x4*)(Os + toff(row, sq, H / 4));
        }
        // Fs / Hp / Os of the next tile are written only after barriers every thread
        // reaches after this store pass; red is rewritten after the next Hp barrier
    }
}

// ======================================================================== decoder + integrator
struct DecArgs {
    const float* x; const float* w0; const float* b0; const float* w1; const float* b1;
    float* delta_norm; const float* out_mean_std; const float* world_pos; const float* phi; const float* node_type;
    float* pred; int n_nodes; int time_dim; int out_dim;
    float* next_world; float* next_phi;      // or null: the last predicted step also written as the next state (may alias world_pos / phi)
};

template <bool HALF>
__global__ __launch_bounds__(NT, 2) void decode_kernel(DecArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short Xp[HALF ? 2 * PLH : 2 * tile_floats(H / 4)];   // x tile: fp16 planes or fp32
    __shared__ __attribute__((aligned(16))) unsigned short Hq[HALF ? 2 * PLH : 2 * tile_floats(H / 4)];   // hidden tile
    __shared__ __attribute__((aligned(16))) float vec[H];
    __shared__ float ob[MGN_MAX_DEC_OUT], om[MGN_MAX_DEC_OUT], os[MGN_MAX_DEC_OUT];
    __shared__ int ott[MGN_MAX_DEC_OUT], okk[MGN_MAX_DEC_OUT];     // output index -> (time step, component): no divisions in the loop
    float* X = (float*)Xp;
    float* Hs = (float*)Hq;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
    const int n_out = a.time_dim * a.out_dim;
    constexpr float BS = HALF ? W_SCALE : 1.0f, US = HALF ? W_UNSCALE : 1.0f;
    f32x4 wa[HALF ? 1 : 16], wb[HALF ? 1 : 16];
    WPlanesH<HALF ? 8 : 1> ha, hb;
    const bool hidden_in = a.w0 == nullptr;          // x already holds the hidden activations (first Linear done by the caller)
    if constexpr (HALF) {
        if (!hidden_in) load_wplanes_h<8>(ha, a.w0, H, 0, 32 * w + j, h);
        load_wplanes_h<8>(hb, a.w1, H, 0, min(j, n_out - 1), h);
        if (!(w == 0 && j < n_out)) {                // rows >= n_out read as zero
#pragma unroll
            for (int c = 0; c < 8; ++c) { hb.p[0][c] = u32x4{0, 0, 0, 0}; hb.p[1][c] = u32x4{0, 0, 0, 0}; }
        }
    } else {
        if (!hidden_in) load_wfrag<16>(wa, a.w0, H, 0, 32 * w + j, h);
        load_wfrag<16>(wb, a.w1, H, 0, min(j, n_out - 1), h, w == 0 && j < n_out);   // rows >= n_out read as zero
    }
    if (t < H) vec[t] = hidden_in ? 0.f : a.b0[t] * BS;
    if (t < MGN_MAX_DEC_OUT) {
        ob[t] = (t < n_out) ? a.b1[t] : 0.f;
        om[t] = (a.pred && t < n_out) ? a.out_mean_std[t] : 0.f;
        os[t] = (a.pred && t < n_out) ? a.out_mean_std[n_out + t] : 1.f;
        ott[t] = t / a.out_dim; okk[t] = t % a.out_dim;
    }
    const int ntiles = (a.n_nodes + TM - 1) / TM;
    const int nb = gridDim.x, rho = range_of_block(blockIdx.x, nb);
    const int t0 = (int)((long long)rho * ntiles / nb), t1 = (int)((long long)(rho + 1) * ntiles / nb);
    const int srow = t >> 5, sq = t & 31;
    f32x4 pre[4];                                    // rows of the next tile, requested one pass ahead
    float rmax = 0.f;                                // largest magnitude this thread has split (range guard)
    if (t0 < t1) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
            pre[it] = *(const f32x4*)(a.x + (size_t)min(t0 * TM + srow + 8 * it, a.n_nodes - 1) * H + 4 * sq);
    }
    __syncthreads();
    for (int tile = t0; tile < t1; ++tile) {
        const int base = tile * TM, nvalid = min(TM, a.n_nodes - base);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = srow + 8 * it;
            if constexpr (HALF) hplanes_store4(hidden_in ? Hq : Xp, PLH, PKH, row, 4 * sq, pre[it], rmax);
            else *(f32x4*)((hidden_in ? Hs : X) + toff(row, sq, H / 4)) = pre[it];
        }
        __syncthreads();
        if (tile + 1 < t1) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                pre[it] = *(const f32x4*)(a.x + (size_t)min(base + TM + srow + 8 * it, a.n_nodes - 1) * H + 4 * sq);
        }
        f32x16 acc = acc_from_vec(vec, w, h);
        if (hidden_in) {
            // nothing to do: the staged rows are the hidden tile
        } else if constexpr (HALF) {
            acc = wave_gemm_h<0, 8, 8>(Xp, PLH, PKH, 0, ha, j, h, acc);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(acc[4 * g + k] * US, 0.f);
                hplanes_store4(Hq, PLH, PKH, j, 32 * w + 8 * g + 4 * h, v, rmax);
            }
        } else {
            acc = wave_gemm<16>(X, H / 4, 0, wa, j, h, acc);
            acc = relu16(acc);
            acc_to_tile(Hs, acc, w, j, h);
        }
        __syncthreads();
        if (w == 0) {
            f32x16 o = zero16();
            if constexpr (HALF) o = wave_gemm_h<0, 8, 8>(Hq, PLH, PKH, 0, hb, j, h, o);
            else o = wave_gemm<16>(Hs, H / 4, 0, wb, j, h, o);
            if (j < nvalid) {
                const int n = base + j;
                // the row's current state, read by both lanes of the row before either writes the next state over it
                float cur3[3] = {0.f, 0.f, 0.f};
                if (a.pred) { cur3[0] = a.world_pos[(size_t)n * 2]; cur3[1] = a.world_pos[(size_t)n * 2 + 1]; cur3[2] = a.phi[n]; }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int oi = (reg & 3) + 8 * (reg >> 2) + 4 * h;     // decoder output index
                    if (oi < n_out) {
                        const int tt = ott[oi], k = okk[oi];
                        const float dn = (o[reg] * US + ob[oi]) * (float)(tt + 1);
                        if (a.delta_norm) a.delta_norm[((size_t)tt * a.n_nodes + n) * a.out_dim + k] = dn;
                        if (a.pred) {
                            float d = os[oi] * dn + om[oi];
                            if (a.node_type[(size_t)n * 5 + k + 1] == 1.0f) d = 0.f;
                            const float nxt = (k == 0 ? cur3[0] : k == 1 ? cur3[1] : cur3[2]) + d;
                            a.pred[((size_t)tt * a.n_nodes + n) * 3 + k] = nxt;
                            if (a.next_world && tt == a.time_dim - 1) {
                                if (k < 2) a.next_world[(size_t)n * 2 + k] = nxt; else a.next_phi[n] = nxt;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    if constexpr (HALF) range_report(rmax);
}

// ======================================================================== features / normaliser
__device__ inline void node_feature_row(const float* __restrict__ world_pos, const float* __restrict__ mesh_pos,
                                        const float* __restrict__ phi, const float* __restrict__ swell, int n_swell,
                                        const float* __restrict__ node_type, const float* __restrict__ mat_param, int n,
                                        float* f) {
    f[0] = world_pos[2 * n] - mesh_pos[2 * n];
    f[1] = world_pos[2 * n + 1] - mesh_pos[2 * n + 1];
    f[2] = phi[n];
    for (int k = 0; k < n_swell; ++k) f[3 + k] = swell[(size_t)n * n_swell + k];
    for (int k = 0; k < 5; ++k) f[3 + n_swell + k] = node_type[(size_t)n * 5 + k];
    if (mat_param) { f[8 + n_swell] = mat_param[0]; f[9 + n_swell] = mat_param[1]; }
}

__global__ void node_features_kernel(const float* __restrict__ world_pos, const float* __restrict__ mesh_pos,
                                     const float* __restrict__ phi, const float* __restrict__ swell, int n_swell,
                                     const float* __restrict__ node_type, const float* __restrict__ mat_param,
                                     float* __restrict__ feat, int n_nodes) {
    const int F = 3 + n_swell + 5 + (mat_param ? 2 : 0);
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < n_nodes; n += gridDim.x * blockDim.x)
        node_feature_row(world_pos, mesh_pos, phi, swell, n_swell, node_type, mat_param, n, feat + (size_t)n * F);
}

__device__ inline void edge_feature_row(const float* __restrict__ world_pos, const float* __restrict__ mesh_pos,
                                        const float* __restrict__ phi, int s, int r, float* f) {
    const float mx = mesh_pos[2 * s] - mesh_pos[2 * r], my = mesh_pos[2 * s + 1] - mesh_pos[2 * r + 1];
    const float wx = world_pos[2 * s] - world_pos[2 * r], wy = world_pos[2 * s + 1] - world_pos[2 * r + 1];
    f[0] = mx; f[1] = my; f[2] = sqrtf(mx * mx + my * my);
    f[3] = wx; f[4] = wy; f[5] = sqrtf(wx * wx + wy * wy);
    f[6] = phi[s] - phi[r];
}

__global__ void edge_features_kernel(const float* __restrict__ world_pos, const float* __restrict__ mesh_pos,
                                     const float* __restrict__ phi, const int* __restrict__ sender,
                                     const int* __restrict__ recv, float* __restrict__ feat, int n_edges) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += gridDim.x * blockDim.x)
        edge_feature_row(world_pos, mesh_pos, phi, sender[e], recv[e], feat + (size_t)e * 7);
}

// ======================================================================== attention weights of the messages
// core/meshgraphnet.py:100-107: alpha_e = leaky_relu(<A[s_e], A[r_e]>) with A = attn_lin(x); the message of edge e
// enters its receiver's sum with weight alpha_e / max(sum of alpha over the receiver's in-edges, 1e-6).
// 32 lanes per edge (16 bytes of each row per lane); then one thread per receiver adds its alphas in CSR order.
__global__ __launch_bounds__(NT) void edge_alpha_kernel(const float* __restrict__ A, const int* __restrict__ sender,
                                                        const int* __restrict__ recv, float* __restrict__ alpha, int n_edges) {
    const int l = threadIdx.x & 31;
    for (long long e = ((long long)blockIdx.x * NT + threadIdx.x) >> 5; e < n_edges; e += ((long long)gridDim.x * NT) >> 5) {
        const f32x4 u = *(const f32x4*)(A + (size_t)sender[e] * H + 4 * l), v = *(const f32x4*)(A + (size_t)recv[e] * H + 4 * l);
        float d = (u[0] * v[0] + u[1] * v[1]) + (u[2] * v[2] + u[3] * v[3]);
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) d += __shfl_xor(d, m, 32);
        if (l == 0) alpha[e] = d > 0.f ? d : 0.01f * d;
    }
}

__global__ void edge_alpha_normalize_kernel(const int* __restrict__ rowptr, const float* __restrict__ alpha,
                                            float* __restrict__ weight, int n_nodes) {
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < n_nodes; n += gridDim.x * blockDim.x) {
        const int b = rowptr[n], e = rowptr[n + 1];
        float s = 0.f;
        for (int k = b; k < e; ++k) s += alpha[k];
        const float d = fmaxf(s, 1e-6f);
        for (int k = b; k < e; ++k) weight[k] = alpha[k] / d;
    }
}

// ---- packed weight planes (fp16-split path): one workgroup per 128-row weight set, thread (wave, lane) writes the
// register quads it will later hold, [wave][k-step][plane][lane], so that the consumers' loads are fully coalesced.
struct PackSlice { const float* w; int ld, col0, ks; unsigned dst; };       // dst: offset into the block, in u32x4
struct PackTable { PackSlice s[MGN_MAX_PACK_SLICES]; };

__global__ __launch_bounds__(NT) void pack_weights_kernel(PackTable tab, u32x4* __restrict__ packed) {
    const PackSlice s = tab.s[blockIdx.x];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
    const float* row = s.w + (size_t)(32 * w + j) * s.ld + s.col0 + 8 * h;
    u32x4* dst = packed + s.dst + (size_t)w * s.ks * 2 * 64 + lane;
    float m = 0.f;
    for (int c = 0; c < s.ks; ++c) {
        const f32x4 lo4 = *(const f32x4*)(row + 16 * c) * W_SCALE, hi4 = *(const f32x4*)(row + 16 * c + 4) * W_SCALE;
        track4(m, lo4); track4(m, hi4);
        u32x2 a1, a2, b1, b2;
        split2_pack4(lo4, a1, a2);
        split2_pack4(hi4, b1, b2);
        dst[(2 * c) * 64] = u32x4{a1[0], a1[1], b1[0], b1[1]};
        dst[(2 * c + 1) * 64] = u32x4{a2[0], a2[1], b2[0], b2[1]};
    }
    range_report(m);
}

constexpr int SUM_BLOCKS_MAX = 1024;

// One block's float64 column sums [sum | sum of squares] from the per-thread sums: fixed order -- butterfly within each
// wave, then the four wave sums in wave order (one barrier in all).
__device__ inline void block_partial_sums(const double* s, const double* q, int F, double* __restrict__ partial) {
    __shared__ double sh[NT / 64][2 * MGN_MAX_FEAT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < MGN_MAX_FEAT; ++k) {
        if (k >= F) continue;
        double a = s[k], b = q[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
        if (lane == 0) { sh[wv][k] = a; sh[wv][MGN_MAX_FEAT + k] = b; }
    }
    __syncthreads();
    const int k = threadIdx.x;
    if (k < 2 * MGN_MAX_FEAT && (k % MGN_MAX_FEAT) < F) {
        double tot = sh[0][k];
#pragma unroll
        for (int v = 1; v < NT / 64; ++v) tot += sh[v][k];
        partial[(size_t)blockIdx.x * 2 * MGN_MAX_FEAT + k] = tot;
    }
}

__global__ __launch_bounds__(NT) void feature_sums_partial_kernel(const float* __restrict__ feat, long long rows,
                                                                  int F, double* __restrict__ partial) {
    const long long per = (rows + gridDim.x - 1) / gridDim.x;
    const long long r0 = (long long)blockIdx.x * per, r1 = min(rows, r0 + per);
    double s[MGN_MAX_FEAT], q[MGN_MAX_FEAT];
#pragma unroll
    for (int k = 0; k < MGN_MAX_FEAT; ++k) { s[k] = 0.0; q[k] = 0.0; }
    for (long long r = r0 + threadIdx.x; r < r1; r += NT) {
#pragma unroll
        for (int k = 0; k < MGN_MAX_FEAT; ++k)
            if (k < F) { const double v = feat[r * F + k]; s[k] += v; q[k] += v * v; }
    }
    block_partial_sums(s, q, F, partial);
}

// Feature rows and their per-block column sums in one pass: the rows of feature_sums_partial_kernel's block, visited in
// its order, summed from the values just written -- the partials are those of a separate mgn_feature_sums pass bit for bit.
__global__ __launch_bounds__(NT) void node_features_sums_kernel(const float* __restrict__ world_pos, const float* __restrict__ mesh_pos,
                                                                const float* __restrict__ phi, const float* __restrict__ swell,
                                                                int n_swell, const float* __restrict__ node_type,
                                                                const float* __restrict__ mat_param, float* __restrict__ feat,
                                                                long long rows, double* __restrict__ partial) {
    const int F = 3 + n_swell + 5 + (mat_param ? 2 : 0);
    const long long per = (rows + gridDim.x - 1) / gridDim.x;
    const long long r0 = (long long)blockIdx.x * per, r1 = min(rows, r0 + per);
    double s[MGN_MAX_FEAT], q[MGN_MAX_FEAT];
#pragma unroll
    for (int k = 0; k < MGN_MAX_FEAT; ++k) { s[k] = 0.0; q[k] = 0.0; }
    for (long long r = r0 + threadIdx.x; r < r1; r += NT) {
        float f[MGN_MAX_FEAT];
        node_feature_row(world_pos, mesh_pos, phi, swell, n_swell, node_type, mat_param, (int)r, f);
#pragma unroll
        for (int k = 0; k < MGN_MAX_FEAT; ++k)
            if (k < F) { feat[r * F + k] = f[k]; const double v = f[k]; s[k] += v; q[k] += v * v; }
    }
    block_partial_sums(s, q, F, partial);
}

__global__ __launch_bounds__(NT) void edge_features_sums_kernel(const float* __restrict__ world_pos, const float* __restrict__ mesh_pos,
                                                                const float* __restrict__ phi, const int* __restrict__ sender,
                                                                const int* __restrict__ recv, float* __restrict__ feat,
                                                                long long rows, double* __restrict__ partial) {
    constexpr int F = 7;
    const long long per = (rows + gridDim.x - 1) / gridDim.x;
    const long long r0 = (long long)blockIdx.x * per, r1 = min(rows, r0 + per);
    double s[MGN_MAX_FEAT], q[MGN_MAX_FEAT];
#pragma unroll
    for (int k = 0; k < MGN_MAX_FEAT; ++k) { s[k] = 0.0; q[k] = 0.0; }
    for (long long r = r0 + threadIdx.x; r < r1; r += NT) {
        float f[F];
        edge_feature_row(world_pos, mesh_pos, phi, sender[r], recv[r], f);
#pragma unroll
        for (int k = 0; k < F; ++k) { feat[r * F + k] = f[k]; const double v = f[k]; s[k] += v; q[k] += v * v; }
    }
    block_partial_sums(s, q, F, partial);
}

// 256 threads: thread (c = t / 32, col = t % 32) adds the partials of blocks c, c+8, c+16, ... of
// padded column col; the eight chunk sums are then added in chunk order (fixed order: deterministic).
__global__ __launch_bounds__(NT) void feature_sums_final_kernel(const double* __restrict__ partial, int nblocks, int F,
                                                                double* __restrict__ sums) {
    __shared__ double sh[8][2 * MGN_MAX_FEAT];
    const int col = threadIdx.x & (2 * MGN_MAX_FEAT - 1), c = threadIdx.x / (2 * MGN_MAX_FEAT);
    double acc = 0.0;
    int b = c;
    for (; b + 56 < nblocks; b += 64) {              // eight loads in flight, added in the order of the plain loop
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(b + 8 * u) * 2 * MGN_MAX_FEAT + col];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; b < nblocks; b += 8) acc += partial[(size_t)b * 2 * MGN_MAX_FEAT + col];
    sh[c][col] = acc;
    __syncthreads();
    if (threadIdx.x < 2 * F) {
        const int k = threadIdx.x, pc = (k < F) ? k : MGN_MAX_FEAT + (k - F);
        double tot = 0.0;
        for (int q = 0; q < 8; ++q) tot += sh[q][pc];
        sums[k] = tot;
    }
}

// feature_sums_final_kernel and the statistics update in one launch: the block partials are added in that kernel's order
// (eight chunk sums, then the chunks), the totals stay in LDS, and threads k < F fold them into the accumulators and
// write [mean | std] exactly as norm_update_kernel / norm_update_counted_kernel do (count: device word or NULL).
__global__ __launch_bounds__(NT) void norm_fold_kernel(const double* __restrict__ partial, int nblocks, int F, float* acc_sum,
                                                       float* acc_sumsq, long long* count, long long host_count,
                                                       long long add, float eps, float* mean_std) {
    __shared__ double sh[8][2 * MGN_MAX_FEAT];
    __shared__ double fin[2 * MGN_MAX_FEAT];
    const int col = threadIdx.x & (2 * MGN_MAX_FEAT - 1), c = threadIdx.x / (2 * MGN_MAX_FEAT);
    double acc = 0.0;
    int b = c;
    for (; b + 56 < nblocks; b += 64) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(b + 8 * u) * 2 * MGN_MAX_FEAT + col];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; b < nblocks; b += 8) acc += partial[(size_t)b * 2 * MGN_MAX_FEAT + col];
    sh[c][col] = acc;
    __syncthreads();
    if (threadIdx.x < 2 * F) {
        const int k = threadIdx.x, pc = (k < F) ? k : MGN_MAX_FEAT + (k - F);
        double tot = 0.0;
        for (int q = 0; q < 8; ++q) tot += sh[q][pc];
        fin[k] = tot;
    }
    const long long cnt = count ? *count + add : host_count;     // host_count already includes the rows of this call
    __syncthreads();
    const int k = threadIdx.x;
    if (k < F) {
        const float safe_count = (float)(cnt > 1 ? cnt : 1);
        float s = acc_sum[k], q = acc_sumsq[k];
        s += (float)fin[k]; q += (float)fin[F + k];
        acc_sum[k] = s; acc_sumsq[k] = q;
        const float mean = s / safe_count;
        const float var = fmaxf(q / safe_count - mean * mean, 0.f);
        mean_std[k] = mean;
        mean_std[F + k] = sqrtf(var + eps);
    }
    if (k == 0 && count && add) *count = cnt;
}

__global__ void norm_update_kernel(float* acc_sum, float* acc_sumsq, const double* sums, int F, float safe_count,
                                   float eps, float* mean_std) {
    const int k = threadIdx.x;
    if (k >= F) return;
    float s = acc_sum[k], q = acc_sumsq[k];
    if (sums) {
        s += (float)sums[k]; q += (float)sums[F + k];
        acc_sum[k] = s; acc_sumsq[k] = q;
    }
    const float mean = s / safe_count;
    const float var = fmaxf(q / safe_count - mean * mean, 0.f);
    mean_std[k] = mean;
    mean_std[F + k] = sqrtf(var + eps);
}

// The same update with the row count kept in device memory (count += add), for callers that replay
// a captured launch sequence and so cannot pass a new scalar per step.
__global__ void norm_update_counted_kernel(float* acc_sum, float* acc_sumsq, const double* sums, int F, long long* count,
                                           long long add, float eps, float* mean_std) {
    const int k = threadIdx.x;
    const long long c = *count + add;
    const float safe_count = (float)(c > 1 ? c : 1);
    if (k < F) {
        float s = acc_sum[k], q = acc_sumsq[k];
        if (sums) {
            s += (float)sums[k]; q += (float)sums[F + k];
            acc_sum[k] = s; acc_sumsq[k] = q;
        }
        const float mean = s / safe_count;
        const float var = fmaxf(q / safe_count - mean * mean, 0.f);
        mean_std[k] = mean;
        mean_std[F + k] = sqrtf(var + eps);
    }
    __syncthreads();                       // every lane has read the old count
    if (k == 0 && add) *count = c;
}

__global__ void normalize_kernel(const float* __restrict__ x, const float* __restrict__ mean_std,
                                 float* __restrict__ out, long long total, int F, bool inverse) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % F);
        const float m = mean_std[k], s = mean_std[F + k];
        out[i] = inverse ? s * x[i] + m : (x[i] - m) / s;
    }
}

// Training-noise injection of one frame (core/datasetclass.py:75-89): z * level * scale added to
// the state, except on Dirichlet nodes.  Unfused multiplies and adds, in the reference's order.
__global__ void inject_noise_kernel(const float* __restrict__ world, const float* __restrict__ phi,
                                    const float* __restrict__ node_type, const float* __restrict__ zw,
                                    const float* __restrict__ zp, float level, float conn, float phi_range,
                                    float* __restrict__ out_world, float* __restrict__ out_phi, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float* nt = node_type + (size_t)i * 5;
        const float nx = nt[1] == 1.f ? 0.f : __fmul_rn(__fmul_rn(zw[2 * i], level), conn);
        const float ny = nt[2] == 1.f ? 0.f : __fmul_rn(__fmul_rn(zw[2 * i + 1], level), conn);
        const float np = nt[3] == 1.f ? 0.f : __fmul_rn(__fmul_rn(zp[i], level), phi_range);
        out_world[2 * i] = __fadd_rn(world[2 * i], nx);
        out_world[2 * i + 1] = __fadd_rn(world[2 * i + 1], ny);
        out_phi[i] = __fadd_rn(phi[i], np);
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const long long* __restrict__ idx,
                                   float* __restrict__ dst, long long rows, int w4, bool scatter) {
    const long long total = rows * w4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / w4; const int c = (int)(i - r * w4);
        const long long o = idx[r];
        if (scatter) ((f32x4*)dst)[o * w4 + c] = ((const f32x4*)src)[i];
        else ((f32x4*)dst)[i] = ((const f32x4*)src)[o * w4 + c];
    }
}

// Matrix path of the processor kernels (environment variable MGN_MATMUL):
//   f16x2  (default) edge and node kernels: two-term fp16 split, three partial products
//   bf16x3           edge kernel: three-term bf16 split, six partial products; node kernel: fp32 MFMA
//   f32              fp32 MFMA everywhere
// All meet the same fp32 tolerance; the fp16 split additionally needs |activations| < 65504.
int matmul_mode() { return mgn_matmul_mode(); }

bool mlp_ok(const mgn_mlp* m, bool need_ln) {
    return m && m->w0 && m->b0 && m->w1 && m->b1 && (!need_ln || (m->ln_w && m->ln_b));
}

}  // namespace

#ifdef MGN_PHASE_STAMPS
extern "C" int mgn_debug_read_stamps(unsigned long long* out32, int reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

extern "C" int mgn_mp_project(const float* x, const float* edge_w0, const float* edge_b0, float* ps, float* pr,
                              int64_t n_nodes, void* stream) {
    MGN_REQUIRE(n_nodes >= 0 && n_nodes < (1ll << 31), "mgn_mp_project");
    if (n_nodes == 0) return 0;
    MGN_REQUIRE(x && edge_w0 && edge_b0 && ps && pr, "mgn_mp_project");
    ProjArgs a{x, edge_w0, edge_b0, ps, pr, (int)n_nodes};
    const int g = grid_for((n_nodes + TM - 1) / TM, 2);
    if (matmul_mode() == 2) hipLaunchKernelGGL((mp_project_kernel<true>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((mp_project_kernel<false>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);
    return check_launch("mgn_mp_project");
}

static int launch_edge(const float* e_in, const float* ps, const float* pr, const int32_t* rowptr,
                       const int32_t* csr_sender, const int32_t* csr_recv, const mgn_mlp* m, const float* edge_weight,
                       float* e_out, float* aggr, int64_t n_edges, int64_t n_nodes, void* stream, const char* name,
                       const void* pk0 = nullptr, const void* pk1 = nullptr) {
    MGN_REQUIRE(n_edges >= 0 && n_edges < (1ll << 31) && n_nodes >= 0 && n_nodes < (1ll << 31), name);
    if (n_edges == 0) return 0;
    MGN_REQUIRE(e_in && ps && pr && rowptr && csr_sender && csr_recv && aggr && mlp_ok(m, true), name);
    // node rows are addressed with 32-bit element offsets
    MGN_REQUIRE(n_nodes * (int64_t)H < (1ll << 31), name);
    EdgeArgs a{e_in, ps, pr, rowptr, csr_sender, csr_recv, m->w0, m->w1, m->b1, m->ln_w, m->ln_b,
               e_out, aggr, (int)n_edges, (int)n_nodes, edge_weight, nullptr, nullptr};
    const hipStream_t st = (hipStream_t)stream;
    if (matmul_mode() == 2) {
        a.pk0 = pk0; a.pk1 = pk1;                    // the other matrix paths read the fp32 matrices
        const int g = grid_for((n_edges + TM - 1) / TM, 2);
        if (edge_weight) hipLaunchKernelGGL((mp_edge_h4_kernel<true>), dim3(g), dim3(NT), 0, st, a);
        else hipLaunchKernelGGL((mp_edge_h4_kernel<false>), dim3(g), dim3(NT), 0, st, a);
    } else if (matmul_mode() == 1) {
        const int g = grid_for((n_edges + TM - 1) / TM, 1);
        if (edge_weight) hipLaunchKernelGGL((mp_edge3_kernel<true>), dim3(g), dim3(NT3), 0, st, a);
        else hipLaunchKernelGGL((mp_edge3_kernel<false>), dim3(g), dim3(NT3), 0, st, a);
    } else {
        const int g = grid_for((n_edges + TM - 1) / TM, 2);
        if (edge_weight) hipLaunchKernelGGL((mp_edge_kernel<true>), dim3(g), dim3(NT), 0, st, a);
        else hipLaunchKernelGGL((mp_edge_kernel<false>), dim3(g), dim3(NT), 0, st, a);
    }
    return check_launch(name);
}

extern "C" int mgn_mp_edge(const float* e_in, const float* ps, const float* pr, const int32_t* rowptr,
                           const int32_t* csr_sender, const int32_t* csr_recv, const mgn_mlp* m, float* e_out,
                           float* aggr, int64_t n_edges, int64_t n_nodes, void* stream) {
    return launch_edge(e_in, ps, pr, rowptr, csr_sender, csr_recv, m, nullptr, e_out, aggr, n_edges, n_nodes, stream,
                       "mgn_mp_edge");
}

extern "C" int mgn_mp_edge_weighted(const float* e_in, const float* ps, const float* pr, const int32_t* rowptr,
                                    const int32_t* csr_sender, const int32_t* csr_recv, const mgn_mlp* m,
                                    const float* edge_weight, float* e_out, float* aggr, int64_t n_edges,
                                    int64_t n_nodes, void* stream) {
    MGN_REQUIRE(edge_weight || n_edges == 0, "mgn_mp_edge_weighted");
    return launch_edge(e_in, ps, pr, rowptr, csr_sender, csr_recv, m, edge_weight, e_out, aggr, n_edges, n_nodes, stream,
                       "mgn_mp_edge_weighted");
}

extern "C" int mgn_mp_edge_packed(const float* e_in, const float* ps, const float* pr, const int32_t* rowptr,
                                  const int32_t* csr_sender, const int32_t* csr_recv, const mgn_mlp* m,
                                  const void* packed_w0e, const void* packed_w1, const float* edge_weight, float* e_out,
                                  float* aggr, int64_t n_edges, int64_t n_nodes, void* stream) {
    MGN_REQUIRE(packed_w0e && packed_w1, "mgn_mp_edge_packed");
    return launch_edge(e_in, ps, pr, rowptr, csr_sender, csr_recv, m, edge_weight, e_out, aggr, n_edges, n_nodes, stream,
                       "mgn_mp_edge_packed", packed_w0e, packed_w1);
}

extern "C" int mgn_edge_attention(const float* a_proj, const int32_t* rowptr, const int32_t* csr_sender,
                                  const int32_t* csr_recv, float* alpha, float* edge_weight, int64_t n_edges,
                                  int64_t n_nodes, void* stream) {
    MGN_REQUIRE(n_edges >= 0 && n_edges < (1ll << 31) && n_nodes >= 0 && n_nodes < (1ll << 31), "mgn_edge_attention");
    if (n_edges == 0 || n_nodes == 0) return 0;
    MGN_REQUIRE(a_proj && rowptr && csr_sender && csr_recv && alpha && edge_weight, "mgn_edge_attention");
    const int g1 = (int)std::min<int64_t>((n_edges * 32 + NT - 1) / NT, 16384);
    hipLaunchKernelGGL(edge_alpha_kernel, dim3(g1), dim3(NT), 0, (hipStream_t)stream, a_proj, csr_sender, csr_recv, alpha,
                       (int)n_edges);
    const int g2 = (int)std::min<int64_t>((n_nodes + NT - 1) / NT, 16384);
    hipLaunchKernelGGL(edge_alpha_normalize_kernel, dim3(g2), dim3(NT), 0, (hipStream_t)stream, rowptr, alpha, edge_weight,
                       (int)n_nodes);
    return check_launch("mgn_edge_attention");
}

static int launch_node(const float* x, const float* aggr, const int32_t* rowptr, const mgn_mlp* m,
                       const float* next_w0, const float* next_b0, float* x_out, float* ps, float* pr,
                       int64_t n_nodes, void* stream, const char* name, const void* pk0 = nullptr,
                       const void* pk1 = nullptr, const void* pks = nullptr, const void* pkr = nullptr) {
    MGN_REQUIRE(n_nodes >= 0 && n_nodes < (1ll << 31), name);
    if (n_nodes == 0) return 0;
    MGN_REQUIRE(x && aggr && rowptr && x_out && mlp_ok(m, true), name);
    NodeArgs a{x, aggr, rowptr, m->w0, m->b0, m->w1, m->b1, m->ln_w, m->ln_b, next_w0, next_b0, x_out, ps, pr,
               (int)n_nodes, nullptr, nullptr, nullptr, nullptr};
    const int g = grid_for((n_nodes + TM - 1) / TM, 1);
    if (matmul_mode() == 2) {
        a.pk0 = pk0; a.pk1 = pk1; a.pks = pks; a.pkr = pkr;      // the other matrix paths read the fp32 matrices
        if (pk0) {
            if (next_w0) hipLaunchKernelGGL((mp_node_h_kernel<true, true>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((mp_node_h_kernel<false, true>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);
        } else {
            if (next_w0) hipLaunchKernelGGL((mp_node_h_kernel<true, false>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((mp_node_h_kernel<false, false>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);
        }
    } else {
        if (next_w0) hipLaunchKernelGGL((mp_node_kernel<true>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((mp_node_kernel<false>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);
    }
    return check_launch(name);
}

extern "C" int mgn_mp_node(const float* x, const float* aggr, const int32_t* rowptr, const mgn_mlp* m, float* x_out,
                           int64_t n_nodes, void* stream) {
    return launch_node(x, aggr, rowptr, m, nullptr, nullptr, x_out, nullptr, nullptr, n_nodes, stream, "mgn_mp_node");
}

extern "C" int mgn_mp_node_project(const float* x, const float* aggr, const int32_t* rowptr, const mgn_mlp* m,
                                   const float* next_edge_w0, const float* next_edge_b0, float* x_out, float* ps,
                                   float* pr, int64_t n_nodes, void* stream) {
    MGN_REQUIRE(next_edge_w0 && next_edge_b0 && ps && pr, "mgn_mp_node_project");
    return launch_node(x, aggr, rowptr, m, next_edge_w0, next_edge_b0, x_out, ps, pr, n_nodes, stream,
                       "mgn_mp_node_project");
}

extern "C" int mgn_mp_node_packed(const float* x, const float* aggr, const int32_t* rowptr, const mgn_mlp* m,
                                  const void* packed_w0, const void* packed_w1, const float* next_edge_w0,
                                  const float* next_edge_b0, const void* packed_next_ws, const void* packed_next_wr,
                                  float* x_out, float* ps, float* pr, int64_t n_nodes, void* stream) {
    MGN_REQUIRE(packed_w0 && packed_w1, "mgn_mp_node_packed");
    if (next_edge_w0) MGN_REQUIRE(next_edge_b0 && ps && pr && packed_next_ws && packed_next_wr, "mgn_mp_node_packed");
    return launch_node(x, aggr, rowptr, m, next_edge_w0, next_edge_w0 ? next_edge_b0 : nullptr, x_out, ps, pr, n_nodes, stream,
                       "mgn_mp_node_packed", packed_w0, packed_w1, packed_next_ws, packed_next_wr);
}

extern "C" int64_t mgn_packed_weight_bytes(int32_t k) {
    return (k > 0 && k % 16 == 0) ? (int64_t)H * k * (int64_t)sizeof(float) : 0;
}

extern "C" int mgn_pack_weights(const mgn_weight_slice* slices, int32_t count, void* packed, void* stream) {
    MGN_REQUIRE(count >= 0 && count <= MGN_MAX_PACK_SLICES && (count == 0 || (slices && packed)), "mgn_pack_weights");
    if (count == 0) return 0;
    PackTable tab;
    unsigned off = 0;
    for (int i = 0; i < count; ++i) {
        const mgn_weight_slice& s = slices[i];
        MGN_REQUIRE(s.w && s.k > 0 && s.k % 16 == 0 && s.k <= 1024 && s.col0 >= 0 && s.col0 % 4 == 0 && s.ld >= s.col0 + s.k &&
                    s.ld % 4 == 0, "mgn_pack_weights");
        tab.s[i] = PackSlice{s.w, s.ld, s.col0, s.k / 16, off};
        off += (unsigned)(H * s.k / 4);              // u32x4 per set: 128 rows x k values, 4 bytes (two fp16) each
    }
    hipLaunchKernelGGL(pack_weights_kernel, dim3(count), dim3(NT), 0, (hipStream_t)stream, tab, (u32x4*)packed);
    return check_launch("mgn_pack_weights");
}

extern "C" int mgn_node_features(const float* world_pos, const float* mesh_pos, const float* phi,
                                 const float* swelling_phi, int32_t n_swell, const float* node_type,
                                 const float* mat_param, float* feat, int64_t n_nodes, void* stream) {
    MGN_REQUIRE(n_nodes >= 0 && n_nodes < (1ll << 31) && n_swell >= 0, "mgn_node_features");
    MGN_REQUIRE(3 + n_swell + 5 + (mat_param ? 2 : 0) <= MGN_MAX_FEAT, "mgn_node_features");
    if (n_nodes == 0) return 0;
    MGN_REQUIRE(world_pos && mesh_pos && phi && node_type && feat && (n_swell == 0 || swelling_phi), "mgn_node_features");
    const int g = (int)std::min<int64_t>((n_nodes + NT - 1) / NT, 4096);
    hipLaunchKernelGGL(node_features_kernel, dim3(g), dim3(NT), 0, (hipStream_t)stream, world_pos, mesh_pos, phi,
                       swelling_phi, n_swell, node_type, mat_param, feat, (int)n_nodes);
    return check_launch("mgn_node_features");
}

extern "C" int mgn_edge_features(const float* world_pos, const float* mesh_pos, const float* phi,
                                 const int32_t* csr_sender, const int32_t* csr_recv, float* feat, int64_t n_edges,
                                 void* stream) {
    MGN_REQUIRE(n_edges >= 0 && n_edges < (1ll << 31), "mgn_edge_features");
    if (n_edges == 0) return 0;
    MGN_REQUIRE(world_pos && mesh_pos && phi && csr_sender && csr_recv && feat, "mgn_edge_features");
    const int g = (int)std::min<int64_t>((n_edges + NT - 1) / NT, 8192);
    hipLaunchKernelGGL(edge_features_kernel, dim3(g), dim3(NT), 0, (hipStream_t)stream, world_pos, mesh_pos, phi,
                       csr_sender, csr_recv, feat, (int)n_edges);
    return check_launch("mgn_edge_features");
}

static int sum_blocks(int64_t rows) {
    const int64_t b = (rows + 4 * NT - 1) / (4 * NT);
    return (int)std::max<int64_t>(1, std::min<int64_t>(b, SUM_BLOCKS_MAX));
}

extern "C" int mgn_node_features_sums(const float* world_pos, const float* mesh_pos, const float* phi,
                                      const float* swelling_phi, int32_t n_swell, const float* node_type,
                                      const float* mat_param, float* feat, int64_t n_nodes, void* scratch, void* stream) {
    MGN_REQUIRE(n_nodes > 0 && n_nodes < (1ll << 31) && n_swell >= 0 && scratch, "mgn_node_features_sums");
    MGN_REQUIRE(3 + n_swell + 5 + (mat_param ? 2 : 0) <= MGN_MAX_FEAT, "mgn_node_features_sums");
    MGN_REQUIRE(world_pos && mesh_pos && phi && node_type && feat && (n_swell == 0 || swelling_phi), "mgn_node_features_sums");
    hipLaunchKernelGGL(node_features_sums_kernel, dim3(sum_blocks(n_nodes)), dim3(NT), 0, (hipStream_t)stream, world_pos,
                       mesh_pos, phi, swelling_phi, n_swell, node_type, mat_param, feat, (long long)n_nodes, (double*)scratch);
    return check_launch("mgn_node_features_sums");
}

extern "C" int mgn_edge_features_sums(const float* world_pos, const float* mesh_pos, const float* phi,
                                      const int32_t* csr_sender, const int32_t* csr_recv, float* feat, int64_t n_edges,
                                      void* scratch, void* stream) {
    MGN_REQUIRE(n_edges > 0 && n_edges < (1ll << 31) && scratch, "mgn_edge_features_sums");
    MGN_REQUIRE(world_pos && mesh_pos && phi && csr_sender && csr_recv && feat, "mgn_edge_features_sums");
    hipLaunchKernelGGL(edge_features_sums_kernel, dim3(sum_blocks(n_edges)), dim3(NT), 0, (hipStream_t)stream, world_pos,
                       mesh_pos, phi, csr_sender, csr_recv, feat, (long long)n_edges, (double*)scratch);
    return check_launch("mgn_edge_features_sums");
}

extern "C" int mgn_norm_fold(const void* scratch, int64_t rows, int32_t n_feat, float* acc_sum, float* acc_sumsq,
                             int64_t* count_dev, int64_t count, float eps, float* mean_std, void* stream) {
    MGN_REQUIRE(scratch && rows > 0 && n_feat > 0 && n_feat <= MGN_MAX_FEAT && acc_sum && acc_sumsq && mean_std && count >= 0,
                "mgn_norm_fold");
    hipLaunchKernelGGL(norm_fold_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, (const double*)scratch, sum_blocks(rows),
                       (int)n_feat, acc_sum, acc_sumsq, (long long*)count_dev, (long long)count, (long long)rows, eps, mean_std);
    return check_launch("mgn_norm_fold");
}

extern "C" int64_t mgn_feature_sums_scratch_bytes(int64_t rows, int32_t n_feat) {
    (void)n_feat;
    return (int64_t)sum_blocks(rows < 0 ? 0 : rows) * 2 * MGN_MAX_FEAT * (int64_t)sizeof(double);
}

extern "C" int mgn_feature_sums(const float* feat, int64_t rows, int32_t n_feat, double* sums, void* scratch,
                                void* stream) {
    MGN_REQUIRE(rows >= 0 && n_feat > 0 && n_feat <= MGN_MAX_FEAT && sums && scratch, "mgn_feature_sums");
    MGN_REQUIRE(rows == 0 || feat, "mgn_feature_sums");
    const int nb = sum_blocks(rows);
    hipLaunchKernelGGL(feature_sums_partial_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, feat,
                       (long long)rows, (int)n_feat, (double*)scratch);
    hipLaunchKernelGGL(feature_sums_final_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream,
                       (const double*)scratch, nb, (int)n_feat, sums);
    return check_launch("mgn_feature_sums");
}

// ---- the same sums with the per-block partials in a buffer the library owns: nothing is allocated
// and no workspace is passed, for callers that cannot allocate (HIP graph capture, serving loops).
// `scratch` is one array per GPU (every device that loads the code object holds its own copy at its
// own address); host code never uses the array's name as an address, only the per-device pointer that
// hipGetSymbolAddress returned with that device current.
static __device__ __align__(256) double scratch[SUM_BLOCKS_MAX * 2 * MGN_MAX_FEAT];    // 256 KiB

static std::mutex g_scratch_mu;
static double* g_scratch_ptr[64] = {nullptr};

static int scratch_of_current_device(double** out, const char* name) {
    int dev = -1;
    const hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) { mgn_set_error("%s: hipGetDevice: %s", name, hipGetErrorString(e0)); return (int)e0; }
    if (dev < 0 || dev >= 64) { mgn_set_error("%s: device ordinal %d outside [0, 64)", name, dev); return MGN_E_RANGE; }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            mgn_set_error("%s: hipGetSymbolAddress(scratch) on device %d: %s", name, dev, hipGetErrorString(e));
            return e != hipSuccess ? (int)e : MGN_E_ARG;
        }
        g_scratch_ptr[dev] = (double*)p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

extern "C" int mgn_feature_sums_static_prepare(void) {
    double* part = nullptr;
    return scratch_of_current_device(&part, "mgn_feature_sums_static_prepare");
}

extern "C" int mgn_feature_sums_static(const float* feat, int64_t rows, int32_t n_feat, double* sums, void* stream) {
    MGN_REQUIRE(rows >= 0 && n_feat > 0 && n_feat <= MGN_MAX_FEAT && sums, "mgn_feature_sums_static");
    if (rows > MGN_FEATURE_SUMS_STATIC_MAX_ROWS) {
        mgn_set_error("mgn_feature_sums_static: %lld rows exceed the limit of %lld", (long long)rows,
                      (long long)MGN_FEATURE_SUMS_STATIC_MAX_ROWS);
        return MGN_E_CAP;
    }
    MGN_REQUIRE(rows == 0 || feat, "mgn_feature_sums_static");
    double* part = nullptr;
    const int rc = scratch_of_current_device(&part, "mgn_feature_sums_static");
    if (rc) return rc;
    const int nb = sum_blocks(rows);
    static_assert(sizeof(scratch) >= (size_t)SUM_BLOCKS_MAX * 2 * MGN_MAX_FEAT * sizeof(double), "scratch too small");
    // every block in [0, nb) writes its columns k < F of both halves and the final kernel uses only those: whatever an
    // earlier (larger or wider) call left in the array is never used, so nothing needs clearing between calls
    hipLaunchKernelGGL(feature_sums_partial_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, feat,
                       (long long)rows, (int)n_feat, part);
    hipLaunchKernelGGL(feature_sums_final_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream,
                       (const double*)part, nb, (int)n_feat, sums);
    return check_launch("mgn_feature_sums_static");
}

extern "C" int mgn_range_check(int32_t* tripped, int32_t reset, void* stream) {
    MGN_REQUIRE(tripped, "mgn_range_check");
    int t = 0;
    int rc = read_range_flag(&t, reset, (hipStream_t)stream, "mgn_range_check");
    if (rc == 0) rc = mgn_train_read_range_flag(&t, reset, stream);
    if (rc == 0) *tripped = t;
    return rc;
}

extern "C" int mgn_range_poll(int32_t* host_words, void* stream) {
    MGN_REQUIRE(host_words, "mgn_range_poll");
    int rc = poll_range_flag(host_words, (hipStream_t)stream, "mgn_range_poll");
    if (rc == 0) rc = mgn_train_poll_range_flag(host_words + 1, stream);
    return rc;
}

extern "C" int mgn_norm_update(float* acc_sum, float* acc_sumsq, const double* sums, int32_t n_feat, int64_t count,
                               float eps, float* mean_std, void* stream) {
    MGN_REQUIRE(acc_sum && acc_sumsq && mean_std && n_feat > 0 && n_feat <= 64 && count >= 0, "mgn_norm_update");
    const float safe = (float)(count > 1 ? count : 1);
    hipLaunchKernelGGL(norm_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc_sum, acc_sumsq, sums,
                       (int)n_feat, safe, eps, mean_std);
    return check_launch("mgn_norm_update");
}

extern "C" int mgn_norm_update_counted(float* acc_sum, float* acc_sumsq, const double* sums, int32_t n_feat,
                                       int64_t* count_dev, int64_t add_rows, float eps, float* mean_std, void* stream) {
    MGN_REQUIRE(acc_sum && acc_sumsq && mean_std && count_dev && n_feat > 0 && n_feat <= 64 && add_rows >= 0,
                "mgn_norm_update_counted");
    MGN_REQUIRE(sums || add_rows == 0, "mgn_norm_update_counted");
    hipLaunchKernelGGL(norm_update_counted_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc_sum, acc_sumsq, sums,
                       (int)n_feat, (long long*)count_dev, (long long)add_rows, eps, mean_std);
    return check_launch("mgn_norm_update_counted");
}

extern "C" int mgn_encode(const float* feat, int32_t n_feat, const float* mean_std, const mgn_mlp* enc, float* out,
                          int64_t rows, void* stream) {
    MGN_REQUIRE(rows >= 0 && rows < (1ll << 31) && n_feat > 0 && n_feat <= MGN_MAX_FEAT, "mgn_encode");
    if (rows == 0) return 0;
    MGN_REQUIRE(feat && out && mlp_ok(enc, true), "mgn_encode");
    EncArgs a{feat, mean_std, enc->w0, enc->b0, enc->w1, enc->b1, enc->ln_w, enc->ln_b, out, (int)rows, (int)n_feat};
    const int g = grid_for((rows + TM - 1) / TM, 2);
    const bool half = matmul_mode() >= 1;            // bf16 split in both split modes (see encode_kernel)
    const int f4 = (n_feat + 3) / 4;
#define MGN_LAUNCH_ENC(F4)                                                                                          \
    do {                                                                                                            \
        if (half) hipLaunchKernelGGL((encode_kernel<F4, true>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);      \
        else hipLaunchKernelGGL((encode_kernel<F4, false>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);          \
    } while (0)
    if (f4 == 1) MGN_LAUNCH_ENC(1); else if (f4 == 2) MGN_LAUNCH_ENC(2); else if (f4 == 3) MGN_LAUNCH_ENC(3); else MGN_LAUNCH_ENC(4);
#undef MGN_LAUNCH_ENC
    return check_launch("mgn_encode");
}

extern "C" int mgn_decode(const float* x, const mgn_mlp* dec, int32_t time_dim, int32_t out_dim, float* delta_norm,
                          const float* out_mean_std, const float* world_pos, const float* phi,
                          const float* node_type, float* pred, int64_t n_nodes, void* stream) {
    MGN_REQUIRE(n_nodes >= 0 && n_nodes < (1ll << 31) && time_dim > 0 && out_dim > 0 &&
                time_dim * out_dim <= MGN_MAX_DEC_OUT, "mgn_decode");
    if (n_nodes == 0) return 0;
    MGN_REQUIRE(x && mlp_ok(dec, false) && (delta_norm || pred), "mgn_decode");
    MGN_REQUIRE(!pred || (out_dim == 3 && out_mean_std && world_pos && phi && node_type), "mgn_decode");
    DecArgs a{x, dec->w0, dec->b0, dec->w1, dec->b1, delta_norm, out_mean_std, world_pos, phi, node_type,
              pred, (int)n_nodes, (int)time_dim, (int)out_dim, nullptr, nullptr};
    const int g = grid_for((n_nodes + TM - 1) / TM, 2);
    if (matmul_mode() == 2) hipLaunchKernelGGL((decode_kernel<true>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((decode_kernel<false>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);
    return check_launch("mgn_decode");
}

extern "C" int mgn_decode_step(const float* x, const mgn_mlp* dec, int32_t time_dim, const float* out_mean_std,
                               const float* world_pos, const float* phi, const float* node_type, float* pred,
                               float* next_world, float* next_phi, int64_t n_nodes, void* stream) {
    MGN_REQUIRE(n_nodes >= 0 && n_nodes < (1ll << 31) && time_dim > 0 && time_dim * 3 <= MGN_MAX_DEC_OUT, "mgn_decode_step");
    if (n_nodes == 0) return 0;
    MGN_REQUIRE(x && mlp_ok(dec, false) && pred && out_mean_std && world_pos && phi && node_type && next_world && next_phi,
                "mgn_decode_step");
    DecArgs a{x, dec->w0, dec->b0, dec->w1, dec->b1, nullptr, out_mean_std, world_pos, phi, node_type,
              pred, (int)n_nodes, (int)time_dim, 3, next_world, next_phi};
    const int g = grid_for((n_nodes + TM - 1) / TM, 2);
    if (matmul_mode() == 2) hipLaunchKernelGGL((decode_kernel<true>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((decode_kernel<false>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);
    return check_launch("mgn_decode_step");
}

extern "C" int mgn_decode_cat(const float* x, const float* x2, const mgn_mlp* dec, int32_t time_dim, int32_t out_dim,
                              float* hidden_ws, float* delta_norm, const float* out_mean_std, const float* world_pos,
                              const float* phi, const float* node_type, float* pred, int64_t n_nodes, void* stream) {
    MGN_REQUIRE(n_nodes >= 0 && n_nodes < (1ll << 31) && time_dim > 0 && out_dim > 0 &&
                time_dim * out_dim <= MGN_MAX_DEC_OUT, "mgn_decode_cat");
    if (n_nodes == 0) return 0;
    MGN_REQUIRE(x && x2 && hidden_ws && mlp_ok(dec, false) && (delta_norm || pred), "mgn_decode_cat");
    MGN_REQUIRE(!pred || (out_dim == 3 && out_mean_std && world_pos && phi && node_type), "mgn_decode_cat");
    // first Linear over cat[x, x2] with its ReLU (the two-block Linear of the training kernels), then the decoder
    // kernel entered at its hidden tile
    const int rc = mgn_train_linear_fwd(x, x2, dec->w0, 2 * H, 0, H, dec->b0, nullptr, nullptr, nullptr, nullptr, 1,
                                        hidden_ws, n_nodes, stream);
    if (rc) return rc;
    DecArgs a{hidden_ws, nullptr, nullptr, dec->w1, dec->b1, delta_norm, out_mean_std, world_pos, phi, node_type,
              pred, (int)n_nodes, (int)time_dim, (int)out_dim, nullptr, nullptr};
    const int g = grid_for((n_nodes + TM - 1) / TM, 2);
    if (matmul_mode() == 2) hipLaunchKernelGGL((decode_kernel<true>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((decode_kernel<false>), dim3(g), dim3(NT), 0, (hipStream_t)stream, a);
    return check_launch("mgn_decode_cat");
}

extern "C" int mgn_normalize(const float* x, const float* mean_std, float* out, int64_t rows, int32_t n_feat,
                             int32_t inverse, void* stream) {
    MGN_REQUIRE(rows >= 0 && n_feat > 0, "mgn_normalize");
    if (rows == 0) return 0;
    MGN_REQUIRE(x && mean_std && out, "mgn_normalize");
    const int64_t total = rows * n_feat;
    const int g = (int)std::min<int64_t>((total + NT - 1) / NT, 16384);
    hipLaunchKernelGGL(normalize_kernel, dim3(g), dim3(NT), 0, (hipStream_t)stream, x, mean_std, out,
                       (long long)total, (int)n_feat, inverse != 0);
    return check_launch("mgn_normalize");
}

extern "C" int mgn_inject_noise(const float* world_pos, const float* phi, const float* node_type, const float* z_world,
                                const float* z_phi, float level, float conn_length, float phi_range, float* out_world,
                                float* out_phi, int64_t n_nodes, void* stream) {
    MGN_REQUIRE(n_nodes >= 0 && n_nodes < (int64_t)1 << 30, "mgn_inject_noise");
    if (n_nodes == 0) return 0;
    MGN_REQUIRE(world_pos && phi && node_type && z_world && z_phi && out_world && out_phi, "mgn_inject_noise");
    const int g = (int)std::min<int64_t>((n_nodes + NT - 1) / NT, 16384);
    hipLaunchKernelGGL(inject_noise_kernel, dim3(g), dim3(NT), 0, (hipStream_t)stream, world_pos, phi, node_type, z_world,
                       z_phi, level, conn_length, phi_range, out_world, out_phi, (int)n_nodes);
    return check_launch("mgn_inject_noise");
}

static int rows_copy(const float* src, const int64_t* idx, float* dst, int64_t rows, int32_t width, void* stream,
                     bool scatter, const char* name) {
    MGN_REQUIRE(rows >= 0 && width > 0 && width % 4 == 0, name);
    if (rows == 0) return 0;
    MGN_REQUIRE(src && idx && dst, name);
    const int64_t total = rows * (width / 4);
    const int g = (int)std::min<int64_t>((total + NT - 1) / NT, 16384);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(g), dim3(NT), 0, (hipStream_t)stream, src, (const long long*)idx, dst,
                       (long long)rows, (int)(width / 4), scatter);
    return check_launch(name);
}

extern "C" int mgn_gather_rows(const float* src, const int64_t* idx, float* dst, int64_t rows, int32_t width,
                               void* stream) {
    return rows_copy(src, idx, dst, rows, width, stream, false, "mgn_gather_rows");
}

extern "C" int mgn_scatter_rows(const float* src, const int64_t* idx, float* dst, int64_t rows, int32_t width,
                                void* stream) {
    return rows_copy(src, idx, dst, rows, width, stream, true, "mgn_scatter_rows");
}
