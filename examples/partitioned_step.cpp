// A host without Python: two message-passing blocks (core/meshgraphnet.py:86-120) on a triangle mesh cut into node
// ranges, driven through the C ABI of libmgn_hip.so alone -- edge build and receiver CSR on the host, projections, halo
// pack, edge update on the interior receivers while the halo rows are copied, boundary receivers, node update -- and
// checked against the same blocks on the whole mesh.  The "ranks" share one GPU here and their exchange is a
// device-to-device copy; with one process per GPU the copy becomes ncclSend / ncclRecv into the same place
// (ps + num_owned * 128 of the receiving rank), nothing else changes.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include examples/partitioned_step.cpp -L gnn_amd -lmgn_hip \
//         -Wl,-rpath,'$ORIGIN/../gnn_amd' -o examples/partitioned_step
//   examples/partitioned_step [nx ny ranks]          exit code 0: partitioned == whole mesh (rel-L2 <= 1e-6)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "mgn_hip.h"

namespace {

constexpr int H = 128;

#define CHECK_HIP(expr)                                                                        \
    do {                                                                                       \
        const hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); std::exit(2); } \
    } while (0)
#define CHECK_MGN(expr)                                                                        \
    do {                                                                                       \
        const int rc_ = (expr);                                                                \
        if (rc_ != 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #expr, rc_, mgn_last_error()); std::exit(2); } \
    } while (0)

template <typename T>
T* to_device(const std::vector<T>& v) {
    T* p = nullptr;
    CHECK_HIP(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (!v.empty()) CHECK_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

float* device_floats(size_t n) {
    float* p = nullptr;
    CHECK_HIP(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(float)));
    return p;
}

std::vector<float> to_host(const float* p, size_t n) {
    std::vector<float> v(n);
    CHECK_HIP(hipMemcpy(v.data(), p, n * sizeof(float), hipMemcpyDeviceToHost));
    return v;
}

std::vector<float> random_floats(size_t n, float scale, std::mt19937& gen) {
    std::normal_distribution<float> d(0.f, scale);
    std::vector<float> v(n);
    for (auto& x : v) x = d(gen);
    return v;
}

// Linear, ReLU, Linear, ReLU, LayerNorm of one processor MLP (core/meshgraphnet.py:64-78), on the device
struct Mlp {
    float *w0, *b0, *w1, *b1, *ln_w, *ln_b;
    mgn_mlp view;
    Mlp(int in, std::mt19937& gen) {
        w0 = to_device(random_floats((size_t)H * in, 1.f / std::sqrt((float)in), gen));
        b0 = to_device(random_floats(H, 0.05f, gen));
        w1 = to_device(random_floats((size_t)H * H, 1.f / std::sqrt((float)H), gen));
        b1 = to_device(random_floats(H, 0.05f, gen));
        std::vector<float> g = random_floats(H, 0.1f, gen);
        for (auto& x : g) x += 1.f;
        ln_w = to_device(g);
        ln_b = to_device(random_floats(H, 0.05f, gen));
        view = mgn_mlp{w0, b0, w1, b1, ln_w, ln_b};
    }
};

struct Csr {
    std::vector<int32_t> rowptr, sender, recv;
    std::vector<int64_t> perm;
};

Csr receiver_csr(const std::vector<int64_t>& snd, const std::vector<int64_t>& rcv, int64_t n_nodes) {
    Csr c;
    const size_t e = snd.size();
    c.rowptr.resize(n_nodes + 1); c.sender.resize(std::max<size_t>(e, 1)); c.recv.resize(std::max<size_t>(e, 1));
    c.perm.resize(std::max<size_t>(e, 1));
    CHECK_MGN(mgn_receiver_csr(snd.data(), rcv.data(), (int64_t)e, n_nodes, c.rowptr.data(), c.sender.data(), c.recv.data(),
                               c.perm.data()));
    c.sender.resize(e); c.recv.resize(e); c.perm.resize(e);
    return c;
}

// One rank's share: owned nodes [lo, hi), every edge whose receiver it owns, senders owned elsewhere as halo rows
struct Rank {
    int64_t lo, hi, n_own, n_halo, n_loc, e0, n_edges;      // e0: first CSR edge of the whole mesh that this rank holds
    std::vector<int64_t> halo_ids;                           // global ids, ascending (hence grouped by owner)
    Csr csr;                                                 // local numbering [owned | halo]
    int64_t interior[2], n_interior_edges;                   // longest run of owned receivers without a halo sender
    std::vector<std::vector<int32_t>> send_rows;             // per peer: owned rows (local ids) the peer needs
    int32_t* d_rowptr; int32_t* d_sender; int32_t* d_recv;
    std::vector<int32_t*> d_send_idx; std::vector<float*> d_send_buf;
    float *x, *e, *ps, *pr, *aggr;
};

double rel_l2(const std::vector<float>& a, const std::vector<float>& b) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); ++i) { const double d = (double)a[i] - b[i]; num += d * d; den += (double)b[i] * b[i]; }
    return std::sqrt(num / std::max(den, 1e-300));
}

}  // namespace

int main(int argc, char** argv) {
    const int nx = argc > 1 ? std::atoi(argv[1]) : 60, ny = argc > 2 ? std::atoi(argv[2]) : 45;
    const int P = argc > 3 ? std::atoi(argv[3]) : 3, blocks = 2;
    if (nx < 2 || ny < 2 || P < 1 || (int64_t)nx * ny < P) { std::fprintf(stderr, "usage: partitioned_step [nx ny ranks]\n"); return 2; }
    if (mgn_abi_version() != MGN_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 2; }
    std::mt19937 gen(7);

    // ---- mesh: nx x ny nodes, two triangles per cell, then the reference's edge list (sorted by sender, receiver)
    std::vector<int64_t> cells;
    for (int y = 0; y + 1 < ny; ++y)
        for (int x = 0; x + 1 < nx; ++x) {
            const int64_t v0 = (int64_t)y * nx + x, v1 = v0 + 1, v2 = v0 + nx, v3 = v2 + 1;
            const int64_t tri[6] = {v0, v1, v3, v0, v3, v2};
            cells.insert(cells.end(), tri, tri + 6);
        }
    const int64_t n_cells = (int64_t)cells.size() / 3;
    std::vector<int64_t> snd(6 * n_cells), rcv(6 * n_cells);
    int64_t E = 0, N = 0;
    CHECK_MGN(mgn_cells_to_edge_index(cells.data(), n_cells, snd.data(), rcv.data(), 6 * n_cells, &E, &N));
    snd.resize(E); rcv.resize(E);
    const Csr whole = receiver_csr(snd, rcv, N);
    std::printf("mesh %d x %d: %lld nodes, %lld directed edges, %d ranks, %d blocks\n", nx, ny, (long long)N, (long long)E, P, blocks);

    // ---- weights of the blocks, latents (edge latents in receiver-major order)
    std::vector<Mlp> edge_mlp, node_mlp;
    for (int b = 0; b < blocks; ++b) { edge_mlp.emplace_back(3 * H, gen); node_mlp.emplace_back(2 * H, gen); }
    const std::vector<float> x0 = random_floats((size_t)N * H, 1.f, gen), e0 = random_floats((size_t)E * H, 1.f, gen);
    hipStream_t stream, copy_stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_HIP(hipStreamCreate(&copy_stream));

    // ---- the whole mesh in one piece
    int32_t* w_rowptr = to_device(whole.rowptr); int32_t* w_sender = to_device(whole.sender); int32_t* w_recv = to_device(whole.recv);
    float* wx = to_device(x0); float* we = to_device(e0);
    float *wps = device_floats((size_t)N * H), *wpr = device_floats((size_t)N * H), *waggr = device_floats((size_t)N * H);
    for (int b = 0; b < blocks; ++b) {
        CHECK_MGN(mgn_mp_project(wx, edge_mlp[b].w0, edge_mlp[b].b0, wps, wpr, N, stream));
        CHECK_MGN(mgn_mp_edge(we, wps, wpr, w_rowptr, w_sender, w_recv, &edge_mlp[b].view, we, waggr, E, N, stream));
        CHECK_MGN(mgn_mp_node(wx, waggr, w_rowptr, &node_mlp[b].view, wx, N, stream));
    }
    CHECK_HIP(hipStreamSynchronize(stream));
    const std::vector<float> x_ref = to_host(wx, (size_t)N * H), e_ref = to_host(we, (size_t)E * H);

    // ---- the plan of every rank (host integers only)
    std::vector<int64_t> bounds(P + 1);
    for (int r = 0; r <= P; ++r) bounds[r] = (int64_t)r * N / P;
    auto owner_of = [&](int64_t id) { return (int)(std::upper_bound(bounds.begin() + 1, bounds.end(), id) - bounds.begin() - 1); };
    std::vector<Rank> ranks(P);
    for (int r = 0; r < P; ++r) {
        Rank& k = ranks[r];
        k.lo = bounds[r]; k.hi = bounds[r + 1]; k.n_own = k.hi - k.lo;
        k.e0 = whole.rowptr[k.lo]; k.n_edges = whole.rowptr[k.hi] - k.e0;
        std::vector<int64_t> ls, lr;                        // the rank's in-edges in the order of the global edge list
        for (int64_t j = 0; j < E; ++j)
            if (rcv[j] >= k.lo && rcv[j] < k.hi) {
                ls.push_back(snd[j]); lr.push_back(rcv[j]);
                if (snd[j] < k.lo || snd[j] >= k.hi) k.halo_ids.push_back(snd[j]);
            }
        std::sort(k.halo_ids.begin(), k.halo_ids.end());
        k.halo_ids.erase(std::unique(k.halo_ids.begin(), k.halo_ids.end()), k.halo_ids.end());
        k.n_halo = (int64_t)k.halo_ids.size(); k.n_loc = k.n_own + k.n_halo;
        for (size_t j = 0; j < ls.size(); ++j) {
            const bool out = ls[j] < k.lo || ls[j] >= k.hi;
            ls[j] = out ? k.n_own + (std::lower_bound(k.halo_ids.begin(), k.halo_ids.end(), ls[j]) - k.halo_ids.begin()) : ls[j] - k.lo;
            lr[j] -= k.lo;
        }
        k.csr = receiver_csr(ls, lr, k.n_loc);              // stable: the rank's slice of the whole mesh's CSR order
        // interior = the longest run of owned receivers none of whose in-edges comes from a halo row
        std::vector<char> needs(k.n_own + 1, 0);
        for (size_t j = 0; j < k.csr.sender.size(); ++j) if (k.csr.sender[j] >= k.n_own) needs[k.csr.recv[j]] = 1;
        needs[k.n_own] = 1;
        int64_t best0 = 0, best1 = 0, start = 0;
        for (int64_t v = 0; v <= k.n_own; ++v)
            if (needs[v]) { if (v - start > best1 - best0) { best0 = start; best1 = v; } start = v + 1; }
        k.interior[0] = best0; k.interior[1] = best1;
        k.n_interior_edges = k.csr.rowptr[best1] - k.csr.rowptr[best0];
    }
    for (int r = 0; r < P; ++r) {                           // what every peer needs from rank r: r's rows in the peer's halo
        ranks[r].send_rows.assign(P, {});
        for (int q = 0; q < P; ++q)
            if (q != r)
                for (int64_t id : ranks[q].halo_ids) if (owner_of(id) == r) ranks[r].send_rows[q].push_back((int32_t)(id - ranks[r].lo));
    }
    for (Rank& k : ranks) {
        k.d_rowptr = to_device(k.csr.rowptr); k.d_sender = to_device(k.csr.sender); k.d_recv = to_device(k.csr.recv);
        k.x = device_floats((size_t)k.n_loc * H); k.ps = device_floats((size_t)k.n_loc * H); k.pr = device_floats((size_t)k.n_loc * H);
        k.aggr = device_floats((size_t)k.n_loc * H); k.e = device_floats((size_t)k.n_edges * H);
        CHECK_HIP(hipMemcpy(k.x, x0.data() + (size_t)k.lo * H, (size_t)k.n_own * H * sizeof(float), hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(k.e, e0.data() + (size_t)k.e0 * H, (size_t)k.n_edges * H * sizeof(float), hipMemcpyHostToDevice));
        for (int q = 0; q < P; ++q) {
            k.d_send_idx.push_back(to_device(k.send_rows[q]));
            k.d_send_buf.push_back(device_floats(k.send_rows[q].size() * H));
        }
    }

    // ---- the same blocks, rank by rank
    hipEvent_t packed, arrived;
    CHECK_HIP(hipEventCreateWithFlags(&packed, hipEventDisableTiming));
    CHECK_HIP(hipEventCreateWithFlags(&arrived, hipEventDisableTiming));
    for (int b = 0; b < blocks; ++b) {
        for (Rank& k : ranks) CHECK_MGN(mgn_mp_project(k.x, edge_mlp[b].w0, edge_mlp[b].b0, k.ps, k.pr, k.n_own, stream));
        // pack the rows the peers need; the copies stand in for the sends and receives, landing in the halo rows of ps
        for (int r = 0; r < P; ++r)
            for (int q = 0; q < P; ++q)
                if (!ranks[r].send_rows[q].empty())
                    CHECK_MGN(mgn_halo_pack(ranks[r].ps, ranks[r].d_send_idx[q], ranks[r].d_send_buf[q],
                                            (int64_t)ranks[r].send_rows[q].size(), H, stream));
        CHECK_HIP(hipEventRecord(packed, stream));
        CHECK_HIP(hipStreamWaitEvent(copy_stream, packed, 0));
        for (int q = 0; q < P; ++q) {
            size_t off = 0;                                 // halo rows of q are grouped by owner, owners ascending
            for (int r = 0; r < P; ++r) {
                const size_t cnt = r == q ? 0 : ranks[r].send_rows[q].size();
                if (cnt) CHECK_HIP(hipMemcpyAsync(ranks[q].ps + ((size_t)ranks[q].n_own + off) * H, ranks[r].d_send_buf[q],
                                                  cnt * H * sizeof(float), hipMemcpyDeviceToDevice, copy_stream));
                off += cnt;
            }
        }
        CHECK_HIP(hipEventRecord(arrived, copy_stream));
        // interior receivers while the halo rows travel: rowptr + n0, that range's counts, everything else absolute
        for (Rank& k : ranks)
            if (k.interior[1] > k.interior[0])
                CHECK_MGN(mgn_mp_edge(k.e, k.ps, k.pr, k.d_rowptr + k.interior[0], k.d_sender, k.d_recv, &edge_mlp[b].view, k.e,
                                      k.aggr, k.n_interior_edges, k.interior[1] - k.interior[0], stream));
        CHECK_HIP(hipStreamWaitEvent(stream, arrived, 0));
        for (Rank& k : ranks) {
            const int64_t cut[2][2] = {{0, k.interior[0]}, {k.interior[1], k.n_own}};
            for (const auto& c : cut)
                if (c[1] > c[0])
                    CHECK_MGN(mgn_mp_edge(k.e, k.ps, k.pr, k.d_rowptr + c[0], k.d_sender, k.d_recv, &edge_mlp[b].view, k.e, k.aggr,
                                          (int64_t)k.csr.rowptr[c[1]] - k.csr.rowptr[c[0]], c[1] - c[0], stream));
            CHECK_MGN(mgn_mp_node(k.x, k.aggr, k.d_rowptr, &node_mlp[b].view, k.x, k.n_own, stream));
        }
    }
    CHECK_HIP(hipStreamSynchronize(stream));
    int32_t tripped = 0;
    CHECK_MGN(mgn_range_check(&tripped, 1, stream));
    if (tripped) { std::fprintf(stderr, "an operand left the fp16 range\n"); return 1; }

    // ---- stitched result against the whole mesh
    std::vector<float> x_got((size_t)N * H), e_got((size_t)E * H);
    for (const Rank& k : ranks) {
        CHECK_HIP(hipMemcpy(x_got.data() + (size_t)k.lo * H, k.x, (size_t)k.n_own * H * sizeof(float), hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(e_got.data() + (size_t)k.e0 * H, k.e, (size_t)k.n_edges * H * sizeof(float), hipMemcpyDeviceToHost));
        std::printf("rank: owned %lld, halo %lld, edges %lld (interior %lld)\n", (long long)k.n_own, (long long)k.n_halo,
                    (long long)k.n_edges, (long long)k.n_interior_edges);
    }
    const double dx = rel_l2(x_got, x_ref), de = rel_l2(e_got, e_ref);
    std::printf("partitioned against whole mesh: node latents rel-L2 %.3e, edge latents rel-L2 %.3e\n", dx, de);
    const bool ok = dx <= 1e-6 && de <= 1e-6 && std::isfinite(dx) && std::isfinite(de);
    std::printf(ok ? "OK\n" : "MISMATCH\n");
    return ok ? 0 : 1;
}
