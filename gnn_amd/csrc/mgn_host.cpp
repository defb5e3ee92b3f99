// Host-side entry points of libmgn_hip.so: error text, edge build, receiver-major CSR.
// Integer work only; results are exact.
#include "mgn_internal.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static thread_local char g_err[512] = "";

void mgn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mgn_last_error(void) { return g_err; }
extern "C" int mgn_abi_version(void) { return MGN_ABI_VERSION; }

// LSD radix sort of non-negative 64-bit keys, 16 bits per pass, skipping passes whose
// digit is constant.  Keys here are sender * N + receiver < N^2.
static void radix_sort_u64(std::vector<uint64_t>& a) {
    const size_t n = a.size();
    if (n < 2) return;
    uint64_t all_or = 0;
    for (uint64_t v : a) all_or |= v;
    std::vector<uint64_t> tmp(n);
    std::vector<size_t> cnt(1u << 16);
    uint64_t* src = a.data();
    uint64_t* dst = tmp.data();
    for (int shift = 0; shift < 64; shift += 16) {
        if (((all_or >> shift) & 0xFFFFu) == 0) continue;
        std::fill(cnt.begin(), cnt.end(), 0);
        for (size_t i = 0; i < n; ++i) ++cnt[(src[i] >> shift) & 0xFFFFu];
        size_t run = 0;
        for (size_t d = 0; d < cnt.size(); ++d) { size_t c = cnt[d]; cnt[d] = run; run += c; }
        for (size_t i = 0; i < n; ++i) dst[cnt[(src[i] >> shift) & 0xFFFFu]++] = src[i];
        std::swap(src, dst);
    }
    if (src != a.data()) std::copy(src, src + n, a.data());
}

extern "C" int mgn_cells_to_edge_index(const int64_t* cells, int64_t n_cells,
                                       int64_t* out_sender, int64_t* out_recv, int64_t cap,
                                       int64_t* out_n_edges, int64_t* out_n_nodes) {
    if (n_cells < 0 || !out_n_edges || !out_n_nodes || (n_cells > 0 && !cells)) {
        mgn_set_error("mgn_cells_to_edge_index: bad argument");
        return MGN_E_ARG;
    }
    *out_n_edges = 0;
    *out_n_nodes = 0;
    if (n_cells == 0) return 0;
    int64_t mx = -1;
    for (int64_t i = 0; i < 3 * n_cells; ++i) {
        if (cells[i] < 0) {
            mgn_set_error("mgn_cells_to_edge_index: negative node id %lld at cell %lld",
                          (long long)cells[i], (long long)(i / 3));
            return MGN_E_RANGE;
        }
        mx = std::max(mx, cells[i]);
    }
    const uint64_t n = (uint64_t)mx + 1;
    if (n > (1ull << 31)) {
        mgn_set_error("mgn_cells_to_edge_index: node id %lld too large", (long long)mx);
        return MGN_E_RANGE;
    }
    std::vector<uint64_t> keys;
    keys.reserve((size_t)6 * n_cells);
    for (int64_t c = 0; c < n_cells; ++c) {
        const uint64_t v0 = cells[3 * c], v1 = cells[3 * c + 1], v2 = cells[3 * c + 2];
        const uint64_t a[3] = {v0, v1, v2}, b[3] = {v1, v2, v0};
        for (int k = 0; k < 3; ++k) {
            keys.push_back(a[k] * n + b[k]);
            keys.push_back(b[k] * n + a[k]);
        }
    }
    radix_sort_u64(keys);
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    const int64_t ne = (int64_t)keys.size();
    *out_n_edges = ne;
    *out_n_nodes = (int64_t)n;
    if (ne > cap || !out_sender || !out_recv) {
        mgn_set_error("mgn_cells_to_edge_index: capacity %lld < %lld edges", (long long)cap, (long long)ne);
        return MGN_E_CAP;
    }
    for (int64_t i = 0; i < ne; ++i) {
        out_sender[i] = (int64_t)(keys[i] / n);
        out_recv[i] = (int64_t)(keys[i] % n);
    }
    return 0;
}

extern "C" int mgn_receiver_csr(const int64_t* sender, const int64_t* recv, int64_t n_edges, int64_t n_nodes,
                                int32_t* rowptr, int32_t* csr_sender, int32_t* csr_recv, int64_t* perm) {
    if (n_edges < 0 || n_nodes < 0 || !rowptr || (n_edges > 0 && (!sender || !recv || !csr_sender || !csr_recv))) {
        mgn_set_error("mgn_receiver_csr: bad argument");
        return MGN_E_ARG;
    }
    if (n_nodes >= (1ll << 31) - 64 || n_edges >= (1ll << 31) - 64) {
        mgn_set_error("mgn_receiver_csr: sizes exceed the int32 index range");
        return MGN_E_RANGE;
    }
    for (int64_t i = 0; i < n_edges; ++i) {
        if (sender[i] < 0 || sender[i] >= n_nodes || recv[i] < 0 || recv[i] >= n_nodes) {
            mgn_set_error("mgn_receiver_csr: edge %lld = (%lld, %lld) outside [0, %lld)", (long long)i,
                          (long long)sender[i], (long long)recv[i], (long long)n_nodes);
            return MGN_E_RANGE;
        }
    }
    std::vector<int64_t> fill((size_t)n_nodes + 1, 0);
    for (int64_t i = 0; i < n_edges; ++i) ++fill[recv[i] + 1];
    for (int64_t v = 0; v < n_nodes; ++v) fill[v + 1] += fill[v];
    for (int64_t v = 0; v <= n_nodes; ++v) rowptr[v] = (int32_t)fill[v];
    for (int64_t i = 0; i < n_edges; ++i) {          // stable counting sort by receiver
        const int64_t j = fill[recv[i]]++;
        csr_sender[j] = (int32_t)sender[i];
        csr_recv[j] = (int32_t)recv[i];
        if (perm) perm[j] = i;
    }
    return 0;
}

static int g_matmul_mode = -1;

int mgn_matmul_mode() {
    if (g_matmul_mode < 0) {
        const char* e = getenv("MGN_MATMUL");
        g_matmul_mode = (e && !strcmp(e, "f32")) ? 0 : (e && !strcmp(e, "bf16x3")) ? 1 : 2;
    }
    return g_matmul_mode;
}

void mgn_set_matmul_mode_internal(int mode) { g_matmul_mode = mode; }

extern "C" int mgn_get_matmul_mode(void) { return mgn_matmul_mode(); }

extern "C" int mgn_set_matmul_mode(int mode) {
    if (mode < 0 || mode > 2) {
        mgn_set_error("mgn_set_matmul_mode: mode %d is not one of MGN_MATMUL_F32 (0), MGN_MATMUL_BF16X3 (1), MGN_MATMUL_F16X2 (2)", mode);
        return MGN_E_ARG;
    }
    g_matmul_mode = mode;
    return 0;
}
