/*
 * mgn_hip.h -- C ABI of libmgn_hip.so, the MI355X (gfx950) implementation of the
 * MeshGraphNets encode-process-decode hot path.
 *
 * The reference is pure Python: nothing native exists there, so each entry point cites
 * the reference Python it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - Every function returns 0 on success, a positive hipError_t value when a HIP call
 *     failed, or a negative MGN_E_* code; mgn_last_error() then holds a message
 *     (thread-local).
 *   - All device pointers are borrowed for the duration of the call.  The library
 *     allocates no device memory and keeps no state between calls.
 *   - Work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and the call returns without synchronising.
 *   - Latent width is fixed at MGN_H = 128 (hidden_size of every shipped config).
 *     Activations and weights are fp32, row-major; nn.Linear weights are [out][in].
 *   - Edges are processed in receiver-major ("CSR") order: edge j of that order has
 *     sender csr_sender[j], receiver csr_recv[j], and the in-edges of node n are
 *     j in [rowptr[n], rowptr[n+1]).  mgn_receiver_csr() builds it from an edge list.
 */
#ifndef MGN_HIP_H
#define MGN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGN_H 128
#define MGN_ABI_VERSION 1
#define MGN_MAX_FEAT 16      /* widest raw feature row an encoder accepts            */
#define MGN_MAX_DEC_OUT 32   /* widest decoder output (node_out_dim * time_dim)      */

#define MGN_E_ARG   (-1)     /* bad argument (null pointer, size out of range, ...)  */
#define MGN_E_RANGE (-2)     /* index outside [0, N)                                 */
#define MGN_E_CAP   (-3)     /* output capacity too small                            */

/* Two-Linear MLP weights as the reference's factories lay them out
 * (core/meshgraphnet.py:33-45 and :64-78): Linear w0/b0, ReLU, Linear w1/b1,
 * optional LayerNorm ln_w/ln_b (NULL when absent). */
typedef struct mgn_mlp {
    const float* w0;   /* [H][in]  */
    const float* b0;   /* [H]      */
    const float* w1;   /* [out][H] */
    const float* b1;   /* [out]    */
    const float* ln_w; /* [H] or NULL */
    const float* ln_b; /* [H] or NULL */
} mgn_mlp;

int mgn_abi_version(void);
const char* mgn_last_error(void);

/* ---- host side: edge build (integer work, bit-exact) ------------------------------
 * cells_to_edge_index, core/datasetclass.py:7-20, including what PyG 2.6.1
 * to_undirected does at :19: emit c0->c1, c1->c2, c2->c0 and the reversed copies, sort
 * by sender * N + receiver with N = max id + 1, drop duplicates.
 * out_sender/out_recv need room for `cap` entries (6 * n_cells always suffices).     */
int mgn_cells_to_edge_index(const int64_t* cells, int64_t n_cells,
                            int64_t* out_sender, int64_t* out_recv, int64_t cap,
                            int64_t* out_n_edges, int64_t* out_n_nodes);

/* Receiver-major view of any directed edge list (stable: edges of one receiver keep
 * their input order).  perm[j] = input position of CSR edge j.  Replaces the implicit
 * ordering torch_scatter.scatter_add imposes at core/meshgraphnet.py:113.            */
int mgn_receiver_csr(const int64_t* sender, const int64_t* recv, int64_t n_edges, int64_t n_nodes,
                     int32_t* rowptr /*[N+1]*/, int32_t* csr_sender /*[E]*/, int32_t* csr_recv /*[E]*/,
                     int64_t* perm /*[E]*/);

/* ---- device side: processor (core/meshgraphnet.py:86-120, attention off) ----------
 * The first edge Linear acts on cat[x[s], x[r], e] (:95); it is evaluated as
 *   ps[s] + pr[r] + e @ W0[:, 2H:3H]^T,  ps = x @ W0[:, 0:H]^T,  pr = x @ W0[:, H:2H]^T + b0
 * with ps/pr computed once per node by mgn_mp_project.                               */
int mgn_mp_project(const float* x /*[N][H]*/, const float* edge_w0 /*[H][3H]*/, const float* edge_b0 /*[H]*/,
                   float* ps /*[N][H]*/, float* pr /*[N][H]*/, int64_t n_nodes, void* stream);

/* Edge update + receiver-segmented sum (:95-113 and the edge residual of :120).
 * e_out = LN(ReLU(W1 ReLU(first Linear) + b1)) + e_in ; aggr[n] = sum of the
 * pre-residual outputs over the in-edges of n, added in CSR order (deterministic).
 * Rows of aggr belonging to nodes without in-edges are left untouched.
 * e_out may alias e_in.                                                               */
int mgn_mp_edge(const float* e_in /*[E][H]*/, const float* ps, const float* pr,
                const int32_t* rowptr, const int32_t* csr_sender, const int32_t* csr_recv,
                const mgn_mlp* edge_mlp, float* e_out /*[E][H]*/, float* aggr /*[N][H]*/,
                int64_t n_edges, int64_t n_nodes, void* stream);

/* Node update (:116-118): x_out = LN(ReLU(W1 ReLU(W0 cat[x, aggr] + b0) + b1)) + x.
 * aggr rows of nodes with rowptr[n] == rowptr[n+1] are read as zero.
 * x_out may alias x.                                                                  */
int mgn_mp_node(const float* x /*[N][H]*/, const float* aggr /*[N][H]*/, const int32_t* rowptr,
                const mgn_mlp* node_mlp, float* x_out /*[N][H]*/, int64_t n_nodes, void* stream);

/* The same node update fused with the NEXT layer's mgn_mp_project applied to x_out (saves one
 * pass over the node latents per layer): ps/pr receive the projections of the updated latents
 * through next_edge_w0 [H][3H] / next_edge_b0 [H].                                          */
int mgn_mp_node_project(const float* x, const float* aggr, const int32_t* rowptr, const mgn_mlp* node_mlp,
                        const float* next_edge_w0, const float* next_edge_b0,
                        float* x_out, float* ps, float* pr, int64_t n_nodes, void* stream);

/* ---- device side: features, normaliser, encoder, decoder --------------------------
 * _build_node_features (core/meshgraphnet.py:212-225): feat row =
 * [world-mesh (2), phi (1), swelling_phi (n_swell), node_type (5), mat_param (2, if non-NULL)]. */
int mgn_node_features(const float* world_pos /*[N][2]*/, const float* mesh_pos /*[N][2]*/,
                      const float* phi /*[N]*/, const float* swelling_phi /*[N][n_swell]*/, int32_t n_swell,
                      const float* node_type /*[N][5]*/, const float* mat_param /*[2] or NULL*/,
                      float* feat /*[N][F]*/, int64_t n_nodes, void* stream);

/* _build_edge_features (:226-235) for CSR edge j: sender minus receiver of mesh_pos (2),
 * its norm, world_pos (2), its norm, phi (1).  feat is [E][7].                        */
int mgn_edge_features(const float* world_pos, const float* mesh_pos, const float* phi,
                      const int32_t* csr_sender, const int32_t* csr_recv,
                      float* feat /*[E][7]*/, int64_t n_edges, void* stream);

/* Column sums and sums of squares of feat[M][F] in float64, fixed summation order
 * (Normalizer._accumulate, core/normalization.py:30-35).  sums = [sum(F) | sumsq(F)].
 * scratch needs mgn_feature_sums_scratch_bytes(M, F) bytes.                           */
int64_t mgn_feature_sums_scratch_bytes(int64_t rows, int32_t n_feat);
int mgn_feature_sums(const float* feat, int64_t rows, int32_t n_feat, double* sums /*[2F]*/,
                     void* scratch, void* stream);

/* Normalizer state update + statistics (core/normalization.py:34-35, :39-48):
 * if sums != NULL: acc_sum += (float)sums[0:F], acc_sumsq += (float)sums[F:2F];
 * then mean = acc_sum / max(count,1), std = sqrt(max(acc_sumsq/max(count,1) - mean^2, 0) + eps).
 * `count` is the accumulated row count AFTER this update (the caller keeps it, as the
 * reference keeps it in a Python int).  mean_std = [mean(F) | std(F)].               */
int mgn_norm_update(float* acc_sum, float* acc_sumsq, const double* sums, int32_t n_feat,
                    int64_t count, float eps, float* mean_std /*[2F]*/, void* stream);

/* Normalizer.forward / inverse applied to rows (core/normalization.py:24, :26-28):
 * out = (x - mean) / std, or with inverse != 0: out = std * x + mean.  x is [rows][F].  */
int mgn_normalize(const float* x, const float* mean_std /*[2F]*/, float* out, int64_t rows, int32_t n_feat,
                  int32_t inverse, void* stream);

/* Encoder (core/meshgraphnet.py:137-138, :177-178): out = LN(W1 ReLU(W0 z + b0) + b1),
 * z = (feat - mean) / std  (Normalizer.forward, core/normalization.py:24).  mean_std may
 * be NULL for z = feat.  n_feat <= MGN_MAX_FEAT.                                      */
int mgn_encode(const float* feat /*[M][F]*/, int32_t n_feat, const float* mean_std /*[2F] or NULL*/,
               const mgn_mlp* enc, float* out /*[M][H]*/, int64_t rows, void* stream);

/* Decoder + integrator (core/meshgraphnet.py:204-211 and predict :305-317).
 * dec = W1 ReLU(W0 x + b0) + b1, n_out = time_dim * out_dim <= MGN_MAX_DEC_OUT.
 * delta_norm[t][n][k] = dec[n][t*out_dim + k] * (t+1)            (written if non-NULL)
 * pred[t][n][k] = cur[n][k] + masked(delta_norm * std[t][k] + mean[t][k])  (if non-NULL;
 * requires out_dim == 3, cur = [world_pos(2), phi(1)], mask: node_type[n][k+1] == 1 -> 0;
 * out_mean_std = [mean(T*3) | std(T*3)] of the output normaliser).                    */
int mgn_decode(const float* x /*[N][H]*/, const mgn_mlp* dec, int32_t time_dim, int32_t out_dim,
               float* delta_norm /*[T][N][out_dim] or NULL*/,
               const float* out_mean_std, const float* world_pos, const float* phi, const float* node_type,
               float* pred /*[T][N][3] or NULL*/, int64_t n_nodes, void* stream);

/* Row gather / scatter used at the module boundary to move edge latents between the
 * caller's edge order and CSR order: dst[i] = src[idx[i]] (gather) or dst[idx[i]] = src[i]. */
int mgn_gather_rows(const float* src, const int64_t* idx, float* dst, int64_t rows, int32_t width, void* stream);
int mgn_scatter_rows(const float* src, const int64_t* idx, float* dst, int64_t rows, int32_t width, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MGN_HIP_H */
