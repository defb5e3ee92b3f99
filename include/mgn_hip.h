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
 *     allocates no device memory.  Its only state between calls: the matrix path
 *     (mgn_set_matmul_mode), the sticky range flag (mgn_range_check) and the partial-sum
 *     buffer of mgn_feature_sums_static.
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
#define MGN_MAX_PACK_SLICES 112   /* weight sets per mgn_pack_weights call           */

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

/* Matrix path of the MLP kernels, one setting per process: MGN_MATMUL_F16X2 (default; fp32 operands as two fp16
 * terms, three partial products), MGN_MATMUL_BF16X3 (three bf16 terms, six partial products) or MGN_MATMUL_F32
 * (fp32 matrix instructions).  All meet the same fp32 tolerance against core/meshgraphnet.py:33-120; the first is
 * the fastest and needs every operand below 65520 in magnitude (weights below 512), see mgn_range_check.  The
 * initial value comes from the environment variable MGN_MATMUL = f16x2 | bf16x3 | f32, read at the first call.
 * A forward pass and the backward pass that belongs to it must run under the same setting. */
#define MGN_MATMUL_F32    0
#define MGN_MATMUL_BF16X3 1
#define MGN_MATMUL_F16X2  2
int mgn_get_matmul_mode(void);
int mgn_set_matmul_mode(int mode);

/* Range guard of MGN_MATMUL_F16X2.  The reference computes in fp32 and has no such limit (core/meshgraphnet.py
 * :33-120); here an operand of 65520 or more would round to an fp16 infinity and the result would be silently
 * wrong, so the kernels that split operands raise a sticky per-GPU flag instead.  *tripped = 1 if any launch on
 * the current device since the last reset saw such an operand (activations, gathered rows, hidden activations,
 * weights after their pre-scale of 128), else 0; reset != 0 clears the flag.  Synchronises `stream`.  Callers
 * check once per forward pass or rollout and switch to MGN_MATMUL_BF16X3 (no range limit) when it trips. */
int mgn_range_check(int32_t* tripped /*host*/, int32_t reset, void* stream);
/* The same guard without waiting: enqueues, on `stream`, copies of the flag's two words into host_words[0..1] (pinned
 * host memory that stays valid until the stream has passed this point) and returns.  A non-zero word, read once the
 * stream has passed (an event recorded after this call), means the flag is up; it stays up until mgn_range_check
 * resets it, so a training loop (run_train_multiscale.py:137-142) can poll after every step and look at the answer
 * one step later without ever stalling the host. */
int mgn_range_poll(int32_t* host_words /*pinned host, 2 words*/, void* stream);

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

/* ---- the same two builds for meshes that already live in device memory -------------------------
 * Same results, integer for integer, as mgn_cells_to_edge_index / mgn_receiver_csr; all array
 * arguments are device pointers, the two size outputs of the first are host pointers.  `workspace`
 * is caller-provided device memory of at least mgn_*_workspace_bytes() bytes.  Because sizes travel
 * back to the host (node count, edge count, range check) these calls synchronise `stream`; they run
 * once per mesh.  Key generation, splitting and the CSR scatter are kernels of the library, the key
 * sorts and the duplicate removal are hipCUB device primitives.                                    */
int64_t mgn_edge_build_workspace_bytes(int64_t n_cells);
int mgn_cells_to_edge_index_dev(const int64_t* cells, int64_t n_cells, void* workspace, int64_t ws_bytes,
                                int64_t* out_sender, int64_t* out_recv, int64_t cap, int64_t* out_n_edges /*host*/,
                                int64_t* out_n_nodes /*host*/, void* stream);
int64_t mgn_receiver_csr_workspace_bytes(int64_t n_edges);
int mgn_receiver_csr_dev(const int64_t* sender, const int64_t* recv, int64_t n_edges, int64_t n_nodes, void* workspace,
                         int64_t ws_bytes, int32_t* rowptr, int32_t* csr_sender, int32_t* csr_recv, int64_t* perm,
                         void* stream);

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
 * e_out may alias e_in, and may be NULL: the updated edge latents are then not stored (the last
 * block of a stack, whose edge output nothing reads; saves the 512 B/edge write).
 * Receiver sub-range: rowptr may point at entry n0 of a larger CSR (rowptr[0] need not be 0);
 * n_nodes / n_edges then count the receivers n0 .. n0+n_nodes-1 and their in-edges, while
 * every other pointer still addresses the full arrays.  Used to run the edges that need no
 * remote sender rows while a halo exchange is in flight.                                    */
int mgn_mp_edge(const float* e_in /*[E][H]*/, const float* ps, const float* pr,
                const int32_t* rowptr, const int32_t* csr_sender, const int32_t* csr_recv,
                const mgn_mlp* edge_mlp, float* e_out /*[E][H]*/, float* aggr /*[N][H]*/,
                int64_t n_edges, int64_t n_nodes, void* stream);

/* The same with weighted messages (attention, core/meshgraphnet.py:100-108): aggr[n] = sum of
 * edge_weight[j] * (pre-residual output of edge j) over the in-edges j of n; e_out is not weighted (:120).
 * edge_weight [E] in CSR order, from mgn_edge_attention. */
int mgn_mp_edge_weighted(const float* e_in, const float* ps, const float* pr,
                         const int32_t* rowptr, const int32_t* csr_sender, const int32_t* csr_recv,
                         const mgn_mlp* edge_mlp, const float* edge_weight /*[E]*/, float* e_out, float* aggr,
                         int64_t n_edges, int64_t n_nodes, void* stream);

/* Attention weights (core/meshgraphnet.py:101-107): a_proj = attn_lin(x) [N][H] (any Linear kernel, e.g.
 * mgn_train_linear_fwd); alpha[j] = leaky_relu(<a_proj[s_j], a_proj[r_j]>, 0.01) (scratch, [E]);
 * edge_weight[j] = alpha[j] / max(sum of alpha over the in-edges of r_j, 1e-6), sums in CSR order. */
int mgn_edge_attention(const float* a_proj, const int32_t* rowptr, const int32_t* csr_sender, const int32_t* csr_recv,
                       float* alpha /*[E]*/, float* edge_weight /*[E]*/, int64_t n_edges, int64_t n_nodes, void* stream);

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

/* ---- prepared weights (optional; same results bit for bit) -------------------------
 * The processor kernels keep their weights in registers for a whole launch and convert them from the
 * row-major fp32 matrices when a workgroup starts (core/meshgraphnet.py:64-78 holds them as nn.Linear
 * parameters).  On meshes of a few hundred nodes (one tile per workgroup) that start is most of the launch:
 * mgn_pack_weights converts any number of weight sets in one launch into the layout the kernels load fully
 * coalesced, and the *_packed entry points take those blocks beside the mgn_mlp.  One set = the 128 rows x k
 * columns [col0, col0 + k) of a row-major matrix with row stride ld (k a multiple of 16, col0 and ld multiples
 * of 4); it occupies mgn_packed_weight_bytes(k) = 512 * k bytes, set i at the sum of the sizes before it.
 * The blocks are valid for the matrix path that was current when they were packed, hold a snapshot of the
 * weights (pack again after an optimiser step), and are ignored by the two paths that read fp32 matrices.
 * The range guard of the fp16-split path (mgn_range_check) sees the weights in mgn_pack_weights.           */
typedef struct mgn_weight_slice {
    const float* w;    /* row-major matrix, 128 rows */
    int32_t ld;        /* row stride in floats       */
    int32_t col0;      /* first column of the set    */
    int32_t k;         /* columns of the set         */
    int32_t reserved;
} mgn_weight_slice;

int64_t mgn_packed_weight_bytes(int32_t k);
int mgn_pack_weights(const mgn_weight_slice* slices /*host array*/, int32_t count /* <= MGN_MAX_PACK_SLICES */,
                     void* packed /*device*/, void* stream);

/* mgn_mp_edge / mgn_mp_edge_weighted (edge_weight NULL = unweighted) with the packed sets of
 * edge_mlp->w0[:, 2H:3H] (k = 128) and edge_mlp->w1 (k = 128); edge_mlp still supplies b1, ln_w, ln_b and the
 * matrices for the paths that ignore the packed sets. */
int mgn_mp_edge_packed(const float* e_in, const float* ps, const float* pr,
                       const int32_t* rowptr, const int32_t* csr_sender, const int32_t* csr_recv,
                       const mgn_mlp* edge_mlp, const void* packed_w0e, const void* packed_w1,
                       const float* edge_weight /*[E] or NULL*/, float* e_out, float* aggr,
                       int64_t n_edges, int64_t n_nodes, void* stream);

/* mgn_mp_node (next_edge_w0 NULL) / mgn_mp_node_project with the packed sets of node_mlp->w0 (k = 256),
 * node_mlp->w1 (k = 128) and, with a next layer, next_edge_w0[:, 0:H] and next_edge_w0[:, H:2H] (k = 128 each). */
int mgn_mp_node_packed(const float* x, const float* aggr, const int32_t* rowptr, const mgn_mlp* node_mlp,
                       const void* packed_w0, const void* packed_w1,
                       const float* next_edge_w0 /*or NULL*/, const float* next_edge_b0,
                       const void* packed_next_ws, const void* packed_next_wr,
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

/* The feature builders and the normaliser's accumulation in two launches instead of four (core/meshgraphnet.py:175-178
 * with core/normalization.py:21-23, :30-48): mgn_node_features_sums / mgn_edge_features_sums write the feature rows and,
 * in the same pass, the per-block partial sums that mgn_feature_sums' first kernel would form from them (scratch:
 * mgn_feature_sums_scratch_bytes(rows, F) bytes); mgn_norm_fold adds the partials of `rows` rows in mgn_feature_sums'
 * order, folds the totals into the accumulators and writes mean_std = [mean | std] as mgn_norm_update does -- with the
 * new row count `count` (host scalar, already including `rows`) or, when count_dev is not NULL, with *count_dev += rows
 * in device memory as mgn_norm_update_counted does.  Same values as the four separate calls, bit for bit. */
int mgn_node_features_sums(const float* world_pos, const float* mesh_pos, const float* phi, const float* swelling_phi,
                           int32_t n_swell, const float* node_type, const float* mat_param /*or NULL*/, float* feat,
                           int64_t n_nodes, void* scratch, void* stream);
int mgn_edge_features_sums(const float* world_pos, const float* mesh_pos, const float* phi, const int32_t* csr_sender,
                           const int32_t* csr_recv, float* feat /*[E][7]*/, int64_t n_edges, void* scratch, void* stream);
int mgn_norm_fold(const void* scratch, int64_t rows, int32_t n_feat, float* acc_sum, float* acc_sumsq,
                  int64_t* count_dev /*or NULL*/, int64_t count, float eps, float* mean_std /*[2F]*/, void* stream);

/* mgn_feature_sums without a workspace argument and without any allocation: the per-block partial
 * sums go to a 256 KiB buffer inside the library's code object (one copy per GPU, held until the
 * process ends).  No counterpart in the reference; it exists for callers that cannot allocate or
 * size a workspace: launch sequences under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  Same kernels and the same results, bit for bit, as mgn_feature_sums.
 * rows <= MGN_FEATURE_SUMS_STATIC_MAX_ROWS, else MGN_E_CAP.  All calls on one GPU share that GPU's
 * buffer: calls that overlap in time on one GPU (different streams) race; calls on one stream, or
 * ordered by events, do not.  The buffer's address on a device is looked up once (hipGetSymbolAddress), which
 * cannot happen under stream capture: call mgn_feature_sums_static_prepare() -- it does only that, for the
 * current device -- before capturing, unless an ordinary call on that device has come first. */
#define MGN_FEATURE_SUMS_STATIC_MAX_ROWS ((int64_t)1 << 31)
int mgn_feature_sums_static(const float* feat, int64_t rows, int32_t n_feat, double* sums /*[2F]*/, void* stream);
int mgn_feature_sums_static_prepare(void);

/* Normalizer state update + statistics (core/normalization.py:34-35, :39-48):
 * if sums != NULL: acc_sum += (float)sums[0:F], acc_sumsq += (float)sums[F:2F];
 * then mean = acc_sum / max(count,1), std = sqrt(max(acc_sumsq/max(count,1) - mean^2, 0) + eps).
 * `count` is the accumulated row count AFTER this update (the caller keeps it, as the
 * reference keeps it in a Python int).  mean_std = [mean(F) | std(F)].               */
int mgn_norm_update(float* acc_sum, float* acc_sumsq, const double* sums, int32_t n_feat,
                    int64_t count, float eps, float* mean_std /*[2F]*/, void* stream);

/* mgn_norm_update with the accumulated row count resident on the device: *count_dev += add_rows, then
 * the statistics as above with the new count.  For launch sequences that are captured once and
 * replayed (HIP graphs), where a per-call scalar would be frozen at capture time; the caller keeps
 * its host-side count (core/normalization.py:36) in step.  add_rows = 0 with sums = NULL only
 * recomputes mean_std. */
int mgn_norm_update_counted(float* acc_sum, float* acc_sumsq, const double* sums, int32_t n_feat, int64_t* count_dev,
                            int64_t add_rows, float eps, float* mean_std /*[2F]*/, void* stream);

/* Normalizer.forward / inverse applied to rows (core/normalization.py:24, :26-28):
 * out = (x - mean) / std, or with inverse != 0: out = std * x + mean.  x is [rows][F].  */
int mgn_normalize(const float* x, const float* mean_std /*[2F]*/, float* out, int64_t rows, int32_t n_feat,
                  int32_t inverse, void* stream);

/* Training-noise injection of one frame (core/datasetclass.py:75-89), on the device:
 * out_world = world_pos + z_world * level * conn_length with the x (y) component left alone
 * where node_type[:,1] (node_type[:,2]) == 1; out_phi = phi + z_phi * level * phi_range except
 * where node_type[:,3] == 1.  z_* are the caller's standard-normal draws ([N,2] and [N]);
 * conn_length is the longest mesh edge (:76), phi_range = max(phi) - min(phi) over the whole
 * trajectory (:83).  Products and sums are rounded one by one, in the reference's order. */
int mgn_inject_noise(const float* world_pos /*[N,2]*/, const float* phi /*[N]*/, const float* node_type /*[N,5]*/,
                     const float* z_world /*[N,2]*/, const float* z_phi /*[N]*/, float level, float conn_length,
                     float phi_range, float* out_world /*[N,2]*/, float* out_phi /*[N]*/, int64_t n_nodes, void* stream);

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

/* mgn_decode in its pred mode plus the state feedback of the rollout loop (core/rollout.py:29-32: the last predicted
 * step becomes the next step's world_pos / phi): pred[T-1][n] is also written to next_world[n][0..1], next_phi[n].
 * next_world / next_phi may be world_pos / phi themselves: a row's current state is read before it is overwritten. */
int mgn_decode_step(const float* x /*[N][H]*/, const mgn_mlp* dec, int32_t time_dim, const float* out_mean_std,
                    const float* world_pos, const float* phi, const float* node_type, float* pred /*[T][N][3]*/,
                    float* next_world /*[N][2]*/, float* next_phi /*[N]*/, int64_t n_nodes, void* stream);

/* mgn_decode for the multiscale model, whose decoder reads cat[x, x2] (fine latents | coarse latents mapped back,
 * core/meshgraphnet.py:192-194, :156): dec->w0 is [H][2H].  hidden_ws [N][H] is caller-provided scratch for the
 * hidden activations. */
int mgn_decode_cat(const float* x /*[N][H]*/, const float* x2 /*[N][H]*/, const mgn_mlp* dec, int32_t time_dim,
                   int32_t out_dim, float* hidden_ws /*[N][H]*/, float* delta_norm, const float* out_mean_std,
                   const float* world_pos, const float* phi, const float* node_type, float* pred, int64_t n_nodes,
                   void* stream);

/* Row gather / scatter used at the module boundary to move edge latents between the
 * caller's edge order and CSR order: dst[i] = src[idx[i]] (gather) or dst[idx[i]] = src[i]. */
int mgn_gather_rows(const float* src, const int64_t* idx, float* dst, int64_t rows, int32_t width, void* stream);
int mgn_scatter_rows(const float* src, const int64_t* idx, float* dst, int64_t rows, int32_t width, void* stream);

/* ---------------------------------------------------------------------------------------
 * Training building blocks: loss(graph) + loss.backward() of run_train_multiscale.py:139-141,
 * i.e. the forward of core/meshgraphnet.py:168-211 with the activations kept and its
 * reverse.  Every Linear is handled as 128 -> 128 blocks; inputs narrower than 128 (encoder
 * features, decoder outputs) are zero padded by the caller.  All activations are [rows,128]
 * fp32, row-major.  Weight / bias gradients are deterministic (per-workgroup partials in
 * `scratch`, reduced in block order).  `scratch` holds at least mgn_train_scratch_bytes().
 * ------------------------------------------------------------------------------------- */
int64_t mgn_train_scratch_bytes(void);

/* One reduction launch per step instead of one per call.  mgn_train_ln_bwd, mgn_train_linear_bwd and mgn_train_edge_bwd
 * each end with a small launch that sums their per-workgroup partials (88 of them in a ten-block step, the same sums
 * torch's autograd forms inside its matrix products at run_train_multiscale.py:141).  While a record list is installed
 * for the calling thread (mgn_train_defer_reductions; recs == NULL removes it), these entry points append a record to
 * the list (*count grows) instead of launching; the gradients they name (dW, db, dgamma, dbeta) are then undefined until
 * mgn_train_reduce_many has run over the records.  Every deferred call needs partial space of its own, alive until then:
 * mgn_train_partial_bytes(kind, rows) bytes at `scratch` (kind 0: linear_bwd, 1: ln_bwd, 2: edge_bwd with rows = edges).
 * A call that finds the list full still reduces at once and returns an error.  Same arithmetic, same order: the
 * results are those of the per-call launches bit for bit.  The list lives in host memory; nothing here allocates. */
typedef struct MgnReduceRec {
    const float* partial;        /* per-workgroup partials */
    float* out;                  /* matrix destination (mcount floats, row stride ld) or NULL */
    float* vec[3];               /* up to three 128-float vector destinations */
    int32_t nblocks, stride, mcount, ld, nvec, reserved;
} MgnReduceRec;
int64_t mgn_train_partial_bytes(int32_t kind, int64_t rows);
int mgn_train_defer_reductions(MgnReduceRec* recs /*host*/, int32_t capacity, int32_t* count /*host, in/out*/);
int mgn_train_reduce_many(const MgnReduceRec* recs /*host*/, int32_t n, void* stream);

/* nn.Linear forward (core/meshgraphnet.py:36-43 as used at :95-96, :116-117):
 * out[r,f] = act( sum_k A[r,k] W[f,koff_a+k] (+ sum_k B[r,k] W[f,koff_b+k]) (+ bias[f])
 *                 (+ P[idxP[r],f] + Q[idxQ[r],f]) ),  act = ReLU if relu else identity.
 * W is [128, ldw] row-major; B, bias and the (P, idxP, Q, idxQ) group may be NULL. */
int mgn_train_linear_fwd(const float* A, const float* B, const float* W, int32_t ldw, int32_t koff_a, int32_t koff_b,
                         const float* bias, const float* P, const int32_t* idxP, const float* Q, const int32_t* idxQ,
                         int32_t relu, float* out, int64_t rows, void* stream);

/* nn.LayerNorm(128) forward (eps 1e-5, biased variance): y = LN(v) * gamma + beta (y may be
 * NULL), yres = y + res (residual of core/meshgraphnet.py:118, :120; NULL to skip). */
int mgn_train_ln_fwd(const float* v, const float* gamma, const float* beta, const float* res, float* y, float* yres,
                     int64_t rows, void* stream);

/* LayerNorm backward.  Upstream gradient of row r is G[r] (+ Gg[idx[r]]); either term may be
 * NULL.  gpre = gradient wrt v, multiplied by (v > 0) when relu_mask (v is then the output of
 * the ReLU that precedes the LayerNorm); dgamma[128], dbeta[128] are overwritten. */
int mgn_train_ln_bwd(const float* G, const float* Gg, const int32_t* idx, const float* v, const float* gamma,
                     int32_t relu_mask, float* gpre, float* dgamma, float* dbeta, void* scratch, int64_t rows,
                     void* stream);

/* Backward of the edge MLP of one processor block (core/meshgraphnet.py:95-96 with the MLP of :64-70)
 * from the block's inputs: the hidden activations are recomputed tile by tile with the arithmetic of
 * mgn_mp_edge's default matrix path, so nothing but e_in has to be kept from the forward pass.
 *   upstream gradient of edge j:  g_e[j] (may be NULL) + g_aggr[recv[j]]       (residual :120 and sum :113)
 *   g1[j]  = gradient wrt the output of the first Linear (after its ReLU mask), [E][128]
 *   dpr[n] = sum of g1 over the in-edges of n (gradient of the receiver projection), [N][128]
 *   dW1 [128][128], db1, dgamma, dbeta [128]: gradients of the second Linear and the LayerNorm.
 * Edges in receiver-major order as for mgn_mp_edge; the gradients of the first Linear's three column
 * blocks follow from g1 with mgn_train_linear_bwd / mgn_segment_sum. */
int mgn_train_edge_bwd(const float* e_in, const float* g_e, const float* g_aggr, const float* ps, const float* pr,
                       const int32_t* sender, const int32_t* recv, const int32_t* rowptr, const float* w0 /*[128][384]*/,
                       const float* w1, const float* b1, const float* ln_w, float* g1, float* dpr, float* dW1, float* db1,
                       float* dgamma, float* dbeta, void* scratch, int64_t n_nodes, int64_t n_edges, void* stream);

/* nn.Linear backward for one 128-column block of the weight.  G = gradient wrt the layer's
 * output rows, A = the block's input rows.
 *   dW[f*lddw + k] = sum_r G[r,f] A[r,k]     (dW points at the block's first column; NULL to skip)
 *   db[f]          = sum_r G[r,f]            (NULL to skip; needs dW)
 *   GA[r,k]        = sum_f G[r,f] W[f,koff+k], then * (mask[r,k] > 0) if mask, then + res[r,k] if res
 *                    (NULL to skip; GA may alias res). */
int mgn_train_linear_bwd(const float* G, const float* A, const float* W, int32_t ldw, int32_t koff, const float* mask,
                         const float* res, float* GA, float* dW, int32_t lddw, float* db, void* scratch, int64_t rows,
                         void* stream);

/* out[n,:] = sum over k in [rowptr[n], rowptr[n+1]) of src[order ? order[k] : k, :], rows added
 * in order of k; empty segments give zero.  The receiver-wise scatter_add of
 * core/meshgraphnet.py:113 (order NULL, CSR edge order) and its sender-wise counterpart in
 * the backward pass. */
int mgn_segment_sum(const float* src, const int32_t* rowptr, const int32_t* order, float* out, int64_t n_segments,
                    void* stream);

/* out[r, 0:n_feat] = (feat[r,:] - mean) / std (mean_std = [mean | std], NULL = copy), out[r, n_feat:128] = 0. */
int mgn_train_pad_features(const float* feat, int32_t n_feat, const float* mean_std, float* out, int64_t rows,
                           void* stream);

/* Loss of core/meshgraphnet.py:293-304 for node_out_dim = 3.  decoded is the padded decoder
 * output [N,128] (column 3t+c = step t, component c; delta = decoded * (t+1), :207-211),
 * target_norm [T,N,3], node_type [N,5].  loss3[c] = mean_t sum over nodes with
 * node_type[:,c+1] != 1 of (delta - target_norm)^2. */
int mgn_train_loss_fwd(const float* decoded, const float* target_norm, const float* node_type, int32_t time_dim,
                       int64_t n_nodes, float* loss3, void* scratch, void* stream);
/* gdecoded [N,128] = gradient of sum_c grad3[c] * loss3[c] wrt decoded (zero in the padding). */
int mgn_train_loss_bwd(const float* decoded, const float* target_norm, const float* node_type, const float* grad3,
                       int32_t time_dim, int64_t n_nodes, float* gdecoded, void* stream);

/* out[n] = sum over k in [rowptr[n], rowptr[n+1]) of weight[k] * src[k]: the weighted receiver sum of the attention option
 * (core/meshgraphnet.py:108, :113) for callers that hold the messages as a tensor. */
int mgn_segment_sum_weighted(const float* src /*[E][H]*/, const int32_t* rowptr, const float* weight /*[E]*/,
                             float* out /*[N][H]*/, int64_t n_segments, void* stream);

/* Backward of the attention option (core/meshgraphnet.py:100-108; forward: mgn_edge_attention + mgn_mp_edge_weighted).
 * mgn_train_attn_edge_bwd: per edge, g_weight[j] = <g_aggr[recv[j]], e_out[j] - e_in[j]> (the block's pre-residual
 * edge output is e_out - e_in; with e_in NULL, e_out is taken to be that output itself) and g_e_out[j] = g_e_in[j] (or 0 when NULL) + edge_weight[j] * g_aggr[recv[j]]: the
 * gradient that reaches the edge MLP's output; pass g_e_out to mgn_train_edge_bwd with a zero g_aggr.
 * mgn_train_attn_alpha_bwd: per receiver, g_dot = gradient wrt <a[s], a[r]> from g_weight: g_weight / D minus
 * in-degree * (sum of g_weight * alpha) / D^2 (nothing through a sum clamped at 1e-6), times the leaky slope 1 / 0.01 by
 * the sign of alpha.  The in-degree factor is what autograd yields for the reference's in-place clamp through the
 * repeated index `col` (:106); training with the reference applies it, so this does too.
 * mgn_train_attn_node_grad: out[v] (+)= sum over v's edge list of g_dot[j] * a_proj[other[j]]; called once with the
 * receiver CSR (order NULL, other = csr_sender) and once with the sender view (order = edges by sender,
 * other = csr_recv, accumulate 1): the gradient of attn_lin's output, which then goes through a Linear backward. */
int mgn_train_attn_edge_bwd(const float* g_aggr /*[N][H]*/, const int32_t* csr_recv, const float* e_out, const float* e_in /*or NULL*/,
                            const float* edge_weight /*[E]*/, const float* g_e_in /*[E][H] or NULL*/,
                            float* g_weight /*[E]*/, float* g_e_out /*[E][H]*/, int64_t n_edges, void* stream);
int mgn_train_attn_alpha_bwd(const int32_t* rowptr, const float* alpha /*[E]*/, const float* g_weight /*[E]*/,
                             float* g_dot /*[E]*/, int64_t n_nodes, void* stream);
int mgn_train_attn_node_grad(const float* a_proj /*[N][H]*/, const int32_t* rowptr, const int32_t* order /*or NULL*/,
                             const int32_t* other /*[E]*/, const float* g_dot /*[E]*/, float* out /*[N][H]*/,
                             int32_t accumulate, int64_t n_nodes, void* stream);

/* One Adam step over many tensors in one launch: optimizer.step() of run_train_multiscale.py:115-119,
 * :142 (torch.optim.Adam, amsgrad off: L2 weight decay added to the gradient, bias-corrected moments).
 * table_dev: device array of n_entries records {float* param; const float* grad; float* exp_avg;
 * float* exp_avg_sq; int64_t count} (40 bytes each), one workgroup per record; step counts from 1. */
int mgn_adam_step(const void* table_dev, int32_t n_entries, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int64_t step, void* stream);

/* The same step for captured (HIP graph) launch sequences, whose kernel arguments are fixed at capture time: the
 * six scalars of the update live in device memory, hyper_dev = {lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t), beta1,
 * beta2, eps, weight_decay}.  mgn_adam_hyper fills that record on the host (the arithmetic of mgn_adam_step); the
 * caller copies it to the device before every step or replay.  Same results as mgn_adam_step bit for bit. */
int mgn_adam_hyper(float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, float* hyper6 /*host*/);
int mgn_adam_step_dev(const void* table_dev, int32_t n_entries, const float* hyper_dev, void* stream);

/* ---------------------------------------------------------------------------------------
 * One-hop halo exchange of a mesh partitioned by contiguous node ranges.  The reference runs one mesh on one device
 * (core/rollout.py:28, run_train_multiscale.py:97-98, :138); split over several GPUs, the edge update of
 * core/meshgraphnet.py:95 needs the latents of senders owned by another rank.  A rank numbers its rows
 * [owned | halo], the halo grouped by owner; the transport between ranks (RCCL send/recv or all-to-all, issued by the
 * caller) moves contiguous buffers and receives the forward halo straight into rows num_owned.. of the tensor.
 * These entry points move rows between the row-indexed tensors and those buffers; none allocates, none uses atomics.
 *
 * mgn_halo_pack: send_buf[i][:] = src[send_idx[i]][:] -- the owned rows the peers need, grouped by peer (per block:
 *   the sender projections `ps`; with the attention option also the attn_lin rows).  Any width > 0.
 * mgn_halo_unpack_add: the reverse exchange of the backward pass (gradients of halo rows return to their owners):
 *   buf[add_rows[d]][:] += recv_buf[add_pos[k]][:] for k in [add_rowptr[d], add_rowptr[d+1]) in that order, where
 *   recv_buf holds the returned rows in the order of send_idx and the lists group its positions by destination row,
 *   peers ascending: a fixed summation order.  width % 4 == 0.
 * mgn_halo_stage_state / mgn_halo_unpack_state: the per-rollout-step exchange of the state that the edge features
 *   read at both end points (core/meshgraphnet.py:226-235; fed back per step at core/rollout.py:29-32): copies the
 *   owned rows of (world_pos [.,2], phi [.,1]) into the local [owned | halo] arrays (skipped when they alias) and
 *   writes the records (x, y, phi) of the rows in send_idx to send_buf [n_send][3]; after the exchange the received
 *   records [n_halo][3] are written to world_halo = world_local + 2 * num_owned, phi_halo = phi_local + num_owned.
 * Every send_idx / add_rows entry must be a valid owned row: the plan that produced them guarantees it, the kernels
 * do not check. */
int mgn_halo_pack(const float* src, const int32_t* send_idx, float* send_buf, int64_t n_send, int32_t width, void* stream);
int mgn_halo_unpack_add(const float* recv_buf, const int32_t* add_rowptr, const int32_t* add_rows, const int32_t* add_pos,
                        float* buf, int64_t n_rows, int32_t width, void* stream);
int mgn_halo_stage_state(const float* world_pos, const float* phi, int64_t n_owned, const int32_t* send_idx, int64_t n_send,
                         float* world_local, float* phi_local, float* send_buf, void* stream);
int mgn_halo_unpack_state(const float* recv_buf, int64_t n_halo, float* world_halo, float* phi_halo, void* stream);

/* ---------------------------------------------------------------------------------------
 * Coarse graph of the multiscale option (core/meshgraphnet.py:237-283), built where the mesh lives.  Both entry points
 * take a block-diagonal batch: block b holds the points [node_ptr[b], node_ptr[b+1]) and is sampled / connected on its
 * own (a single mesh is one block).  All arrays are device memory; neither call synchronises or allocates.
 *
 * mgn_fps: farthest-point sampling (:243-254, called at :262).  Writes the sampled point ids (global rows of `points`)
 *   of block b to out_idx[sample_ptr[b] .. sample_ptr[b+1]), the first being node_ptr[b] + starts[b] (the reference
 *   draws it with torch.randint, :248), each further one torch.argmax of the running minimum of
 *   torch.norm(points - last, dim=-1): float32, fused multiply-add chain over the coordinates, rounded square root,
 *   lowest index among equal distances -- the choices of the reference's expressions on the CPU.  A block whose start
 *   lies outside [0, block size) gets -1 in all its slots.  One workgroup per block (four waves up to 1024 points, sixteen beyond); blocks of up to
 *   16 384 points keep everything in registers, larger ones keep the running distances in dist_ws (one float per
 *   point of the batch, may be NULL if max_block_nodes <= 16 384).  dim is 2 or 3; coordinates must be finite.
 *   Cost: (number of samples) rounds over the block's points on one compute unit.
 * mgn_knn_graph: torch_geometric.nn.knn_graph(sampled_points, k, loop=False) (:268; torch-cluster, absent from the
 *   reference tree: restated).  The coarse points of block b are rows first[coarse_ptr[b] .. coarse_ptr[b+1]) of
 *   `points` (first == NULL: the rows themselves).  For every coarse point c the min(k, block size - 1) nearest other
 *   points of its block, by float64 squared distance (products and sums rounded one by one), nearest first, equal
 *   distances by lower index; edge j of point c goes to position edge_ptr[b] + (c - coarse_ptr[b]) * kk + j as
 *   out_sender = neighbour, out_recv = c, both in coarse numbering (positions in `first`).  k in {1, 2, 3, 4, 6, 8}
 *   (the reference uses 3 in 2-D, 4 in 3-D, :259).  Brute force: (block size)^2 distance evaluations. */
int mgn_fps(const float* points /*[N][dim]*/, int32_t dim, const int32_t* node_ptr /*[n_blocks+1]*/,
            const int32_t* sample_ptr /*[n_blocks+1]*/, const int32_t* starts /*[n_blocks]*/, int32_t n_blocks,
            int64_t max_block_nodes, float* dist_ws /*[N] or NULL*/, int64_t* out_idx, void* stream);
/* mgn_fps for ONE mesh with every compute unit at work: two launches per sample (all workgroups update the running
 * distances of their slice and report its best point; one workgroup picks the winner), issued back to back on `stream`
 * with the state in `workspace` (mgn_fps_workspace_floats(n_points) floats) -- nothing waits inside a kernel and the
 * host does not synchronise.  The same samples as mgn_fps, index for index; pays from about 30 000 points on
 * (one round ≈ 8 µs at 100k and ≈ 10 µs at 1M points, against 25 and 270 µs in the single-workgroup kernel). */
int64_t mgn_fps_workspace_floats(int64_t n_points);
int mgn_fps_single(const float* points /*[N][dim]*/, int32_t dim, int64_t n_points, int64_t n_samples, int64_t start,
                   float* workspace, int64_t* out_idx /*[n_samples]*/, void* stream);
int mgn_knn_graph(const float* points /*[N][dim]*/, int32_t dim, const int64_t* first /*[M] or NULL*/,
                  const int32_t* coarse_ptr /*[n_blocks+1]*/, const int64_t* edge_ptr /*[n_blocks+1]*/, int32_t n_blocks,
                  int64_t max_block_points, int32_t k, int64_t* out_sender, int64_t* out_recv, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MGN_HIP_H */
