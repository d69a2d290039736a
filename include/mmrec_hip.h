/*
 * mmrec_hip.h -- C ABI of libmmrec_hip.so, the MI355X (gfx950) implementation of the MMRec hot path.
 *
 * The reference (enoche/MMRec) has no native layer and no FFI: its hot path is a set of stock torch
 * calls inside each model class.  Each entry point below replaces one of
 * those call sites; the citation after "replaces" is the reference file:line under
 * /root/reference/src.  A maintainer binds them with ctypes (see INTEGRATION.md and
 * mmrec_amd/_lib.py), which is what a Python reference would use as its FFI.
 *
 * Contract:
 *   - plain pointers and sizes only; no torch types; all pointers are DEVICE pointers unless a
 *     parameter is documented as host;
 *   - every call ENQUEUES work on `stream` (a hipStream_t passed as void*) and returns at once; it
 *     never synchronises, never allocates or frees device memory and keeps no global mutable state
 *     (re-entrant).  The caller owns every buffer, including workspaces whose size the matching
 *     *_workspace_bytes() call reports;
 *   - return value: 0 on success, otherwise a hipError_t value (mmrec_error_string() names it), or
 *     MMREC_ERR_* below for argument errors detected on the host;
 *   - row-major contiguous fp32 matrices; int32 CSR indices; int64 ids where the reference hands
 *     over torch.LongTensor ids (batches, masks, top-K output).
 */
#ifndef MMREC_HIP_H
#define MMREC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMREC_ABI_VERSION 14
#define MMREC_EMB_DIM 64 /* embedding_size the SpMM / BPR / top-K kernels are specialised for (overall.yaml:16) */

#define MMREC_ERR_BAD_ARG 10001      /* null pointer / negative size / unsupported d or k */
#define MMREC_ERR_UNSUPPORTED 10002  /* shape outside what the kernel is built for */

typedef void* mmrec_stream_t; /* hipStream_t */

int mmrec_abi_version(void);
const char* mmrec_error_string(int err);

/* ------------------------------------------------------------------------------------------------
 * P2  sparse propagation  Y = alpha * A X (+ beta * Z), optional fused running layer sum.
 * replaces: torch.sparse.mm(adj, ego) -- models/freedom.py:167,172  bm3.py:90  layergcn.py:131
 *           lightgcn.py:120  lattice.py:169,188  common/encoders.py:99,122 ; and the
 *           stack(...).mean(dim=1) epilogue -- freedom.py:175-176  bm3.py:92-93  lightgcn.py:122-123.
 *
 * A is CSR (rowptr[n_rows+1], colidx[nnz], vals[nnz]) with n_rows local rows whose column ids index
 * rows of X (any number of X rows; a rank of a row-sharded graph passes its row block here).
 * d must be a multiple of 64, at most 384 (64 for the d = 64 graph models; 256 / 384 for MMGCN's
 * modality layers) -- or 8, 16 or 32: ONE FEATURE SLICE of a 64-wide table (X, Y, Z, acc_* are [rows, d] row-major slices;
 * the feature-sliced multi-GPU layout, where a rank owns 64 / P columns of every table and the whole graph: a column's
 * sum is the d = 64 launch's, operation for operation, so the P slices ARE the columns of the single-GPU result).
 * Rows are summed in CSR order, a fixed order that does not depend on how rows are
 * partitioned over GPUs, so sharded == single-GPU bit for bit.
 *
 * Rows longer than `long_row_threshold` are not handled by the row kernel; the caller lists them
 * in a plan (host-built, see mmrec_spmm_plan_*): long_rows[n_long] (row ids, ascending),
 * long_chunk_ptr[n_long+1] (prefix sum of ceil(deg / MMREC_SPMM_CHUNK) per long row).  Each chunk
 * is reduced by one workgroup into `partials` (n_chunks x 64 fp32 workspace) and the chunks of a row
 * are then summed in order -- deterministic, no float atomics.  partials = n_chunks * d floats.
 * long_tickets (ABI 7; may be NULL): n_long int32 counters, ZERO before the first call and left at zero by every call.
 * With them, graphs of at most MMREC_SPMM_FUSED_REDUCE_MAX_ROWS rows finish a multi-chunk row inside the launch (the chunk
 * block that arrives last sums the partials, in the same order: same bits) instead of in a second launch -- these graphs
 * are cache resident and latency bound (Amazon-Baby: 18.5 -> 15.4 us per layer).  One graph's tickets / partials must not be
 * used by two launches at the same time.
 *
 * Epilogue per row r (y = alpha * sum + beta * Z[r], Z may be NULL):
 *      Y[r] = y                         (Y may be NULL when only the running sum is wanted)
 *      acc_out[r] = acc_scale * (acc_in[r] + y)      (when acc_out != NULL; acc_in may alias acc_out)
 * which is how the LightGCN layer mean (1/(L+1) * sum_l E_l) is accumulated without a stack+mean pass.
 * ---------------------------------------------------------------------------------------------- */
#ifndef MMREC_SPMM_CHUNK
#define MMREC_SPMM_CHUNK 512            /* nnz per long-row chunk (one workgroup) */
#endif
#define MMREC_SPMM_FUSED_REDUCE_MAX_ROWS (1 << 18)
#define MMREC_SPMM_LONG_ROW_DEFAULT 32  /* long_row_threshold of HBM-sized graphs; cache-resident ones (<= 2^18 columns) run
                                          * 30 % faster with 16: mmrec_amd/hip_ops.py default_long_row_threshold */

int mmrec_spmm_csr_f32(const int32_t* rowptr, const int32_t* colidx, const float* vals,
                       const float* X, float* Y, const float* Z, const float* acc_in, float* acc_out,
                       int32_t n_rows, int32_t d, float alpha, float beta, float acc_scale,
                       int32_t long_row_threshold, const int32_t* long_rows,
                       const int32_t* long_chunk_ptr, int32_t n_long, int32_t n_chunks,
                       float* partials, int32_t* long_tickets, mmrec_stream_t stream);

/* The same product at LISTED rows only (ABI 10): Y[i] = (A X)[rows[i]] + Z[rows[i]] (z_compact = 0; Z may be NULL) or
 * + Z[i] (z_compact = 1: Z is [n_list, d] like Y), i < n_list, rows int64 (duplicates allowed), Y [n_list, d] compact,
 * d = 8 / 16 / 32 / 64.  For the training step of a model that consumes a propagated table at its batch rows only -- FREEDOM's
 * item-item layer, freedom.py:173-177 read at :197-199: 4096 of 500,000 rows at config 5.
 * Bits: those of mmrec_spmm_csr_f32 with the same long_row_threshold for every row that does not span several chunks
 * (n_chunks == n_long in the graph's plan; callers keep the full launch otherwise).
 * mmrec_spmm_push_rows_f32 is the transposed product of the listed rows without a transposed graph:
 * dX[c] += A[r, c] * g_scale * G[i] over the nonzeros (r, c) of the listed rows r = rows[i], and dZ[r] += g_scale * G[i]
 * (either may be NULL; dZ may alias dX); one workgroup per listed row; fp32 atomics into caller-zeroed / accumulating buffers,
 * so the order of the sums differs from a pull launch's in the last ulp (as the sampled-scoring backward's scatters do). */
int mmrec_spmm_rows_f32(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* X,
                        const float* Z, int32_t z_compact, const int64_t* rows, int32_t n_list, int32_t d,
                        int32_t long_row_threshold, float* Y, mmrec_stream_t stream);
/* ABI 12: the same for ANY listed row of a d = 64 graph, rows spanning several chunks included (one workgroup per such listed
 * row sums its chunks in the full launch's order: its bits).  max_row_chunks: the largest number of 512-nonzero chunks a row of
 * the graph's long-row plan spans (1: identical to mmrec_spmm_rows_f32; up to 480 = 245,760 nonzeros; more: MMREC_ERR_UNSUPPORTED).
 * For a training step that reads the LAST user-item layer (freedom.py:169-177) at its batch rows only: popular items are in
 * every batch and their rows span dozens of chunks. */
int mmrec_spmm_rows_any_f32(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* X,
                            const float* Z, int32_t z_compact, const int64_t* rows, int32_t n_list, int32_t d,
                            int32_t long_row_threshold, int32_t max_row_chunks, float* Y, mmrec_stream_t stream);
int mmrec_spmm_push_rows_f32(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* G,
                             float g_scale, const int64_t* rows, int32_t n_list, int32_t d, float* dX, float* dZ,
                             mmrec_stream_t stream);

/* One LayerGCN layer in one launch (layergcn.py:131-135): y = A x ; w[row] = cosine_similarity(y[row], ego[row]) with
 * eps 1e-8 per norm ; scaled = w * y (the next layer's input) ; acc_out = acc_in + scaled (acc_in NULL: acc_out = scaled;
 * acc_out NULL: no sum).  Y (the unscaled product, needed by the backward) may be NULL.  d must be 64.  Plan arguments
 * as in mmrec_spmm_csr_f32.  Results are bit-identical to mmrec_spmm_csr_f32 followed by mmrec_cos_scale_fwd_f32. */
int mmrec_spmm_csr_f32_layergcn(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* X,
                                float* Y, const float* ego, float* scaled, float* w, const float* acc_in,
                                float* acc_out, int32_t n_rows, int32_t d, int32_t long_row_threshold,
                                const int32_t* long_rows, const int32_t* long_chunk_ptr, int32_t n_long,
                                int32_t n_chunks, float* partials, int32_t* long_tickets, mmrec_stream_t stream);
/* Host-side plan helpers (pure CPU, rowptr is a HOST pointer).  count: returns n_long and n_chunks;
 * fill: writes long_rows[n_long] and long_chunk_ptr[n_long+1] (host arrays the caller copies to the
 * device).  partials workspace = n_chunks * 64 * 4 bytes. */
int mmrec_spmm_plan_count(const int32_t* rowptr_host, int32_t n_rows, int32_t long_row_threshold,
                          int32_t* n_long, int32_t* n_chunks);
int mmrec_spmm_plan_fill(const int32_t* rowptr_host, int32_t n_rows, int32_t long_row_threshold,
                         int32_t* long_rows, int32_t* long_chunk_ptr);

/* LayerGCN per-layer re-weighting: w[r] = cos(E[r], Ego[r]) with eps 1e-8 (torch >= 2 semantics:
 * each norm clamped), Out[r] = w[r] * E[r]; optional running sum Acc[r] += Out[r].
 * replaces: F.cosine_similarity + einsum('a,ab->ab') -- models/layergcn.py:132-134 (+ the layer sum :136).
 * Backward: given dOut (gradient w.r.t. Out, already including the layer-sum branch), E, Ego, w:
 *   dE, dEgo_add (accumulated into dEgo).  d must be 64. */
int mmrec_cos_scale_fwd_f32(const float* E, const float* Ego, float* Out, float* w, float* acc,
                            int32_t n_rows, int32_t d, mmrec_stream_t stream);
int mmrec_cos_scale_bwd_f32(const float* dOut, const float* E, const float* Ego, const float* w,
                            float* dE, float* dEgo_accum, int32_t n_rows, int32_t d,
                            mmrec_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * P4  sampled scoring: fused gather - dot - (log)sigmoid and its scatter-add backward.
 * replaces: FREEDOM.bpr_loss freedom.py:180-187 (variant LOGSIG, mean) ; LayerGCN.bpr_loss
 *           layergcn.py:140-152 (LOGSIG, sum) ; BPRLoss common/loss.py:33-35 (GAMMA, mean; used by
 *           vbpr.py:94, lightgcn.py:142) ; the row gathers ua[users], ia[pos], ia[neg]
 *           freedom.py:197-199.
 * U [n_u, d], P and N tables [n_i, d] (P and N may be the same table; row strides = d; d a multiple
 * of 64: 64 for the graph models, 128 for VBPR's cat(id, visual) embeddings vbpr.py:71).
 * ids: users[B], pos[B], neg[B] int64.  Outputs: loss_out[1] = scale * sum_b l_b  (scale = 1/B for the
 * mean variants, 1 for sum), coef[B] = d l_b / d x_b  (x_b = <u,p> - <u,n>), for the backward.
 * workspace: mmrec_bpr_workspace_bytes(B).  The loss reduction is a fixed-order tree (deterministic).
 * Backward: dU[users[b]] += g*coef[b]*(p - n) ; dP[pos[b]] += g*coef[b]*u ; dN[neg[b]] -= g*coef[b]*u
 * with g = *grad_scalar (device pointer, upstream gradient of the scalar loss) * scale;
 * duplicate ids are combined with fp32 atomics (order-dependent in the last ulp).
 * ---------------------------------------------------------------------------------------------- */
#define MMREC_BPR_LOGSIG 0 /* l = -logsigmoid(x) */
#define MMREC_BPR_GAMMA 1  /* l = -log(1e-10 + sigmoid(x)) */

size_t mmrec_bpr_workspace_bytes(int32_t batch);
int mmrec_bpr_fwd_f32(const float* U, const float* P, const float* N, const int64_t* users,
                      const int64_t* pos, const int64_t* neg, int32_t batch, int32_t d,
                      int32_t variant, float scale, float* loss_out, float* coef, void* workspace,
                      mmrec_stream_t stream);
int mmrec_bpr_bwd_f32(const float* U, const float* P, const float* N, const int64_t* users,
                      const int64_t* pos, const int64_t* neg, int32_t batch, int32_t d,
                      const float* coef, const float* grad_scalar, float scale, float* dU, float* dP,
                      float* dN, mmrec_stream_t stream);

/* The same loss on a COLUMN SLICE of the tables (feature-sliced multi-GPU layout, SURVEY.md 8e: a rank holds d / P = 8, 16
 * or 32 columns of every table; whole tables, d a multiple of 64, work too).  <u, p> and <u, n> of freedom.py:183-184 are sums
 * over the ranks of partial dot products: mmrec_bpr_dots_f32 writes dots[0..B) = <u, p>, dots[B..2B) = <u, n> over the
 * columns given; the caller sums `dots` over the ranks (one all-reduce of 2B floats per term), then
 * mmrec_bpr_loss_from_dots_f32 computes loss_out / coef from the sums exactly as mmrec_bpr_fwd_f32 does (same workspace
 * size), and mmrec_bpr_bwd_f32 takes that coef on the rank's own columns (d = 8 / 16 / 32 accepted there as well). */
int mmrec_bpr_dots_f32(const float* U, const float* P, const float* N, const int64_t* users,
                       const int64_t* pos, const int64_t* neg, int32_t batch, int32_t d, float* dots,
                       mmrec_stream_t stream);
int mmrec_bpr_loss_from_dots_f32(const float* dots, int32_t batch, int32_t variant, float scale,
                                 float* loss_out, float* coef, void* workspace, mmrec_stream_t stream);

/* Sum of squared L2 norms of gathered rows: out[0] = sum_b ||E[ids[b]]||^2 (fixed-order tree).
 * replaces the gathers + norms of EmbLoss / L2Loss -- common/loss.py:46-51,58-62 as used at
 * lightgcn.py:145-149, layergcn.py:154-161, vbpr.py:95.  Backward: dE[ids[b]] += coef * E[ids[b]]
 * (coef is a device scalar: 2*g for ||.||^2, g for 0.5||.||^2, g/||.|| for the unsquared norm). */
int mmrec_gather_sqnorm_fwd_f32(const float* E, const int64_t* ids, int32_t batch, int32_t d,
                                float* out, void* workspace, mmrec_stream_t stream);
int mmrec_gather_scale_add_bwd_f32(const float* E, const int64_t* ids, int32_t batch, int32_t d,
                                   const float* coef_scalar, float* dE, mmrec_stream_t stream);
/* ABI 14 -- the whole regulariser of a training step in one forward call (two launches) and one backward launch:
 *   out[0] = scale * sum_t f(S_t),  S_t = sum_b ||E[t][ids[t][b]]||^2,  mode 0: f = identity (the L2 regulariser on the
 *   batch's rows: layergcn.py:154-161, lattice.py:214-216), mode 1: f = sqrt (EmbLoss: common/loss.py:46-51 as used at
 *   vbpr.py:95, lightgcn.py:145-149); n_terms <= MMREC_ROWS_REG_MAX_TERMS, batch[t] rows per term, rows of d = 64 k floats.
 *   coef[t] (out, device) = the factor of the backward: dE[t][ids[t][b]] += g[0] * coef[t] * E[t][ids[t][b]] (fp32 atomics;
 *   two terms may name the same E / dE: the item table's positive and negative rows).  A term whose S_t is 0 gets coef 0 in
 *   mode 1, as torch.norm's backward does.  ids[t] NULL: rows 0 .. batch[t] - 1 of E[t] (a whole table: bm3.py:146's
 *   EmbLoss(u, i)).  E, ids, batch, dE are HOST arrays of n_terms entries (copied into the launch).
 * replaces: the per-term mmrec_gather_sqnorm_fwd_f32 / mmrec_gather_scale_add_bwd_f32 calls and the ~20 elementwise
 * launches between them (a quarter of a LayerGCN / VBPR step at Amazon-Baby size). */
#define MMREC_ROWS_REG_MAX_TERMS 6
size_t mmrec_rows_reg_workspace_bytes(int32_t n_terms, int32_t max_batch);
int mmrec_rows_reg_fwd_f32(const float* const* E, const int64_t* const* ids, const int32_t* batch, int32_t n_terms, int32_t d,
                           int32_t mode, float scale, float* out, float* coef, void* workspace, mmrec_stream_t stream);
int mmrec_rows_reg_bwd_f32(const float* const* E, const int64_t* const* ids, const int32_t* batch, int32_t n_terms, int32_t d,
                           const float* coef, const float* g, float* const* dE, mmrec_stream_t stream);

/* ABI 14 -- several BPR terms over the SAME user rows in one forward call (two launches) and one backward launch:
 *   losses[t] = scale * sum_b loss(<U[users[b]], I[t][pos[t][b]]> - <U[users[b]], I[t][neg[t][b]]>)  (losses may be NULL),
 *   total[0] = sum_t w[t] * losses[t];  coef [n_terms][batch] as mmrec_bpr_fwd_f32's per term;  n_terms <= MMREC_BPR_MAX_TERMS.
 *   bwd: dU[users[b]] += c (p - n), dI[t][pos] += c u, dI[t][neg] -= c u with c = g[0] scale w[t] coef[t][b] (fp32 atomics; dU or
 *   any dI[t] may be NULL).  I, pos, neg, w, dI are HOST arrays of n_terms entries.
 * replaces: bpr_loss(id) + reg_weight * (bpr_loss(text) + bpr_loss(image)) of freedom.py:197-211 (per-term form:
 * mmrec_bpr_fwd_f32 / mmrec_bpr_bwd_f32 and the scalar launches that weight and add the three losses). */
#define MMREC_BPR_MAX_TERMS 4
size_t mmrec_bpr_multi_workspace_bytes(int32_t n_terms, int32_t batch);
int mmrec_bpr_multi_fwd_f32(const float* U, const int64_t* users, const float* const* I, const int64_t* const* pos,
                            const int64_t* const* neg, const float* w, int32_t n_terms, int32_t batch, int32_t d, int32_t variant,
                            float scale, float* total, float* losses, float* coef, void* workspace, mmrec_stream_t stream);
int mmrec_bpr_multi_bwd_f32(const float* U, const int64_t* users, const float* const* I, const int64_t* const* pos,
                            const int64_t* const* neg, const float* w, int32_t n_terms, int32_t batch, int32_t d, const float* coef,
                            const float* grad_scalar, float scale, float* dU, float* const* dI, mmrec_stream_t stream);

/* ABI 14 -- several mean-cosine terms in one forward call (two launches) and one backward launch:
 *   out[0] = sum_t w[t] * mean_b cos(X[t][ix[t][b]], Y[t][iy[t][b]])   (ix[t] / iy[t] NULL: row b; Y constant; F.cosine_similarity's
 *   1e-8 clamp; n_terms <= MMREC_COSINE_MAX_TERMS, rows of d = 64 k floats), coef [n_terms][max_batch][2] as mmrec_cosine_fwd_f32's;
 *   bwd: dX[t][ix[t][b]] += g[0] w[t] / batch[t] (coef.x y - coef.y x) by fp32 atomics (dX[t] NULL: no gradient for that term;
 *   terms may share X / dX).  X, ix, Y, iy, w, batch, dX are HOST arrays of n_terms entries.
 * replaces: the six `1 - cosine_similarity(...).mean()` terms of bm3.py:129-144 (per-term form: mmrec_cosine_fwd/bwd_f32). */
#define MMREC_COSINE_MAX_TERMS 8
size_t mmrec_cosine_multi_workspace_bytes(int32_t n_terms, int32_t max_batch);
int mmrec_cosine_multi_fwd_f32(const float* const* X, const int64_t* const* ix, const float* const* Y, const int64_t* const* iy,
                               const float* w, const int32_t* batch, int32_t n_terms, int32_t d, float* out, float* coef,
                               void* workspace, mmrec_stream_t stream);
int mmrec_cosine_multi_bwd_f32(const float* const* X, const int64_t* const* ix, const float* const* Y, const int64_t* const* iy,
                               const float* w, const int32_t* batch, int32_t n_terms, int32_t d, const float* coef,
                               const float* grad_scalar, float* const* dX, mmrec_stream_t stream);

/* ABI 14 -- F.normalize(x, p=2, dim=1) in one launch each way (lattice.py:165, mmgcn.py:167): Y = X / max(||X_row||, eps),
 * inv[row] = +- 1 / max(||X_row||, eps) (out, for the backward; negative: the clamp was active); bwd: dX = |inv| (G - Y (Y . G))
 * (clamp active: |inv| G).  Rows of d % 4 == 0 floats. */
int mmrec_row_normalize_fwd_f32(const float* X, int64_t n, int32_t d, float eps, float* Y, float* inv, mmrec_stream_t stream);
int mmrec_row_normalize_bwd_f32(const float* Y, const float* G, const float* inv, int64_t n, int32_t d, float* dX,
                                mmrec_stream_t stream);

/* ABI 14 -- the elementwise tail of an MMGCN layer (mmgcn.py:170-173, :176-179, :182-185) in one launch each way:
 *   fwd: out [n, wa + wb] = [ leaky_relu(A [n, wa]) | leaky_relu(B [n, wb]) + R [n, wb] ]   (R may be NULL; wa, wb % 4 == 0)
 *   bwd: dA = dOut[:, :wa] * (A > 0 ? 1 : slope), dB likewise from dOut[:, wa:], dR = dOut[:, wa:] (each may be NULL)
 * replaces: F.leaky_relu x 2, the `+ id_embedding`, torch.cat((h, x_hat), dim=1) and their four backward launches. */
int mmrec_cat_leaky_fwd_f32(const float* A, const float* B, const float* R, int64_t n, int32_t wa, int32_t wb, float slope,
                            float* out, mmrec_stream_t stream);
int mmrec_cat_leaky_bwd_f32(const float* A, const float* B, const float* dOut, int64_t n, int32_t wa, int32_t wb, float slope,
                            float* dA, float* dB, float* dR, mmrec_stream_t stream);


/* In-batch InfoNCE between two views of the same ids (d = 64), logits never materialised:
 *   v1 = normalize(E1[ids]), v2 = normalize(E2[ids])  (F.normalize, eps 1e-12)
 *   loss[0] = mean_i -log( exp(<v1_i,v2_i>/tau) / sum_j exp(<v1_i,v2_j>/tau) )
 * replaces MGCN.InfoNCE -- mgcn.py:224-231 as called at mgcn.py:252-253 (and its autograd).
 * The backward call takes the workspace the forward call filled and accumulates into dE1 / dE2
 * (dense tables, either may be NULL); grad_loss is a device scalar. */
size_t mmrec_infonce_workspace_bytes(int32_t batch);
int mmrec_infonce_fwd_f32(const float* E1, const float* E2, const int64_t* ids, int32_t batch,
                          int32_t d, float tau, float* loss, void* workspace, mmrec_stream_t stream);
int mmrec_infonce_bwd_f32(const int64_t* ids, int32_t batch, int32_t d, float tau,
                          const float* grad_loss, float* dE1, float* dE2, void* workspace,
                          mmrec_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * P3  modal feature projection (fp32 MFMA, exact fp32):  Y[n,64] = X[n,F] W[64,F]^T + b
 * replaces: nn.Linear image_trs / text_trs / item_linear -- freedom.py:205,208  bm3.py:102,104
 *           lattice.py:134,136  vbpr.py:70 ; and autograd's dW = dY^T X, db = sum dY, dX = dY W.
 * F must be a multiple of 4; out features must be 64 (mmrec_linear_bwd_w_f32 also takes out = 64 j:
 * dY [n, out], dW [out, F], db [out]).  `workspace` (split-K partials) size from
 * mmrec_linear_workspace_bytes.  Deterministic (partials are summed in order).

 * Wider layers (MMGCN's 4096 -> 256 MLP and 256 x 256 / 384 x 384 convolution weights, mmgcn.py:46-
 * 60,164-188): forward and dX are mmrec_gemm_nt_f32, dW / db the out = 64 j form above.
 * ---------------------------------------------------------------------------------------------- */
size_t mmrec_linear_workspace_bytes(int32_t n, int32_t F, int32_t out);
int mmrec_linear_fwd_f32(const float* X, const float* W, const float* b, float* Y, int32_t n,
                         int32_t F, int32_t out, void* workspace, mmrec_stream_t stream);
/* ABI 8 -- the same projection on the 16-bit matrix cores with SPLIT operands: x = hi + 2^-11 lo' in fp16, three fp16 MFMA
 * products per 16 k instead of eight fp32 ones (fp32-input MFMA is 1/16 of the 16-bit rate and bounds the fp32 form), fp32
 * accumulators, error <= 2^-21 |x||w| per term -- as accurate against float64 as the fp32 form.
 * The whole fp32 range is served (round 5): rows of X / W outside what fp16 can hold -- |.| >= 65520, inf, NaN (found as a
 * non-finite result), or a row whose largest magnitude is below 2^-10 and not 0 -- are detected ON THE DEVICE and their
 * 128-row blocks recomputed by the fp32 kernel inside the same call (flags in the workspace, a second launch that returns at
 * once for unflagged blocks; no host synchronisation, capture-safe).  Unflagged rows: |err| <= 2^-21 sum |x w| + 2^-25 max|x_row|
 * sum |w_row|.  Arguments, workspace (mmrec_linear_workspace_bytes) and results (to rounding) as mmrec_linear_fwd_f32;
 * F % 32 != 0 is served by it. */
int mmrec_linear_fwd_split_f32(const float* X, const float* W, const float* b, float* Y, int32_t n, int32_t F,
                               int32_t out, void* workspace, mmrec_stream_t stream);
/* ABI 11 -- the projection's BACKWARD with split operands, one call: dW [64, F] = dY^T X, db [64] = column sums of dY (fp32,
 * fixed order), dX [n, F] = dY W  (autograd of nn.Linear: freedom.py:58-62,205,208; bm3.py:51-56; lattice.py:90-92).  dW (with
 * db or without) and dX may each be NULL (not wanted).  inf / NaN give non-finite results where F.linear's backward has them.
 *   dW  both operands as THREE bf16 parts (x = b1 + b2 + b3 exactly to 2^-24; bf16 has fp32's exponent range: no scales, no
 *       guard, no second launch), six products per 16 items in one fp32 accumulator set:
 *       |err| <= 2^-22 sum |dy x| per output + fp32 accumulation, for any magnitudes; entries below 2^-110 are carried to an
 *       absolute 2^-133 (bf16's smallest denormal) instead of a relative 2^-24.
 *       (The first form used two fp16 halves and one power-of-two scale per column of dY behind a range guard: the gradient
 *       columns of a real batch span more than any one scale holds and the guard's fp32 fix-up ran on most training steps.)
 *   dX  two fp16 halves of dY and W, three products, the operands brought into fp16's range by exact power-of-two scales (per
 *       row of dY, in registers; per column of W): |err| <= 2^-21 sum |dy w| per output as long as the entries of a dY ROW / a
 *       W COLUMN lie within 2^28 of that row's / column's maximum; smaller entries carry an absolute error of 2^-36 of the
 *       maximum's power of two (negligible against the sum unless the large entries' partners are exactly zero).  Stated, not
 *       guarded.
 * out == 64 and F % 128 == 0 run these kernels, every other shape is handed to mmrec_linear_bwd_w_f32 / mmrec_linear_bwd_x_f32.
 * Deterministic (no float atomics).
 * workspace: mmrec_linear_bwd_split_workspace_bytes (>= mmrec_linear_workspace_bytes). */
size_t mmrec_linear_bwd_split_workspace_bytes(int32_t n, int32_t F, int32_t out);
int mmrec_linear_bwd_split_f32(const float* dY, const float* X, const float* W, float* dW, float* db, float* dX, int32_t n,
                               int32_t F, int32_t out, void* workspace, mmrec_stream_t stream);
int mmrec_linear_bwd_w_f32(const float* dY, const float* X, float* dW, float* db, int32_t n,
                           int32_t F, int32_t out, void* workspace, mmrec_stream_t stream);
int mmrec_linear_bwd_x_f32(const float* dY, const float* W, float* dX, int32_t n, int32_t F,
                           int32_t out, mmrec_stream_t stream);
/* C[M, :N] = A[M, K] B[N, K]^T (+ bias[N], may be NULL); K % 32 == 0, N <= ldc <= 2 Mi (32-bit byte offsets within a 128-row block).
 * F.linear(x, W, b) is (A, B) = (x, W); its dX is (A, B) = (dY, W^T). */
int mmrec_gemm_nt_f32(const float* A, const float* B, const float* bias, float* C, int32_t M, int32_t N,
                      int32_t K, int32_t ldc, mmrec_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * P5 / P6  fused scoring + mask + top-K:  for every query row q: top-k over c of <Q[q], C[c]>,
 *          skipping candidates listed for q in a CSR mask (train positives), never materialising
 *          the [nq, nc] score matrix.
 * replaces: torch.matmul(u, i^T) freedom.py:219 (& bm3.py:153 layergcn.py:185 lightgcn.py:160
 *           vbpr.py:105) + scores[mask]=-1e10 + torch.topk(scores, 50) common/trainer.py:307-309 ;
 *           with Q=C=row-normalised features and k=knn_k it is the kNN build freedom.py:79-82.
 * Q [nq, kd], C [nc, kd] fp32 row-major, kd a multiple of 4.  mask_rowptr[nq+1] (int32) /
 * mask_col[...] (int32, ASCENDING within a row: the kernels binary-search it); NULL = no mask.  k <= MMREC_TOPK_MAX.
 * Output sorted by score descending (ties: lower candidate id first): out_idx[nq,k] int64,
 * out_val[nq,k] fp32 (may be NULL).  Masked candidates score -1e10 like the reference, so they can
 * only appear when fewer than k candidates are unmasked.  workspace: mmrec_topk_workspace_bytes.
 * Scores are fp32 dot products in every implementation behind this entry point: kd == 64 or 128 with >= 4096 candidates
 * runs an fp16 matrix-core FILTER with a proven error bound and rescores the ~k survivors per query exactly
 * (topk_filter.hip; `flags & MMREC_TOPK_NO_FILTER` keeps the materialised fp32 path for A/B measurements -- an
 * ARGUMENT since ABI 5: the library reads no environment and keeps no state), the other shapes materialise
 * fp32-MFMA score blocks inside the workspace (topk.hip).  Unknown flag bits: MMREC_ERR_BAD_ARG.
 * ---------------------------------------------------------------------------------------------- */
#define MMREC_TOPK_MAX 128      /* kd % 32 == 0 AND nc <= 2,097,152 candidates (full-sort evaluations, e.g. topk: [10, 20, 50, 100]; the fp16 paths serve kd = 64 / 128 up to 128 and wider rows up to 32, the fp32 block path the rest) */
#define MMREC_TOPK_MAX_OTHER 64 /* every other shape -- row widths that are not a multiple of 32, or more than 2,097,152 candidates -- runs the fused fp32 path, which ranks k <= 64: MMREC_ERR_UNSUPPORTED above it (callers fall back to their dense path; `strict_fused_eval` then raises) */
#define MMREC_TOPK_NO_FILTER 1
size_t mmrec_topk_workspace_bytes(int32_t nq, int32_t nc, int32_t kd, int32_t k);
int mmrec_score_topk_f32(const float* Q, const float* C, int32_t nq, int32_t nc, int32_t kd,
                         const int32_t* mask_rowptr, const int32_t* mask_col, int32_t k,
                         int64_t* out_idx, float* out_val, void* workspace, int32_t flags,
                         mmrec_stream_t stream);
/* Several query blocks against the SAME candidate table (ABI 7) -- the batches of one evaluation and its valid / test pair
 * (trainer.py:262,271,298-310: the item table is frozen while evaluating): the candidate side of the fp16 filter (column
 * means, centred fp16 copy of C, its norms, from 131,072 candidates on the clipping of the <= 32 rows of outlying norm:
 * 0.6 ms at 500K candidates, 8 % of a 65,536-query block) is computed once into
 * a caller-owned buffer of mmrec_topk_prepared_bytes(nc, kd) bytes (0: the filter does not serve this (nc, kd); use
 * mmrec_score_topk_f32) and handed to every call.  The buffer is valid as long as C is unchanged; same results as
 * mmrec_score_topk_f32 bit for bit.  `C` itself is still needed: the exact refinement reads the fp32 rows. */
size_t mmrec_topk_prepared_bytes(int32_t nc, int32_t kd);
int mmrec_topk_prepare_f32(const float* C, int32_t nc, int32_t kd, void* prepared, mmrec_stream_t stream);
int mmrec_score_topk_prepared_f32(const float* Q, const float* C, const void* prepared, int32_t nq, int32_t nc,
                                  int32_t kd, const int32_t* mask_rowptr, const int32_t* mask_col, int32_t k,
                                  int64_t* out_idx, float* out_val, void* workspace, int32_t flags,
                                  mmrec_stream_t stream);
/* WARM evaluation (ABI 12): the same ranking with the filter's threshold taken from a per-query LIST instead of a first pass
 * over all products -- ONE matrix-core pass per call instead of two.
 * replaces: the second and later runs of common/trainer.py:298-310 over the same users: the TEST pass that follows the VALID
 *           pass on the same frozen tables with the same train-positive mask (trainer.py:262,271; utils/dataloader.py:347,
 *           370-391), and every later epoch's evaluation, whose rankings move little from the previous one's.
 * hint [.., hint_k] int32, IN / OUT, k <= hint_k <= MMREC_TOPK_MAX (64 is the natural width for k <= 64, 128 above); hint_rows
 * (int64 [nq], may be NULL): query q uses hint row hint_rows[q] (NULL: row q) -- e.g. a table with one row per user.
 *   IN : candidate ids expected to rank high.  Any k DISTINCT, UNMASKED, in-range ids of the row bound the k-th best score from
 *        below; their approximate scores under the CURRENT Q and C give the threshold (topk_filter.hip:
 *        filter_hint_bound_kernel; with more than k usable ids the k-th largest), so the output is the exact top-k for ANY
 *        list -- bit-identical to mmrec_score_topk_f32's: a stale list only lets more candidates through to the exact
 *        refinement, a row with fewer than k usable ids (-1 padding, duplicates, masked ids) sends its query to the exact slow
 *        queue.  flags & MMREC_TOPK_HINT_COLD: the row is not read (the call runs the two-pass form) -- the FIRST evaluation.
 *   OUT: the call's own ranking of the query -- its top-k followed by the runners-up the refinement ranked anyway (up to
 *        hint_k ids, -1 beyond) -- ready to be the next call's list; flags & MMREC_TOPK_HINT_KEEP leaves the row untouched.
 * queue_counts (int32 [2] on the device, may be NULL): the call ADDS the number of queries the slow queue / the overflow queue
 * served, so that a caller can return to the cold call when its lists have gone stale.
 * prepared: as for mmrec_score_topk_prepared_f32, or NULL (the call prepares C itself).  Served shapes: those of the fp16
 * filter (kd = 64 / 128, 4096 <= nc <= 1,000,000, k <= 128); others: MMREC_ERR_UNSUPPORTED.  Other flag bits: MMREC_ERR_BAD_ARG.
 * workspace: mmrec_topk_workspace_bytes. */
#define MMREC_TOPK_HINT_COLD 1
#define MMREC_TOPK_HINT_KEEP 2
int mmrec_score_topk_hinted_f32(const float* Q, const float* C, const void* prepared, int32_t nq, int32_t nc,
                                int32_t kd, const int32_t* mask_rowptr, const int32_t* mask_col, int32_t k,
                                int32_t* hint, int32_t hint_k, const int64_t* hint_rows,
                                int64_t* out_idx, float* out_val, void* workspace, int32_t* queue_counts,
                                int32_t flags, mmrec_stream_t stream);

/* Deterministic scatter-add of per-sample gradient rows (ABI 7; config `hip_deterministic`): out[ids[b]] += rows[b], the
 * occurrences of an id summed in position order by one owner -- no float atomics, so a batch with duplicated ids gives the
 * same bits run after run (the reference's CPU scatter is deterministic).  `order` = STABLE argsort of ids
 * (int64[n], the caller sorts); ids < 0 are skipped; d a multiple of 64.  The fused backward kernels (bpr / cosine / infonce /
 * gather_scale_add) scatter with hardware fp32 atomics: in deterministic mode they are called on the batch's GATHERED rows
 * with identity ids (every output row is written once) and this entry point does the scatter into the tables. */
int mmrec_scatter_add_rows_sorted_f32(const int64_t* order, const int64_t* ids, const float* rows, int32_t n, int32_t d,
                                      float* out, mmrec_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * P1  graph build on device.
 * replaces: get_norm_adj_mat freedom.py:102-126 (structure from de-duplicated train pairs) ;
 *           _normalize_adj_m freedom.py:145-154 ; the masked COO of pre_epoch_processing :136-143.
 * ---------------------------------------------------------------------------------------------- */
/* counts[i] += number of ids equal to i (int32 histogram, integer atomics: exact). */
int mmrec_degree_count_i32(const int64_t* ids, int64_t n, int32_t* counts, int32_t n_bins,
                           mmrec_stream_t stream);
/* per-edge symmetric normalisation: val[e] = (du[u_e]+1e-7)^-1/2 * (di[i_e]+1e-7)^-1/2 in fp32
 * (freedom.py:148-154). deg_* are int32 counts. */
int mmrec_edge_norm_f32(const int64_t* eu, const int64_t* ei, int64_t n_edges, const int32_t* deg_u,
                        const int32_t* deg_i, float* val, mmrec_stream_t stream);
/* Symmetric bipartite COO in the reference's layout cat(edges, flipped edges), cat(w, w)
 * (freedom.py:139-143): rows/cols/vals have 2*n_edges entries; item node ids are offset by n_users. */
int mmrec_bipartite_expand(const int64_t* eu, const int64_t* ei, const float* w, int64_t n_edges,
                           int32_t n_users, int32_t* rows, int32_t* cols, float* vals,
                           mmrec_stream_t stream);
/* Stable COO -> CSR: entries of a row keep their COO order (radix sort on the row key), so the SpMM
 * summation order is a pure function of the edge list.  rowptr[n_rows+1], colidx[nnz], vals_out[nnz].
 * workspace: mmrec_coo_to_csr_workspace_bytes(nnz, n_rows). */
size_t mmrec_coo_to_csr_workspace_bytes(int64_t nnz, int32_t n_rows);
int mmrec_coo_to_csr(const int32_t* rows, const int32_t* cols, const float* vals, int64_t nnz,
                     int32_t n_rows, int32_t* rowptr, int32_t* colidx, float* vals_out,
                     void* workspace, mmrec_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Rows next to the hot path.
 * ---------------------------------------------------------------------------------------------- */
/* f1  one uniform negative per sample from cand_items[n_cand] (train-seen items), rejected while it
 * is in the user's training history (CSR hist_rowptr/hist_col, item ids sorted inside a row).
 * replaces: TrainDataLoader._sample_neg_ids utils/dataloader.py:267-275 (python `random` loop).
 * Counter-based RNG (splitmix64 of seed, counter, sample index): reproducible, but not the host
 * stream -- the host sampler remains the bit-exact parity mode.  At most 4096 draws per sample. */
int mmrec_sample_negatives_i64(const int64_t* users, int32_t batch, const int32_t* hist_rowptr,
                               const int32_t* hist_col, const int32_t* cand_items, int32_t n_cand,
                               uint64_t seed, uint64_t counter, int64_t* out_neg,
                               mmrec_stream_t stream);
/* f2  hit matrix + per-user Recall / NDCG / Precision / MAP at the cut-offs ks[n_ks] (ascending,
 * <= k) from the top-k ids and the ground-truth CSR (item ids sorted inside a row).
 * replaces: the Python double loop topk_evaluator.py:88-93 and the per-user part of metrics.py:12-105.
 * discount[j] = 1/log2(j+2), idcg_cum[j] = cumulative sum of discount (host-computed doubles, k
 * entries).  Per-user sums run sequentially in the order of numpy's cumsum: out_per_user[u][m][t]
 * (m: 0 recall, 1 ndcg, 2 precision, 3 map) is bit-identical to the reference's per-user value.
 * hit_out [n_users, k] uint8 may be NULL. */
int mmrec_topk_metrics_f64(const int64_t* topk_idx, int32_t n_users, int32_t k,
                           const int32_t* gt_rowptr, const int32_t* gt_col, const double* discount,
                           const double* idcg_cum, const int32_t* ks, int32_t n_ks, uint8_t* hit_out,
                           double* out_per_user, mmrec_stream_t stream);

/* f3  fused dense Adam step on one tensor (16-byte aligned): exp_avg / exp_avg_sq updated in place,
 * bias corrections from `step` (>= 1), optional L2 weight_decay folded into the gradient.
 * replaces: torch.optim.Adam.step common/trainer.py:111-128,189 (same update formula). */
int mmrec_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int64_t step,
                        mmrec_stream_t stream);

/* Graph-replay-safe Adam: the step count (int64) and the learning rate (fp32) live in device memory.
 * mmrec_adam_prepare increments *step_dev and writes hyper_dev[2] = {lr/(1-b1^t), 1/sqrt(1-b2^t)} once
 * per optimizer step; mmrec_adam_step_dev_f32 applies the update to one tensor reading hyper_dev.  A
 * captured hipGraph of a training step therefore replays with correct bias corrections. */
int mmrec_adam_prepare(int64_t* step_dev, const float* lr_dev, float beta1, float beta2, float* hyper_dev,
                       mmrec_stream_t stream);
int mmrec_adam_step_dev_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev,
                            float beta1, float beta2, float eps, float weight_decay,
                            mmrec_stream_t stream);

/* Multi-tensor Adam: ALL parameter tensors of one optimizer step in one launch per 24 tensors (a training
 * step otherwise pays 8-32 tiny launches).  p / g / m / v / n (and lr / step) are HOST arrays of n_tensors
 * entries holding device pointers and element counts; the table is copied into the kernel-argument segment,
 * so the arrays only need to live for the call.  Entries with n == 0 are skipped.  Same update as
 * mmrec_adam_step_f32 (per-tensor lr and step, as torch keeps them) / mmrec_adam_step_dev_f32 (one device
 * hyper pair for all tensors: call mmrec_adam_prepare first).
 * replaces: torch.optim.Adam.step common/trainer.py:111-128,189. */
int mmrec_adam_multi_step_f32(float* const* p, const float* const* g, float* const* m, float* const* v,
                              const int64_t* n, int32_t n_tensors, const float* lr, const int64_t* step,
                              float beta1, float beta2, float eps, float weight_decay, mmrec_stream_t stream);
int mmrec_adam_multi_step_dev_f32(float* const* p, const float* const* g, float* const* m, float* const* v,
                                  const int64_t* n, int32_t n_tensors, const float* hyper_dev, float beta1,
                                  float beta2, float eps, float weight_decay, mmrec_stream_t stream);

/* a14  HOST function (no GPU involved): the reference's negative sampler bit for bit.  Continues CPython's Mersenne
 * Twister from `random.getstate()` (mt_state[624] + *mt_index, both updated in place) and draws, per user of the batch,
 * `all_items[_randbelow(n_all_items)]` until the item is not in the user's history (hist_rowptr [n_users + 1] /
 * hist_items ascending per user, int64) -- the exact output sequence and stream consumption of
 * `_sample_neg_ids` dataloader.py:267-275 (`random.sample(self.all_items, 1)[0]` in a rejection loop).
 * replaces: the per-sample Python loop (0.75 ms per 2048-user batch; ~10 us here). */
int mmrec_host_sample_negatives(uint32_t* mt_state, int32_t* mt_index, const int64_t* users, int32_t batch,
                                const int64_t* hist_rowptr, const int64_t* hist_items, const int64_t* all_items,
                                int32_t n_all_items, int64_t* out);

/* a4  HOST function: `random.sample(range(n), k)` of CPython 3.10 bit for bit (same result list, same generator state
 * afterwards) -- the uniform edge pruning of LayerGCN's alternate epochs, layergcn.py:56-58.  mt_state / mt_index as in
 * mmrec_host_sample_negatives; out [k] int64; scratch [n] int32. */
int mmrec_host_random_sample_range(uint32_t* mt_state, int32_t* mt_index, int32_t n, int32_t k, int64_t* out,
                                   int32_t* scratch);

/* a9'  BM3's BYOL terms: out[0] = scale * sum_b cos(X[ix[b]], Y[iy[b]]) with F.cosine_similarity's clamp (each norm at
 * least 1e-8); rows of d = 64 j floats; ix / iy NULL = row b.  The targets are detached in the reference, so the backward
 * produces the gradient w.r.t. X only: dX[ix[b]] += grad * scale * dcos_b/dx (atomic: duplicate ids).  coef [batch][2] fp32
 * and workspace (mmrec_cosine_workspace_bytes) are written by the forward, coef is read by the backward.
 * replaces: the six `1 - cosine_similarity(p, z.detach(), dim=-1).mean()` terms bm3.py:129-144 (and selfcfed_lgn.py:57-58). */
size_t mmrec_cosine_workspace_bytes(int32_t batch);
int mmrec_cosine_fwd_f32(const float* X, const int64_t* ix, const float* Y, const int64_t* iy, int32_t batch, int32_t d,
                         float scale, float* out, float* coef, void* workspace, mmrec_stream_t stream);
int mmrec_cosine_bwd_f32(const float* X, const int64_t* ix, const float* Y, const int64_t* iy, int32_t batch, int32_t d,
                         const float* coef, const float* grad_scalar, float scale, float* dX, mmrec_stream_t stream);

/* f3  Row-lazy EXACT Adam for a trainable [n_rows, F] table of which a step touches few rows (F % 4 == 0).  Rows
 * with a zero gradient are not visited; their postponed updates (dense-Adam semantics: the moments keep decaying,
 * the parameter keeps moving) are replayed in registers, with the dense kernel's instructions in its order, when the
 * row is next needed -- results equal the dense update bit for bit.
 *   hist     [capacity][2] fp32: step-dependent scalars of optimizer step t, written by mmrec_adam_hist_set(t)
 *   last_step[n_rows] int32: steps already applied per row (0 initially)
 *   ids may hold -1 = "no row" (a slot served elsewhere, e.g. by another rank's shard of the table): skipped everywhere.
 *   owner    [n_rows] int32, INT_MAX where idle: mmrec_adam_rows_owner marks the first position of every row in `ids`
 *            (duplicates allowed); catchup / step consume the marks (entries are INT_MAX again afterwards)
 *   catchup: bring the rows of `ids` (ids == NULL: all n_rows rows) to step t_now
 *   step:    optimizer step t (= t_now + 1) on the rows of `ids`; g [n_ids][F]: row i holds the gradient of OCCURRENCE i;
 *            the occurrences of a table row are summed in position order by the workgroup of its first occurrence
 *            (n_ids <= MMREC_ADAM_ROWS_MAX_IDS: the position list lives in LDS).  presummed != 0: row i already holds the
 *            SUMMED gradient of the table row whose first occurrence is position i (other positions are ignored)
 * replaces: torch.optim.Adam.step on image_embedding / text_embedding (freedom.py:58,61; trainer.py:111-128,189). */
#define MMREC_ADAM_ROWS_MAX_IDS 16000
int mmrec_adam_hist_set(float* hist, int32_t t, float lr, float beta1, float beta2, mmrec_stream_t stream);
int mmrec_adam_rows_owner(const int64_t* ids, int32_t n, int32_t* owner, mmrec_stream_t stream);
int mmrec_adam_rows_catchup_f32(float* p, float* m, float* v, const int64_t* ids, int32_t* owner, int32_t n_ids,
                                int32_t n_rows, int32_t F, int32_t* last_step, const float* hist, int32_t t_now,
                                float beta1, float beta2, float eps, float weight_decay, mmrec_stream_t stream);
int mmrec_adam_rows_step_f32(float* p, float* m, float* v, const int64_t* ids, int32_t* owner, const float* g,
                             int32_t n_ids, int32_t F, int32_t* last_step, int32_t t, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int32_t presummed, mmrec_stream_t stream);
/* The same three calls for a step replayed as a hipGraph (hip_graph_step): nothing step-dependent comes from the host.
 * step_dev [1] int64 and hyper_dev [2] fp32 are the device scalars mmrec_adam_prepare maintains (step count; lr / (1 -
 * b1^t), 1 / sqrt(1 - b2^t)).  Order inside a step: catchup_dev (before prepare: step_dev = steps taken so far) ...
 * backward ... mmrec_adam_prepare ... hist_set_dev ... rows_step_dev.  hist_set_dev does not write beyond `capacity`
 * entries: it raises *overflow (sticky, device int32) instead, which the host checks between epochs; from then on
 * catchup_dev / rows_step_dev (ABI 7: they take the same `capacity`) leave the rows untouched -- there is no table entry to
 * replay from -- so the run can be resumed from the state before the overflowing step. */
int mmrec_adam_hist_set_dev(float* hist, int32_t capacity, const int64_t* step_dev, const float* hyper_dev,
                            int32_t* overflow, mmrec_stream_t stream);
int mmrec_adam_rows_catchup_dev_f32(float* p, float* m, float* v, const int64_t* ids, int32_t* owner, int32_t n_ids,
                                    int32_t n_rows, int32_t F, int32_t* last_step, const float* hist, int32_t capacity,
                                    const int64_t* step_dev, float beta1, float beta2, float eps, float weight_decay,
                                    mmrec_stream_t stream);
int mmrec_adam_rows_step_dev_f32(float* p, float* m, float* v, const int64_t* ids, int32_t* owner, const float* g,
                                 int32_t n_ids, int32_t F, int32_t* last_step, int32_t capacity, const int64_t* step_dev,
                                 const float* hyper_dev, float beta1, float beta2, float eps, float weight_decay,
                                 int32_t presummed, mmrec_stream_t stream);
/* ABI 13 -- OPT-IN fast-forward (config key `lazy_adam_fast_forward`, default off): the arguments and the bookkeeping of
 * catchup / catchup_dev, but a row skipped for more than 12 steps is brought to step t_now in CLOSED FORM instead of being
 * replayed step by step: m *= b1^n, v *= b2^n, p -= m0 * sum_j w_j / (sqrt(v0) d_j + eps) with the sum evaluated as a
 * six-term series around the row's weighted-mean d (csrc/adam.hip; the row scalars W, dbar, M_2..M_6 come from `hist`).
 * ~16 instructions per element whatever the gap (the exact replay: 7 per element AND skipped step).  NOT bit-identical to
 * dense Adam: p within 2e-6 of the distance it moved, m / v within 1e-6 sqrt(n) relative (tests) -- inside north_star's
 * 1e-4 tolerance, outside the "lazy == dense bit for bit" property, hence opt-in.  What the series does not serve to 1e-7
 * (steps before optimizer step 128, where the bias correction of v moves by per cents per step: such a row is replayed to
 * step 128 and advanced in closed form from there; b1^256 > 1e-9; weight decay; short gaps) takes the exact replay inside
 * the same launch.
 * replaces: the same torch.optim.Adam.step as above (trainer.py:111-128,189). */
int mmrec_adam_rows_fastforward_f32(float* p, float* m, float* v, const int64_t* ids, int32_t* owner, int32_t n_ids,
                                    int32_t n_rows, int32_t F, int32_t* last_step, const float* hist, int32_t t_now,
                                    float beta1, float beta2, float eps, float weight_decay, mmrec_stream_t stream);
int mmrec_adam_rows_fastforward_dev_f32(float* p, float* m, float* v, const int64_t* ids, int32_t* owner, int32_t n_ids,
                                        int32_t n_rows, int32_t F, int32_t* last_step, const float* hist,
                                        int32_t capacity, const int64_t* step_dev, float beta1, float beta2, float eps,
                                        float weight_decay, mmrec_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MMREC_HIP_H */
