/*
 * mirec -- C ABI of the MI355X (gfx950) recommender core.
 *
 * The reference has no FFI of its own: its numeric core is reached through the
 * Python APIs of implicit 0.7.2 and lightfm 1.17.  Each entry point below is
 * the device-side replacement of one of those calls and cites the reference
 * call site it stands behind (paths relative to the reference repository).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name starts with `host_`;
 *    the caller owns all memory and the library allocates nothing that outlives
 *    a call.  Its only state besides the thread-local error string is a per-device
 *    record of one-off kernel set-up (dynamic-LDS attributes, the device's
 *    compute-unit count), filled on a device's first launch;
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*); no entry
 *    point synchronises, except mirec_als_plan_build (it returns counts to the
 *    host) and mirec_lightfm_warp_epoch (it uploads a small table);
 *  - return value 0 = success, negative = error (MIREC_E_*); the message is
 *    available from mirec_last_error() on the calling thread;
 *  - factor matrices are row-major fp32 with row stride `f` floats, where `f`
 *    must be a width the kernels are built for (mirec_factor_width(): multiples
 *    of 32 up to 192 for the ALS kernels; multiples of 8 up to 192 for
 *    scoring).  Callers with another width zero-pad columns: zero columns
 *    change no result bit.
 *  - CSR matrices: `indptr` int64[rows+1] (absolute offsets), `indices`
 *    int32[nnz], `data` fp32[nnz].
 */
#ifndef MIREC_H
#define MIREC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIREC_OK 0
#define MIREC_E_INVALID (-1)     /* bad argument (null pointer, size, unsupported width) */
#define MIREC_E_WORKSPACE (-2)   /* workspace too small */
#define MIREC_E_HIP (-3)         /* a HIP runtime call failed */
#define MIREC_E_UNSUPPORTED (-4) /* valid request the kernels are not built for */

typedef void* mirec_stream_t; /* hipStream_t */

int mirec_version(void);
const char* mirec_last_error(void);

/* Smallest supported factor width >= f for the ALS kernels (`for_scoring` = 0)
 * or for the scoring / top-K kernels (`for_scoring` = 1); -1 if f is too large. */
int mirec_factor_width(int f, int for_scoring);

/* ------------------------------------------------------------------------
 * Shared Gram  G = Y^T Y + reg * I   (fp32 MFMA partials, fixed-order reduce).
 * Replaces the `np.dot(Y.T, Y) + regularization * eye` that opens every
 * half-sweep of implicit's `_least_squares_cg`, reached from
 * als/als.py:174 and recommenders/als_recommender.py:136 (`model.fit`).
 * `y_absmax` (one device float, may be null) receives the largest |entry| of Y,
 * found on the same pass: mirec_als_half_sweep takes it beside G to scale the
 * matrix-core operands of its explicit-Gram kernel.
 * `ws` must hold mirec_gram_workspace_bytes(n, f) bytes. */
size_t mirec_gram_workspace_bytes(int64_t n, int f);
int mirec_gram(const float* Y, int64_t n, int f, float reg, float* G, float* y_absmax, void* ws, size_t ws_bytes,
               mirec_stream_t stream);

/* mirec_gram without a workspace argument and without any allocation.  It has
 * no counterpart in the reference: it exists for callers that cannot allocate
 * device memory -- code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  The block partials go to a buffer inside the library
 * (a static device array of MIREC_GRAM_STATIC_BYTES that every GPU the library
 * runs on holds until the process ends; its address on a device is looked up
 * on the first call there).  Same kernels, same result bits as mirec_gram.
 * Inputs with mirec_gram_workspace_bytes(n, f) above that size are refused
 * with MIREC_E_WORKSPACE (at f = 128: more than 131072 rows; at f <= 64 every
 * n fits).  All calls on one GPU share the buffer: calls that overlap there
 * (different streams or threads) race; calls on one stream are ordered. */
#define MIREC_GRAM_STATIC_BYTES (((size_t)32 << 20) + 4096)   /* 512 slots of 128 x 128 partial sums + one float each */
size_t mirec_gram_static_max_bytes(void);
int mirec_gram_static(const float* Y, int64_t n, int f, float reg, float* G, float* y_absmax, mirec_stream_t stream);

/* ------------------------------------------------------------------------
 * ALS half-sweep (implicit `_least_squares_cg`, cg_steps = 3 by default):
 * for every row u in [row_begin, row_end) of the CSR matrix, warm-started from
 * X[u]: build A_u = G + sum_k (|c_k|-1) y_k y_k^T and b_u = sum_{c_k>0} c_k y_k
 * from the gathered rows y_k = Y[indices[k]], run `cg_steps` conjugate-gradient
 * steps, store X[u].  Rows without entries are set to zero.
 * Reference call sites: als/als.py:174, recommenders/als_recommender.py:136.
 *
 * Two kernels share the rows.  Rows with at most `short_max` stored entries
 * (empty rows included) are solved matrix-free, one wavefront per row, re-reading
 * the gathered rows once per CG step (on chip for the first few of them,
 * mirec_als_short_cache_rows(f)): the right form while the gathered table stays
 * cache-resident.  Longer rows build A_u explicitly on the matrix cores; a
 * plan splits them into chunks of `chunk_nnz` stored entries so that heavy
 * rows spread over many workgroups.  The plan depends only on `indptr` and can
 * be reused for every iteration of a fit.
 *   nnz         stored entries in rows [row_begin, row_end)
 *   y_rows      rows of Y; every stored index is below it (0 = not known)
 *   plan        device buffer of mirec_als_plan_bytes(nrows, nnz, chunk_nnz)
 *   chunk_nnz   in: rows with more stored entries than this are split (into at
 *               most 32 equal chunks); 0 = choose from nnz; out: the value used
 *   short_max   >= 0; the same value must be given to the sweep (a negative
 *               value there skips the matrix-free kernel: for profiling only)
 *   host_counts out: {number of chunks, number of split rows, number of
 *               partial-result slots, number of rows left to the matrix-free
 *               kernel}
 * mirec_als_plan_build synchronises `stream` once (to read the counts) and
 * leaves the plan's last kernel enqueued on it.  The sweep takes the plan,
 * `chunk_nnz` and `host_counts` exactly as the build left them and a workspace
 * of mirec_als_sweep_workspace_bytes(host_counts, f) bytes (needed whenever the
 * plan has chunks).
 *   G           must be the Gram of Y plus reg * I (mirec_gram)
 *   y_absmax    one device float, the largest |entry| of Y as mirec_gram returns it,
 *               or null.  The explicit-Gram kernel multiplies on the matrix cores:
 *               every gathered row, scaled by sqrt(||c|-1|) and by one power of two
 *               per launch, enters as two fp16 planes (22 significant bits where
 *               fp32 has 24; sums and solves are fp32), and that power of two puts
 *               y_absmax * sqrt(conf_max - 1) just below the fp16 maximum.  Null:
 *               sqrt(max G_jj) stands in -- a bound of every |entry| that is up to the
 *               square root of Y's row count too large, which costs no accuracy
 *               until the confidences spread over more than about 2^10.  An entry
 *               of Y beyond the value given (a y_absmax or G that belongs to another
 *               matrix) overflows its operand and the row comes out non-finite.
 *   conf_max    the largest |c| over the stored entries -- of the whole matrix,
 *               so that every shard of a multi-GPU fit scales alike; an upper
 *               bound does; 0 = not known: the sweep then finds it with one pass
 *               over the values of its rows
 *   cg_steps    >= 0: that many conjugate-gradient steps from the warm start;
 *               MIREC_SOLVE_EXACT: solve A_u x = b_u exactly instead (implicit's
 *               use_cg=False path, `_least_squares`: LAPACK posv) by an fp32
 *               Cholesky factorisation in the explicit-Gram kernels; the plan must
 *               then be built with short_max = 0, so that every row with entries
 *               takes that kernel.  A row whose A_u is not positive definite
 *               (possible only with confidences below 1) comes out as NaN; the
 *               library's LU fallback for that case is not reproduced.
 * A sweep over an empty row range (row_begin == row_end: a shard that owns no
 * rows) returns MIREC_OK without reading any other argument. */
#define MIREC_SOLVE_EXACT (-1)
int mirec_als_short_cache_rows(int f);
size_t mirec_als_plan_bytes(int64_t nrows, int64_t nnz, int32_t chunk_nnz);
int mirec_als_plan_build(const int64_t* indptr, int64_t row_begin, int64_t row_end, int64_t nnz, int f,
                         void* plan, size_t plan_bytes, int32_t* chunk_nnz, int32_t short_max,
                         int64_t host_counts[4], mirec_stream_t stream);
size_t mirec_als_sweep_workspace_bytes(const int64_t host_counts[4], int f);
int mirec_als_half_sweep(const int64_t* indptr, const int32_t* indices, const float* data, int64_t row_begin,
                         int64_t row_end, int64_t nnz, float* X, const float* Y, int64_t y_rows, const float* G,
                         const float* y_absmax, int f, int cg_steps, const void* plan, int32_t chunk_nnz,
                         int32_t short_max, const int64_t host_counts[4], float conf_max, void* ws, size_t ws_bytes,
                         mirec_stream_t stream);

/* ------------------------------------------------------------------------
 * Interaction-matrix build (`build_matrix`, als/als.py:107-123, after the ids
 * have been mapped to indices on the host): duplicate (user, item) events
 * collapse to one stored entry of value 1.0 (the groupby at :118 and the ones at
 * :121); rows come out with ascending column indices.  `indices` / `data` must
 * hold n_events entries; the number of unique pairs is written to the device
 * scalar `nnz_out`, and `item_counts` (optional, n_items ints) receives the
 * number of users per item -- the column sums behind `most_popular_movies`
 * (als/als.py:175-176, recommenders/als_recommender.py:138-139). */
size_t mirec_events_to_csr_workspace_bytes(int64_t n_events, int64_t n_users);
int mirec_events_to_csr(const int32_t* user_idx, const int32_t* item_idx, int64_t n_events, int64_t n_users,
                        int64_t n_items, int64_t* indptr, int32_t* indices, float* data, int32_t* item_counts,
                        int64_t* nnz_out, void* ws, size_t ws_bytes, mirec_stream_t stream);

/* ------------------------------------------------------------------------
 * CSR transpose (`Ciu = Cui.T.tocsr()` inside implicit's fit; same call
 * sites).  Output rows keep ascending column order.  `ws` must hold
 * mirec_csr_transpose_workspace_bytes(rows, cols, nnz) bytes. */
size_t mirec_csr_transpose_workspace_bytes(int64_t rows, int64_t cols, int64_t nnz);
int mirec_csr_transpose(const int64_t* indptr, const int32_t* indices, const float* data, int64_t rows,
                        int64_t cols, int64_t nnz, int64_t* t_indptr, int32_t* t_indices, float* t_data,
                        void* ws, size_t ws_bytes, mirec_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused score + top-K (implicit `recommend` / `similar_items` -> topk):
 * for each of the nq query rows, score every item with the fp32 chain
 * s = fma(q[f-1], y[f-1], ... fma(q[0], y[0], +0)), optionally divide by
 * y_norm[item] and then q_norm[query], replace the entries listed in the
 * query's mask row by -FLT_MAX, and keep the k best under the order
 * (score descending, item index ascending).  The score matrix is never
 * written to memory.  out_ids int32[nq,k], out_scores fp32[nq,k]; slots past
 * the item count get id -1 and score -FLT_MAX.  k <= 224.
 * Reference call sites: recommenders/als_recommender.py:181-186 (recommend,
 * filter_already_liked_items=True), :237 (similar_items),
 * hybrid_docker/inference.py:197 (argsort top-k, via mirec_lightfm_*).
 * mask_indptr may be null (no filter); it is indexed by query row, and every
 * mask row must list its items in ascending order.
 * ws (device memory, 256-byte aligned, ws_bytes >= mirec_topk_workspace_bytes)
 * is optional scratch: with it, plain scoring (no norms) at any f that is a
 * multiple of 16 up to 128 and k <= 24 runs the selection on bf16 matrix-core scores with an error
 * margin and re-scores the survivors with the fp32 chain above, so the output
 * is the same bit for bit and arrives several times sooner (ten times with
 * mirec_topk_ordered at 1e6 x 27e3, f = 128).  ws may be null (or the size
 * query 0): the fp32 kernel then does all the work.  The contents of ws mean
 * nothing to the caller before or after the call; two calls that overlap in
 * time need a workspace each. */
size_t mirec_topk_workspace_bytes(int64_t nq, int64_t ni, int f, int k);
int mirec_topk(const float* Q, int64_t nq, const float* Y, int64_t ni, int f, const int64_t* mask_indptr,
               const int32_t* mask_indices, const float* y_norm, const float* q_norm, int k, int32_t* out_ids,
               float* out_scores, void* ws, size_t ws_bytes, mirec_stream_t stream);

/* The same contract for a handful of queries -- the reference's serving calls score ONE
 * user (recommenders/als_recommender.py:181-186), one item (:237) or one user
 * representation (hybrid_docker/inference.py:156-165,197) per request.  mirec_topk gives
 * a workgroup 128 or 256 queries and walks it through all items, which leaves the chip idle
 * below a few thousand queries; this entry point splits the items over the workgroups
 * instead (256 per workgroup, each lane the fp32 chain of one item), ranks every tile's
 * scores and merges the tiles' k best per query.  Arguments and result bits as mirec_topk
 * (one difference: a NaN score is never listed, and -0 counts as +0), any f that is a
 * multiple of 8.  ws (device memory, 16-byte aligned) must hold
 * mirec_topk_small_workspace_bytes(nq, ni, k) = 8 * nq * k * ceil(ni / 256) bytes; its
 * contents mean nothing before or after the call.  nq <= MIREC_TOPK_SMALL_MAX_QUERIES.
 * Pointers may address rows inside larger resident arrays (Q = X + u * f,
 * mask_indptr = indptr + u, q_norm = norms + u), so a request uploads nothing. */
#define MIREC_TOPK_SMALL_MAX_QUERIES 65536
size_t mirec_topk_small_workspace_bytes(int64_t nq, int64_t ni, int k);
int mirec_topk_small(const float* Q, int64_t nq, const float* Y, int64_t ni, int f, const int64_t* mask_indptr,
                     const int32_t* mask_indices, const float* y_norm, const float* q_norm, int k,
                     int32_t* out_ids, float* out_scores, void* ws, size_t ws_bytes, mirec_stream_t stream);

/* mirec_topk with the items swept in a given order.  Results are those of mirec_topk, bit for
 * bit, whatever the order: the order only decides how much work the all-users sweep has --
 * least when the items most queries score highest come first, which an item's number of
 * interactions predicts well (a wavefront of 32 queries then finds nothing to look at in about
 * nineteen item tiles out of twenty, against one in two, and skips two thirds of their
 * products).  The all-users pass behind `model.recommend` over every user of `train_matrix`
 * (recommenders/als_recommender.py:181-186 in a loop; als/als.py:177-185 for evaluation).
 *   item_order   int32[ni]: position -> item, a permutation of 0 .. ni-1
 *   mask_blocks  with a filter: int32 pairs (position, row set), one list per block of 32 filter
 *                rows -- the positions any row of the block holds, ascending, bit r of the row set
 *                = row 32 b + r holds it; the list of block b starts at pair mask_indptr[32 b]
 *   block_len    int32 per block: pairs in its list
 * mask_indptr / mask_indices stay the filter in item numbering (queries the sweep hands to the
 * fp32 kernel use it).  Query 0 must be the first row of a block: to score rows [a, b) of the
 * matrix pass mask_indptr + a and block_len + a / 32 with a % 32 == 0.
 * mirec_topk_order_build fills all three from the filter matrix, once per matrix and not per
 * model: items by descending number of stored entries, ties by ascending index; mask_blocks has
 * room for nnz pairs (8 nnz bytes, 8-byte aligned), block_len for ceil(nrows / 32) entries; it
 * needs mirec_topk_order_workspace_bytes(ni) of scratch and ni <= MIREC_TOPK_ORDER_MAX_ITEMS.
 * Without a filter any permutation may be given (e.g. by descending norm) and the two arrays are
 * NULL.  Entries of item_order outside 0 .. ni-1 are clamped, never dereferenced; an item_order
 * that is not a permutation, or lists that do not belong to it, give unspecified results.  Where
 * the sweep kernel does not run (no workspace, f not a multiple of 16 or above 128, k > 24,
 * fewer than 64 items) the order is ignored. */
#define MIREC_TOPK_ORDER_MAX_ITEMS 262144
size_t mirec_topk_order_workspace_bytes(int64_t ni);
int mirec_topk_order_build(const int64_t* mask_indptr, const int32_t* mask_indices, int64_t nrows, int64_t nnz,
                           int64_t ni, int32_t* item_order, int32_t* mask_blocks, int32_t* block_len, void* ws,
                           size_t ws_bytes, mirec_stream_t stream);
int mirec_topk_ordered(const float* Q, int64_t nq, const float* Y, int64_t ni, int f, const int64_t* mask_indptr,
                       const int32_t* mask_indices, const int32_t* item_order, const int32_t* mask_blocks,
                       const int32_t* block_len, int k, int32_t* out_ids, float* out_scores, void* ws,
                       size_t ws_bytes, mirec_stream_t stream);

/* ------------------------------------------------------------------------
 * LightFM representations (lightfm `compute_representation`, reached from
 * `model.predict` at hybrid_docker/inference.py:159-165,188-194):
 *   out[r, emb_col + j] = sum_k (w_k * scale) * E[indices[k], j]   (j < nc)
 *   out[r, bias_col]    = sum_k (w_k * scale) * bias[indices[k]]
 *   out[r, one_col]     = 1                      (skipped when one_col < 0)
 * accumulated in stored order with one fma per entry; other columns of `out`
 * (row stride `ld`) are left untouched.  With (bias_col, one_col, emb_col) =
 * (0, 1, 2) for users and (1, 0, 2) for items, mirec_topk over the two outputs
 * evaluates LightFM's score  (b_u + b_i) + sum_j u_j * i_j  in LightFM's order. */
int mirec_lightfm_repr(const int64_t* indptr, const int32_t* indices, const float* weights, int64_t rows,
                       const float* E, const float* bias, float scale, int nc, float* out, int ld, int emb_col,
                       int bias_col, int one_col, mirec_stream_t stream);

/* LightFM.predict for explicit pairs (hybrid_docker/inference.py:159-165):
 * urepr/irepr are [*, nc+1] with the bias in column nc. */
int mirec_lightfm_predict(const float* urepr, const float* irepr, const int32_t* user_ids,
                          const int32_t* item_ids, int64_t n, int nc, float* out, mirec_stream_t stream);

/* lightfm `predict_ranks`, the kernel under `auc_score`, `precision_at_k` and
 * `recall_at_k` as called at hybrid_docker/utils.py:111-155: for every stored
 * entry (u, i) of the test matrix (CSR, int64 row pointers), ranks[entry] = the
 * number of items j != i that are not stored in the user's train row and score
 * >= score(u, i), as a float.  urepr / irepr are [*, nc+1] with the bias in
 * column nc; scores use the chain of mirec_lightfm_predict.  train_indptr may
 * be null (nothing excluded). */
int mirec_lightfm_ranks(const float* urepr, const float* irepr, int64_t nu, int64_t ni, int nc,
                        const int64_t* test_indptr, const int32_t* test_indices, const int64_t* train_indptr,
                        const int32_t* train_indices, float* ranks, mirec_stream_t stream);

/* One epoch of LightFM WARP training with the adagrad schedule
 * (`model.fit_partial(train, user_features=, item_features=, epochs=1,
 * num_threads=4)` at hybrid_docker/utils.py:100-107; lightfm 1.17 `fit_warp`).
 * item / user features and the positives (the train interactions, rows sorted)
 * are CSR with int64 row pointers; the examples are (user_ids[e], item_ids[e])
 * with y[e] > 0, visited in `order`; sample_weight may be null (ones).
 * Parameters and their squared-gradient accumulators (initially one) are updated
 * in place; no_components <= 63.  Examples run concurrently and update shared
 * parameters without locks, as the library's OpenMP threads do -- in phases:
 * one workgroup at first, then as many examples side by side as 1/256 of those
 * the call has already worked through (more at the start, where every
 * accumulator is one, keeps the model from learning).  examples_seen: how many
 * examples the model has worked through in earlier calls (0 for a fresh model,
 * or when not known: the call then starts again at one workgroup); they count
 * like the call's own.  serial != 0 runs the examples one after the other in
 * `order` (the sequential statement, for checking).
 * Negatives are drawn from a counter-based generator keyed by (seed, example).
 * ws: device scratch of mirec_lightfm_warp_workspace_bytes(max_sampled) bytes.
 * The call synchronises the stream once (it uploads a small table).
 * n_examples == 0 is a valid epoch that changes nothing. */
size_t mirec_lightfm_warp_workspace_bytes(int max_sampled);
int mirec_lightfm_warp_epoch(
    const int64_t* if_indptr, const int32_t* if_indices, const float* if_data, int64_t n_items, int64_t n_item_features,
    const int64_t* uf_indptr, const int32_t* uf_indices, const float* uf_data, int64_t n_user_features,
    const int64_t* pos_indptr, const int32_t* pos_indices, const int32_t* user_ids, const int32_t* item_ids,
    const float* y, const float* sample_weight, const int32_t* order, int64_t n_examples, float* item_emb,
    float* item_emb_grad, float* item_bias, float* item_bias_grad, float* user_emb, float* user_emb_grad,
    float* user_bias, float* user_bias_grad, int nc, double learning_rate, double item_alpha, double user_alpha,
    int max_sampled, uint32_t seed, int serial, int64_t examples_seen, void* ws, size_t ws_bytes, mirec_stream_t stream);

/* ------------------------------------------------------------------------
 * Held-out ranking evaluation over the top-k output (the evaluation cell of
 * notebooks/baseline_als.ipynb, cells 66ca94ea / fd92678d; LightFM's
 * precision_at_k / recall_at_k as called at hybrid_docker/utils.py:111-155).
 * rec_ids is [n_users, k] (negative = no recommendation), the held-out matrix a
 * CSR with ascending column indices per row.  Per user: hits = |set(recs) n
 * held-out|, n_rec = recommendations that are not negative.  The averages are
 * formed by the caller from these counts. */
int mirec_ranking_hits(const int32_t* rec_ids, int64_t n_users, int k, const int64_t* held_indptr,
                       const int32_t* held_indices, int32_t* hits, int32_t* n_rec, mirec_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MIREC_H */
