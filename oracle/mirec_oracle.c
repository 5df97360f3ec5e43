/*
 * CPU oracle for the recommender hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the shipped package may import, link or call this file: it is the
 * checker used by tests/, by __graft_entry__.smoke() and by bench.py's
 * cpu_baseline leg, never the thing measured as the product or shipped.
 *
 * PARITY UNPINNED.  The arithmetic of this path does not live in the reference
 * tree: the reference calls implicit==0.7.2 and lightfm==1.17
 * (requirements.txt:2,4), neither of which is vendored, installed or
 * installable here, and the reference's own tests replace both with mocks
 * (src/shrekommender_system/tests/test_als_recommender.py:30-32,
 * src/shrekommender_system/tests/test_pipeline.py:24), so no golden vector of
 * the reference pins these results.  This file restates the published
 * algorithms of those two releases (Hu/Koren/Volinsky implicit ALS with the
 * Takacs/Pilaszy/Tikk conjugate-gradient solver; LightFM's linear-in-features
 * predict) and anchors them on the reference's call sites:
 *   als/als.py:166-177                         ctor + fit(matrix * alpha)
 *   src/shrekommender_system/recommenders/als_recommender.py:117-140  ctor + fit
 *   src/shrekommender_system/recommenders/als_recommender.py:181-186  recommend
 *   src/shrekommender_system/recommenders/als_recommender.py:237      similar_items
 *   hybrid_docker/inference.py:159-165,188-194,197                    predict + top-k
 *
 * All arithmetic is fp32, as in implicit's float32 path.  Dots and symv's are
 * written as plain loops; the scoring functions use explicit fmaf chains in
 * ascending k so that a device that evaluates the same chain (gfx950's f32
 * MFMA does, bit for bit) can be compared for exact equality.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* YtY = Y^T Y + reg*I.  implicit forms this with one numpy dot (a blocked
 * sgemm) per half-sweep; a blocked fp32 accumulation is restated here: 256-row
 * blocks summed in fp32, block sums combined in fp64 and rounded once. */
void oracle_gram(const float* Y, int64_t n, int f, float reg, float* out) {
    double* acc = (double*)calloc((size_t)f * f, sizeof(double));
    float* blk = (float*)malloc((size_t)f * f * sizeof(float));
    for (int64_t r0 = 0; r0 < n; r0 += 256) {
        int64_t r1 = r0 + 256 < n ? r0 + 256 : n;
        memset(blk, 0, (size_t)f * f * sizeof(float));
        for (int64_t r = r0; r < r1; ++r) {
            const float* y = Y + r * f;
            for (int i = 0; i < f; ++i) {
                float yi = y[i];
                float* row = blk + (size_t)i * f;
                for (int j = 0; j < f; ++j) row[j] += yi * y[j];
            }
        }
        for (int i = 0; i < f * f; ++i) acc[i] += (double)blk[i];
    }
    for (int i = 0; i < f; ++i)
        for (int j = 0; j < f; ++j)
            out[(size_t)i * f + j] = (float)(acc[(size_t)i * f + j] + (i == j ? (double)reg : 0.0));
    free(acc);
    free(blk);
}

static inline float dotf(const float* a, const float* b, int n) {
    float s = 0.f;
#pragma omp simd reduction(+ : s)
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

static inline void axpyf(float a, const float* x, float* y, int n) {
#pragma omp simd
    for (int i = 0; i < n; ++i) y[i] += a * x[i];
}

/* out = alpha * A x for a dense symmetric f x f matrix. */
static inline void symvf(float alpha, const float* A, const float* x, float* out, int f) {
    for (int i = 0; i < f; ++i) out[i] = alpha * dotf(A + (size_t)i * f, x, f);
}

/*
 * One half-sweep of implicit's CG solver (cpu/_als.pyx `_least_squares_cg`):
 * for every row u of the CSR matrix, warm-started from X[u], run `cg_steps`
 * conjugate-gradient steps on (YtY + Y^T (C_u - I) Y) x = Y^T C_u p(u) without
 * forming the matrix.  Rows without entries are zeroed.  Negative stored
 * values mean "confidence |c|, preference 0".  X is updated in place.
 */
void oracle_als_half_sweep(const int64_t* indptr, const int32_t* indices, const float* data,
                           int64_t row_begin, int64_t row_end, float* X, const float* Y,
                           const float* YtY, int f, int cg_steps) {
#pragma omp parallel
    {
        float* Ap = (float*)malloc(sizeof(float) * f);
        float* p = (float*)malloc(sizeof(float) * f);
        float* r = (float*)malloc(sizeof(float) * f);
#pragma omp for schedule(dynamic, 8)
        for (int64_t u = row_begin; u < row_end; ++u) {
            float* x = X + u * f;
            int64_t s = indptr[u], e = indptr[u + 1];
            if (s == e) {
                memset(x, 0, sizeof(float) * f);
                continue;
            }
            symvf(-1.0f, YtY, x, r, f);
            for (int64_t k = s; k < e; ++k) {
                const float* y = Y + (int64_t)indices[k] * f;
                float confidence = data[k];
                float temp;
                if (confidence > 0) {
                    temp = confidence;
                } else {
                    temp = 0;
                    confidence = -1 * confidence;
                }
                temp = temp - (confidence - 1) * dotf(y, x, f);
                axpyf(temp, y, r, f);
            }
            memcpy(p, r, sizeof(float) * f);
            float rsold = dotf(r, r, f);
            if (rsold < 1e-20f) continue;
            for (int it = 0; it < cg_steps; ++it) {
                symvf(1.0f, YtY, p, Ap, f);
                for (int64_t k = s; k < e; ++k) {
                    const float* y = Y + (int64_t)indices[k] * f;
                    float confidence = data[k];
                    if (confidence < 0) confidence = -1 * confidence;
                    float temp = (confidence - 1) * dotf(y, p, f);
                    axpyf(temp, y, Ap, f);
                }
                float alpha = rsold / dotf(p, Ap, f);
                axpyf(alpha, p, x, f);
                axpyf(-alpha, Ap, r, f);
                float rsnew = dotf(r, r, f);
                if (rsnew < 1e-20f) break;
                float beta = rsnew / rsold;
                for (int i = 0; i < f; ++i) p[i] = r[i] + beta * p[i];
                rsold = rsnew;
            }
        }
        free(Ap);
        free(p);
        free(r);
    }
}

/* Exact per-row solve (implicit's use_cg=False path, `_least_squares`):
 * (YtY + Y^T (C_u - I) Y) x = Y^T C_u p(u) by Cholesky, fp32 storage, the
 * factorisation itself carried in fp64 so that it can serve as a converged
 * reference for the CG iterates. */
void oracle_als_half_sweep_exact(const int64_t* indptr, const int32_t* indices, const float* data,
                                 int64_t row_begin, int64_t row_end, float* X, const float* Y,
                                 const float* YtY, int f) {
#pragma omp parallel
    {
        double* A = (double*)malloc(sizeof(double) * f * f);
        double* b = (double*)malloc(sizeof(double) * f);
#pragma omp for schedule(dynamic, 8)
        for (int64_t u = row_begin; u < row_end; ++u) {
            float* x = X + u * f;
            int64_t s = indptr[u], e = indptr[u + 1];
            if (s == e) {
                memset(x, 0, sizeof(float) * f);
                continue;
            }
            for (int i = 0; i < f * f; ++i) A[i] = YtY[i];
            memset(b, 0, sizeof(double) * f);
            for (int64_t k = s; k < e; ++k) {
                const float* y = Y + (int64_t)indices[k] * f;
                double c = data[k];
                double pos = c > 0 ? c : 0.0;
                if (c < 0) c = -c;
                for (int i = 0; i < f; ++i) {
                    b[i] += pos * y[i];
                    double w = (c - 1.0) * y[i];
                    for (int j = 0; j < f; ++j) A[(size_t)i * f + j] += w * y[j];
                }
            }
            /* in-place Cholesky A = L L^T (lower) */
            for (int j = 0; j < f; ++j) {
                double d = A[(size_t)j * f + j];
                for (int k = 0; k < j; ++k) d -= A[(size_t)j * f + k] * A[(size_t)j * f + k];
                d = sqrt(d);
                A[(size_t)j * f + j] = d;
                for (int i = j + 1; i < f; ++i) {
                    double v = A[(size_t)i * f + j];
                    for (int k = 0; k < j; ++k) v -= A[(size_t)i * f + k] * A[(size_t)j * f + k];
                    A[(size_t)i * f + j] = v / d;
                }
            }
            for (int i = 0; i < f; ++i) {
                double v = b[i];
                for (int k = 0; k < i; ++k) v -= A[(size_t)i * f + k] * b[k];
                b[i] = v / A[(size_t)i * f + i];
            }
            for (int i = f - 1; i >= 0; --i) {
                double v = b[i];
                for (int k = i + 1; k < f; ++k) v -= A[(size_t)k * f + i] * b[k];
                b[i] = v / A[(size_t)i * f + i];
            }
            for (int i = 0; i < f; ++i) x[i] = (float)b[i];
        }
        free(A);
        free(b);
    }
}

/* Implicit-feedback ALS loss in fp64 (Hu/Koren/Volinsky eq. 3): used only for
 * the "loss does not increase per half-sweep" invariant. */
double oracle_als_loss(const int64_t* indptr, const int32_t* indices, const float* data, int64_t nu,
                       int64_t ni, const float* X, const float* Y, int f, float reg) {
    double* YtY = (double*)calloc((size_t)f * f, sizeof(double));
    for (int64_t i = 0; i < ni; ++i)
        for (int a = 0; a < f; ++a)
            for (int b = 0; b < f; ++b) YtY[(size_t)a * f + b] += (double)Y[i * f + a] * Y[i * f + b];
    double loss = 0.0;
    for (int64_t u = 0; u < nu; ++u) {
        const float* x = X + u * f;
        /* sum over all items of (0 - x.y)^2 = x^T YtY x, then correct the stored ones */
        double q = 0.0;
        for (int a = 0; a < f; ++a)
            for (int b = 0; b < f; ++b) q += x[a] * YtY[(size_t)a * f + b] * x[b];
        loss += q;
        for (int64_t k = indptr[u]; k < indptr[u + 1]; ++k) {
            const float* y = Y + (int64_t)indices[k] * f;
            double c = data[k], pref = c > 0 ? 1.0 : 0.0;
            if (c < 0) c = -c;
            double s = 0.0;
            for (int a = 0; a < f; ++a) s += (double)x[a] * y[a];
            loss += c * (pref - s) * (pref - s) - s * s;
        }
        double n2 = 0.0;
        for (int a = 0; a < f; ++a) n2 += (double)x[a] * x[a];
        loss += reg * n2;
    }
    for (int64_t i = 0; i < ni; ++i) {
        double n2 = 0.0;
        for (int a = 0; a < f; ++a) n2 += (double)Y[i * f + a] * Y[i * f + a];
        loss += reg * n2;
    }
    free(YtY);
    return loss;
}

/* score(q, y) = fmaf chain over ascending k starting from +0. */
static inline float score_chain(const float* q, const float* y, int f) {
    float s = 0.f;
    for (int k = 0; k < f; ++k) s = fmaf(q[k], y[k], s);
    return s;
}

/* Strict ordering used for every top-K in this build: higher score first,
 * equal scores by ascending item index. */
static inline int better(float sa, int32_t ia, float sb, int32_t ib) {
    return sa > sb || (sa == sb && ia < ib);
}

static void select_topk(const float* scores, int64_t ni, int k, int32_t* out_ids, float* out_scores) {
    int filled = 0;
    for (int64_t i = 0; i < ni; ++i) {
        float s = scores[i];
        if (filled == k && !better(s, (int32_t)i, out_scores[k - 1], out_ids[k - 1])) continue;
        int pos = filled < k ? filled : k - 1;
        while (pos > 0 && better(s, (int32_t)i, out_scores[pos - 1], out_ids[pos - 1])) {
            out_scores[pos] = out_scores[pos - 1];
            out_ids[pos] = out_ids[pos - 1];
            --pos;
        }
        out_scores[pos] = s;
        out_ids[pos] = (int32_t)i;
        if (filled < k) ++filled;
    }
    for (int j = filled; j < k; ++j) {
        out_ids[j] = -1;
        out_scores[j] = -FLT_MAX;
    }
}

/*
 * Batch recommend (implicit's MatrixFactorizationBase.recommend + topk, as
 * called at als_recommender.py:181-186): scores = Q . Y^T, entries of the
 * query's own CSR row set to -FLT_MAX when filtering, top-k per query.  When
 * q_norm/y_norm are given every score is divided by y_norm[i] and then by
 * q_norm[q] (similar_items).  Slots beyond the item count get id -1.
 */
void oracle_topk(const float* Q, int64_t nq, const float* Y, int64_t ni, int f,
                 const int64_t* mask_indptr, const int32_t* mask_indices, int k,
                 const float* y_norm, const float* q_norm, int32_t* out_ids, float* out_scores) {
#pragma omp parallel
    {
        float* scores = (float*)malloc(sizeof(float) * (size_t)(ni > 0 ? ni : 1));
#pragma omp for schedule(dynamic, 8)
        for (int64_t q = 0; q < nq; ++q) {
            const float* qv = Q + q * f;
            for (int64_t i = 0; i < ni; ++i) {
                float s = score_chain(qv, Y + i * f, f);
                if (y_norm) s = s / y_norm[i];
                if (q_norm) s = s / q_norm[q];
                scores[i] = s;
            }
            if (mask_indptr)
                for (int64_t m = mask_indptr[q]; m < mask_indptr[q + 1]; ++m) scores[mask_indices[m]] = -FLT_MAX;
            select_topk(scores, ni, k, out_ids + q * k, out_scores + q * k);
        }
        free(scores);
    }
}

/* Dense score matrix with the same chain (small cases only). */
void oracle_scores(const float* Q, int64_t nq, const float* Y, int64_t ni, int f, float* out) {
    for (int64_t q = 0; q < nq; ++q)
        for (int64_t i = 0; i < ni; ++i) out[q * ni + i] = score_chain(Q + q * f, Y + i * f, f);
}

/*
 * LightFM representation (lightfm/_lightfm_fast.pyx `compute_representation`):
 * out[r, j] = sum over the row's features of (weight*scale) * E[feature, j] for
 * j < nc, and out[r, nc] the same sum over the feature biases; entries are
 * accumulated in stored order, one fmaf per entry.
 */
void oracle_lightfm_repr(const int64_t* indptr, const int32_t* indices, const float* weights,
                         int64_t rows, const float* E, const float* bias, float scale, int nc, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        float* o = out + r * (nc + 1);
        for (int j = 0; j <= nc; ++j) o[j] = 0.f;
        for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k) {
            int32_t feat = indices[k];
            float w = weights[k] * scale;
            const float* e = E + (int64_t)feat * nc;
            for (int j = 0; j < nc; ++j) o[j] = fmaf(w, e[j], o[j]);
            o[nc] = fmaf(w, bias[feat], o[nc]);
        }
    }
}

/* lightfm `compute_prediction_from_repr`: bias_u + bias_i, then the component
 * products added in ascending order. */
static inline float lightfm_score(const float* u, const float* it, int nc) {
    float s = u[nc] + it[nc];
    for (int j = 0; j < nc; ++j) s = fmaf(u[j], it[j], s);
    return s;
}

/* LightFM.predict for explicit (user, item) pairs, as called at
 * hybrid_docker/inference.py:159-165. */
void oracle_lightfm_predict(const float* urepr, const float* irepr, const int32_t* user_ids,
                            const int32_t* item_ids, int64_t n, int nc, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p)
        out[p] = lightfm_score(urepr + (int64_t)user_ids[p] * (nc + 1), irepr + (int64_t)item_ids[p] * (nc + 1), nc);
}

/* All items for each user, top-k as `scores.argsort()[::-1][:k]` at
 * hybrid_docker/inference.py:197 (no liked-item filter), ties by ascending
 * item index. */
void oracle_lightfm_topk(const float* urepr, int64_t nu, const float* irepr, int64_t ni, int nc, int k,
                         int32_t* out_ids, float* out_scores) {
#pragma omp parallel
    {
        float* scores = (float*)malloc(sizeof(float) * (size_t)(ni > 0 ? ni : 1));
#pragma omp for schedule(dynamic, 8)
        for (int64_t u = 0; u < nu; ++u) {
            for (int64_t i = 0; i < ni; ++i) scores[i] = lightfm_score(urepr + u * (nc + 1), irepr + i * (nc + 1), nc);
            select_topk(scores, ni, k, out_ids + u * k, out_scores + u * k);
        }
        free(scores);
    }
}

/* lightfm `predict_ranks` (lightfm/_lightfm_fast.pyx.template): for each user with test interactions, the
 * predictions of its test items first, then every item that is not a train positive of the user adds one to
 * the rank of each test interaction it is not itself and whose prediction it reaches (>=). */
void oracle_lightfm_ranks(const float* urepr, const float* irepr, int64_t nu, int64_t ni, int nc,
                          const int64_t* test_indptr, const int32_t* test_indices, const int64_t* train_indptr,
                          const int32_t* train_indices, float* ranks) {
#pragma omp parallel for schedule(dynamic, 10)
    for (int64_t u = 0; u < nu; ++u) {
        int64_t t0 = test_indptr[u], t1 = test_indptr[u + 1];
        if (t0 == t1) continue;
        const float* ur = urepr + u * (nc + 1);
        float* pred = (float*)malloc(sizeof(float) * (size_t)(t1 - t0));
        for (int64_t i = t0; i < t1; ++i) {
            pred[i - t0] = lightfm_score(ur, irepr + (int64_t)test_indices[i] * (nc + 1), nc);
            ranks[i] = 0.f;
        }
        int64_t r = train_indptr ? train_indptr[u] : 0, r1 = train_indptr ? train_indptr[u + 1] : 0;
        for (int64_t item = 0; item < ni; ++item) {
            /* in_positives: the train row is sorted, a cursor replaces the binary search */
            while (r < r1 && train_indices[r] < item) ++r;
            if (r < r1 && train_indices[r] == item) continue;
            float p = lightfm_score(ur, irepr + item * (nc + 1), nc);
            for (int64_t i = t0; i < t1; ++i)
                if (item != test_indices[i] && p >= pred[i - t0]) ranks[i] += 1.0f;
        }
        free(pred);
    }
}

static int flt_compare(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* lightfm `calculate_auc_from_rank`: per user, ranks sorted ascending, the i-th has i better positives above
 * it; auc = mean over positives of 1 - (rank - i)/num_negatives in float32; 0.5 when a class is missing.
 * rank_data is sorted in place, as in the library. */
void oracle_lightfm_auc_from_ranks(const int64_t* indptr, int64_t nu, int64_t ni, const int32_t* num_train_positives,
                                   float* rank_data, float* auc) {
    for (int64_t u = 0; u < nu; ++u) {
        int64_t a = indptr[u], b = indptr[u + 1];
        int64_t num_pos = b - a;
        int64_t num_neg = ni - (num_pos + num_train_positives[u]);
        auc[u] = 0.f;
        if (num_pos == 0 || num_neg == ni) {
            auc[u] = 0.5f;
            continue;
        }
        qsort(rank_data + a, (size_t)num_pos, sizeof(float), flt_compare);
        for (int64_t i = 0; i < num_pos; ++i) {
            float rank = rank_data[a + i] - (float)i;
            if (rank < 0.f) rank = 0.f;
            auc[u] += 1.0f - rank / (float)num_neg;
        }
        auc[u] /= (float)num_pos;
    }
}

/* ---------------------------------------------------------------------------------------------------------
 * LightFM WARP training, one epoch, sequential (lightfm 1.17 `fit_warp` / `warp_update` / `update_features` /
 * `update_biases` of lightfm/_lightfm_fast.pyx.template with the adagrad schedule, reached from
 * `model.fit_partial(..., epochs=1, num_threads=4)` at hybrid_docker/utils.py:100-107).  The library runs its
 * examples Hogwild over OpenMP threads and draws negatives with rand_r; this statement runs them in the given
 * order and draws negatives from a counter-based generator keyed by (seed, example), so that a run is a
 * function of its inputs.  Arithmetic types follow the Cython source: parameters float32, the update
 * expressions double.
 *
 * Per example (u, i+): representations with the lazy-regularisation scales, positive prediction; up to
 * max_sampled negatives j: skip when score(j) <= score(i+) - 1; skip (and count) when j is a positive of u; else
 * loss = weight * log(max(1, floor((n_items - 1) / sampled))) clipped at 10 and one adagrad step on the item
 * biases (i+: -loss, j: +loss), the item embeddings (i+: -loss*u_c, j: +loss*u_c) and the user embeddings
 * (loss * (j_c - i+_c)); the scales grow by (1 + alpha * mean learning rate).  `regularize` at the end of the
 * epoch divides the parameters by the scales and resets them.  (Representations multiply by the scale and
 * `regularize` divides by it: that is how the source is recalled; with the reference's item_alpha = 1e-6 and
 * user_alpha = 0, hybrid_docker/train.py:157, the scales stay within 1e-4 of one either way.)
 */
static inline uint32_t warp_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

static void warp_repr(const int64_t* indptr, const int32_t* indices, const float* data, int64_t row,
                      const float* emb, const float* bias, double scale, int nc, float* out) {
    for (int j = 0; j <= nc; ++j) out[j] = 0.f;
    for (int64_t k = indptr[row]; k < indptr[row + 1]; ++k) {
        int32_t f = indices[k];
        float w = (float)(data[k] * scale);
        for (int j = 0; j < nc; ++j) out[j] = fmaf(w, emb[(int64_t)f * nc + j], out[j]);
        out[nc] = fmaf(w, bias[f], out[nc]);
    }
}

static double warp_update_row(const int64_t* indptr, const int32_t* indices, const float* data, int64_t row,
                              float* param, float* grad, int stride, int col, double gradient, double lr, double alpha) {
    double sum_lr = 0.0;
    for (int64_t k = indptr[row]; k < indptr[row + 1]; ++k) {
        int64_t at = (int64_t)indices[k] * stride + col;
        double fw = data[k];
        double local = lr / sqrt((double)grad[at]);
        param[at] = (float)(param[at] - local * fw * gradient);
        grad[at] = (float)(grad[at] + (gradient * fw) * (gradient * fw));
        param[at] = (float)(param[at] * (1.0 + alpha * local));
        sum_lr += local;
    }
    return sum_lr;
}

void oracle_lightfm_warp_epoch(
    const int64_t* if_indptr, const int32_t* if_indices, const float* if_data, int64_t n_items,
    const int64_t* uf_indptr, const int32_t* uf_indices, const float* uf_data,
    const int64_t* pos_indptr, const int32_t* pos_indices,
    const int32_t* user_ids, const int32_t* item_ids, const float* y, const float* sample_weight,
    const int32_t* order, int64_t n_examples,
    float* item_emb, float* item_emb_grad, float* item_bias, float* item_bias_grad, int64_t n_item_features,
    float* user_emb, float* user_emb_grad, float* user_bias, float* user_bias_grad, int64_t n_user_features,
    int nc, double learning_rate, double item_alpha, double user_alpha, int max_sampled, uint32_t seed,
    double* scales /* [item_scale, user_scale] in/out */) {
    float* ur = (float*)malloc(sizeof(float) * 3 * (size_t)(nc + 1));
    float* pr = ur + nc + 1;
    float* nr = pr + nc + 1;
    double item_scale = scales[0], user_scale = scales[1];
    for (int64_t e = 0; e < n_examples; ++e) {
        int32_t row = order[e];
        if (!(y[row] > 0.f)) continue;
        int32_t u = user_ids[row], ip = item_ids[row];
        double weight = sample_weight ? sample_weight[row] : 1.0;
        warp_repr(uf_indptr, uf_indices, uf_data, u, user_emb, user_bias, user_scale, nc, ur);
        warp_repr(if_indptr, if_indices, if_data, ip, item_emb, item_bias, item_scale, nc, pr);
        float ppred = lightfm_score(ur, pr, nc);
        uint32_t rng = warp_hash(seed ^ warp_hash((uint32_t)row + 0x9e3779b9u));
        int sampled = 0;
        while (sampled < max_sampled) {
            ++sampled;
            rng = warp_hash(rng + 0x6d2b79f5u);
            int32_t neg = (int32_t)(rng % (uint32_t)n_items);
            warp_repr(if_indptr, if_indices, if_data, neg, item_emb, item_bias, item_scale, nc, nr);
            float npred = lightfm_score(ur, nr, nc);
            if (npred > ppred - 1.f) {
                int64_t a = pos_indptr[u], b = pos_indptr[u + 1];
                while (a < b) {
                    int64_t m = a + (b - a) / 2;
                    if (pos_indices[m] < neg) a = m + 1; else b = m;
                }
                if (a < pos_indptr[u + 1] && pos_indices[a] == neg) continue;
                double loss = weight * log(fmax(1.0, floor((double)(n_items - 1) / sampled)));
                if (loss > 10.0) loss = 10.0;
                double avg = 0.0;
                avg += warp_update_row(if_indptr, if_indices, if_data, ip, item_bias, item_bias_grad, 1, 0, -loss, learning_rate, item_alpha);
                avg += warp_update_row(if_indptr, if_indices, if_data, neg, item_bias, item_bias_grad, 1, 0, loss, learning_rate, item_alpha);
                for (int c = 0; c < nc; ++c) {
                    double uc = ur[c], pc = pr[c], ncp = nr[c];
                    avg += warp_update_row(if_indptr, if_indices, if_data, ip, item_emb, item_emb_grad, nc, c, -loss * uc, learning_rate, item_alpha);
                    avg += warp_update_row(if_indptr, if_indices, if_data, neg, item_emb, item_emb_grad, nc, c, loss * uc, learning_rate, item_alpha);
                    avg += warp_update_row(uf_indptr, uf_indices, uf_data, u, user_emb, user_emb_grad, nc, c, loss * (ncp - pc), learning_rate, user_alpha);
                }
                double denom = (double)(nc + 1) * (double)((uf_indptr[u + 1] - uf_indptr[u]) + (if_indptr[ip + 1] - if_indptr[ip]) +
                                                           (if_indptr[neg + 1] - if_indptr[neg]));
                avg /= denom;
                item_scale *= 1.0 + item_alpha * avg;
                user_scale *= 1.0 + user_alpha * avg;
                break;
            }
        }
    }
    /* regularize(): fold the scales into the parameters */
    for (int64_t i = 0; i < n_item_features; ++i) {
        for (int j = 0; j < nc; ++j) item_emb[i * nc + j] = (float)(item_emb[i * nc + j] / item_scale);
        item_bias[i] = (float)(item_bias[i] / item_scale);
    }
    for (int64_t i = 0; i < n_user_features; ++i) {
        for (int j = 0; j < nc; ++j) user_emb[i * nc + j] = (float)(user_emb[i * nc + j] / user_scale);
        user_bias[i] = (float)(user_bias[i] / user_scale);
    }
    scales[0] = 1.0;
    scales[1] = 1.0;
    free(ur);
}
