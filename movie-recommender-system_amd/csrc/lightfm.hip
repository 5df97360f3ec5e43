// LightFM inference pieces: feature-embedding segment-sum and pair scoring.
//
// The segment-sum is a short CSR gather (a handful of embedding rows per user
// or item), bandwidth-trivial next to the all-pairs scoring that follows; one
// thread owns one output column of one row and adds the row's entries in
// stored order, one fma each, so every run and every launch shape gives the
// same bits.
#include "common.h"

namespace mirec {

__global__ __launch_bounds__(256) void lightfm_repr_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ weights,
    int64_t rows, const float* __restrict__ E, const float* __restrict__ bias, float scale, int nc,
    float* __restrict__ out, int ld, int emb_col, int bias_col, int one_col) {
    const int cols = nc + 1;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows * cols) return;
    const int64_t r = gid / cols;
    const int j = (int)(gid % cols);
    float acc = 0.f;
    const int64_t s = indptr[r], e = indptr[r + 1];
    if (j < nc) {
        for (int64_t k = s; k < e; ++k) {
            const float w = weights[k] * scale;
            acc = fmaf(w, E[(int64_t)indices[k] * nc + j], acc);
        }
        out[r * ld + emb_col + j] = acc;
    } else {
        for (int64_t k = s; k < e; ++k) {
            const float w = weights[k] * scale;
            acc = fmaf(w, bias[indices[k]], acc);
        }
        out[r * ld + bias_col] = acc;
        if (one_col >= 0) out[r * ld + one_col] = 1.f;
    }
}

__global__ __launch_bounds__(256) void lightfm_predict_kernel(
    const float* __restrict__ urepr, const float* __restrict__ irepr, const int32_t* __restrict__ user_ids,
    const int32_t* __restrict__ item_ids, int64_t n, int nc, float* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float* u = urepr + (int64_t)user_ids[p] * (nc + 1);
    const float* it = irepr + (int64_t)item_ids[p] * (nc + 1);
    float s = u[nc] + it[nc];
    for (int j = 0; j < nc; ++j) s = fmaf(u[j], it[j], s);
    out[p] = s;
}

// Rank of every test interaction among the items that are not train positives of its user
// (lightfm `predict_ranks`): the number of such items, other than the interaction's own, whose score is >= its
// score.  One wavefront per user.  Its test positives are taken 64 at a time (lane p holds positive p); all
// items are scored 64 per step (lane = item) with the same fma chain as the predictions, and a step adds, per
// positive, the population count of the lanes that reach its score.  The user's train positives are scored once
// more at the end and taken off again, which is the same as having skipped them.  Counts are integers: the
// result does not depend on any order.
constexpr int RANK_WAVES = 4;

__device__ __forceinline__ float lightfm_score_dev(const float* __restrict__ u, const float* __restrict__ it, int nc) {
    float s = u[nc] + it[nc];
    for (int j = 0; j < nc; ++j) s = fmaf(u[j], it[j], s);
    return s;
}

__global__ __launch_bounds__(64 * RANK_WAVES) void lightfm_ranks_kernel(
    const float* __restrict__ urepr, const float* __restrict__ irepr, int64_t nu, int64_t ni, int nc,
    const int64_t* __restrict__ test_indptr, const int32_t* __restrict__ test_indices,
    const int64_t* __restrict__ train_indptr, const int32_t* __restrict__ train_indices,
    float* __restrict__ ranks) {
    extern __shared__ float rk_smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* urow = rk_smem + w * (nc + 1 + 128);  // user representation, then 64 predictions and 64 item ids
    float* ppred = urow + nc + 1;
    int* pid = reinterpret_cast<int*>(ppred + 64);
    const int64_t u = (int64_t)blockIdx.x * RANK_WAVES + w;
    if (u >= nu) return;
    const int64_t t0 = test_indptr[u], t1 = test_indptr[u + 1];
    if (t0 == t1) return;
    for (int j = lane; j <= nc; j += 64) urow[j] = urepr[u * (nc + 1) + j];
    __builtin_amdgcn_wave_barrier();
    const int64_t r0 = train_indptr ? train_indptr[u] : 0, r1 = train_indptr ? train_indptr[u + 1] : 0;
    for (int64_t c0 = t0; c0 < t1; c0 += 64) {
        const int np = (int)((t1 - c0) < 64 ? (t1 - c0) : 64);
        int my_id = -1;
        float my_pred = 0.f;
        if (lane < np) {
            my_id = test_indices[c0 + lane];
            my_pred = lightfm_score_dev(urow, irepr + (int64_t)my_id * (nc + 1), nc);
        }
        __builtin_amdgcn_wave_barrier();
        ppred[lane] = my_pred;
        pid[lane] = my_id;
        __builtin_amdgcn_wave_barrier();
        int cnt = 0;  // lane p: rank of positive p
        for (int64_t base = 0; base < ni; base += 64) {
            const int64_t item = base + lane;
            const bool ok = item < ni;
            const float s = ok ? lightfm_score_dev(urow, irepr + item * (nc + 1), nc) : 0.f;
            for (int p = 0; p < np; ++p) {
                const int pc = __builtin_popcountll(__ballot(ok && (int)item != pid[p] && s >= ppred[p]));
                cnt += (lane == p) ? pc : 0;
            }
        }
        for (int64_t tb = r0; tb < r1; tb += 64) {
            const bool ok = tb + lane < r1;
            const int item = ok ? train_indices[tb + lane] : 0;
            const float s = ok ? lightfm_score_dev(urow, irepr + (int64_t)item * (nc + 1), nc) : 0.f;
            for (int p = 0; p < np; ++p) {
                const int pc = __builtin_popcountll(__ballot(ok && item != pid[p] && s >= ppred[p]));
                cnt -= (lane == p) ? pc : 0;
            }
        }
        if (lane < np) ranks[c0 + lane] = (float)cnt;
    }
}

}  // namespace mirec

extern "C" int mirec_lightfm_ranks(const float* urepr, const float* irepr, int64_t nu, int64_t ni, int nc,
                                   const int64_t* test_indptr, const int32_t* test_indices,
                                   const int64_t* train_indptr, const int32_t* train_indices, float* ranks,
                                   mirec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MIREC_REQUIRE(nu >= 0 && ni >= 0 && nc >= 1, MIREC_E_INVALID, "mirec_lightfm_ranks: bad sizes");
    MIREC_REQUIRE(ni < ((int64_t)1 << 31), MIREC_E_UNSUPPORTED, "mirec_lightfm_ranks: more than 2^31 items");
    if (nu == 0) return MIREC_OK;
    MIREC_REQUIRE(urepr && irepr && test_indptr && ranks, MIREC_E_INVALID, "mirec_lightfm_ranks: null argument");
    MIREC_REQUIRE((train_indptr == nullptr) == (train_indices == nullptr) || train_indptr != nullptr, MIREC_E_INVALID,
                  "mirec_lightfm_ranks: train indices without row pointers");
    const size_t smem = (size_t)mirec::RANK_WAVES * (nc + 1 + 128) * sizeof(float);
    MIREC_REQUIRE(smem <= 64 * 1024, MIREC_E_UNSUPPORTED, "mirec_lightfm_ranks: %d components exceed the supported maximum", nc);
    const int64_t blocks = (nu + mirec::RANK_WAVES - 1) / mirec::RANK_WAVES;
    hipLaunchKernelGGL(mirec::lightfm_ranks_kernel, dim3((unsigned)blocks), dim3(64 * mirec::RANK_WAVES), smem, stream,
                       urepr, irepr, nu, ni, nc, test_indptr, test_indices, train_indptr, train_indices, ranks);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}

extern "C" int mirec_lightfm_repr(const int64_t* indptr, const int32_t* indices, const float* weights, int64_t rows,
                                  const float* E, const float* bias, float scale, int nc, float* out, int ld,
                                  int emb_col, int bias_col, int one_col, mirec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MIREC_REQUIRE(rows >= 0 && nc >= 1, MIREC_E_INVALID, "mirec_lightfm_repr: bad rows/nc");
    if (rows == 0) return MIREC_OK;
    MIREC_REQUIRE(indptr && E && bias && out, MIREC_E_INVALID, "mirec_lightfm_repr: null argument");
    MIREC_REQUIRE(emb_col >= 0 && emb_col + nc <= ld && bias_col >= 0 && bias_col < ld && one_col < ld,
                  MIREC_E_INVALID, "mirec_lightfm_repr: columns do not fit the row stride %d", ld);
    MIREC_REQUIRE((bias_col < emb_col || bias_col >= emb_col + nc) &&
                      (one_col < 0 || ((one_col < emb_col || one_col >= emb_col + nc) && one_col != bias_col)),
                  MIREC_E_INVALID, "mirec_lightfm_repr: output columns overlap");
    const int64_t total = rows * (nc + 1);
    hipLaunchKernelGGL(mirec::lightfm_repr_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, indptr,
                       indices, weights, rows, E, bias, scale, nc, out, ld, emb_col, bias_col, one_col);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}

extern "C" int mirec_lightfm_predict(const float* urepr, const float* irepr, const int32_t* user_ids,
                                     const int32_t* item_ids, int64_t n, int nc, float* out,
                                     mirec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MIREC_REQUIRE(n >= 0 && nc >= 1, MIREC_E_INVALID, "mirec_lightfm_predict: bad n/nc");
    if (n == 0) return MIREC_OK;
    MIREC_REQUIRE(urepr && irepr && user_ids && item_ids && out, MIREC_E_INVALID,
                  "mirec_lightfm_predict: null argument");
    hipLaunchKernelGGL(mirec::lightfm_predict_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, urepr,
                       irepr, user_ids, item_ids, n, nc, out);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}
