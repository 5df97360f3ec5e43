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

}  // namespace mirec

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
