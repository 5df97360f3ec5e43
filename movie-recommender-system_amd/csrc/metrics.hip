// Held-out ranking evaluation over the top-K output: hits per user.
//
// One thread per user: each recommended id is looked up by binary search in the
// user's held-out row (ascending column indices).  Everything is integer work;
// the averages are formed on the host from the per-user counts, in user order,
// so they equal the CPU evaluation bit for bit.
#include "common.h"

namespace mirec {

__global__ __launch_bounds__(256) void ranking_hits_kernel(const int32_t* __restrict__ rec, int64_t nu, int k,
                                                           const int64_t* __restrict__ hp,
                                                           const int32_t* __restrict__ hi, int32_t* __restrict__ hits,
                                                           int32_t* __restrict__ nrec) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nu) return;
    const int64_t b = hp[u], e = hp[u + 1];
    const int32_t* r = rec + u * k;
    int h = 0, n = 0;
    for (int j = 0; j < k; ++j) {
        const int32_t id = r[j];
        if (id < 0) continue;
        ++n;
        int64_t lo = b, hi_ = e;
        while (lo < hi_) {
            const int64_t mid = (lo + hi_) >> 1;
            if (hi[mid] < id) lo = mid + 1;
            else hi_ = mid;
        }
        // an id recommended twice counts once, as in a set intersection
        bool dup = false;
        for (int p = 0; p < j; ++p) dup |= (r[p] == id);
        if (lo < e && hi[lo] == id && !dup) ++h;
    }
    hits[u] = h;
    nrec[u] = n;
}

}  // namespace mirec

extern "C" int mirec_ranking_hits(const int32_t* rec_ids, int64_t n_users, int k, const int64_t* held_indptr,
                                  const int32_t* held_indices, int32_t* hits, int32_t* n_rec,
                                  mirec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MIREC_REQUIRE(n_users >= 0 && k >= 1, MIREC_E_INVALID, "mirec_ranking_hits: bad n_users or k");
    if (n_users == 0) return MIREC_OK;
    MIREC_REQUIRE(rec_ids && held_indptr && hits && n_rec, MIREC_E_INVALID, "mirec_ranking_hits: null argument");
    const unsigned blocks = (unsigned)((n_users + 255) / 256);
    hipLaunchKernelGGL(mirec::ranking_hits_kernel, dim3(blocks), dim3(256), 0, stream, rec_ids, n_users, k, held_indptr,
                       held_indices, hits, n_rec);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}
