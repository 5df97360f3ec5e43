// CSR transpose by a stable radix sort on the column index.
//
// Entries are visited in row-major order, so a stable sort by column leaves the
// rows of every output row (= input column) ascending, which is what scipy's
// `.T.tocsr()` produces and what keeps the summation order of the item
// half-sweep fixed from run to run.  One-off per fit; rocPRIM's device radix
// sort moves the keys, the kernels here build the pointers and gather.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace mirec {

__global__ void iota_kernel(int32_t* v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (int32_t)i;
}

// t_indptr[c] = first sorted position whose column is >= c.
__global__ void transpose_ptr_kernel(const int32_t* __restrict__ sorted_cols, int64_t nnz, int64_t cols,
                                     int64_t* __restrict__ t_indptr) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > cols) return;
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)sorted_cols[mid] < c) lo = mid + 1; else hi = mid;
    }
    t_indptr[c] = lo;
}

// For every sorted position: the source entry's row (upper bound in indptr) and value.
__global__ void transpose_gather_kernel(const int64_t* __restrict__ indptr, int64_t rows,
                                        const float* __restrict__ data, const int32_t* __restrict__ src_pos,
                                        int64_t nnz, int32_t* __restrict__ t_indices, float* __restrict__ t_data) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const int64_t k = src_pos[p];
    int64_t lo = 0, hi = rows;  // largest r with indptr[r] <= k
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (indptr[mid] <= k) lo = mid; else hi = mid;
    }
    t_indices[p] = (int32_t)lo;
    t_data[p] = data[k];
}

static inline unsigned bits_for(int64_t cols) {
    unsigned b = 1;
    while (b < 32 && ((int64_t)1 << b) < cols) ++b;
    return b;
}

static size_t sort_temp_bytes(int64_t nnz, int64_t cols) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr,
                                    (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)nnz, 0u, bits_for(cols));
    return bytes;
}

}  // namespace mirec

using namespace mirec;

extern "C" size_t mirec_csr_transpose_workspace_bytes(int64_t rows, int64_t cols, int64_t nnz) {
    (void)rows;
    if (nnz < 0 || cols < 0) return 0;
    const size_t arr = align_up((size_t)(nnz > 0 ? nnz : 1) * 4, 256);
    return 3 * arr + align_up(sort_temp_bytes(nnz, cols), 256) + 256;
}

extern "C" int mirec_csr_transpose(const int64_t* indptr, const int32_t* indices, const float* data, int64_t rows,
                                   int64_t cols, int64_t nnz, int64_t* t_indptr, int32_t* t_indices, float* t_data,
                                   void* ws, size_t ws_bytes, mirec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MIREC_REQUIRE(rows >= 0 && cols >= 0 && nnz >= 0, MIREC_E_INVALID, "mirec_csr_transpose: negative size");
    MIREC_REQUIRE(indptr && t_indptr, MIREC_E_INVALID, "mirec_csr_transpose: null pointer array");
    MIREC_REQUIRE(nnz == 0 || (indices && data && t_indices && t_data), MIREC_E_INVALID,
                  "mirec_csr_transpose: null entry array");
    MIREC_REQUIRE(nnz < ((int64_t)1 << 31) && rows < ((int64_t)1 << 31), MIREC_E_UNSUPPORTED,
                  "mirec_csr_transpose: more than 2^31 entries or rows");
    MIREC_REQUIRE(ws && ws_bytes >= mirec_csr_transpose_workspace_bytes(rows, cols, nnz), MIREC_E_WORKSPACE,
                  "mirec_csr_transpose: workspace of %zu bytes needed, %zu given",
                  mirec_csr_transpose_workspace_bytes(rows, cols, nnz), ws_bytes);
    const size_t arr = align_up((size_t)(nnz > 0 ? nnz : 1) * 4, 256);
    char* base = (char*)ws;
    int32_t* keys_out = (int32_t*)base;
    int32_t* vals_in = (int32_t*)(base + arr);
    int32_t* vals_out = (int32_t*)(base + 2 * arr);
    void* temp = base + 3 * arr;
    size_t temp_bytes = sort_temp_bytes(nnz, cols);
    if (nnz > 0) {
        hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, stream, vals_in, nnz);
        MIREC_LAUNCH_CHECK();
        MIREC_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, indices, keys_out, (const int32_t*)vals_in, vals_out,
                                            (size_t)nnz, 0u, bits_for(cols), stream));
        hipLaunchKernelGGL(transpose_gather_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, stream, indptr,
                           rows, data, vals_out, nnz, t_indices, t_data);
        MIREC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(transpose_ptr_kernel, dim3((unsigned)((cols + 1 + 255) / 256)), dim3(256), 0, stream, keys_out,
                       nnz, cols, t_indptr);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}
