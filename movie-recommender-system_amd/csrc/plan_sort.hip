// Row order of the matrix-free half-sweep kernel: the rows it solves (at most short_max stored entries), longest
// first, equal lengths in ascending row order.  Wavefronts of a workgroup that advance in step (they share the
// matrix-core evaluation of G v) then work on rows of the same length, and the longest rows start first.
// A stable radix sort (rocPRIM) on the key short_max - length; rows left to the explicit-Gram kernel get the
// largest key and end up behind the n_short rows that count.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "als_internal.h"

namespace mirec {

__global__ void short_keys_kernel(const int64_t* __restrict__ indptr, int64_t row_begin, int64_t nrows,
                                  int32_t short_max, uint32_t* __restrict__ keys, int32_t* __restrict__ rows) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const int64_t n = indptr[row_begin + r + 1] - indptr[row_begin + r];
    keys[r] = n <= (int64_t)short_max ? (uint32_t)((int64_t)short_max - n) : 0xffffffffu;
    rows[r] = (int32_t)r;
}

static size_t sort_temp(int64_t nrows) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                    (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)(nrows > 0 ? nrows : 1), 0u, 32u);
    return bytes;
}

size_t short_order_tmp_bytes(int64_t nrows) {
    const size_t arr = align_up((size_t)(nrows > 0 ? nrows : 1) * 4, 256);
    return 3 * arr + align_up(sort_temp(nrows), 256);
}

int build_short_order(const int64_t* indptr, int64_t row_begin, int64_t nrows, int32_t short_max, int32_t* order,
                      void* tmp, hipStream_t stream) {
    if (nrows <= 0) return MIREC_OK;
    const size_t arr = align_up((size_t)nrows * 4, 256);
    char* base = (char*)tmp;
    uint32_t* keys_in = (uint32_t*)base;
    uint32_t* keys_out = (uint32_t*)(base + arr);
    int32_t* rows_in = (int32_t*)(base + 2 * arr);
    void* temp = base + 3 * arr;
    size_t temp_bytes = sort_temp(nrows);
    hipLaunchKernelGGL(short_keys_kernel, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, stream, indptr,
                       row_begin, nrows, short_max, keys_in, rows_in);
    MIREC_LAUNCH_CHECK();
    MIREC_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t*)keys_in, keys_out, (const int32_t*)rows_in,
                                        order, (size_t)nrows, 0u, 32u, stream));
    return MIREC_OK;
}

}  // namespace mirec
