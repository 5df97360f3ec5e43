// Interaction-matrix build: (user index, item index) events -> binary CSR.
//
// The reference groups its watch events by (user, movie) and stores a one per
// pair (als/als.py:118-121).  Here the pairs are packed into 64-bit keys,
// radix-sorted and made unique (rocPRIM moves the keys); the kernels build the
// keys, the row pointers, the column indices and the per-item counts (the
// popularity the reference derives from the column sums, als/als.py:175-176).
// Rows come out with ascending column indices, as scipy's csr_matrix((data,
// (rows, cols))) + sort_indices does.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace mirec {

__global__ void pack_keys_kernel(const int32_t* __restrict__ u, const int32_t* __restrict__ m, int64_t n,
                                 uint64_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = ((uint64_t)(uint32_t)u[i] << 32) | (uint64_t)(uint32_t)m[i];
}

// Entries of the unique pairs: column index, value one, and the item's count.
__global__ void unpack_entries_kernel(const uint64_t* __restrict__ keys, const size_t* __restrict__ count,
                                      int32_t* __restrict__ indices, float* __restrict__ data,
                                      int32_t* __restrict__ item_counts, int64_t n_items, int64_t* __restrict__ nnz_out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t cnt = (int64_t)*count;
    if (j == 0) *nnz_out = cnt;
    if (j >= cnt) return;
    const int32_t col = (int32_t)(uint32_t)keys[j];
    indices[j] = col;
    data[j] = 1.0f;
    if (item_counts && col >= 0 && col < n_items) atomicAdd(&item_counts[col], 1);
}

// indptr[r] = first unique pair whose user is >= r.
__global__ void row_ptr_kernel(const uint64_t* __restrict__ keys, const size_t* __restrict__ count, int64_t n_users,
                               int64_t* __restrict__ indptr) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_users) return;
    const uint64_t target = (uint64_t)r << 32;
    int64_t lo = 0, hi = (int64_t)*count;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    indptr[r] = lo;
}

static inline unsigned key_bits(int64_t n_users) {
    unsigned b = 1;
    while (b < 31 && ((int64_t)1 << b) < n_users) ++b;
    return 32 + b;
}

static size_t events_temp_bytes(int64_t n, int64_t n_users) {
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_keys(nullptr, a, (const uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)n, 0u,
                                   key_bits(n_users));
    (void)rocprim::unique(nullptr, b, (const uint64_t*)nullptr, (uint64_t*)nullptr, (size_t*)nullptr, (size_t)n);
    return a > b ? a : b;
}

}  // namespace mirec

using namespace mirec;

extern "C" size_t mirec_events_to_csr_workspace_bytes(int64_t n_events, int64_t n_users) {
    if (n_events < 0 || n_users < 0) return 0;
    const size_t arr = align_up((size_t)(n_events > 0 ? n_events : 1) * 8, 256);
    return 2 * arr + 256 + align_up(events_temp_bytes(n_events, n_users), 256) + 256;
}

extern "C" int mirec_events_to_csr(const int32_t* user_idx, const int32_t* item_idx, int64_t n_events, int64_t n_users,
                                   int64_t n_items, int64_t* indptr, int32_t* indices, float* data,
                                   int32_t* item_counts, int64_t* nnz_out, void* ws, size_t ws_bytes,
                                   mirec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MIREC_REQUIRE(n_events >= 0 && n_users >= 0 && n_items >= 0, MIREC_E_INVALID, "mirec_events_to_csr: negative size");
    MIREC_REQUIRE(n_events < ((int64_t)1 << 31) && n_users < ((int64_t)1 << 31) && n_items < ((int64_t)1 << 31),
                  MIREC_E_UNSUPPORTED, "mirec_events_to_csr: more than 2^31 events, users or items");
    MIREC_REQUIRE(indptr && nnz_out, MIREC_E_INVALID, "mirec_events_to_csr: null output");
    MIREC_REQUIRE(n_events == 0 || (user_idx && item_idx && indices && data), MIREC_E_INVALID,
                  "mirec_events_to_csr: null event or entry array");
    MIREC_REQUIRE(ws && ws_bytes >= mirec_events_to_csr_workspace_bytes(n_events, n_users), MIREC_E_WORKSPACE,
                  "mirec_events_to_csr: workspace of %zu bytes needed, %zu given",
                  mirec_events_to_csr_workspace_bytes(n_events, n_users), ws_bytes);
    const size_t arr = align_up((size_t)(n_events > 0 ? n_events : 1) * 8, 256);
    char* base = (char*)ws;
    uint64_t* keys_a = (uint64_t*)base;
    uint64_t* keys_b = (uint64_t*)(base + arr);
    size_t* count = (size_t*)(base + 2 * arr);
    void* temp = base + 2 * arr + 256;
    size_t temp_bytes = events_temp_bytes(n_events, n_users);
    MIREC_HIP(hipMemsetAsync(count, 0, sizeof(size_t), stream));
    if (item_counts && n_items > 0) MIREC_HIP(hipMemsetAsync(item_counts, 0, (size_t)n_items * sizeof(int32_t), stream));
    const unsigned eb = (unsigned)((n_events + 255) / 256);
    if (n_events > 0) {
        hipLaunchKernelGGL(pack_keys_kernel, dim3(eb), dim3(256), 0, stream, user_idx, item_idx, n_events, keys_a);
        MIREC_LAUNCH_CHECK();
        MIREC_HIP(rocprim::radix_sort_keys(temp, temp_bytes, (const uint64_t*)keys_a, keys_b, (size_t)n_events, 0u,
                                           key_bits(n_users), stream));
        MIREC_HIP(rocprim::unique(temp, temp_bytes, (const uint64_t*)keys_b, keys_a, count, (size_t)n_events,
                                  rocprim::equal_to<uint64_t>(), stream));
    }
    hipLaunchKernelGGL(unpack_entries_kernel, dim3(eb > 0 ? eb : 1), dim3(256), 0, stream, (const uint64_t*)keys_a,
                       (const size_t*)count, indices, data, item_counts, n_items, nnz_out);
    MIREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(row_ptr_kernel, dim3((unsigned)((n_users + 1 + 255) / 256)), dim3(256), 0, stream,
                       (const uint64_t*)keys_a, (const size_t*)count, n_users, indptr);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}
