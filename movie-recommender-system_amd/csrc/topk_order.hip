// Sweep order of the all-users top-K pass.
//
// The bf16-plane kernel walks the item tiles once per query and keeps, per query, a threshold below which a score
// is not looked at.  The order of the walk cannot change what comes out (the survivors are scored again exactly and
// ordered by score and item), but it decides how much selection work there is: with the items that most users score
// highest met first, the thresholds are near their final values after a few tiles and a wavefront of 32 queries finds
// nothing to look at in nineteen tiles out of twenty, against one in two for an arbitrary order.  An item's number
// of interactions is such an order, and it belongs to the interaction matrix, not to the model: it is built once per
// matrix, together with the filter rows renumbered to positions of the order and sorted (the kernel's walk over a
// filter row needs it ascending in the order of the sweep).
//
//   order_count_kernel   interactions per item (one thread per stored entry, atomic adds)
//   order_rank_kernel    position of an item = number of items with more interactions, or as many and a smaller
//                        index; one thread per item, the counts pass through LDS in slabs
//   order_rows_kernel    one wavefront per filter row: the row's positions are marked in a bitmap of the positions
//                        in LDS and read back in ascending order -- a counting sort whose cost does not depend on
//                        how the row's length compares with a wavefront
#include "common.h"

namespace mirec {

constexpr int64_t TOPK_ORDER_MAX_ITEMS = 1 << 18;  // bitmap of a row: 32 KB of LDS per wavefront at most

__global__ __launch_bounds__(256) void order_count_kernel(const int32_t* __restrict__ indices, int64_t nnz, int ni,
                                                          int32_t* __restrict__ count) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nnz; e += stride) {
        const int i = indices[e];
        if (i >= 0 && i < ni) atomicAdd(count + i, 1);
    }
}

__global__ __launch_bounds__(256) void order_rank_kernel(const int32_t* __restrict__ count, int ni,
                                                         int32_t* __restrict__ item_order,
                                                         int32_t* __restrict__ position) {
    __shared__ int32_t slab[1024];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int mine = i < ni ? count[i] : 0;
    int rank = 0;
    for (int j0 = 0; j0 < ni; j0 += 1024) {
        __syncthreads();
        for (int t = threadIdx.x; t < 1024; t += 256) slab[t] = j0 + t < ni ? count[j0 + t] : -1;
        __syncthreads();
        // items before i in index order win ties, items after lose them: one compare each
        const int n = ni - j0 < 1024 ? ni - j0 : 1024;
        const int before = i - j0;  // slab entries below this index are items with a smaller index than i
        for (int t = 0; t < n; ++t) {
            const int c = slab[t];
            rank += (c > mine || (c == mine && t < before)) ? 1 : 0;
        }
    }
    if (i < ni) {
        item_order[rank] = i;
        position[i] = rank;
    }
}

__global__ __launch_bounds__(256) void order_rows_kernel(const int64_t* __restrict__ indptr,
                                                         const int32_t* __restrict__ indices, int64_t nrows, int ni,
                                                         const int32_t* __restrict__ position,
                                                         int32_t* __restrict__ out) {
    extern __shared__ uint32_t order_lds[];
    const int words = (ni + 31) >> 5;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* bm = order_lds + (size_t)w * words;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t r = (int64_t)blockIdx.x * 4 + w; r < nrows; r += stride) {
        const int64_t b = indptr[r];
        const int n = (int)(indptr[r + 1] - b);
        if (n <= 0) continue;
        for (int t = lane; t < words; t += 64) bm[t] = 0u;
        __builtin_amdgcn_wave_barrier();
        int last = ni;  // a position past every item: what fills a row without a valid entry
        for (int e = lane; e < n; e += 64) {
            const int i = indices[b + e];
            if (i >= 0 && i < ni) {
                const int p = position[i];
                atomicOr(bm + (p >> 5), 1u << (p & 31));
            }
        }
        __builtin_amdgcn_wave_barrier();
        int base = 0;
        for (int t0 = 0; t0 < words; t0 += 64) {
            const int t = t0 + lane;
            const uint32_t word0 = t < words ? bm[t] : 0u;
            uint32_t word = word0;
            if (__ballot(word != 0u) == 0ull) continue;
            const int c = __builtin_popcount(word);
            int incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off, 64);
                incl += lane >= off ? v : 0;
            }
            int at = base + incl - c;
            while (word != 0u) {
                const int bit = __builtin_ctz(word);
                word &= word - 1u;
                out[b + at++] = (t << 5) + bit;
            }
            base += __shfl(incl, 63, 64);
            const int top = 63 - (int)__builtin_clzll(__ballot(c != 0));  // highest lane with a bit
            const uint32_t wt = (uint32_t)__shfl((int)word0, top, 64);
            last = ((t0 + top) << 5) + 31 - __builtin_clz(wt);
        }
        // a row that names an item twice (or one out of range) has fewer positions than entries: its tail repeats
        // the last one, which keeps the row ascending and flags nothing new
        for (int e = base + lane; e < n; e += 64) out[b + e] = last;
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace mirec

using namespace mirec;

extern "C" size_t mirec_topk_order_workspace_bytes(int64_t ni) {
    if (ni <= 0 || ni > TOPK_ORDER_MAX_ITEMS) return 0;
    return align_up((size_t)ni * 4, 256) * 2;
}

extern "C" int mirec_topk_order_build(const int64_t* mask_indptr, const int32_t* mask_indices, int64_t nrows,
                                      int64_t nnz, int64_t ni, int32_t* item_order, int32_t* mask_positions, void* ws,
                                      size_t ws_bytes, mirec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MIREC_REQUIRE(nrows >= 0 && nnz >= 0 && ni >= 1, MIREC_E_INVALID, "mirec_topk_order_build: bad sizes");
    MIREC_REQUIRE(ni <= TOPK_ORDER_MAX_ITEMS, MIREC_E_UNSUPPORTED,
                  "mirec_topk_order_build: %lld items, at most %lld are ordered; use mirec_topk", (long long)ni,
                  (long long)TOPK_ORDER_MAX_ITEMS);
    MIREC_REQUIRE(item_order && ws && (nnz == 0 || (mask_indptr && mask_indices && mask_positions)), MIREC_E_INVALID,
                  "mirec_topk_order_build: null argument");
    const size_t part = align_up((size_t)ni * 4, 256);
    MIREC_REQUIRE(ws_bytes >= 2 * part, MIREC_E_INVALID,
                  "mirec_topk_order_build: workspace of %zu bytes given, mirec_topk_order_workspace_bytes asks for %zu",
                  ws_bytes, 2 * part);
    int32_t* count = static_cast<int32_t*>(ws);
    int32_t* position = reinterpret_cast<int32_t*>(static_cast<char*>(ws) + part);
    MIREC_HIP(hipMemsetAsync(count, 0, (size_t)ni * 4, stream));
    if (nnz > 0) {
        const int64_t blocks = (nnz + 255) / 256;
        hipLaunchKernelGGL(order_count_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream,
                           mask_indices, nnz, (int)ni, count);
        MIREC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(order_rank_kernel, dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, stream, count, (int)ni,
                       item_order, position);
    MIREC_LAUNCH_CHECK();
    if (nnz > 0 && nrows > 0) {
        const size_t lds = (size_t)((ni + 31) / 32) * 4 * 4;
        static OncePerDevice once;
        bool* attr_done = once.slot();
        MIREC_REQUIRE(attr_done != nullptr, MIREC_E_HIP, "cannot read the current device ordinal (or it is >= %d)", kMaxDevices);
        if (!*attr_done) {
            MIREC_HIP(hipFuncSetAttribute((const void*)order_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(TOPK_ORDER_MAX_ITEMS / 32 * 4 * 4)));
            *attr_done = true;
        }
        const int64_t blocks = (nrows + 3) / 4;
        hipLaunchKernelGGL(order_rows_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), lds,
                           stream, mask_indptr, mask_indices, nrows, (int)ni, position, mask_positions);
        MIREC_LAUNCH_CHECK();
    }
    return MIREC_OK;
}
