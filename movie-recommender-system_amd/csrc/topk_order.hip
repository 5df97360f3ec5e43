// Sweep order of the all-users top-K pass.
//
// The bf16-plane kernel walks the item tiles once per query and keeps, per query, a threshold below which a score
// is not looked at.  The order of the walk cannot change what comes out (the survivors are scored again exactly and
// ordered by score and item), but it decides how much selection work there is: with the items that most users score
// highest met first, the thresholds are near their final values after a few tiles and a wavefront of 32 queries finds
// nothing to look at in nineteen tiles out of twenty, against one in two for an arbitrary order.  An item's number
// of interactions is such an order, and it belongs to the interaction matrix, not to the model: it is built once per
// matrix, together with the filter renumbered to positions of the order and sorted (the kernel's walk over the
// filter needs it ascending in the order of the sweep).
//
//   order_count_kernel   interactions per item (one thread per stored entry, atomic adds)
//   order_rank_kernel    position of an item = number of items with more interactions, or as many and a smaller
//                        index; one thread per item, the counts pass through LDS in slabs
//   order_blocks_kernel  one list per block of 32 filter rows (the queries of one wavefront of the sweep): the
//                        positions any of the rows holds, ascending, each with the set of rows that hold it -- a
//                        counting sort through a word per position in LDS.  The sweep then walks one list per
//                        wavefront: a cursor all lanes share, coalesced loads, one compare for a tile nobody
//                        of the 32 has a liked item in.
#include "common.h"

namespace mirec {

constexpr int64_t TOPK_ORDER_MAX_ITEMS = 1 << 18;  // eight passes of the block kernel over its entries at most

__global__ __launch_bounds__(256) void order_count_kernel(const int32_t* __restrict__ indices, int64_t nnz, int ni,
                                                          int32_t* __restrict__ count) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nnz; e += stride) {
        const int i = indices[e];
        if (i >= 0 && i < ni) atomicAdd(count + i, 1);
    }
}

// The same through a histogram per workgroup in LDS (catalogues of up to ORDER_COUNT_LDS items): the popular items,
// which every workgroup meets thousands of times, cost an LDS atomic each time and one global atomic per workgroup.
constexpr int ORDER_COUNT_LDS = 36864;  // words of LDS (144 KB)

__global__ __launch_bounds__(1024) void order_count_lds_kernel(const int32_t* __restrict__ indices, int64_t nnz, int ni,
                                                               int32_t* __restrict__ count) {
    extern __shared__ uint32_t order_lds[];
    int32_t* hist = reinterpret_cast<int32_t*>(order_lds);
    for (int t = threadIdx.x; t < ni; t += 1024) hist[t] = 0;
    __syncthreads();
    const int64_t per = (nnz + gridDim.x - 1) / gridDim.x;
    const int64_t e0 = (int64_t)blockIdx.x * per, e1 = e0 + per < nnz ? e0 + per : nnz;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 1024) {
        const int i = indices[e];
        if (i >= 0 && i < ni) atomicAdd(hist + i, 1);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ni; t += 1024) {
        const int v = hist[t];
        if (v != 0) atomicAdd(count + t, v);
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

// One workgroup per block of 32 filter rows at a time.  LDS holds one word per position of a range of the order:
// bit r of the word = row r of the block has the item at that position.  The block's entries (contiguous in the CSR
// arrays) set their bits; then every thread compacts a stretch of the words: the ones that are not zero, in
// ascending position, are the block's list.  The words are cleared as they are read, so the array is zero again
// for the next block.  Catalogues longer than the array take several passes over the block's entries.
constexpr int ORDER_RANGE = 32768;

__global__ __launch_bounds__(256) void order_blocks_kernel(const int64_t* __restrict__ indptr,
                                                           const int32_t* __restrict__ indices, int64_t nrows, int ni,
                                                           const int32_t* __restrict__ position,
                                                           int2* __restrict__ out, int32_t* __restrict__ block_len) {
    extern __shared__ uint32_t order_lds[];
    __shared__ int64_t ptrs[33];
    __shared__ int wave_total[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int range_words = ni < ORDER_RANGE ? ni : ORDER_RANGE;
    for (int t = tid; t < range_words; t += 256) order_lds[t] = 0u;
    const int per = ((range_words + 255) / 256) | 1;  // words per thread; odd: the threads of a wavefront on distinct banks
    const int64_t nblk = (nrows + 31) / 32;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t r0 = blk * 32;
        const int nr = nrows - r0 < 32 ? (int)(nrows - r0) : 32;
        __syncthreads();
        if (tid <= nr) ptrs[tid] = indptr[r0 + tid];
        __syncthreads();
        const int64_t base = ptrs[0], end = ptrs[nr];
        int total = 0;
        for (int range0 = 0; range0 < ni; range0 += ORDER_RANGE) {
            const int rlen = ni - range0 < ORDER_RANGE ? ni - range0 : ORDER_RANGE;
            for (int64_t e = base + tid; e < end; e += 256) {
                int r = 0;
                for (int j = 1; j < nr; ++j) r += ptrs[j] <= e ? 1 : 0;  // the row that holds entry e
                const int i = indices[e];
                if (i >= 0 && i < ni) {
                    const int p = position[i] - range0;
                    if (p >= 0 && p < rlen) atomicOr(order_lds + p, 1u << r);
                }
            }
            __syncthreads();
            const int w0 = tid * per;
            int mine = 0;
            for (int j = 0; j < per; ++j) mine += (w0 + j < rlen && order_lds[w0 + j] != 0u) ? 1 : 0;
            int incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off, 64);
                incl += lane >= off ? v : 0;
            }
            if (lane == 63) wave_total[wv] = incl;
            __syncthreads();
            int before = 0, all = 0;
            for (int j = 0; j < 4; ++j) {
                before += j < wv ? wave_total[j] : 0;
                all += wave_total[j];
            }
            int64_t at = base + total + before + incl - mine;
            for (int j = 0; j < per; ++j) {
                if (w0 + j < rlen) {
                    const uint32_t v = order_lds[w0 + j];
                    if (v != 0u) {
                        out[at++] = make_int2(range0 + w0 + j, (int)v);
                        order_lds[w0 + j] = 0u;
                    }
                }
            }
            total += all;
            __syncthreads();
        }
        if (tid == 0) block_len[blk] = total;
    }
}

}  // namespace mirec

using namespace mirec;

extern "C" size_t mirec_topk_order_workspace_bytes(int64_t ni) {
    if (ni <= 0 || ni > TOPK_ORDER_MAX_ITEMS) return 0;
    return align_up((size_t)ni * 4, 256) * 2;
}

extern "C" int mirec_topk_order_build(const int64_t* mask_indptr, const int32_t* mask_indices, int64_t nrows,
                                      int64_t nnz, int64_t ni, int32_t* item_order, int32_t* mask_blocks,
                                      int32_t* block_len, void* ws, size_t ws_bytes, mirec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MIREC_REQUIRE(nrows >= 0 && nnz >= 0 && ni >= 1, MIREC_E_INVALID, "mirec_topk_order_build: bad sizes");
    MIREC_REQUIRE(ni <= TOPK_ORDER_MAX_ITEMS, MIREC_E_UNSUPPORTED,
                  "mirec_topk_order_build: %lld items, at most %lld are ordered; use mirec_topk", (long long)ni,
                  (long long)TOPK_ORDER_MAX_ITEMS);
    MIREC_REQUIRE(item_order && ws && (nrows == 0 || (mask_indptr && block_len)) &&
                      (nnz == 0 || (mask_indices && mask_blocks)),
                  MIREC_E_INVALID, "mirec_topk_order_build: null argument");
    MIREC_REQUIRE(((uintptr_t)mask_blocks & 7) == 0, MIREC_E_INVALID, "mirec_topk_order_build: mask_blocks must be 8-byte aligned");
    const size_t part = align_up((size_t)ni * 4, 256);
    MIREC_REQUIRE(ws_bytes >= 2 * part, MIREC_E_INVALID,
                  "mirec_topk_order_build: workspace of %zu bytes given, mirec_topk_order_workspace_bytes asks for %zu",
                  ws_bytes, 2 * part);
    int32_t* count = static_cast<int32_t*>(ws);
    int32_t* position = reinterpret_cast<int32_t*>(static_cast<char*>(ws) + part);
    MIREC_HIP(hipMemsetAsync(count, 0, (size_t)ni * 4, stream));
    static OncePerDevice once;
    bool* attr_done = once.slot();
    MIREC_REQUIRE(attr_done != nullptr, MIREC_E_HIP, "cannot read the current device ordinal (or it is >= %d)", kMaxDevices);
    if (!*attr_done) {
        MIREC_HIP(hipFuncSetAttribute((const void*)order_blocks_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      ORDER_RANGE * 4));
        MIREC_HIP(hipFuncSetAttribute((const void*)order_count_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      ORDER_COUNT_LDS * 4));
        *attr_done = true;
    }
    if (nnz > 0 && ni <= ORDER_COUNT_LDS && nnz >= 65536) {
        // a workgroup per CU: each takes a contiguous share of the entries
        const int64_t share = (nnz + 255) / 256;
        const unsigned blocks = (unsigned)((nnz + share - 1) / share);
        hipLaunchKernelGGL(order_count_lds_kernel, dim3(blocks), dim3(1024), (size_t)ni * 4, stream, mask_indices, nnz,
                           (int)ni, count);
        MIREC_LAUNCH_CHECK();
    } else if (nnz > 0) {
        const int64_t blocks = (nnz + 255) / 256;
        hipLaunchKernelGGL(order_count_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream,
                           mask_indices, nnz, (int)ni, count);
        MIREC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(order_rank_kernel, dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, stream, count, (int)ni,
                       item_order, position);
    MIREC_LAUNCH_CHECK();
    if (nrows > 0) {
        const size_t lds = (size_t)(ni < ORDER_RANGE ? ni : ORDER_RANGE) * 4;
        const int64_t nblk = (nrows + 31) / 32;
        hipLaunchKernelGGL(order_blocks_kernel, dim3((unsigned)(nblk < 2048 ? nblk : 2048)), dim3(256), lds, stream,
                           mask_indptr, mask_indices, nrows, (int)ni, position, reinterpret_cast<int2*>(mask_blocks),
                           block_len);
        MIREC_LAUNCH_CHECK();
    }
    return MIREC_OK;
}
