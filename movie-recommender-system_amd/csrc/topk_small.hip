// Score + top-K for a handful of queries: the serving form of the pass, one user (or one item, for
// similar-items) per request -- recommenders/als_recommender.py:181-186 and :237,
// hybrid_docker/inference.py:156-165,197 in the reference.
//
// The all-users sweep (topk_kernel.h, topk_planes_kernel.h) gives a workgroup 128 or 256 queries and walks it
// through every item; with one query that is one workgroup on a chip of 256 compute units.  Here the *items*
// are split instead: workgroup (tile, query group) scores 256 items, one per lane, for up to QB queries with
// the reference chain s = fma(q[c], y[c], s), c ascending -- the item rows pass through LDS in slabs of CK
// columns so that the loads are coalesced and every lane then reads its own row without bank conflicts, the
// query values are wave-uniform -- and ranks its 256 scores by counting.  The k best of the tile go to the
// workspace as 64-bit keys (score bits made monotone, then the complemented item index: a larger key is a
// better entry under (score descending, item ascending), and no two items share a key).  A second kernel, one
// workgroup per query, merges the tiles' lists: it takes a lower bound of the k-th best key from the tiles'
// best entries, keeps what reaches it and ranks the survivors.  Same result bits as the sweep kernels.
#include <float.h>

#include "common.h"

namespace mirec {

constexpr int TS_TI = 256;          // items per tile = threads per workgroup
constexpr int TS_CAP = 1024;        // survivor buffer of the merge
constexpr int TS_ROUND = 512;       // keys looked at between two room checks of the merge

__device__ __forceinline__ uint32_t ts_ord(float s) {
    const uint32_t b = __float_as_uint(s);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ float ts_unord(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// 0 marks "no entry": a NaN score is never recommended (the sweep kernels never admit one either), and every
// other score, -inf included, maps to a high word of at least 0x007fffff.
__device__ __forceinline__ uint64_t ts_key(float s, uint32_t item) {
    if (s != s) return 0ull;
    s = (s == 0.f) ? 0.f : s;  // -0 and +0 are one score
    return ((uint64_t)ts_ord(s) << 32) | (uint64_t)(0xffffffffu - item);
}

__device__ __forceinline__ int ts_count_above(const uint64_t* keys, int n2, uint64_t key) {
    // keys: 16-byte aligned LDS array of 2 * n2 entries; every lane reads the same address (broadcast)
    const ulonglong2* kv = reinterpret_cast<const ulonglong2*>(keys);
    int rank = 0;
#pragma unroll 8
    for (int j = 0; j < n2; ++j) {
        const ulonglong2 v = kv[j];
        rank += (v.x > key ? 1 : 0) + (v.y > key ? 1 : 0);
    }
    return rank;
}

template <int CK, int QB>
__global__ __launch_bounds__(TS_TI) void topk_small_tile_kernel(
    const float* __restrict__ Q, int64_t nq, const float* __restrict__ Y, int64_t ni, int f,
    const int64_t* __restrict__ mask_indptr, const int32_t* __restrict__ mask_indices,
    const float* __restrict__ y_norm, const float* __restrict__ q_norm, int k, uint64_t* __restrict__ part,
    int64_t ntiles) {
    constexpr int LD = CK + 1;   // odd stride: the lanes' row reads fall into distinct banks
    constexpr int C4 = CK / 4;   // 16-byte pieces per row of a slab
    constexpr int NV = C4;       // pieces per thread: 256 rows * C4 pieces / 256 threads
    __shared__ float ys[TS_TI * LD];
    __shared__ __attribute__((aligned(16))) uint64_t keys[TS_TI];
    __shared__ unsigned char liked[QB][TS_TI];

    const int tid = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t i0 = tile * TS_TI;
    const int64_t q0 = (int64_t)blockIdx.y * QB;
    const int nqb = (int)((nq - q0) < (int64_t)QB ? (nq - q0) : (int64_t)QB);

    f32x4 pre[NV];
    auto load_slab = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int e = tid + v * TS_TI;
            int64_t item = i0 + e / C4;
            item = item < ni ? item : ni - 1;  // rows past the last item repeat it; their keys are dropped below
            pre[v] = *reinterpret_cast<const f32x4*>(Y + item * f + kc + 4 * (e % C4));
        }
    };
    load_slab(0);

    // liked items of this tile, per query of the group
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) liked[qi][tid] = 0;
    __syncthreads();
    if (mask_indptr != nullptr) {
        for (int qi = 0; qi < nqb; ++qi) {
            const int64_t mb = mask_indptr[q0 + qi], me = mask_indptr[q0 + qi + 1];
            for (int64_t e = mb + tid; e < me; e += TS_TI) {
                const int64_t it = (int64_t)mask_indices[e] - i0;
                if (it >= 0 && it < TS_TI) liked[qi][it] = 1;
            }
        }
    }

    int64_t qrow[QB];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) qrow[qi] = (qi < nqb ? q0 + qi : q0) * (int64_t)f;

    float acc[QB];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) acc[qi] = 0.f;

    for (int kc = 0; kc < f; kc += CK) {
        __syncthreads();  // the previous slab has been consumed (first pass: the liked flags are in place)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int e = tid + v * TS_TI;
            float* d = ys + (e / C4) * LD + 4 * (e % C4);
            d[0] = pre[v][0];
            d[1] = pre[v][1];
            d[2] = pre[v][2];
            d[3] = pre[v][3];
        }
        __syncthreads();
        if (kc + CK < f) load_slab(kc + CK);
        const float* yr = ys + tid * LD;
#pragma unroll
        for (int c = 0; c < CK; ++c) {
            const float y = yr[c];
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) acc[qi] = __builtin_fmaf(Q[qrow[qi] + kc + c], y, acc[qi]);
        }
    }

    const int64_t item = i0 + tid;
    const bool item_ok = item < ni;
    const float yn = (y_norm != nullptr) ? y_norm[item_ok ? item : 0] : 1.f;
    for (int qi = 0; qi < nqb; ++qi) {
        // select the lane's accumulator without indexing the register array by a loop variable
        float s = acc[0];
#pragma unroll
        for (int j = 1; j < QB; ++j) s = (qi == j) ? acc[j] : s;
        if (y_norm != nullptr) s = s / yn;
        if (q_norm != nullptr) s = s / q_norm[q0 + qi];
        if (liked[qi][tid]) s = -FLT_MAX;
        const uint64_t key = item_ok ? ts_key(s, (uint32_t)item) : 0ull;
        keys[tid] = key;
        const int nvalid = __syncthreads_count(key != 0ull);
        const int rank = ts_count_above(keys, TS_TI / 2, key);
        uint64_t* dst = part + (q0 + qi) * (int64_t)k * ntiles + tile;   // entry of rank r at dst[r * ntiles]
        if (key != 0ull && rank < k) dst[(int64_t)rank * ntiles] = key;
        if (tid < k && tid >= nvalid) dst[(int64_t)tid * ntiles] = 0ull;
        __syncthreads();  // keys[] is reused by the next query
    }
}

__global__ __launch_bounds__(256) void topk_small_merge_kernel(const uint64_t* __restrict__ part, int64_t ntiles, int k,
                                                               int32_t* __restrict__ out_ids,
                                                               float* __restrict__ out_scores) {
    __shared__ __attribute__((aligned(16))) uint64_t buf[2][TS_CAP];
    __shared__ int cnt_s;
    __shared__ uint64_t tau_s;
    const int tid = threadIdx.x;
    const int64_t q = blockIdx.x;
    const uint64_t* p = part + q * (int64_t)k * ntiles;
    const int64_t total = (int64_t)k * ntiles;
    int cur = 0;

    // A lower bound of the k-th best key: the k-th best of the first 256 entries.  The workspace is laid out
    // rank-major, so those are the best entries of the first tiles (all tiles, up to 256 of them).
    const uint64_t mine = tid < total ? p[tid] : 0ull;
    buf[0][tid] = mine;
    if (tid == 0) {
        cnt_s = 0;
        tau_s = 0ull;
    }
    __syncthreads();
    if (mine != 0ull && ts_count_above(buf[0], 128, mine) == k - 1) tau_s = mine;  // keys are distinct: one writer
    __syncthreads();
    uint64_t tau = tau_s;
    int cnt = 0;

    // Keep min(cnt, k) best entries of buf[cur], sorted, at the front of the other buffer; raises tau.
    auto compact = [&]() __attribute__((always_inline)) {
        const int n = cnt;
        for (int e = tid; e < n; e += 256) {
            const uint64_t key = buf[cur][e];
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += buf[cur][j] > key ? 1 : 0;
            if (rank < k) buf[cur ^ 1][rank] = key;
        }
        __syncthreads();
        cur ^= 1;
        cnt = n < k ? n : k;
        if (n >= k) {
            const uint64_t kth = buf[cur][k - 1];
            tau = kth > tau ? kth : tau;
        }
        if (tid == 0) cnt_s = cnt;
        __syncthreads();
    };

    for (int64_t base = 0; base < total; base += TS_ROUND) {
        if (cnt + TS_ROUND > TS_CAP) compact();  // cnt is the same in every thread
#pragma unroll
        for (int v = 0; v < TS_ROUND / 256; ++v) {
            const int64_t e = base + tid + 256 * v;
            const uint64_t key = e < total ? p[e] : 0ull;
            if (key != 0ull && key >= tau) buf[cur][atomicAdd(&cnt_s, 1)] = key;
        }
        __syncthreads();
        cnt = cnt_s;
    }
    compact();
    for (int j = tid; j < k; j += 256) {
        const uint64_t key = j < cnt ? buf[cur][j] : 0ull;
        out_ids[q * k + j] = j < cnt ? (int32_t)(0xffffffffu - (uint32_t)key) : -1;
        out_scores[q * k + j] = j < cnt ? ts_unord((uint32_t)(key >> 32)) : -FLT_MAX;
    }
}

template <int CK>
static void launch_tiles(int64_t groups8, const float* Q, int64_t nq, const float* Y, int64_t ni, int f,
                         const int64_t* mask_indptr, const int32_t* mask_indices, const float* y_norm,
                         const float* q_norm, int k, uint64_t* part, int64_t ntiles, hipStream_t stream) {
    if (nq == 1)
        hipLaunchKernelGGL((topk_small_tile_kernel<CK, 1>), dim3((unsigned)ntiles, 1), dim3(TS_TI), 0, stream, Q, nq, Y, ni,
                           f, mask_indptr, mask_indices, y_norm, q_norm, k, part, ntiles);
    else
        hipLaunchKernelGGL((topk_small_tile_kernel<CK, 8>), dim3((unsigned)ntiles, (unsigned)groups8), dim3(TS_TI), 0,
                           stream, Q, nq, Y, ni, f, mask_indptr, mask_indices, y_norm, q_norm, k, part, ntiles);
}

}  // namespace mirec

extern "C" size_t mirec_topk_small_workspace_bytes(int64_t nq, int64_t ni, int k) {
    if (nq <= 0 || ni < 0 || k < 1) return 0;
    const int64_t ntiles = (ni + mirec::TS_TI - 1) / mirec::TS_TI;
    const size_t need = (size_t)nq * (size_t)k * (size_t)ntiles * sizeof(uint64_t);
    return need < 256 ? 256 : need;
}

extern "C" int mirec_topk_small(const float* Q, int64_t nq, const float* Y, int64_t ni, int f,
                                const int64_t* mask_indptr, const int32_t* mask_indices, const float* y_norm,
                                const float* q_norm, int k, int32_t* out_ids, float* out_scores, void* ws,
                                size_t ws_bytes, mirec_stream_t stream_) {
    using namespace mirec;
    hipStream_t stream = (hipStream_t)stream_;
    MIREC_REQUIRE(nq >= 0 && ni >= 0, MIREC_E_INVALID, "mirec_topk_small: negative size");
    MIREC_REQUIRE(k >= 1, MIREC_E_INVALID, "mirec_topk_small: k must be >= 1 (got %d)", k);
    MIREC_REQUIRE(k <= 224, MIREC_E_UNSUPPORTED, "mirec_topk_small: k = %d exceeds the supported maximum of 224", k);
    if (nq == 0) return MIREC_OK;
    MIREC_REQUIRE(nq <= MIREC_TOPK_SMALL_MAX_QUERIES, MIREC_E_UNSUPPORTED,
                  "mirec_topk_small: %lld queries; beyond %d use mirec_topk", (long long)nq, MIREC_TOPK_SMALL_MAX_QUERIES);
    MIREC_REQUIRE(Q && out_ids && out_scores && (ni == 0 || Y), MIREC_E_INVALID, "mirec_topk_small: null argument");
    MIREC_REQUIRE(ni < ((int64_t)1 << 31), MIREC_E_UNSUPPORTED, "mirec_topk_small: more than 2^31 items");
    MIREC_REQUIRE(f >= 8 && f % 8 == 0, MIREC_E_UNSUPPORTED,
                  "mirec_topk_small: factor width %d is not a multiple of 8; zero-pad the columns", f);
    MIREC_REQUIRE(((uintptr_t)Y & 15) == 0 && ((uintptr_t)Q & 3) == 0, MIREC_E_INVALID,
                  "mirec_topk_small: Y must be 16-byte aligned");
    const size_t need = mirec_topk_small_workspace_bytes(nq, ni, k);
    MIREC_REQUIRE(ws != nullptr && ((uintptr_t)ws & 15) == 0 && ws_bytes >= need, MIREC_E_WORKSPACE,
                  "mirec_topk_small: workspace of %zu bytes needed (16-byte aligned), %zu given", need, ws_bytes);
    const int64_t ntiles = (ni + TS_TI - 1) / TS_TI;
    uint64_t* part = static_cast<uint64_t*>(ws);
    if (ntiles > 0) {
        const int64_t groups8 = (nq + 7) / 8;
        if (f % 32 == 0)
            launch_tiles<32>(groups8, Q, nq, Y, ni, f, mask_indptr, mask_indices, y_norm, q_norm, k, part, ntiles, stream);
        else if (f % 16 == 0)
            launch_tiles<16>(groups8, Q, nq, Y, ni, f, mask_indptr, mask_indices, y_norm, q_norm, k, part, ntiles, stream);
        else
            launch_tiles<8>(groups8, Q, nq, Y, ni, f, mask_indptr, mask_indices, y_norm, q_norm, k, part, ntiles, stream);
        MIREC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(topk_small_merge_kernel, dim3((unsigned)nq), dim3(256), 0, stream, part, ntiles, k, out_ids,
                       out_scores);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}
