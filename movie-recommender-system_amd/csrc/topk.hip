// C entry point of the fused score + top-K; the kernel template lives in
// topk_kernel.h and is instantiated per factor width in topk_inst_*.hip.
#include "common.h"
#include "topk_workspace.h"

namespace mirec {
template <int F>
int launch_topk(const float* Q, int64_t nq, const float* Y, int64_t ni, const int64_t* mask_indptr,
                const int32_t* mask_indices, const int32_t* item_order, const int32_t* mask_blocks,
                const int32_t* block_len, const float* y_norm, const float* q_norm, int k, int32_t* out_ids,
                float* out_scores, void* ws, size_t ws_bytes, hipStream_t stream);
}  // namespace mirec

extern "C" size_t mirec_topk_workspace_bytes(int64_t nq, int64_t ni, int f, int k) {
    if (nq <= 0 || ni <= 0 || !mirec::topk_planes_supported(f, k)) return 0;
    return mirec::topk_workspace_layout(nq, ni, f).bytes;
}

static int topk_entry(const char* who, const float* Q, int64_t nq, const float* Y, int64_t ni, int f,
                      const int64_t* mask_indptr, const int32_t* mask_indices, const int32_t* item_order,
                      const int32_t* mask_blocks, const int32_t* block_len, const float* y_norm, const float* q_norm,
                      int k, int32_t* out_ids, float* out_scores, void* ws, size_t ws_bytes, hipStream_t stream) {
    MIREC_REQUIRE(nq >= 0 && ni >= 0, MIREC_E_INVALID, "%s: negative size", who);
    MIREC_REQUIRE(k >= 1, MIREC_E_INVALID, "%s: k must be >= 1 (got %d)", who, k);
    MIREC_REQUIRE(k <= 224, MIREC_E_UNSUPPORTED, "%s: k = %d exceeds the supported maximum of 224", who, k);
    if (nq == 0) return MIREC_OK;
    MIREC_REQUIRE(Q && out_ids && out_scores && (ni == 0 || Y), MIREC_E_INVALID, "%s: null argument", who);
    MIREC_REQUIRE(ni < ((int64_t)1 << 31), MIREC_E_UNSUPPORTED, "%s: more than 2^31 items", who);
    MIREC_REQUIRE(((uintptr_t)Y & 15) == 0 && ((uintptr_t)Q & 3) == 0, MIREC_E_INVALID, "%s: Y must be 16-byte aligned",
                  who);
    MIREC_REQUIRE(item_order == nullptr || mask_indptr == nullptr || (mask_blocks != nullptr && block_len != nullptr),
                  MIREC_E_INVALID,
                  "%s: an item order with a filter needs the filter as lists per 32 rows in positions of that order "
                  "(mirec_topk_order_build)", who);
#define MIREC_TOPK_CASE(FW)                                                                                     \
    case FW:                                                                                                    \
        return mirec::launch_topk<FW>(Q, nq, Y, ni, mask_indptr, mask_indices, item_order, mask_blocks, block_len, \
                                      y_norm, q_norm, k, out_ids, out_scores, ws, ws_bytes, stream);
    switch (f) {
        MIREC_TOPK_CASE(8)
        MIREC_TOPK_CASE(16)
        MIREC_TOPK_CASE(24)
        MIREC_TOPK_CASE(32)
        MIREC_TOPK_CASE(48)
        MIREC_TOPK_CASE(64)
        MIREC_TOPK_CASE(96)
        MIREC_TOPK_CASE(128)
        MIREC_TOPK_CASE(160)
        MIREC_TOPK_CASE(192)
    }
#undef MIREC_TOPK_CASE
    mirec::set_error("%s: factor width %d is not one of 8,16,24,32,48,64,96,128,160,192; zero-pad the columns", who, f);
    return MIREC_E_UNSUPPORTED;
}

extern "C" int mirec_topk(const float* Q, int64_t nq, const float* Y, int64_t ni, int f,
                          const int64_t* mask_indptr, const int32_t* mask_indices, const float* y_norm,
                          const float* q_norm, int k, int32_t* out_ids, float* out_scores, void* ws, size_t ws_bytes,
                          mirec_stream_t stream_) {
    return topk_entry("mirec_topk", Q, nq, Y, ni, f, mask_indptr, mask_indices, nullptr, nullptr, nullptr, y_norm, q_norm,
                      k, out_ids, out_scores, ws, ws_bytes, (hipStream_t)stream_);
}

// mirec_topk with the items swept in a caller-given order: the same results, less selection work when the order
// puts the items most queries score highest first (topk_order.hip).  The order only reaches the kernel that
// sweeps; where another kernel answers (no workspace, widths or k it does not serve, a handful of items) it is
// not looked at.
extern "C" int mirec_topk_ordered(const float* Q, int64_t nq, const float* Y, int64_t ni, int f,
                                  const int64_t* mask_indptr, const int32_t* mask_indices, const int32_t* item_order,
                                  const int32_t* mask_blocks, const int32_t* block_len, int k, int32_t* out_ids,
                                  float* out_scores, void* ws, size_t ws_bytes, mirec_stream_t stream_) {
    return topk_entry("mirec_topk_ordered", Q, nq, Y, ni, f, mask_indptr, mask_indices, item_order, mask_blocks,
                      block_len, nullptr, nullptr, k, out_ids, out_scores, ws, ws_bytes, (hipStream_t)stream_);
}
