#include "topk_planes_kernel.h"

namespace mirec {
template int launch_topk<128>(const float*, int64_t, const float*, int64_t, const int64_t*, const int32_t*,
                              const int32_t*, const int32_t*, const int32_t*, const float*, const float*, int, int32_t*, float*, void*, size_t, hipStream_t);
}  // namespace mirec

