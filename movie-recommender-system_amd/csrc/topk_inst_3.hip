#include "topk_planes_kernel.h"

namespace mirec {
template int launch_topk<128>(const float*, int64_t, const float*, int64_t, const int64_t*, const int32_t*,
                              const float*, const float*, int, int32_t*, float*, void*, size_t, hipStream_t);
}  // namespace mirec

#ifdef MIREC_DEBUG_TIMING
extern "C" int mirec_topk_debug_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mirec::mirec_topk_dbg), sizeof(unsigned long long) * 16) != hipSuccess)
        return -3;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mirec::mirec_topk_dbg), z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
#endif
