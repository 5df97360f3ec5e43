// Shared Gram matrix  G = Y^T Y + reg*I  on the fp32 matrix cores.
//
// Row blocks of Y are contracted with v_mfma_f32_32x32x2_f32 (an exact fp32
// fma chain over the rows), one 32-row band of G per wavefront; block partials
// are then summed in a fixed order (fp64 accumulation, one rounding), so the
// result does not depend on scheduling.
#include "common.h"

namespace mirec {

// Rows handled by one workgroup: at most ~1024 workgroups, at least 256 rows.
static inline int gram_rows_per_block(int64_t n) {
    int64_t r = (n + 1023) / 1024;
    if (r < 256) r = 256;
    r = (r + 7) / 8 * 8;
    return (int)r;
}

template <int NT>
__global__ __launch_bounds__(64 * NT) void gram_partial_kernel(const float* __restrict__ Y, int64_t n,
                                                               int rows_per_block,
                                                               float* __restrict__ partials) {
    constexpr int F = 32 * NT;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // band of G rows
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;  // which of the two rows of a k-step this lane feeds
    const int c = lane & 31;

    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < n) ? r0 + rows_per_block : n;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    for (int64_t r = r0; r < r1; r += 8) {
        float a[4];
        float b[4][NT];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int64_t row = r + 2 * p + h;
            const bool valid = row < r1;
            const float* yr = Y + (valid ? row : r0) * F;
            const float av = yr[w * 32 + c];
            a[p] = valid ? av : 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float bv = yr[t * 32 + c];
                b[p][t] = valid ? bv : 0.f;
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p], b[p][t], acc[t], 0, 0, 0);
    }

    float* out = partials + (size_t)blockIdx.x * F * F;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            out[(size_t)(w * 32 + row) * F + t * 32 + c] = acc[t][i];
        }
}

__global__ void gram_reduce_kernel(const float* __restrict__ partials, int nblk, int f, float reg,
                                   float* __restrict__ G) {
    const int ff = f * f;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ff) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)partials[(size_t)b * ff + e];
    if (e / f == e % f) s += (double)reg;
    G[e] = (float)s;
}

template <int NT>
static int launch_gram(const float* Y, int64_t n, float reg, float* G, float* partials, hipStream_t stream) {
    constexpr int F = 32 * NT;
    const int rpb = gram_rows_per_block(n);
    const int nblk = (int)((n + rpb - 1) / rpb);
    if (nblk > 0) {
        hipLaunchKernelGGL(gram_partial_kernel<NT>, dim3(nblk), dim3(64 * NT), 0, stream, Y, n, rpb, partials);
        MIREC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((F * F + 255) / 256), dim3(256), 0, stream, partials, nblk, F, reg,
                       G);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}

}  // namespace mirec

extern "C" size_t mirec_gram_workspace_bytes(int64_t n, int f) {
    if (n < 0 || f <= 0) return 0;
    const int rpb = mirec::gram_rows_per_block(n);
    const int64_t nblk = (n + rpb - 1) / rpb;
    return (size_t)(nblk > 0 ? nblk : 1) * f * f * sizeof(float);
}

extern "C" int mirec_gram(const float* Y, int64_t n, int f, float reg, float* G, void* ws, size_t ws_bytes,
                          mirec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MIREC_REQUIRE(G != nullptr && n >= 0, MIREC_E_INVALID, "mirec_gram: null output or negative n");
    MIREC_REQUIRE(n == 0 || Y != nullptr, MIREC_E_INVALID, "mirec_gram: null input");
    MIREC_REQUIRE(f % 32 == 0 && f >= 32 && f <= 192, MIREC_E_UNSUPPORTED,
                  "mirec_gram: factor width %d is not a multiple of 32 in [32,192]; zero-pad the columns", f);
    MIREC_REQUIRE(ws != nullptr && ws_bytes >= mirec_gram_workspace_bytes(n, f), MIREC_E_WORKSPACE,
                  "mirec_gram: workspace of %zu bytes needed, %zu given", mirec_gram_workspace_bytes(n, f), ws_bytes);
    float* partials = (float*)ws;
    switch (f / 32) {
        case 1: return mirec::launch_gram<1>(Y, n, reg, G, partials, stream);
        case 2: return mirec::launch_gram<2>(Y, n, reg, G, partials, stream);
        case 3: return mirec::launch_gram<3>(Y, n, reg, G, partials, stream);
        case 4: return mirec::launch_gram<4>(Y, n, reg, G, partials, stream);
        case 5: return mirec::launch_gram<5>(Y, n, reg, G, partials, stream);
        case 6: return mirec::launch_gram<6>(Y, n, reg, G, partials, stream);
    }
    return MIREC_E_UNSUPPORTED;
}
