// ALS half-sweep, matrix-free form, one wavefront per row.
//
// When a row has far fewer entries than F^2/16, forming A_u explicitly costs
// more than applying it: A_u v = G v + sum_k w_k (y_k . v) y_k needs one pass
// over the row's gathered factor rows per product, and the solver needs four
// (the start residual and three CG steps).  This kernel runs that form for
// every row with at most `short_max` entries.  It pays off when the gathered
// table (the other side's factor matrix) stays resident in L2 / Infinity
// Cache, as the item factors do in the user half-sweep; rows beyond
// `short_max` go to the explicit-Gram kernel in als_sweep.hip.
//
// Layout: persistent workgroups of 16 wavefronts (8 beyond F = 128); G (F x F)
// sits in LDS, shared by all of them; the rest of the LDS is split into one
// cache per wavefront that keeps the first CAP gathered rows of the current
// CSR row, so short rows are fetched once and re-read on chip.  Within a
// wavefront the two 32-lane halves take alternate gathered rows (16-byte
// loads at F = 128: lane c owns columns [c*NT, c*NT+NT)), GV rows each per
// step; their GV dot products are reduced across the 32 lanes by one
// reduce-scatter butterfly (about 2 shuffles per row instead of 5), after
// which a group of 32/GV adjacent lanes holds row j's dot.
#include "als_internal.h"

namespace mirec {

// Wavefronts per workgroup and gathered rows per half-wave per step.  The kernel is bound by latency and by
// LDS instruction issue, not by registers well spent: measured on the 1 M x 27 k user half-sweep at F = 128,
// (waves, rows/step) = (8,16) 12.8 ms, (8,8) 14.3, (12,8) 10.7, (16,8) 10.5 -- so up to F = 128 the kernel runs
// 16 waves (4 per SIMD, 128 VGPRs) with 8 rows per step; wider rows keep 8 waves and 16 rows per step.
#ifndef MIREC_SHORT_GV
#define MIREC_SHORT_GV 8
#endif
constexpr int GV = MIREC_SHORT_GV;  // gathered rows per half-wave per step (8 or 16)
template <int NT>
struct ShortWaves {
    static constexpr int SW = NT <= 4 ? 16 : 8;
};

template <int NT>
struct ShortLds {
    static constexpr int SW = ShortWaves<NT>::SW;
    static constexpr int F = 32 * NT;
    static constexpr int TOTAL = 160 * 1024 / 4;                       // floats of LDS per CU
    static constexpr int AVAIL = TOTAL - F * F - SW * F - 64;
    static constexpr int CAP_RAW = AVAIL / (SW * F);
    static constexpr int CAP = CAP_RAW >= 64 ? 64 : (CAP_RAW & ~1);    // even: both halves agree on "cached"
    static constexpr int FLOATS = F * F + SW * (CAP * F + F);
    static constexpr size_t BYTES = (size_t)FLOATS * 4;
};

template <int NT>
struct VecN {
    float v[NT];
};

template <int NT>
__device__ __forceinline__ void load_n(float (&dst)[NT], const float* src) {
    if constexpr (NT % 4 == 0) {
#pragma unroll
        for (int q = 0; q < NT / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(src + 4 * q);
            dst[4 * q] = t.x; dst[4 * q + 1] = t.y; dst[4 * q + 2] = t.z; dst[4 * q + 3] = t.w;
        }
    } else if constexpr (NT % 2 == 0) {
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) {
            const float2 t = *reinterpret_cast<const float2*>(src + 2 * q);
            dst[2 * q] = t.x; dst[2 * q + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int q = 0; q < NT; ++q) dst[q] = src[q];
    }
}

template <int NT>
__device__ __forceinline__ void store_n(float* dst, const float (&src)[NT]) {
    if constexpr (NT % 4 == 0) {
#pragma unroll
        for (int q = 0; q < NT / 4; ++q)
            *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(src[4 * q], src[4 * q + 1], src[4 * q + 2], src[4 * q + 3]);
    } else if constexpr (NT % 2 == 0) {
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) *reinterpret_cast<float2*>(dst + 2 * q) = make_float2(src[2 * q], src[2 * q + 1]);
    } else {
#pragma unroll
        for (int q = 0; q < NT; ++q) dst[q] = src[q];
    }
}

// Sum over the 32 lanes of a half-wave (all lanes get the result).
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// GV values per lane -> after the butterfly, lane c of each half holds the 32-lane sum of value
// (c >> 1) & 15 (GV = 16) or (c >> 2) & 7 (GV = 8).
__device__ __forceinline__ float reduce_scatter(float (&p)[GV], int c) {
    if constexpr (GV == 16) {
        const bool up = (c & 16) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float keep = up ? p[i + 8] : p[i];
            const float send = up ? p[i] : p[i + 8];
            p[i] = keep + __shfl_xor(send, 16, 64);
        }
    }
    constexpr int S1 = GV == 16 ? 8 : 16;  // lane bit that halves 8 values to 4
    {
        const bool up = (c & S1) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float keep = up ? p[i + 4] : p[i];
            const float send = up ? p[i] : p[i + 4];
            p[i] = keep + __shfl_xor(send, S1, 64);
        }
    }
    {
        const bool up = (c & (S1 >> 1)) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float keep = up ? p[i + 2] : p[i];
            const float send = up ? p[i] : p[i + 2];
            p[i] = keep + __shfl_xor(send, S1 >> 1, 64);
        }
    }
    {
        const bool up = (c & (S1 >> 2)) != 0;
        const float keep = up ? p[1] : p[0];
        const float send = up ? p[0] : p[1];
        p[0] = keep + __shfl_xor(send, S1 >> 2, 64);
    }
    float d = p[0];
#pragma unroll
    for (int off = S1 >> 3; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
    return d;
}

constexpr int DSH = GV == 16 ? 1 : 2;  // lane c holds the dot of gathered row (c >> DSH) of its half's step

// acc (per half: the sum over that half's gathered rows) of
//   FIRST:  (max(c,0) - (|c|-1)(y.v)) y        (start residual, v = x)
//   else :  (|c|-1)(y.v) y                     (operator, v = p)
template <int NT, int CAP, bool FIRST>
__device__ __forceinline__ void sparse_pass(const float (&v)[NT], float (&acc)[NT], const int32_t* __restrict__ indices,
                                            const float* __restrict__ data, const float* __restrict__ Y, float* cache,
                                            int n, int lane, int h, int c) {
    constexpr int F = 32 * NT;
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = 0.f;
    for (int g0 = 0; g0 < n; g0 += 2 * GV) {
        const bool need_idx = FIRST || (g0 + 2 * GV > CAP);
        int myidx = 0;
        if (need_idx) {
            const int myk = g0 + (lane & 31);
            if (myk < n) myidx = indices[myk];
        }
        float yv[GV][NT];
        float part[GV];
#pragma unroll
        for (int j = 0; j < GV; ++j) {
            const int e = g0 + 2 * j;  // wave-uniform; this half takes e + h
            const int kk = e + h;
            const bool cached = !FIRST && (e < CAP);
            if (cached) {
                if (kk < n) {
                    load_n<NT>(yv[j], cache + kk * F + c * NT);
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t) yv[j][t] = 0.f;
                }
            } else {
                const int idx = __shfl(myidx, 2 * j + h, 64);
                if (kk < n) {
                    load_n<NT>(yv[j], Y + (int64_t)idx * F + c * NT);
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t) yv[j][t] = 0.f;
                }
                if (FIRST && e < CAP && kk < n) store_n<NT>(cache + kk * F + c * NT, yv[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < GV; ++j) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) s = fmaf(yv[j][t], v[t], s);
            part[j] = s;
        }
        const float d = reduce_scatter(part, c);
        const int kkm = g0 + 2 * (c >> DSH) + h;
        float coef = 0.f;
        if (kkm < n) {
            const float cv = data[kkm];
            const float w = fabsf(cv) - 1.f;
            coef = FIRST ? ((cv > 0.f ? cv : 0.f) - w * d) : (w * d);
        }
#pragma unroll
        for (int j = 0; j < GV; ++j) {
            const float cj = __shfl(coef, (h << 5) | (j << DSH), 64);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = fmaf(cj, yv[j][t], acc[t]);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] += __shfl_xor(acc[t], 32, 64);
}

// out = G v (both halves end with the full result); the k range is split over the halves.
template <int NT>
__device__ __forceinline__ void symv_short(const float* Gs, float* vec, const float (&v)[NT], float (&out)[NT], int h,
                                           int c) {
    constexpr int F = 32 * NT;
    __builtin_amdgcn_wave_barrier();
    if (h == 0) store_n<NT>(vec + c * NT, v);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < NT; ++t) out[t] = 0.f;
    const int k0 = h * (F / 2);
#pragma unroll 2
    for (int k = k0; k < k0 + F / 2; k += 4) {
        const float4 pk = *reinterpret_cast<const float4*>(vec + k);
        float g[4][NT];
#pragma unroll
        for (int q = 0; q < 4; ++q) load_n<NT>(g[q], Gs + (k + q) * F + c * NT);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            out[t] = fmaf(g[0][t], pk.x, out[t]);
            out[t] = fmaf(g[1][t], pk.y, out[t]);
            out[t] = fmaf(g[2][t], pk.z, out[t]);
            out[t] = fmaf(g[3][t], pk.w, out[t]);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) out[t] += __shfl_xor(out[t], 32, 64);
}

template <int NT>
__global__ __launch_bounds__(64 * ShortWaves<NT>::SW) void als_short_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ data,
    int64_t row_begin, int64_t nrows, float* X, const float* __restrict__ Y, const float* __restrict__ G,
    int32_t short_max, int cg_steps) {
    constexpr int F = 32 * NT;
    using L = ShortLds<NT>;
    constexpr int CAP = L::CAP;
    constexpr int SW = L::SW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Gs = smem;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int h = lane >> 5;
    const int c = lane & 31;
    float* cache = smem + F * F + w * (CAP * F);
    float* vec = smem + F * F + SW * (CAP * F) + w * F;

    for (int i = tid * 4; i < F * F; i += 64 * SW * 4)
        *reinterpret_cast<float4*>(Gs + i) = *reinterpret_cast<const float4*>(G + i);
    __syncthreads();

    const int64_t stride = (int64_t)gridDim.x * SW;
    for (int64_t r = (int64_t)blockIdx.x * SW + w; r < nrows; r += stride) {
        const int64_t row = row_begin + r;
        const int64_t rs = indptr[row];
        const int64_t nn = indptr[row + 1] - rs;
        if (nn > (int64_t)short_max) continue;
        const int n = (int)nn;
        float* xg = X + row * F + c * NT;
        float x[NT], rr[NT], p[NT], Ap[NT], acc[NT];
        if (n == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) x[t] = 0.f;
            if (h == 0) store_n<NT>(xg, x);
            continue;
        }
        load_n<NT>(x, xg);
        const int32_t* ind = indices + rs;
        const float* dat = data + rs;

        sparse_pass<NT, CAP, true>(x, acc, ind, dat, Y, cache, n, lane, h, c);
        symv_short<NT>(Gs, vec, x, Ap, h, c);
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            rr[t] = acc[t] - Ap[t];
            p[t] = rr[t];
            part = fmaf(rr[t], rr[t], part);
        }
        float rsold = half_sum(part);
        if (rsold < 1e-20f) continue;
        for (int it = 0; it < cg_steps; ++it) {
            sparse_pass<NT, CAP, false>(p, acc, ind, dat, Y, cache, n, lane, h, c);
            symv_short<NT>(Gs, vec, p, Ap, h, c);
            part = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                Ap[t] += acc[t];
                part = fmaf(p[t], Ap[t], part);
            }
            const float alpha = rsold / half_sum(part);
            part = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                x[t] = fmaf(alpha, p[t], x[t]);
                rr[t] = fmaf(-alpha, Ap[t], rr[t]);
                part = fmaf(rr[t], rr[t], part);
            }
            const float rsnew = half_sum(part);
            if (rsnew < 1e-20f) break;
            const float beta = rsnew / rsold;
#pragma unroll
            for (int t = 0; t < NT; ++t) p[t] = fmaf(beta, p[t], rr[t]);
            rsold = rsnew;
        }
        if (h == 0) store_n<NT>(xg, x);
    }
}

template <int NT>
static int launch_short(const int64_t* indptr, const int32_t* indices, const float* data, int64_t row_begin,
                        int64_t nrows, float* X, const float* Y, const float* G, int32_t short_max, int cg_steps,
                        hipStream_t stream) {
    using L = ShortLds<NT>;
    constexpr int SW = L::SW;
    static bool attr_done = false;  // idempotent; a race only repeats the call
    static int cus = 0;
    if (!attr_done) {
        MIREC_HIP(hipFuncSetAttribute((const void*)als_short_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)L::BYTES));
        int dev = 0;
        MIREC_HIP(hipGetDevice(&dev));
        MIREC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        attr_done = true;
    }
    int64_t blocks = (nrows + SW - 1) / SW;
    if (blocks > cus) blocks = cus;
    if (blocks < 1) return MIREC_OK;
    hipLaunchKernelGGL(als_short_kernel<NT>, dim3((unsigned)blocks), dim3(64 * SW), L::BYTES, stream, indptr, indices,
                       data, row_begin, nrows, X, Y, G, short_max, cg_steps);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}

int launch_als_short(int nt, const int64_t* indptr, const int32_t* indices, const float* data, int64_t row_begin,
                     int64_t nrows, float* X, const float* Y, const float* G, int32_t short_max, int cg_steps,
                     hipStream_t stream) {
    switch (nt) {
        case 1: return launch_short<1>(indptr, indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps, stream);
        case 2: return launch_short<2>(indptr, indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps, stream);
        case 3: return launch_short<3>(indptr, indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps, stream);
        case 4: return launch_short<4>(indptr, indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps, stream);
        case 5: return launch_short<5>(indptr, indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps, stream);
        case 6: return launch_short<6>(indptr, indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps, stream);
    }
    set_error("launch_als_short: unsupported width");
    return MIREC_E_UNSUPPORTED;
}

int als_short_cache_rows(int nt) {
    switch (nt) {
        case 1: return ShortLds<1>::CAP;
        case 2: return ShortLds<2>::CAP;
        case 3: return ShortLds<3>::CAP;
        case 4: return ShortLds<4>::CAP;
        case 5: return ShortLds<5>::CAP;
        case 6: return ShortLds<6>::CAP;
    }
    return 0;
}

}  // namespace mirec
