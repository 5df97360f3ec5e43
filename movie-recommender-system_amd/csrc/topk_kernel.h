// Fused score + top-K: scores = Q . Y^T on the fp32 matrix cores, selection in
// LDS, nothing but the k winners per query ever reaches memory.
//
// A workgroup owns 32*WAVES query rows and sweeps all items in tiles of
// BI = 32*NTI rows of Y staged in LDS.  Each wavefront keeps its 32 queries as
// MFMA A operands in registers and produces a 32 x BI score tile per item tile
// with v_mfma_f32_32x32x2_f32, which evaluates, per (query, item), exactly the
// chain s = fma(q[k], y[k], s) for ascending k.  Every wavefront owns the
// selection state of its 32 queries (no cross-wave traffic): a buffer of CAP
// (score, item) entries per query, of which the first k after a compaction are
// the current best, and a threshold tau = score of the k-th best.  A score
// enters the buffer only if it beats tau; a full buffer is compacted by a
// wave-wide rank sort under the order (score desc, item asc).  Items arrive in
// ascending index order, so a later item that ties the k-th best never
// displaces it and `score > tau` is exact.
#pragma once
#include <float.h>
#include <math.h>

#include "common.h"

namespace mirec {

__device__ __forceinline__ bool better(float sa, int ia, float sb, int ib) {
    return sa > sb || (sa == sb && ia < ib);
}

// Keep the best min(n, k) entries of one query's buffer, sorted, at the front.
template <int CAP>
__device__ __forceinline__ void compact_wave(float* ss, int* ii, int* cntp, float* taup, int k, int lane) {
    constexpr int E = CAP / 64;
    const int n = __builtin_amdgcn_readfirstlane(*cntp);
    float ms[E];
    int mi[E];
    int rank[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = lane + 64 * e;
        ms[e] = idx < n ? ss[idx] : -INFINITY;
        mi[e] = idx < n ? ii[idx] : 0x7fffffff;
        rank[e] = 0;
    }
    for (int j = 0; j < n; ++j) {
        const float sj = ss[j];
        const int ij = ii[j];
#pragma unroll
        for (int e = 0; e < E; ++e) rank[e] += better(sj, ij, ms[e], mi[e]) ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = lane + 64 * e;
        if (idx < n && rank[e] < k) {
            ss[rank[e]] = ms[e];
            ii[rank[e]] = mi[e];
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int newn = n < k ? n : k;
    const float tau = newn >= k ? ss[k - 1] : -INFINITY;
    if (lane == 0) {
        *cntp = newn;
        *taup = tau;
    }
    __builtin_amdgcn_wave_barrier();
}

template <int F, int WAVES, int NTI, int CAP>
struct TopkLds {
    static constexpr int BI = 32 * NTI;
    static constexpr int LDY = F + 1;  // odd stride: column reads of a tile hit 32 distinct banks
    static constexpr int YT = BI * LDY;
    static constexpr int PER_WAVE = 32 * CAP * 2 + 64;  // scores, ids, counts, thresholds
    static constexpr size_t BYTES = (size_t)(YT + WAVES * PER_WAVE) * 4;
};

template <int F, int WAVES, int NTI, int CAP>
__global__ __launch_bounds__(64 * WAVES) void topk_kernel(
    const float* __restrict__ Q, int64_t nq, const float* __restrict__ Y, int64_t ni,
    const int64_t* __restrict__ mask_indptr, const int32_t* __restrict__ mask_indices,
    const float* __restrict__ y_norm, const float* __restrict__ q_norm, int k, int32_t* __restrict__ out_ids,
    float* __restrict__ out_scores) {
    using L = TopkLds<F, WAVES, NTI, CAP>;
    constexpr int KS = F / 2;
    constexpr int BI = L::BI;
    constexpr int LDY = L::LDY;
    constexpr int NTHR = 64 * WAVES;
    constexpr int TILE_V4 = BI * F / 4;
    constexpr int NV = (TILE_V4 + NTHR - 1) / NTHR;
    constexpr int F4 = F / 4;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ytile = smem;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int h = lane >> 5;
    const int c = lane & 31;
    float* sel_s = smem + L::YT + w * L::PER_WAVE;
    int* sel_i = reinterpret_cast<int*>(sel_s + 32 * CAP);
    int* cnt = sel_i + 32 * CAP;
    float* tau = reinterpret_cast<float*>(cnt + 32);

    const int64_t u0 = (int64_t)blockIdx.x * (32 * WAVES) + w * 32;  // first query of this wave
    if (lane < 32) {
        cnt[lane] = 0;
        tau[lane] = -INFINITY;
    }

    // A operands: query row u0 + c, elements k = 2s + h
    float qa[KS];
    {
        const int64_t u = u0 + c;
        const bool uv = u < nq;
        const float* qr = Q + (uv ? u : 0) * F + h;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float v = qr[2 * s];
            qa[s] = uv ? v : 0.f;
        }
    }

    float4 pre[NV];
    auto load_y = [&](int64_t i0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int e = tid + v * NTHR;
            const int j = e / F4;
            const int k4 = e % F4;
            const int64_t item = i0 + j;
            if (e < TILE_V4 && item < ni)
                pre[v] = *reinterpret_cast<const float4*>(Y + item * F + 4 * k4);
            else
                pre[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_y = [&]() {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int e = tid + v * NTHR;
            if (e < TILE_V4) {
                const int j = e / F4;
                const int k4 = e % F4;
                float* d = ytile + j * LDY + 4 * k4;
                d[0] = pre[v].x;
                d[1] = pre[v].y;
                d[2] = pre[v].z;
                d[3] = pre[v].w;
            }
        }
    };

    load_y(0);
    for (int64_t i0 = 0; i0 < ni; i0 += BI) {
        __syncthreads();  // previous tile fully consumed
        store_y();
        __syncthreads();
        if (i0 + BI < ni) load_y(i0 + BI);

        f32x16 acc[NTI];
#pragma unroll
        for (int t = 0; t < NTI; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int t = 0; t < NTI; ++t) {
                const float b = ytile[(t * 32 + c) * LDY + 2 * s + h];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[s], b, acc[t], 0, 0, 0);
            }
        }

        // ---- selection: acc[t][i] = score(query u0 + row(i,h), item i0 + 32t + c)
#pragma unroll
        for (int t = 0; t < NTI; ++t) {
            const int64_t item = i0 + t * 32 + c;
            const bool item_ok = item < ni;
            float yn = 1.f;
            if (y_norm != nullptr) yn = y_norm[item_ok ? item : 0];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rbase = (i & 3) + 8 * (i >> 2);
                const int ur = rbase + 4 * h;
                const int64_t ug = u0 + ur;
                float s = acc[t][i];
                if (y_norm != nullptr) s = s / yn;
                if (q_norm != nullptr) s = s / q_norm[ug < nq ? ug : 0];
                bool pass = item_ok && ug < nq && s > tau[ur];
                if (__ballot(pass) == 0ull) continue;
                if (pass && mask_indptr != nullptr) {
                    int64_t lo = mask_indptr[ug], hi = mask_indptr[ug + 1];
                    while (lo < hi) {
                        const int64_t mid = (lo + hi) >> 1;
                        if (mask_indices[mid] < (int32_t)item) lo = mid + 1; else hi = mid;
                    }
                    if (lo < mask_indptr[ug + 1] && mask_indices[lo] == (int32_t)item) {
                        s = -FLT_MAX;
                        pass = s > tau[ur];
                    }
                }
                unsigned long long bal = __ballot(pass);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    unsigned m = (unsigned)(bal >> (32 * half));
                    if (m == 0u) continue;
                    const int urh = rbase + 4 * half;
                    int base = __builtin_amdgcn_readfirstlane(cnt[urh]);
                    int n = __popc(m);
                    if (base + n > CAP) {
                        compact_wave<CAP>(sel_s + urh * CAP, sel_i + urh * CAP, cnt + urh, tau + urh, k, lane);
                        if (h == half) pass = pass && s > tau[urh];
                        bal = __ballot(pass);
                        m = (unsigned)(bal >> (32 * half));
                        n = __popc(m);
                        base = __builtin_amdgcn_readfirstlane(cnt[urh]);
                    }
                    if (h == half && pass) {
                        const int pos = base + __popc(m & ((1u << c) - 1u));
                        sel_s[urh * CAP + pos] = s;
                        sel_i[urh * CAP + pos] = (int)item;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) cnt[urh] = base + n;
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }

    // ---- final compaction and output
    for (int ur = 0; ur < 32; ++ur) {
        const int64_t ug = u0 + ur;
        if (ug >= nq) break;
        compact_wave<CAP>(sel_s + ur * CAP, sel_i + ur * CAP, cnt + ur, tau + ur, k, lane);
        const int n = __builtin_amdgcn_readfirstlane(cnt[ur]);
        for (int j = lane; j < k; j += 64) {
            out_ids[ug * k + j] = j < n ? sel_i[ur * CAP + j] : -1;
            out_scores[ug * k + j] = j < n ? sel_s[ur * CAP + j] : -FLT_MAX;
        }
    }
}

template <int F, int WAVES, int NTI, int CAP>
static int launch_topk_cfg(const float* Q, int64_t nq, const float* Y, int64_t ni, const int64_t* mask_indptr,
                           const int32_t* mask_indices, const float* y_norm, const float* q_norm, int k,
                           int32_t* out_ids, float* out_scores, hipStream_t stream) {
    using L = TopkLds<F, WAVES, NTI, CAP>;
    static bool attr_done = false;  // idempotent; a race only repeats the call
    if (!attr_done) {
        MIREC_HIP(hipFuncSetAttribute((const void*)topk_kernel<F, WAVES, NTI, CAP>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES));
        attr_done = true;
    }
    const int64_t blocks = (nq + 32 * WAVES - 1) / (32 * WAVES);
    hipLaunchKernelGGL((topk_kernel<F, WAVES, NTI, CAP>), dim3((unsigned)blocks), dim3(64 * WAVES), L::BYTES, stream,
                       Q, nq, Y, ni, mask_indptr, mask_indices, y_norm, q_norm, k, out_ids, out_scores);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}

template <int F>
int launch_topk(const float* Q, int64_t nq, const float* Y, int64_t ni, const int64_t* mask_indptr,
                       const int32_t* mask_indices, const float* y_norm, const float* q_norm, int k,
                       int32_t* out_ids, float* out_scores, hipStream_t stream) {
    if (k <= 32)
        return launch_topk_cfg<F, 4, 2, 64>(Q, nq, Y, ni, mask_indptr, mask_indices, y_norm, q_norm, k, out_ids,
                                            out_scores, stream);
    return launch_topk_cfg<F, 2, 1, 256>(Q, nq, Y, ni, mask_indptr, mask_indices, y_norm, q_norm, k, out_ids,
                                         out_scores, stream);
}

}  // namespace mirec

