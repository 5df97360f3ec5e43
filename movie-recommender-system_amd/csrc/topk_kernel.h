// Fused score + top-K: scores = Q . Y^T on the fp32 matrix cores, selection in
// LDS, nothing but the k winners per query ever reaches memory.
//
// A workgroup owns 32*WAVES query rows and sweeps all items in tiles of 32
// rows of Y staged in LDS.  Each wavefront keeps its 32 queries as MFMA A
// operands in registers and produces a 32 x 32 score tile per item tile with
// v_mfma_f32_32x32x2_f32, which evaluates, per (query, item), exactly the chain
// s = fma(q[k], y[k], s) for ascending k.
//
// Selection state is private to a wavefront (no cross-wave traffic): per query
// a buffer of CAP (score, item) entries, of which the first k after a
// compaction are the current best, and a threshold tau = score of the k-th
// best, mirrored in registers.  A score enters the buffer only if it beats tau
// (one LDS atomic reserves the slot); when a reservation fails the full buffers
// are compacted by a wave-wide rank sort under the order (score desc, item
// asc), which also raises tau, and the push is retried (CAP >= k + 32, so all
// 32 lanes of a tile row fit after a compaction).  Tiles arrive in ascending index order, so an item that ties the
// k-th best of earlier tiles never displaces it and the register test `score > tau` is exact; within a tile the
// push order is arbitrary, so the push itself compares (score, index) with the k-th best.  Liked items are
// found by a per-query cursor over the (ascending) mask row and turned into
// -FLT_MAX only when they come up as candidates.
//
// Two workgroups share a CU at the usual widths, so one wavefront's selection
// and staging overlap the other's MFMA phase.
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
    constexpr int E = (CAP + 63) / 64;
    int n = __builtin_amdgcn_readfirstlane(*cntp);
    n = n < CAP ? n : CAP;  // failed reservations may have pushed the counter past the capacity
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

constexpr int TOPK_BI = 32;  // item rows per tile

template <int F, int WAVES, int CAP>
struct TopkLds {
    static constexpr int LDY = F + 1;  // odd stride: column reads of a tile hit 32 distinct banks
    static constexpr int YT = TOPK_BI * LDY;
    static constexpr int PER_WAVE = 32 * CAP * 2 + 32 * 3;  // scores, ids, counts, thresholds, liked bits
    static constexpr size_t BYTES = (size_t)(YT + WAVES * PER_WAVE) * 4;
};

template <int F, int WAVES, int CAP>
__global__ __launch_bounds__(64 * WAVES) void topk_kernel(
    const float* __restrict__ Q, int64_t nq, const float* __restrict__ Y, int64_t ni,
    const int64_t* __restrict__ mask_indptr, const int32_t* __restrict__ mask_indices,
    const float* __restrict__ y_norm, const float* __restrict__ q_norm, int k, int32_t* __restrict__ out_ids,
    float* __restrict__ out_scores, const int32_t* __restrict__ qlist, const int32_t* __restrict__ qcount) {
    // List mode (qlist != nullptr): the workgroups process the *qcount queries whose indices are listed in
    // qlist, in list order; everything per query (factor row, mask row, norm, output row) is addressed
    // through the list.  The launch covers the worst case, surplus workgroups leave at once.
    using L = TopkLds<F, WAVES, CAP>;
    constexpr int KS = F / 2;
    constexpr int BI = TOPK_BI;
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
    unsigned* lmask = reinterpret_cast<unsigned*>(tau + 32);  // [32]: liked bits of the current item tile

    const int64_t u0 = (int64_t)blockIdx.x * (32 * WAVES) + w * 32;  // first query (slot) of this wave
    if (qlist != nullptr) {
        nq = *qcount;
        if ((int64_t)blockIdx.x * (32 * WAVES) >= nq) return;
    }
    auto query_of = [&](int64_t slot) __attribute__((always_inline)) {
        return (qlist != nullptr && slot < nq) ? (int64_t)qlist[slot] : slot;
    };
    if (lane < 32) {
        cnt[lane] = 0;
        tau[lane] = -INFINITY;
        lmask[lane] = 0u;
    }

    // A operands: query row u0 + c, elements k = 2s + h
    float qa[KS];
    {
        const int64_t u = u0 + c;
        const bool uv = u < nq;
        const float* qr = Q + (uv ? query_of(u) : 0) * F + h;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float v = qr[2 * s];
            qa[s] = uv ? v : 0.f;
        }
    }

    // Score acc[i] of this lane belongs to query row (i&3) + 8*(i>>2) + 4h; its threshold and divisor:
    float tau_r[16];
    float qn_r[16];
    bool urow_ok[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t ug = u0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        tau_r[i] = -INFINITY;
        urow_ok[i] = ug < nq;
        qn_r[i] = (q_norm != nullptr && ug < nq) ? q_norm[query_of(ug)] : 1.f;
    }

    // Liked-item filter: lane u < 32 walks the (ascending) mask row of query u0 + u with a cursor, one
    // entry prefetched, and turns the entries that fall into the current item tile into a bit mask.
    int64_t cur = 0, cend = 0;
    int nxt = 0x7fffffff;
    if (mask_indptr != nullptr && lane < 32 && u0 + lane < nq) {
        const int64_t uq = query_of(u0 + lane);
        cur = mask_indptr[uq];
        cend = mask_indptr[uq + 1];
        if (cur < cend) nxt = mask_indices[cur];
    }

    // Item tiles are fetched two tiles ahead into registers (pre0 / pre1 alternate), so that an L2 or
    // Infinity-Cache round trip is covered by two tiles of MFMA work even at small widths.
    f32x4 pre0[NV], pre1[NV];
    auto load_y = [&](f32x4 (&pre)[NV], int64_t i0) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            // unconditional (a branch around a load makes the compiler wait for it on the spot): threads past
            // the tile repeat its last element, rows past the last item repeat the last item (never candidates)
            int e = tid + v * NTHR;
            e = e < TILE_V4 ? e : TILE_V4 - 1;
            const int j = e / F4;
            const int k4 = e % F4;
            int64_t item = i0 + j;
            item = item < ni ? item : ni - 1;
            pre[v] = *reinterpret_cast<const f32x4*>(Y + (item > 0 ? item : 0) * F + 4 * k4);
        }
    };
    auto store_y = [&](const f32x4 (&pre)[NV]) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int e = tid + v * NTHR;
            if (e < TILE_V4) {
                const int j = e / F4;
                const int k4 = e % F4;
                float* d = ytile + j * LDY + 4 * k4;
                d[0] = pre[v][0];
                d[1] = pre[v][1];
                d[2] = pre[v][2];
                d[3] = pre[v][3];
            }
        }
    };

    // MFMA + selection of the tile that sits in LDS
    auto process_tile = [&](int64_t i0) __attribute__((always_inline)) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        // B operands are fetched one batch of k-steps ahead of the MFMAs that consume them; the scheduling
        // barriers keep the compiler from sinking each read next to its use (which would stall every MFMA
        // pair on an LDS round trip).
        {
            constexpr int KB = 8;                  // k-steps per batch
            constexpr int NB = (KS + KB - 1) / KB;
            const float* yrow = ytile + c * LDY + h;
            float bq[2][KB];
#pragma unroll
            for (int j = 0; j < KB; ++j) bq[0][j] = (j < KS) ? yrow[2 * j] : 0.f;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                if (bt + 1 < NB) {
#pragma unroll
                    for (int j = 0; j < KB; ++j) {
                        const int s = (bt + 1) * KB + j;
                        bq[(bt + 1) & 1][j] = (s < KS) ? yrow[2 * s] : 0.f;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    const int s = bt * KB + j;
                    if (s < KS) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[s], bq[bt & 1][j], acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        if (mask_indptr != nullptr) {
            if (lane < 32) {
                unsigned mw = 0u;
                const int64_t lim = i0 + BI;
                while ((int64_t)nxt < lim) {
                    const int bit = nxt - (int)i0;
                    if (bit >= 0) mw |= 1u << bit;
                    ++cur;
                    nxt = cur < cend ? mask_indices[cur] : 0x7fffffff;
                }
                lmask[lane] = mw;
            }
            __builtin_amdgcn_wave_barrier();
        }

        // ---- selection: acc[i] = score(query u0 + row(i,h), item i0 + c)
        const int64_t item = i0 + c;
        const bool item_ok = item < ni;
        float yn = 1.f;
        if (y_norm != nullptr) yn = y_norm[item_ok ? item : 0];
        unsigned pm = 0u;
        float sc[16];
        if (y_norm != nullptr || q_norm != nullptr) {
            // similar-items mode only; the empty asm keeps this a real (wave-uniform) branch: if-converted,
            // the 32 IEEE divisions would run on every tile of every plain scoring pass
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[i] = (acc[i] / yn) / qn_r[i];
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[i] = acc[i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) pm |= (item_ok && urow_ok[i] && sc[i] > tau_r[i]) ? (1u << i) : 0u;
        // Candidates: a lane rarely holds more than one per tile, so they are taken one at a time (lowest
        // row first) with a select chain.  A slot is reserved with one LDS atomic; when the buffer is full
        // the push fails, the full buffers are compacted (which also raises tau) and the lane retries.
        while (__ballot(pm != 0u) != 0ull) {
            bool failed = false;
            if (pm != 0u) {
                const int i = __builtin_ctz(pm);
                float s = sc[0];
#pragma unroll
                for (int j = 1; j < 16; ++j) {
                    s = (i == j) ? sc[j] : s;
                    asm volatile("" : "+v"(s));  // keep it a select chain (not a spilled table lookup)
                }
                const int ur = (i & 3) + 8 * (i >> 2) + 4 * h;
                if (mask_indptr != nullptr && ((lmask[ur] >> c) & 1u)) s = -FLT_MAX;
                // A compaction in the middle of a tile can leave a later item of this same tile as the k-th
                // best; an equal score with a smaller index still beats it.
                const float t = tau[ur];
                if (s > t || (s == t && (int)item < sel_i[ur * CAP + k - 1])) {
                    const int pos = atomicAdd(&cnt[ur], 1);
                    if (pos < CAP) {
                        sel_s[ur * CAP + pos] = s;
                        sel_i[ur * CAP + pos] = (int)item;
                    } else {
                        failed = true;
                    }
                }
                if (!failed) pm &= pm - 1u;
            }
            if (__ballot(failed) != 0ull) {
                __builtin_amdgcn_wave_barrier();
                unsigned long long bal = __ballot(lane < 32 && cnt[lane] >= CAP);
                while (bal != 0ull) {
                    const int u = __builtin_ctzll(bal);
                    bal &= bal - 1ull;
                    compact_wave<CAP>(sel_s + u * CAP, sel_i + u * CAP, cnt + u, tau + u, k, lane);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) tau_r[i] = tau[(i & 3) + 8 * (i >> 2) + 4 * h];
            }
        }
    };

    if (ni > 0) {
        load_y(pre0, 0);
        load_y(pre1, BI);
    }
    for (int64_t i0 = 0; i0 < ni; i0 += 2 * BI) {
        __syncthreads();  // previous tile fully consumed
        store_y(pre0);
        __syncthreads();
        load_y(pre0, i0 + 2 * BI);
        process_tile(i0);
        if (i0 + BI >= ni) break;
        __syncthreads();
        store_y(pre1);
        __syncthreads();
        load_y(pre1, i0 + 3 * BI);
        process_tile(i0 + BI);
    }

    // ---- final compaction and output
    for (int ur = 0; ur < 32; ++ur) {
        if (u0 + ur >= nq) break;
        const int64_t ug = query_of(u0 + ur);
        compact_wave<CAP>(sel_s + ur * CAP, sel_i + ur * CAP, cnt + ur, tau + ur, k, lane);
        const int n = __builtin_amdgcn_readfirstlane(cnt[ur]);
        for (int j = lane; j < k; j += 64) {
            out_ids[ug * k + j] = j < n ? sel_i[ur * CAP + j] : -1;
            out_scores[ug * k + j] = j < n ? sel_s[ur * CAP + j] : -FLT_MAX;
        }
    }
}

template <int F, int WAVES, int CAP>
static int launch_topk_cfg(const float* Q, int64_t nq, const float* Y, int64_t ni, const int64_t* mask_indptr,
                           const int32_t* mask_indices, const float* y_norm, const float* q_norm, int k,
                           int32_t* out_ids, float* out_scores, const int32_t* qlist, const int32_t* qcount,
                           hipStream_t stream) {
    using L = TopkLds<F, WAVES, CAP>;
    static OncePerDevice once;  // idempotent; a race only repeats the call
    bool* attr_done = once.slot();
    MIREC_REQUIRE(attr_done != nullptr, MIREC_E_HIP, "cannot read the current device ordinal (or it is >= %d)", kMaxDevices);
    if (!*attr_done) {
        MIREC_HIP(hipFuncSetAttribute((const void*)topk_kernel<F, WAVES, CAP>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES));
        *attr_done = true;
    }
    const int64_t blocks = (nq + 32 * WAVES - 1) / (32 * WAVES);
    hipLaunchKernelGGL((topk_kernel<F, WAVES, CAP>), dim3((unsigned)blocks), dim3(64 * WAVES), L::BYTES, stream, Q, nq,
                       Y, ni, mask_indptr, mask_indices, y_norm, q_norm, k, out_ids, out_scores, qlist, qcount);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}

// Buffer size by k: CAP - 32 >= k.  56 entries (k <= 24) lets two 4-wave workgroups share a CU.
template <int F>
int launch_topk_fused(const float* Q, int64_t nq, const float* Y, int64_t ni, const int64_t* mask_indptr,
                      const int32_t* mask_indices, const float* y_norm, const float* q_norm, int k, int32_t* out_ids,
                      float* out_scores, const int32_t* qlist, const int32_t* qcount, hipStream_t stream) {
    if (k <= 24)
        return launch_topk_cfg<F, 4, 56>(Q, nq, Y, ni, mask_indptr, mask_indices, y_norm, q_norm, k, out_ids,
                                         out_scores, qlist, qcount, stream);
    if (k <= 96)
        return launch_topk_cfg<F, 4, 128>(Q, nq, Y, ni, mask_indptr, mask_indices, y_norm, q_norm, k, out_ids,
                                          out_scores, qlist, qcount, stream);
    return launch_topk_cfg<F, 2, 256>(Q, nq, Y, ni, mask_indptr, mask_indices, y_norm, q_norm, k, out_ids,
                                      out_scores, qlist, qcount, stream);
}

}  // namespace mirec
