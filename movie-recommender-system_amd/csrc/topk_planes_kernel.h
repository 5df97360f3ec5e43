// Fused score + top-K on the bf16 matrix pipe with exact fp32 results (factor
// widths that are multiples of 16 up to 128, k <= 24, plain scoring, caller
// workspace given).
//
// The fp32 MFMA keeps the vector ALU of its SIMD busy for its whole duration,
// so selection work cannot hide behind it.  Here the scores that drive the
// selection come from v_mfma_f32_32x32x16_bf16 instead: both factor matrices are
// split into two bf16 planes (x ~ hi + mid, 16 significant bits) and a tile
// accumulates the three plane products hi*hi, hi*mid and mid*hi.  Such a score
// s~ differs from the reference chain s = fma(q[f-1], y[f-1], ... ) by at most
// eps = (c_f * 2^-24 + 2^-17 + 2^-18) * ||q|| * max||y||, far below the spacing
// of the scores around rank k.  The selection is therefore run with a margin
// and the survivors are scored again, exactly:
//
//   * tau~ = k-th best s~ seen by a query so far.  At most k - 1 items have
//     s > s_(k), so tau~ <= s_(k) + eps at all times, while every member of the
//     exact top k has s~ >= s_(k) - eps: a buffer that holds every item with
//     s~ > tau~ - 2 eps holds the exact top k, ties at the boundary included.
//   * At the end the buffered candidates (k plus the few inside the margin) are
//     scored with the fp32 fma chain in ascending k -- bit for bit what the fp32
//     MFMA kernel computes -- and the k best under (score desc, item asc) are
//     written.
//   * A query whose margin set outgrows its buffer (heavy ties), whose margin is
//     not finite, or whose filter leaves fewer than k items is not answered here:
//     it is appended to a device-side list and the fp32 kernel answers it in list
//     mode afterwards.  Liked items are simply never buffered.
//
// A workgroup owns 32*WM query rows.  Its WM "matrix" waves keep 32 queries each
// as split MFMA A operands in registers, sweep the items in tiles of 32 and do
// nothing but plane staging (global -> registers -> LDS, double buffered),
// operand reads, 3*F/16 MFMAs and four 16-byte stores of the score tile per item
// tile.  Two "selector" waves per matrix wave consume the score tile of the
// previous item tile and own all selection state: thresholds in registers, fill
// counts as packed scalars, the candidates of a query row straight from the
// compare's lane mask (which also assigns their slots -- no atomics), liked
// items from a per-query cursor fed by a small LDS ring, buffers compacted by a
// digit-serial k-th-best search (no sort) when a tile's candidates would not
// fit and on a tile schedule that all selectors share.  One workgroup barrier
// per item tile separates producers and consumers of the double-buffered images.
#pragma once
#include "topk_kernel.h"
#include "topk_workspace.h"

namespace mirec {

// Optional phase timing (-DMIREC_DEBUG_TIMING): cycle counts of one matrix wave and one selector wave of every
// workgroup, summed into a per-translation-unit counter array; the F = 128 unit exports
// mirec_topk_debug_read (tools/probes/topk_phase_timing.py).
#ifdef MIREC_DEBUG_TIMING
static __device__ unsigned long long mirec_topk_dbg[16];
static __device__ int mirec_topk_dbg_mode;  // bits 8 and up: the workgroup whose tiles are traced
static __device__ unsigned mirec_topk_trace[20][1024];  // per tile, of workgroup mirec_topk_dbg_mode >> 8
#define TKS_T() __builtin_readcyclecounter()
#else
#define TKS_T() 0ull
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 tk_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 tk_bf16x2;
typedef __attribute__((ext_vector_type(2))) float tk_f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t tk_u32x4;  // one MFMA operand: 8 bf16

__device__ __forceinline__ uint32_t tk_cvt_pk_bf16(float a, float b) {
    const tk_f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, tk_bf16x2));
}

// (a, b) -> packed bf16 pairs hi, mid with |a - hi.a - mid.a| <= 2^-18 |a| (two roundings to nearest)
__device__ __forceinline__ void tk_split2(float a, float b, uint32_t& hi, uint32_t& mid) {
    hi = tk_cvt_pk_bf16(a, b);
    a -= __uint_as_float(hi << 16);
    b -= __uint_as_float(hi & 0xffff0000u);
    mid = tk_cvt_pk_bf16(a, b);
}

// eight consecutive factors (a, b) -> one MFMA operand per plane
__device__ __forceinline__ void tk_split8(const f32x4& a, const f32x4& b, tk_u32x4& hi, tk_u32x4& mid) {
    uint32_t h[4], m[4];
    tk_split2(a[0], a[1], h[0], m[0]);
    tk_split2(a[2], a[3], h[1], m[1]);
    tk_split2(b[0], b[1], h[2], m[2]);
    tk_split2(b[2], b[3], h[3], m[3]);
    hi = tk_u32x4{h[0], h[1], h[2], h[3]};
    mid = tk_u32x4{m[0], m[1], m[2], m[3]};
}

__device__ __forceinline__ f32x16 tk_mma(const tk_u32x4& a, const tk_u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tk_bf16x8, a), __builtin_bit_cast(tk_bf16x8, b), c,
                                                   0, 0, 0);
}

// |s~ - s| <= topk_eps_coef<F>() * ||q|| * max||y||: the factors' third bf16 planes, which are dropped
// (|x - hi - mid| <= 2^-18 |x| on either side: 2^-17 in all), the mid * mid plane product, which is not
// formed (|mid| <= 2^-9 |x|: 2^-18), the F roundings of the fp32 chain and the
// roundings of the matrix pipe's accumulation -- the latter with a wide safety factor on the typical
// (random-walk) error, which is below 2^-24 of sum |q_k y_k|.
template <int F>
constexpr float topk_eps_coef() {
    return ((float)(8 * F + 64) * 5.9604644775390625e-8f + 7.62939453125e-6f + 3.814697265625e-6f) * 1.01f;
}

// ---------------------------------------------------------------------------------------------------
// Item factors -> bf16 planes (one workgroup per item tile, one thread per 8 consecutive factors) and
// max ||y|| (atomic max on the bits of the non-negative norm; a NaN norm has the largest bits).
template <int F>
__global__ __launch_bounds__(32 * (F / 8)) void topk_split_items_kernel(const float* __restrict__ Y, int64_t ni,
                                                                       uint32_t* __restrict__ planes,
                                                                       uint32_t* __restrict__ ymax_bits) {
    constexpr int F8 = F / 8;
    __shared__ float part[32 * F8];
    __shared__ float rown[32];
    const int tid = threadIdx.x;
    const int j = tid / F8;
    const int kc = tid % F8;
    const int64_t item = (int64_t)blockIdx.x * 32 + j;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (item < ni) {
        a = *reinterpret_cast<const f32x4*>(Y + item * F + 8 * kc);
        b = *reinterpret_cast<const f32x4*>(Y + item * F + 8 * kc + 4);
    }
    tk_u32x4 hi, mid;
    tk_split8(a, b, hi, mid);
    // [tile][plane][32][F] bf16 = [tile][plane][32][F8] chunks of 16 bytes
    tk_u32x4* dst = reinterpret_cast<tk_u32x4*>(planes) + (size_t)blockIdx.x * (TOPK_NPLANES * 32 * F8) + j * F8 + kc;
    dst[0] = hi;
    dst[32 * F8] = mid;
    part[tid] = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3] + b[0] * b[0] + b[1] * b[1] + b[2] * b[2] +
                b[3] * b[3];
    __syncthreads();
    if (tid < 32) {
        float s = 0.f;
        for (int q = 0; q < F8; ++q) s += part[tid * F8 + q];
        rown[tid] = fabsf(sqrtf(s));
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t m = 0u;
        for (int q = 0; q < 32; ++q) {
            const uint32_t v = __float_as_uint(rown[q]);
            m = v > m ? v : m;
        }
        atomicMax(ymax_bits, m);
    }
}

// Sort one query's buffer (n <= CAP entries, wave-uniform n) under (score desc, item asc) and keep the sorted
// prefix of the entries that beat  (k-th best score) - marg,  at most `limit` of them.  Returns the number
// kept; thr_out = the new entry threshold, over = the margin set did not fit.
template <int CAP>
__device__ __forceinline__ int compact_margin(float* ss, int* ii, int n_in, int k, int lane, float marg, int limit,
                                              float& thr_out, bool& over) {
    constexpr int E = (CAP + 63) / 64;
    const int n = __builtin_amdgcn_readfirstlane(n_in);
    __builtin_amdgcn_wave_barrier();
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
    float tau = -INFINITY;
    if (n >= k) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const unsigned long long mm = __ballot(lane + 64 * e < n && rank[e] == k - 1);
            if (mm != 0ull)
                tau = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ms[e]), __builtin_ctzll(mm)));
        }
    }
    const float thr = tau - marg;
    int keepn = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) keepn += __builtin_popcountll(__ballot(lane + 64 * e < n && ms[e] > thr));
    const int nk = n < k ? n : k;
    keepn = keepn > nk ? keepn : nk;  // a zero margin (zero query row: every score is exactly 0) keeps the k best
    over = keepn > limit;
    const int keep = keepn < limit ? keepn : limit;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = lane + 64 * e;
        if (idx < n && rank[e] < keep) {
            ss[rank[e]] = ms[e];
            ii[rank[e]] = mi[e];
        }
    }
    __builtin_amdgcn_wave_barrier();
    thr_out = thr;
    return keep;
}

// Sweep-time compaction of one query's buffer (n <= CAP <= 64 entries, one per lane, n >= k): find the k-th
// best score tau by a digit-serial search on an order-preserving integer image of the scores (three compares
// and population counts per two bits, no sort), then keep -- in place, in buffer order, which is ascending item
// order and stays so -- the entries that beat tau - marg; among entries equal to tau the earliest ones
// complete the k when the margin does not cover them (zero margin).  At most `limit` entries are kept.
// Returns the number kept; thr_out = the new entry threshold, over = the margin set did not fit.
template <int CAP>
__device__ __forceinline__ int compact_select(float* ss, int* ii, int n_in, int k, int lane, float marg, int limit,
                                              float& thr_out, bool& over) {
    static_assert(CAP <= 64, "one entry per lane");
    const int n = __builtin_amdgcn_readfirstlane(n_in);
    __builtin_amdgcn_wave_barrier();
    const bool valid = lane < n;
    const float sv = valid ? ss[lane] : -INFINITY;
    const int iv = valid ? ii[lane] : 0;
    const uint32_t bits = __float_as_uint(sv);
    const uint32_t key = valid ? (bits ^ ((uint32_t)((int32_t)bits >> 31) | 0x80000000u)) : 0u;
    // Two bits per step: the three trial values are independent compares.  The search stops at bit 12: the
    // result is then a lower bound of the k-th best score, at most 2^-11 below it in relative terms, which is
    // all a threshold needs (the entries between the two are simply kept).
    auto count_ge = [&](uint32_t trial) __attribute__((always_inline)) {
        return __builtin_popcountll(__builtin_amdgcn_uicmp(key, trial, 35 /* unsigned >= */));
    };
    uint32_t prefix = 0u;
#pragma unroll
    for (int b = 30; b >= 12; b -= 2) {
        const int c1 = count_ge(prefix | (1u << b));
        const int c2 = count_ge(prefix | (2u << b));
        const int c3 = count_ge(prefix | (3u << b));
        const uint32_t add = c3 >= k ? 3u : (c2 >= k ? 2u : (c1 >= k ? 1u : 0u));
        prefix |= add << b;
    }
    const uint32_t tb = (prefix & 0x80000000u) ? (prefix ^ 0x80000000u) : ~prefix;
    const float tau = __uint_as_float(tb);
    const float thr = tau - marg;
    const unsigned long long gtm = __ballot(valid && sv > tau);
    const unsigned long long eqm = __ballot(valid && sv == tau);
    const int need = k - __builtin_popcountll(gtm);
    const int eq_rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(eqm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)eqm, 0u));
    const bool keep = valid && (sv > thr || (sv == tau && eq_rank < need));
    const unsigned long long km = __ballot(keep);
    const int keepn = __builtin_popcountll(km);
    over = keepn > limit;
    const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u));
    __builtin_amdgcn_wave_barrier();
    if (keep && pos < limit) {
        ss[pos] = sv;
        ii[pos] = iv;
    }
    __builtin_amdgcn_wave_barrier();
    thr_out = thr;
    return keepn < limit ? keepn : limit;
}

// Keep the best min(n, k) entries of one query's buffer (n <= CAP, wave-uniform), sorted, at the front;
// returns the new fill count.
template <int CAP>
__device__ __forceinline__ int compact_sorted(float* ss, int* ii, int n_in, int k, int lane) {
    constexpr int E = (CAP + 63) / 64;
    const int n = __builtin_amdgcn_readfirstlane(n_in);
    __builtin_amdgcn_wave_barrier();
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
    return n < k ? n : k;
}

constexpr int TOPK_PLANES_SEL = 2;      // selector waves per matrix wave
constexpr int TOPK_EARLY_COMPACT = 8;   // entries beyond k that make a buffer worth compacting on the schedule

template <int F, int WM, int CAP>
struct TopkPlanesLds {
    static constexpr int SEL = TOPK_PLANES_SEL;
    static constexpr int NW = WM * (1 + SEL);
    // One plane of an item tile: 32 rows of F bf16, padded by 16 bytes: the row stride in dwords is an odd
    // multiple of four, which puts the 16 rows of every ds_read_b128 group on disjoint banks.
    static constexpr int ROWB = 2 * F + 16;
    static constexpr int PLANEB = TOPK_BI * ROWB;
    static constexpr int TILEB = TOPK_NPLANES * PLANEB;
    static constexpr int ST = 64 * 16;  // floats of one score tile: [i >> 2][lane][i & 3]
    // per matrix wave: scores, ids, liked bits, count / threshold / margin words, liked-item rings
    static constexpr int RING = 8;  // liked-item look-ahead per query (a power of two, at least 8)
    static constexpr int PER_M = 32 * CAP * 2 + 32 * 4 + 32 * RING;
    static constexpr size_t BYTES = (size_t)2 * TILEB + (size_t)(2 * WM * ST + WM * PER_M) * 4;
};

template <int F, int WM, int CAP>
__global__ __launch_bounds__(64 * WM * (1 + TOPK_PLANES_SEL)) void topk_planes_kernel(
    const float* __restrict__ Q, int64_t nq, const float* __restrict__ Y, int64_t ni,
    const uint32_t* __restrict__ planes, const uint32_t* __restrict__ ymax_bits,
    const int64_t* __restrict__ mask_indptr, const int32_t* __restrict__ mask_indices, int k,
    int32_t* __restrict__ out_ids, float* __restrict__ out_scores, int32_t* __restrict__ qlist,
    int32_t* __restrict__ qcount) {
    using L = TopkPlanesLds<F, WM, CAP>;
    constexpr int KS = F / 16;  // MFMA k-steps
    constexpr int F8 = F / 8;   // 16-byte chunks per plane row
    constexpr int BI = TOPK_BI;
    constexpr int SEL = L::SEL;
    constexpr int NROW = 16 / SEL;  // score rows (of each half-wave) per selector

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* ybuf = reinterpret_cast<char*>(smem);
    float* sbuf = smem + 2 * L::TILEB / 4;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int h = lane >> 5;
    const int c = lane & 31;
    const int ntile = (int)((ni + BI - 1) / BI);

    if (w < WM) {
        // ------------------------------------------------------------ matrix wave
        const int64_t u0 = (int64_t)blockIdx.x * (32 * WM) + w * 32;
        // A operands: query row u0 + c, factors 16*ks + 8*h .. + 7, as two bf16 planes
        tk_u32x4 qh[KS], qm[KS];
        {
            const int64_t u = u0 + c;
            const bool uv = u < nq;
            const float* qr = Q + (uv ? u : 0) * F + 8 * h;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                f32x4 a = *reinterpret_cast<const f32x4*>(qr + 16 * s);
                f32x4 b = *reinterpret_cast<const f32x4*>(qr + 16 * s + 4);
                if (!uv) {
                    a = f32x4{0.f, 0.f, 0.f, 0.f};
                    b = a;
                }
                tk_split8(a, b, qh[s], qm[s]);
            }
        }
        // Plane staging by the matrix waves (the selectors' time per tile varies, the matrix waves' does not):
        // tile t + 2 is in registers during the MFMAs of tile t and goes to LDS at the start of tile t + 1.
        constexpr int NTS = 64 * WM;
        constexpr int TILE_CH = TOPK_NPLANES * BI * F8;  // 16-byte chunks of one tile
        constexpr int NV = (TILE_CH + NTS - 1) / NTS;
        tk_u32x4 pre[NV];
        // Unconditional loads (a branch around a load makes the compiler wait for it on the spot): threads past
        // the tile repeat its last chunk.
        auto load_y = [&](int t) __attribute__((always_inline)) {
            const tk_u32x4* src = reinterpret_cast<const tk_u32x4*>(planes) + (size_t)t * TILE_CH;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                int e = tid + v * NTS;
                e = e < TILE_CH ? e : TILE_CH - 1;
                pre[v] = src[e];
            }
        };
        auto store_y = [&](char* ytile) __attribute__((always_inline)) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                int e = tid + v * NTS;
                e = e < TILE_CH ? e : TILE_CH - 1;  // duplicates write the same values
                const int row = e / F8;             // plane * 32 + item
                const int kc = e % F8;
                *reinterpret_cast<tk_u32x4*>(ytile + row * L::ROWB + 16 * kc) = pre[v];
            }
        };
        if (ntile > 0) {
            load_y(0);
            store_y(ybuf);
        }
        load_y(ntile > 1 ? 1 : 0);
        __syncthreads();  // tile 0 is staged
        unsigned long long dm = 0, db = 0;
        for (int t = 0; t < ntile; ++t) {
            const unsigned long long t0 = TKS_T();
            // rows past the last tile repeat it: the loads stay unconditional, their stores land in the idle image
            store_y(ybuf + ((t + 1) & 1) * L::TILEB);
            load_y(t + 2 < ntile ? t + 2 : ntile - 1);
            const char* yb = ybuf + (t & 1) * L::TILEB + c * L::ROWB + 16 * h;
            f32x16 acc0, acc1;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc0[i] = 0.f;
                acc1[i] = 0.f;
            }
            // B operands are fetched one k-step ahead of the MFMAs that consume them
            tk_u32x4 bq[2][TOPK_NPLANES];
            auto fetch = [&](tk_u32x4 (&dst)[TOPK_NPLANES], int s) __attribute__((always_inline)) {
#pragma unroll
                for (int p = 0; p < TOPK_NPLANES; ++p) dst[p] = *reinterpret_cast<const tk_u32x4*>(yb + p * L::PLANEB + 32 * s);
            };
            fetch(bq[0], 0);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) fetch(bq[(s + 1) & 1], s + 1);
                __builtin_amdgcn_sched_barrier(0);
                const tk_u32x4(&b)[TOPK_NPLANES] = bq[s & 1];
                // mid * mid is below 2^-18 of |q_k y_k| and left to the margin
                if (s & 1) {
                    acc1 = tk_mma(qh[s], b[0], acc1);  // hi hi
                    acc0 = tk_mma(qh[s], b[1], acc0);  // hi mid
                    acc1 = tk_mma(qm[s], b[0], acc1);  // mid hi
                } else {
                    acc0 = tk_mma(qh[s], b[0], acc0);  // hi hi
                    acc1 = tk_mma(qh[s], b[1], acc1);  // hi mid
                    acc0 = tk_mma(qm[s], b[0], acc0);  // mid hi
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            float* st = sbuf + ((t & 1) * WM + w) * L::ST + lane * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(st + q * 256) =
                    f32x4{acc0[4 * q] + acc1[4 * q], acc0[4 * q + 1] + acc1[4 * q + 1], acc0[4 * q + 2] + acc1[4 * q + 2],
                          acc0[4 * q + 3] + acc1[4 * q + 3]};
            const unsigned long long t1 = TKS_T();
            __syncthreads();  // score tile t is complete; plane tile t + 1 is staged
            dm += t1 - t0;
            db += TKS_T() - t1;
#ifdef MIREC_DEBUG_TIMING
            if (lane == 0 && (int)blockIdx.x == (mirec_topk_dbg_mode >> 8) && t < 1024) {
                mirec_topk_trace[16 + w][t] = (unsigned)(t1 - t0);
                if (w == 0) mirec_topk_trace[8][t] = (unsigned)(TKS_T() - t0);
            }
#endif
        }
#ifdef MIREC_DEBUG_TIMING
        if (tid == 0) {
            atomicAdd(&mirec_topk_dbg[0], dm);
            atomicAdd(&mirec_topk_dbg[1], db);
            atomicAdd(&mirec_topk_dbg[2], (unsigned long long)ntile);
        }
#endif
        return;
    }

    // ---------------------------------------------------------------- selector wave
    const int sw = w - WM;             // selector index in the workgroup
    const int m = sw / SEL;            // the matrix wave it serves
    const int r0 = (sw % SEL) * NROW;  // its score rows: i in [r0, r0 + NROW)
    const int64_t u0 = (int64_t)blockIdx.x * (32 * WM) + m * 32;
    float* sel_s = smem + 2 * L::TILEB / 4 + 2 * WM * L::ST + m * L::PER_M;
    int* sel_i = reinterpret_cast<int*>(sel_s + 32 * CAP);
    unsigned* lmask = reinterpret_cast<unsigned*>(sel_i + 32 * CAP);  // [32]: liked bits of the current item tile
    int* st_cnt = reinterpret_cast<int*>(lmask + 32);                 // [32]: fill counts
    float* st_thr = reinterpret_cast<float*>(st_cnt + 32);            // [32]: entry thresholds
    float* st_marg = st_thr + 32;                                     // [32]: margins
    int* ring = reinterpret_cast<int*>(st_marg + 32);                 // [32][RING]: upcoming liked items per query
    constexpr int RING = L::RING;
    auto row_of = [&](int i, int hh) __attribute__((always_inline)) { return (i & 3) + 8 * (i >> 2) + 4 * hh; };

    // Score row i of this lane belongs to query row_of(i, h).  The entry threshold (k-th best score so far minus
    // the margin) of a row is identical in the 32 lanes of a half-wave and mirrored in registers for the
    // per-tile compare; rows past the last query get +inf: nothing ever beats it.  Counts, thresholds and
    // margins live in LDS: they are touched only for rows that have a candidate.
    float thr_r[NROW];
    {
        const float ymax = __uint_as_float(*ymax_bits);
#pragma unroll
        for (int j = 0; j < NROW; ++j) {
            const int ur = row_of(r0 + j, h);
            const int64_t ug = u0 + ur;
            const bool uv = ug < nq;
            float s = 0.f;
#pragma unroll
            for (int e = c; e < F; e += 32) {
                const float v = Q[(uv ? ug : 0) * F + e];
                s += v * v;
            }
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            // zero only for a zero query row or all-zero items (all scores exactly 0 on both pipes); otherwise
            // at least the absolute error that products in the denormal range can add
            const float qn = sqrtf(s);
            const float marg = (qn > 0.f && ymax > 0.f)
                                   ? 2.f * topk_eps_coef<F>() * qn * ymax + (float)(8 * F + 64) * 2.4e-38f
                                   : qn * ymax;
            thr_r[j] = uv ? -INFINITY : INFINITY;
            if (c == 0) {
                st_marg[ur] = marg;
                st_thr[ur] = thr_r[j];
                st_cnt[ur] = 0;
            }
        }
    }
    unsigned ovf = 0u;  // wave-uniform, bit ur: the margin set of that query outgrew its buffer

    // Liked-item filter: lane l < 2*NROW walks the (ascending) mask row of one of this selector's queries.  The
    // row's upcoming entries sit in a small LDS ring, [cur, tail) of the row, refilled four at a time by loads
    // that are issued with the plane staging and committed a tile later -- the walk itself never waits for
    // memory unless one query has more liked items in a tile than the ring delivers.
    const int my_ur = row_of(r0 + (lane % NROW), (lane / NROW) & 1);
    int* my_ring = ring + my_ur * RING;
    const int32_t* mrow = mask_indices;
    int cur = 0, tail = 0, mlen = 0, pend_base = -1;
    int nxt = 0x7fffffff, nxt2 = 0x7fffffff;  // the row's entries cur and cur + 1 (0x7fffffff: none, or not read yet)
    int rf[4] = {0, 0, 0, 0};
    const bool walker = mask_indptr != nullptr && lane < 2 * NROW && u0 + my_ur < nq;
    if (walker) {
        const int64_t b = mask_indptr[u0 + my_ur];
        mrow = mask_indices + b;
        mlen = (int)(mask_indptr[u0 + my_ur + 1] - b);
        tail = mlen < RING ? mlen : RING;
        for (int e = 0; e < tail; ++e) my_ring[e] = mrow[e];
        // read back from the ring, not from memory: a register that may still be waiting for a memory load
        // would put a wait for all loads in flight (the refill's among them) into the walk
        __builtin_amdgcn_wave_barrier();
        if (mlen > 0) nxt = my_ring[0];
        if (mlen > 1) nxt2 = my_ring[1];
    }
    if (lane < 2 * NROW) lmask[my_ur] = 0u;
    auto mask_refill_issue = [&]() __attribute__((always_inline)) {
        // room for four more and something left to fetch; loads are unconditional (clamped) inside the branch
        if (walker && tail < mlen && tail - cur <= RING - 4) {
            pend_base = tail;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = tail + e;
                rf[e] = mrow[idx < mlen ? idx : mlen - 1];
            }
        }
    };
    auto mask_refill_commit = [&]() __attribute__((always_inline)) {
        if (pend_base >= 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) my_ring[(pend_base + e) & (RING - 1)] = rf[e];
            int nt = pend_base + 4;
            nt = nt < mlen ? nt : mlen;
            tail = tail > nt ? tail : nt;
            pend_base = -1;
        }
    };

    const int stid = tid - 64 * WM;

    // Fill counts: wave-uniform, one byte per row, packed per half-wave (a count never exceeds CAP <= 255)
    static_assert(CAP <= 255, "fill counts are kept in bytes");
    unsigned cnt_p[2][NROW / 4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int q = 0; q < NROW / 4; ++q) cnt_p[hh][q] = 0u;
    auto cnt_get = [&](int hh, int j) __attribute__((always_inline)) { return (int)((cnt_p[hh][j >> 2] >> (8 * (j & 3))) & 0xffu); };
    auto cnt_add = [&](int hh, int j, int d) __attribute__((always_inline)) { cnt_p[hh][j >> 2] += (unsigned)d << (8 * (j & 3)); };
    // the same with a (wave-uniform) run-time row
    auto cnt_get_dyn = [&](int hh, int j) __attribute__((always_inline)) {
        unsigned wv = 0u;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < NROW / 4; ++q) wv = (a == hh && q == (j >> 2)) ? cnt_p[a][q] : wv;
        return (int)((wv >> (8 * (j & 3))) & 0xffu);
    };
    auto cnt_set_dyn = [&](int hh, int j, int v) __attribute__((always_inline)) {
        const unsigned sh = 8u * (unsigned)(j & 3);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < NROW / 4; ++q)
                cnt_p[a][q] = (a == hh && q == (j >> 2)) ? ((cnt_p[a][q] & ~(0xffu << sh)) | ((unsigned)v << sh)) : cnt_p[a][q];
    };
    int sched_next = 5;  // next item tile with a scheduled compaction: 5, 13, 33, 83, 208, 521, ...
    auto counts_to_lds = [&]() __attribute__((always_inline)) {
        if (lane == 0) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int j = 0; j < NROW; ++j) st_cnt[row_of(r0 + j, hh)] = cnt_get(hh, j);
        }
        __builtin_amdgcn_wave_barrier();
    };

    unsigned long long dbg_a = 0, dbg_b = 0;
    // Selection over the score tile of item tile tt
    auto select_tile = [&](int tt) __attribute__((always_inline)) {
        const int i0 = tt * BI;
        const int item = i0 + c;
        // the scores are requested first: their way from LDS overlaps the liked-item walk
        const float* sb = sbuf + ((tt & 1) * WM + m) * L::ST + lane * 4;
        f32x4 scv[NROW / 4];
#pragma unroll
        for (int q = 0; q < NROW / 4; ++q) scv[q] = *reinterpret_cast<const f32x4*>(sb + (r0 / 4 + q) * 256);
        bool tile_liked = false;  // wave-uniform: some query of this selector has a liked item in the tile
        unsigned mw = 0u;         // walker lane hh * NROW + j: liked bits of the tile for query row_of(r0 + j, hh)
        if (mask_indptr != nullptr) {
            if (lane < 2 * NROW) {
                const int lim = i0 + BI;
                bool dry = false;
                // The common case -- at most one liked item of a query in a tile -- without a loop and without
                // waiting for LDS: the entry after the one consumed is already in a register, the one after that is
                // requested now and not looked at before the next tile.
                bool more = false;
                if (nxt < lim) {
                    const int bit = nxt - i0;
                    if (bit >= 0) mw |= 1u << bit;
                    ++cur;
                    nxt = nxt2;
                    nxt2 = cur + 1 < tail ? my_ring[(cur + 1) & (RING - 1)] : 0x7fffffff;
                    more = nxt < lim || (nxt == 0x7fffffff && cur < mlen);
                }
                if (__ballot(more) != 0ull) {
                if (more && nxt == 0x7fffffff && cur < mlen) {  // not read when it was prefetched: it may be there by now
                    if (cur < tail)
                        nxt = my_ring[cur & (RING - 1)];
                    else
                        dry = true;
                }
                while (more && nxt < lim) {
                    const int bit = nxt - i0;
                    if (bit >= 0) mw |= 1u << bit;
                    ++cur;
                    nxt = nxt2;  // read from the ring one consumption ago
                    if (nxt == 0x7fffffff && cur < mlen) {  // not in the ring then: it may be by now
                        if (cur < tail)
                            nxt = my_ring[cur & (RING - 1)];
                        else
                            dry = true;
                    }
                    nxt2 = cur + 1 < tail ? my_ring[(cur + 1) & (RING - 1)] : 0x7fffffff;
                }
                // The ring ran dry inside a tile: fetch directly, and wait.  Kept out of the loop above so that
                // no wait for memory lands on its path.
                if (__ballot(dry) != 0ull) {
                    if (dry) {
                        for (;;) {
                            const int v = mrow[cur];
                            if (v >= lim) {
                                nxt = v;
                                nxt2 = 0x7fffffff;
                                tail = cur + 1;
                                break;
                            }
                            const int bit = v - i0;
                            if (bit >= 0) mw |= 1u << bit;
                            if (++cur >= mlen) {
                                nxt = 0x7fffffff;
                                nxt2 = 0x7fffffff;
                                tail = mlen;
                                break;
                            }
                        }
                    }
                }
                }
            }
            tile_liked = __ballot(mw != 0u) != 0ull;
        }
        float sc[NROW];
#pragma unroll
        for (int q = 0; q < NROW / 4; ++q) {
            sc[4 * q] = scv[q][0];
            sc[4 * q + 1] = scv[q][1];
            sc[4 * q + 2] = scv[q][2];
            sc[4 * q + 3] = scv[q][3];
        }
        dbg_a = TKS_T();
        const unsigned long long okm = (tt == ntile - 1) ? __ballot(item < ni) : ~0ull;
        // Row by row, the compare result is the lane mask of the candidates, which also gives every candidate its
        // slot; counts are scalars.  A row whose buffer cannot take its candidates is deferred to the code below.
        unsigned defer = 0u;
#pragma unroll
        for (int j = 0; j < NROW; ++j) {
            unsigned long long mm = __ballot(sc[j] > thr_r[j]) & okm;
            if (mm == 0ull) continue;
            const int ur = row_of(r0 + j, h);
            if (tile_liked) {
                const unsigned l0 = (unsigned)__builtin_amdgcn_readlane((int)mw, j);
                const unsigned l1 = (unsigned)__builtin_amdgcn_readlane((int)mw, NROW + j);
                mm &= ~((unsigned long long)l0 | ((unsigned long long)l1 << 32));
                if (mm == 0ull) continue;
            }
            const unsigned m0 = (unsigned)mm, m1 = (unsigned)(mm >> 32);
            const int pc0 = __builtin_popcount(m0), pc1 = __builtin_popcount(m1);
            const int n0 = cnt_get(0, j), n1 = cnt_get(1, j);
            if (n0 + pc0 > CAP || n1 + pc1 > CAP) {
                defer |= 1u << j;
                continue;
            }
            const int pos = h ? n1 + (int)__builtin_amdgcn_mbcnt_hi(m1, 0u) : n0 + (int)__builtin_amdgcn_mbcnt_lo(m0, 0u);
            if ((mm >> lane) & 1ull) {
                sel_s[ur * CAP + pos] = sc[j];
                sel_i[ur * CAP + pos] = item;
            }
            cnt_add(0, j, pc0);
            cnt_add(1, j, pc1);
        }
        // Buffers to compact now, bit 8*hh + j: those that cannot take this tile's candidates -- and, at tiles
        // that every selector of the workgroup computes alike (a geometric schedule), all that have gathered
        // enough new entries.  A compaction stalls every wave of the workgroup at the next barrier; on the
        // schedule the selectors pay that together instead of one after the other, and the forced ones become rare.
        dbg_b = TKS_T();
        unsigned todo = 0u;
        if (defer != 0u) {
            if (tile_liked) {  // the deferred rows take the liked bits from LDS
                if (lane < 2 * NROW) lmask[my_ur] = mw;
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int j = 0; j < NROW; ++j) {
                if (!((defer >> j) & 1u)) continue;
                // recount this row's candidates per half (uniform): a half is compacted only if it would overflow
                const unsigned long long mm = __ballot(sc[j] > thr_r[j] && item < ni &&
                                                       !(tile_liked && ((lmask[row_of(r0 + j, h)] >> c) & 1u)));
                if (cnt_get(0, j) + __builtin_popcount((unsigned)mm) > CAP) todo |= 1u << j;
                if (cnt_get(1, j) + __builtin_popcount((unsigned)(mm >> 32)) > CAP) todo |= 1u << (8 + j);
            }
        }
        if (tt == sched_next) {
            sched_next = tt * 5 / 2 + 1;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int j = 0; j < NROW; ++j)
                    if (cnt_get(hh, j) >= k + TOPK_EARLY_COMPACT) todo |= 1u << (8 * hh + j);
        }
        if (todo == 0u) return;
        while (todo != 0u) {
            const int b = __builtin_ctz(todo);
            todo &= todo - 1u;
            const int j = b & 7, hh = b >> 3;
            const int urh = row_of(r0 + j, hh);
            float tnew;
            bool over;
            const int newn = compact_select<CAP>(sel_s + urh * CAP, sel_i + urh * CAP, cnt_get_dyn(hh, j), k, lane,
                                                 st_marg[urh], CAP - 32, tnew, over);
            ovf |= over ? (1u << urh) : 0u;
            cnt_set_dyn(hh, j, newn);
            if (lane == 0) st_thr[urh] = tnew;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < NROW; ++j) thr_r[j] = st_thr[row_of(r0 + j, h)];
        // the deferred rows' candidates, against the raised thresholds
        while (defer != 0u) {
            const int j = __builtin_ctz(defer);
            defer &= defer - 1u;
            const int ur = row_of(r0 + j, h);
            const float s = sb[((r0 + j) >> 2) * 256 + ((r0 + j) & 3)];
            bool cand = item < ni && s > st_thr[ur];
            if (tile_liked) cand = cand && !((lmask[ur] >> c) & 1u);
            const unsigned long long mm = __ballot(cand);
            if (mm == 0ull) continue;
            const unsigned m0 = (unsigned)mm, m1 = (unsigned)(mm >> 32);
            const int n0 = cnt_get_dyn(0, j), n1 = cnt_get_dyn(1, j);
            const int pos = h ? n1 + (int)__builtin_amdgcn_mbcnt_hi(m1, 0u) : n0 + (int)__builtin_amdgcn_mbcnt_lo(m0, 0u);
            if (cand) {
                sel_s[ur * CAP + pos] = s;
                sel_i[ur * CAP + pos] = item;
            }
            cnt_set_dyn(0, j, n0 + __builtin_popcount(m0));
            cnt_set_dyn(1, j, n1 + __builtin_popcount(m1));
        }
    };

    __syncthreads();  // tile 0 is staged
    unsigned long long ds = 0, dsel = 0, dbar = 0, dst = 0;
    for (int t = 0; t < ntile; ++t) {
        const unsigned long long t0 = TKS_T();
        mask_refill_commit();
        const unsigned long long t05 = TKS_T();
        mask_refill_issue();
        const unsigned long long t1 = TKS_T();
        dst += t05 - t0;
        if (t >= 1) select_tile(t - 1);
        const unsigned long long t2 = TKS_T();
        __syncthreads();  // score tile t is complete; plane tile t + 1 is staged
        ds += t1 - t0;
        dsel += t2 - t1;
        dbar += TKS_T() - t2;
#ifdef MIREC_DEBUG_TIMING
        if (lane == 0 && (int)blockIdx.x == (mirec_topk_dbg_mode >> 8) && t < 1024) {
            mirec_topk_trace[sw][t] = (unsigned)(t2 - t1);
            mirec_topk_trace[9 + (sw & 3)][t] = (unsigned)(t1 - t0);
            if (sw == 0 && t >= 1) {
                mirec_topk_trace[13][t] = (unsigned)(dbg_a - t1);
                mirec_topk_trace[14][t] = (unsigned)(dbg_b - dbg_a);
                mirec_topk_trace[15][t] = (unsigned)(t2 - dbg_b);
            }
        }
#endif
    }
#ifdef MIREC_DEBUG_TIMING
    if (stid == 0) {
        atomicAdd(&mirec_topk_dbg[4], ds);
        atomicAdd(&mirec_topk_dbg[5], dsel);
        atomicAdd(&mirec_topk_dbg[6], dbar);
        atomicAdd(&mirec_topk_dbg[7], (unsigned long long)ntile);
        atomicAdd(&mirec_topk_dbg[8], dst);
    }
#endif
    mask_refill_commit();
    if (ntile >= 1) select_tile(ntile - 1);

    // ---- this selector's queries: margin set -> exact scores -> the k best
    const unsigned long long t3 = TKS_T();
    counts_to_lds();
    for (int l = 0; l < 2 * NROW; ++l) {
        const int ur = row_of(r0 + (l % NROW), l / NROW);
        const int64_t ug = u0 + ur;
        if (ug >= nq) continue;
        const int n0 = __builtin_amdgcn_readfirstlane(st_cnt[ur]);
        const float marg = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(st_marg[ur])));
        const bool hand_over = ((ovf >> ur) & 1u) || !(marg < INFINITY) || (mask_indptr != nullptr && n0 < k);
        if (hand_over) {
            if (lane == 0) qlist[atomicAdd(qcount, 1)] = (int32_t)ug;
            continue;
        }
        float* ss = sel_s + ur * CAP;
        int* ii = sel_i + ur * CAP;
        float tnew;
        bool over;
        const int nm = compact_margin<CAP>(ss, ii, n0, k, lane, marg, CAP, tnew, over);
        // exact score of every survivor: the fp32 chain s = fma(q[k], y[k], s), ascending k, from +0
        for (int e0 = 0; e0 < nm; e0 += 64) {
            const int e = e0 + lane;
            const int id = e < nm ? ii[e] : 0;
            const float* yr = Y + (int64_t)id * F;
            const float* qr = Q + ug * F;
            float s = 0.f;
#pragma unroll 4
            for (int k4 = 0; k4 < F / 4; ++k4) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(qr + 4 * k4);
                const f32x4 yv = *reinterpret_cast<const f32x4*>(yr + 4 * k4);
                s = __builtin_fmaf(xv[0], yv[0], s);
                s = __builtin_fmaf(xv[1], yv[1], s);
                s = __builtin_fmaf(xv[2], yv[2], s);
                s = __builtin_fmaf(xv[3], yv[3], s);
            }
            if (e < nm) ss[e] = s;
        }
        const int n = compact_sorted<CAP>(ss, ii, nm, k, lane);
        for (int j = lane; j < k; j += 64) {
            out_ids[ug * k + j] = j < n ? ii[j] : -1;
            out_scores[ug * k + j] = j < n ? ss[j] : -FLT_MAX;
        }
    }
#ifdef MIREC_DEBUG_TIMING
    if (stid == 0) atomicAdd(&mirec_topk_dbg[9], TKS_T() - t3);
#endif
}

template <int F, int WM, int CAP>
static int launch_topk_planes_cfg(const float* Q, int64_t nq, const float* Y, int64_t ni, const int64_t* mask_indptr,
                                  const int32_t* mask_indices, int k, int32_t* out_ids, float* out_scores, void* ws,
                                  hipStream_t stream) {
    using L = TopkPlanesLds<F, WM, CAP>;
    static_assert(L::BYTES <= 160 * 1024, "top-K plane kernel: LDS image too large");
    static OncePerDevice once;  // idempotent; a race only repeats the call
    bool* attr_done = once.slot();
    MIREC_REQUIRE(attr_done != nullptr, MIREC_E_HIP, "cannot read the current device ordinal (or it is >= %d)", kMaxDevices);
    if (!*attr_done) {
        MIREC_HIP(hipFuncSetAttribute((const void*)topk_planes_kernel<F, WM, CAP>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES));
        *attr_done = true;
    }
    const TopkWorkspace lay = topk_workspace_layout(nq, ni, F);
    char* base = static_cast<char*>(ws);
    uint32_t* ymax_bits = reinterpret_cast<uint32_t*>(base);
    int32_t* qcount = reinterpret_cast<int32_t*>(base + 4);
    int32_t* qlist = reinterpret_cast<int32_t*>(base + lay.qlist_off);
    uint32_t* planes = reinterpret_cast<uint32_t*>(base + lay.planes_off);
    const int ntile = (int)((ni + TOPK_BI - 1) / TOPK_BI);

    MIREC_HIP(hipMemsetAsync(base, 0, 256, stream));
    hipLaunchKernelGGL((topk_split_items_kernel<F>), dim3((unsigned)ntile), dim3(32 * (F / 8)), 0, stream, Y, ni, planes,
                       ymax_bits);
    MIREC_LAUNCH_CHECK();
    const int64_t blocks = (nq + 32 * WM - 1) / (32 * WM);
    hipLaunchKernelGGL((topk_planes_kernel<F, WM, CAP>), dim3((unsigned)blocks), dim3(64 * L::NW), L::BYTES, stream, Q, nq,
                       Y, ni, planes, ymax_bits, mask_indptr, mask_indices, k, out_ids, out_scores, qlist, qcount);
    MIREC_LAUNCH_CHECK();
    // the queries handed over (none, as a rule): the fp32 kernel in list mode, sized for the worst case
    return launch_topk_fused<F>(Q, nq, Y, ni, mask_indptr, mask_indices, nullptr, nullptr, k, out_ids, out_scores, qlist,
                                qcount, stream);
}

// Entry of a width.  The bf16-plane kernel needs the caller's workspace; without one, for other widths, for
// k > 24, for the similar-items mode (per-score divisions) and for tiny item sets the fp32 kernel runs alone.
template <int F>
int launch_topk(const float* Q, int64_t nq, const float* Y, int64_t ni, const int64_t* mask_indptr,
                const int32_t* mask_indices, const float* y_norm, const float* q_norm, int k, int32_t* out_ids,
                float* out_scores, void* ws, size_t ws_bytes, hipStream_t stream) {
    if constexpr (F % 16 == 0 && F <= 128) {
        if (ws != nullptr && y_norm == nullptr && q_norm == nullptr && topk_planes_supported(F, k) && ni >= 2 * TOPK_BI) {
            MIREC_REQUIRE(ws_bytes >= topk_workspace_layout(nq, ni, F).bytes, MIREC_E_INVALID,
                          "mirec_topk: workspace of %zu bytes given, mirec_topk_workspace_bytes asks for %zu", ws_bytes,
                          topk_workspace_layout(nq, ni, F).bytes);
            MIREC_REQUIRE(((uintptr_t)ws & 255) == 0, MIREC_E_INVALID, "mirec_topk: workspace must be 256-byte aligned");
            return launch_topk_planes_cfg<F, 4, 64>(Q, nq, Y, ni, mask_indptr, mask_indices, k, out_ids, out_scores, ws,
                                                    stream);
        }
    }
    return launch_topk_fused<F>(Q, nq, Y, ni, mask_indptr, mask_indices, y_norm, q_norm, k, out_ids, out_scores, nullptr,
                                nullptr, stream);
}

}  // namespace mirec
