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
//   * tau~ = a lower bound of the k-th best s~ seen by a query so far.  At most
//     k - 1 items have s > s_(k), so tau~ <= s_(k) + eps at all times, while every
//     member of the exact top k has s~ >= s_(k) - eps: a buffer that holds every
//     item with s~ > tau~ - 2 eps holds the exact top k, ties at the boundary
//     included.
//   * At the end the buffered candidates (k plus the few inside the margin) are
//     scored with the fp32 fma chain in ascending k -- bit for bit what the fp32
//     MFMA kernel computes -- and the k best under (score desc, item asc) are
//     written.
//   * A query whose margin set outgrows its buffer (heavy ties), whose margin is
//     zero or not finite (a zero query row, all-zero items, overflow), or whose
//     filter leaves fewer than k items is not answered here: it is appended to a
//     device-side list and the fp32 kernel answers it in list mode afterwards.
//     Liked items are simply never buffered.
//
// Every wavefront does the same work.  A workgroup of NW wavefronts owns 32*NW
// queries, 32 per wavefront, which the wavefront keeps as split MFMA B operands
// in registers; the item tiles (32 items, both planes) pass through a ring of
// images in LDS and are the A operands.  The 32 x 32 result then has one QUERY
// per lane column: lanes c and c + 32 hold the scores of query c for 16 items
// each.  Everything the selection needs is therefore per lane -- the entry
// threshold is one register, "does this tile hold a candidate" is a register
// maximum and one compare, and the candidates of different queries are appended
// side by side under the execution mask (one LDS atomic reserves the slot) -- so
// the scores never leave the registers and no wavefront waits for another's
// selection.  When an append finds a buffer full the wavefront compacts the
// buffers of all its queries at once, again one query per lane pair: a bisection
// on the value range finds a lower bound of the k-th best entry, the entries
// above it (minus the margin) are kept and the threshold is raised.
//
// No workgroup barrier inside the sweep: every wavefront publishes its progress
// in LDS (images stored, images read) and waits only for what it is about to
// touch, and the ring lets a wavefront that compacts fall an image behind
// without stopping the rest.
//
// Measured (1 M queries x 27 k items, F = 128, k = 20, liked-item filter on, one MI355X): 29.5 ms, against 33.1 ms
// for the previous design (four matrix wavefronts handing score tiles through LDS to eight selector wavefronts,
// item-per-lane).  Counters: 385 instructions per wavefront and tile (180 vector, 24 matrix, 137 scalar, 39 LDS),
// the matrix pipe 28 % busy -- the sweep is bound by instruction issue of the selection, not by the products.
// Builds that leave out one part each (wrong results, timing only; tools/probes/patches): without the matrix
// instructions 13.7 ms, without selection and walk 17.4 ms.
#pragma once
#include <type_traits>

#include "topk_kernel.h"
#include "topk_workspace.h"

namespace mirec {


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

// |s~ - s| <= topk_eps_coef<F>() * ||q|| * max||y||.  Rounding to bf16 leaves |x - hi| <= 2^-8 |x| and, of
// that remainder, |x - hi - mid| <= 2^-17 |x|; so the factors' third planes, which are dropped, account for
// 2^-16 (2^-17 on either side, Cauchy-Schwarz over the row), the mid * mid plane product, which is not formed, for
// another 2^-16, and the roundings of the fp32 chain and of the matrix pipe's accumulation for (8 F + 64) 2^-24,
// several times what F sequential and 3 F pipelined additions can lose.
template <int F>
constexpr float topk_eps_coef() {
    return ((float)(8 * F + 64) * 5.9604644775390625e-8f + 1.52587890625e-5f + 1.52587890625e-5f) * 1.01f;
}
// |s~ - hi.hi| <= topk_coarse_coef<F>() * ||q|| * ||y||: the two products with a mid plane,
// |hi_q . mid_y| + |mid_q . hi_y| <= 2 * 2^-8 (1 + 2^-8) ||q|| ||y||, and the roundings of the accumulation.
template <int F>
constexpr float topk_coarse_coef() {
    return (0.0078125f * 1.00390625f + (float)(8 * F + 64) * 5.9604644775390625e-8f) * 1.01f;
}

// ---------------------------------------------------------------------------------------------------
// Item factors -> bf16 planes (one workgroup per item tile, one thread per 8 consecutive factors), the largest
// row norm of every tile and max ||y|| (atomic max on the bits of the non-negative norm; a NaN norm has the
// largest bits).
template <int F>
__global__ __launch_bounds__(32 * (F / 8)) void topk_split_items_kernel(const float* __restrict__ Y, int64_t ni,
                                                                       const int32_t* __restrict__ order,
                                                                       uint32_t* __restrict__ planes,
                                                                       float* __restrict__ tile_norm,
                                                                       uint32_t* __restrict__ ymax_bits) {
    constexpr int F8 = F / 8;
    __shared__ float part[32 * F8];
    __shared__ float rown[32];
    const int tid = threadIdx.x;
    const int j = tid / F8;
    const int kc = tid % F8;
    // row j of the tile is the item at that position of the sweep (its own index without an order)
    int64_t item = (int64_t)blockIdx.x * 32 + j;
    if (order != nullptr && item < ni) {
        const int64_t o = order[item];
        item = o < 0 ? 0 : (o < ni ? o : ni - 1);
    }
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (item < ni) {
        a = *reinterpret_cast<const f32x4*>(Y + item * F + 8 * kc);
        b = *reinterpret_cast<const f32x4*>(Y + item * F + 8 * kc + 4);
    }
    tk_u32x4 hi, mid;
    tk_split8(a, b, hi, mid);
    // three sections of 32 F8 chunks of 16 bytes: hi by rows; hi and mid by chunks (chunk kc of all 32 rows, then
    // chunk kc + 1)
    tk_u32x4* dst = reinterpret_cast<tk_u32x4*>(planes) + (size_t)blockIdx.x * (TOPK_TILE_SECTIONS * 32 * F8);
    dst[j * F8 + kc] = hi;
    dst[32 * F8 + kc * 32 + j] = hi;
    dst[2 * 32 * F8 + kc * 32 + j] = mid;
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
        // the tile's largest row norm, rounded up past what the sum of squares above may have lost; a norm
        // that is not a number (the largest bits) stays one
        tile_norm[blockIdx.x] = __uint_as_float(m) * 1.0001f;
    }
}

// A sweep order made inside the call, for a call that brings none and has no filter (whose rows an order would
// have to be renumbered to): the items by descending row norm, as exactly as 4096 bins of the norm's leading bits
// tell -- which is all the sweep needs, since any order gives the same results and a nearly sorted one nearly
// all of the saving.  Three small launches: count per bin, offsets (one workgroup), place.  Items of one bin land in
// the order the atomics decide; the results of the call do not depend on it.
__device__ __forceinline__ int tk_norm_bin(const float* __restrict__ y, int f) {
    float s = 0.f;
    for (int j = 0; j < f; ++j) s = fmaf(y[j], y[j], s);
    // a norm is not negative: its bits order like the value; a NaN has the largest
    return (TOPK_ORDER_BINS - 1) - (int)(__float_as_uint(fabsf(s)) >> 19);  // descending: bin 0 = largest
}
template <int F>
__global__ __launch_bounds__(256) void topk_order_count_kernel(const float* __restrict__ Y, int64_t ni, int32_t* bins) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < ni) atomicAdd(bins + tk_norm_bin(Y + i * F, F), 1);
}
template <int F>
__global__ __launch_bounds__(1024) void topk_order_scan_kernel(int32_t* bins) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    int v[TOPK_ORDER_BINS / 1024], sum = 0;
    for (int j = 0; j < TOPK_ORDER_BINS / 1024; ++j) {
        v[j] = bins[t * (TOPK_ORDER_BINS / 1024) + j];
        sum += v[j];
    }
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int add = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    int at = part[t] - sum;
    for (int j = 0; j < TOPK_ORDER_BINS / 1024; ++j) {
        bins[t * (TOPK_ORDER_BINS / 1024) + j] = at;
        at += v[j];
    }
}
template <int F>
__global__ __launch_bounds__(256) void topk_order_place_kernel(const float* __restrict__ Y, int64_t ni, int32_t* bins,
                                                               int32_t* __restrict__ order) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < ni) order[atomicAdd(bins + tk_norm_bin(Y + i * F, F), 1)] = (int32_t)i;
}

// Register i of a lane in half h of a 32 x 32 MFMA result is row tk_rowpos(i) + 4 * h.
__device__ __forceinline__ constexpr int tk_rowpos(int i) { return (i & 3) + 8 * (i >> 2); }

// The values that the two lanes of a pair (c, c + 32) hold: .x of the lane in the first half-wave, .y of the other
// (the same in both lanes of the pair; a vector-ALU lane swap, not an LDS access).
struct tk_pair_u32 {
    uint32_t x, y;
};
__device__ __forceinline__ tk_pair_u32 tk_pair(uint32_t v) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return tk_pair_u32{r[0], r[1]};
}
__device__ __forceinline__ float tk_pair_min(float v) {
    const tk_pair_u32 p = tk_pair(__float_as_uint(v));
    return fminf(__uint_as_float(p.x), __uint_as_float(p.y));
}
__device__ __forceinline__ float tk_pair_max(float v) {
    const tk_pair_u32 p = tk_pair(__float_as_uint(v));
    return fmaxf(__uint_as_float(p.x), __uint_as_float(p.y));
}
__device__ __forceinline__ int tk_pair_sum(int v) {
    const tk_pair_u32 p = tk_pair((uint32_t)v);
    return (int)(p.x + p.y);
}

__device__ __forceinline__ int tk_load_acquire(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void tk_store_release(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Wait until each of the NW progress words at p (one per wavefront of the workgroup, in LDS, only growing) has
// reached `need`.  Lane l looks at the word of wavefront l.
template <int NW>
__device__ __forceinline__ void tk_wait_all(const int* p, int need, int lane) {
    while (__ballot(lane < NW && tk_load_acquire(p + (lane < NW ? lane : 0)) < need) != 0ull) __builtin_amdgcn_s_sleep(1);
}

#ifndef MIREC_TOPK_AHEAD
#define MIREC_TOPK_AHEAD 3
#endif
#ifndef MIREC_TOPK_STRETCH
#define MIREC_TOPK_STRETCH 64
#endif
// Tiles at the start of the sweep that every wavefront works without testing them first (a power of two): 0 / 8 / 16 /
// 32 / 64 ran the 1 M x 27 k pass in 9.17 / 9.22 / 9.08 / 8.97 / 9.15 ms.
#ifndef MIREC_TOPK_EARLY
#define MIREC_TOPK_EARLY 32
#endif

template <int F, int NW, int CAP, int PD>
struct TopkPlanesLds {
    static constexpr int NQ = 32 * NW;  // queries of a workgroup
    // The hi plane of an item tile: 32 rows of F bf16, padded by 16 bytes: the row stride in dwords is an odd
    // multiple of four, which puts the 16 rows of every ds_read_b128 group on disjoint banks.  An image is
    // filled by loads that write LDS directly (lanes x 16 bytes, consecutive, per instruction), in pieces of whole
    // padded rows, as few as fit a wavefront's 64 lanes -- the same number for every sweeping wavefront, so the wait
    // for its own pieces is the same instruction everywhere.
    static constexpr int NS = NW / 2;  // wavefronts that sweep: each for its own 32 queries and for those of wavefront + NS
    static_assert(NW % 2 == 0 && NS >= 1, "sweeping wavefronts pair up with the others");
    static constexpr int ROWB = 2 * F + 16;
    static constexpr int IMGB = TOPK_BI * ROWB;
    static constexpr int piece_count(int npw) { return (TOPK_BI / (npw * NS)) * (ROWB / 16) <= 64 ? npw : piece_count(2 * npw); }
    static constexpr int NPW = piece_count(1);               // pieces a sweeping wavefront issues per tile
    static constexpr int PROWS = TOPK_BI / (NPW * NS);       // image rows per piece
    static constexpr int PLANES = PROWS * (ROWB / 16);       // lanes of a piece
    static constexpr int PIECEB = PROWS * ROWB;
    static_assert(PROWS >= 1 && TOPK_BI % (NPW * NS) == 0 && PLANES <= 64, "pieces of whole rows, at most a wavefront wide");
    static constexpr int RING = 8;  // liked-item look-ahead per query (a power of two, at least 8)
    // in floats: the item images; candidate scores and ids, entry-major ([entry][query]: the lanes of a
    // half-wave, one query each, touch 32 consecutive words); fill counts; liked-item rings; the progress words
    static constexpr int SEL_S = PD * IMGB / 4;
    static constexpr int SEL_I = SEL_S + CAP * NQ;
    static constexpr int CNT = SEL_I + CAP * NQ;
    static constexpr int RINGS = CNT + NQ;
    static constexpr int SYNC = RINGS + NQ * RING;
    static constexpr int SYNC_WORDS = 32;  // a pair per sweeping wavefront: images landed, images read
    static_assert(NS <= 8, "progress words");
    // what a sweeping wavefront needs of its partner's queries (entry thresholds, coarse margins), and what it hands
    // back (the marks of a stretch, 64 bits per wavefront)
    static constexpr int XTHR = SYNC + SYNC_WORDS;
    static constexpr int XCM = XTHR + NQ;
    static constexpr int XMARK = XCM + NQ;
    // a byte per tile of a stretch and group, set where a lane's test fails: 64 + 64 per sweeping wavefront
    static constexpr int XFLAG = XMARK + 2 * NW;
    static constexpr size_t BYTES = (size_t)(XFLAG + 32 * NS) * 4;
    static constexpr int LIMIT = CAP - 16;  // entries a compaction may leave (at least k)
    static_assert(CAP % 2 == 0 && LIMIT >= TOPK_PLANE_KMAX && LIMIT <= 32, "buffer size");
};

// Compaction of the buffers of all 32 queries of a wavefront at once.  Lane pair (c, c + 32) owns query slot q;
// the lane in half h holds the entries 2j + h in registers.  For a query with at least k entries a bisection
// on [min, max] finds lo with  #(entries >= lo) >= k  (so lo <= the k-th best entry), the smallest entry >= lo
// takes its place (same count, tighter), the entry threshold becomes max(thr, lo - marg) and the entries above
// it move to the front of the buffer.  Returns the number kept; over = more than LIMIT.
template <int CAP, int NQ, int LIMIT>
__device__ __forceinline__ int tk_compact_all(float* sel_s, int* sel_i, int* cnt, int q, int h, int k, float marg,
                                              float& thr, bool& over) {
    constexpr int NE = CAP / 2;
    constexpr int STEPS = 9;
    __builtin_amdgcn_wave_barrier();
    int n = cnt[q];
    n = n < CAP ? n : CAP;  // appends that found the buffer full have counted past it
    float v[NE];
    int id[NE];
    float lo = INFINITY, hi = -INFINITY;
    const float* my_s = sel_s + (h * NQ + q);  // entry 2j + h of the query: a constant offset from here
    const int* my_i = sel_i + (h * NQ + q);
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int e = 2 * j + h;
        const float sv = my_s[2 * j * NQ];
        id[j] = my_i[2 * j * NQ];
        v[j] = e < n ? sv : -INFINITY;
        lo = e < n ? fminf(lo, sv) : lo;
        hi = fmaxf(hi, v[j]);
    }
    lo = tk_pair_min(lo);
    hi = tk_pair_max(hi);
    if (n >= k) {
#pragma unroll 1
        for (int step = 0; step < STEPS; ++step) {
            const float mid = 0.5f * lo + 0.5f * hi;
            int cge = 0;
#pragma unroll
            for (int j = 0; j < NE; ++j) cge += v[j] >= mid ? 1 : 0;
            cge = tk_pair_sum(cge);
            lo = cge >= k ? mid : lo;
            hi = cge >= k ? hi : mid;
        }
        float t = INFINITY;
#pragma unroll
        for (int j = 0; j < NE; ++j) t = v[j] >= lo ? fminf(t, v[j]) : t;
        lo = tk_pair_min(t);
        thr = fmaxf(thr, lo - marg);
    }
    int mine = 0;
#pragma unroll
    for (int j = 0; j < NE; ++j) mine += v[j] > thr ? 1 : 0;
    const tk_pair_u32 cn = tk_pair((uint32_t)mine);
    const int total = (int)(cn.x + cn.y);
    over = total > LIMIT;
    int pos = h ? (int)cn.x : 0;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        if (v[j] > thr) {
            sel_s[pos * NQ + q] = v[j];
            sel_i[pos * NQ + q] = id[j];
            ++pos;
        }
    }
    if (h == 0) cnt[q] = total;
    __builtin_amdgcn_wave_barrier();
    return total;
}

// BLOCKS: the filter comes as one list per 32 queries (a wavefront's), sorted by position of the sweep, each entry
// a position and the 32-bit set of the block's rows that hold it (mask_indices then points to those pairs and
// block_len to the lists' lengths; the list of the block that starts at query row r begins at entry
// mask_indptr[r]).  Otherwise the filter is the CSR rows themselves, ascending, one walked per query.
template <int F, int NW, int CAP, int PD, bool BLOCKS>
__global__ __launch_bounds__(64 * NW) void topk_planes_kernel(
    const float* __restrict__ Q, int64_t nq, const float* __restrict__ Y, int64_t ni,
    const uint32_t* __restrict__ planes, const float* __restrict__ tile_norm, const uint32_t* __restrict__ ymax_bits,
    const int64_t* __restrict__ mask_indptr, const int32_t* __restrict__ mask_indices,
    const int32_t* __restrict__ block_len, const int32_t* __restrict__ order, int k, int32_t* __restrict__ out_ids,
    float* __restrict__ out_scores, int32_t* __restrict__ qlist, int32_t* __restrict__ qcount,
    int32_t* __restrict__ stash_flags, int stash_slots, tk_u32x4* __restrict__ stash) {
    using L = TopkPlanesLds<F, NW, CAP, PD>;
    static_assert(L::NQ == TOPK_WG_QUERIES, "the workspace is sized for workgroups of this many queries");
    constexpr int KS = F / 16;  // MFMA k-steps
    constexpr int F8 = F / 8;   // 16-byte chunks per plane row
    constexpr int BI = TOPK_BI;
    constexpr int NQ = L::NQ;
    constexpr int RING = L::RING;
    // Images a loading wavefront keeps on their way at a time (the ring itself bounds how far the loads run ahead of
    // the sweep: PD - 1 tiles).
    constexpr int D = MIREC_TOPK_AHEAD;
    static_assert(D >= 1 && D <= PD - 1 && D * L::NPW <= 48, "staging depth");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* ybuf = reinterpret_cast<char*>(smem);
    float* sel_s = smem + L::SEL_S;
    int* sel_i = reinterpret_cast<int*>(smem + L::SEL_I);
    int* cnt = reinterpret_cast<int*>(smem + L::CNT);
    int* rings = reinterpret_cast<int*>(smem + L::RINGS);
    int* sync = reinterpret_cast<int*>(smem + L::SYNC);
    float* xthr = smem + L::XTHR;
    float* xcm = smem + L::XCM;
    unsigned long long* xmark = reinterpret_cast<unsigned long long*>(smem + L::XMARK);
    unsigned char* xflag = reinterpret_cast<unsigned char*>(smem + L::XFLAG);
    // sync[l]: the shares of loading wavefront NS + l of the images of the tiles below this number are in LDS;
    // sync[8 + w]: sweeping wavefront w has its operands of the tiles below this number in registers
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int h = lane >> 5;
    const int c = lane & 31;
    const int ntile = (int)((ni + BI - 1) / BI);
    const int q = w * 32 + c;  // this lane pair's query slot
    constexpr int NS = L::NS;
    const bool sweeper = w < NS;  // (the same for a whole wavefront)
    const int64_t ug = (int64_t)blockIdx.x * NQ + q;
    const bool uv = ug < nq;

    // The first tiles are not swept (further down): the ring, and the counts of tiles in the progress words, start
    // behind them.
    constexpr int EARLY = MIREC_TOPK_EARLY;
    static_assert(EARLY >= 0 && (EARLY & (EARLY - 1)) == 0 && EARLY <= MIREC_TOPK_STRETCH, "whole stretches, doubling");
    const int t_early = EARLY < ntile ? EARLY : ntile;
    if (h == 0) cnt[q] = 0;
    if (tid < L::SYNC_WORDS) sync[tid] = t_early;
    __syncthreads();
    // A slot of the workspace for this workgroup, for as long as it runs: there are several times as many as
    // workgroups in flight, so the probe finds a free one at once.
    if (tid == 0) {
        int sl = (int)(blockIdx.x % (unsigned)stash_slots);
        while (atomicCAS(stash_flags + sl, 0, 1) != 0) {
            sl = sl + 1 == stash_slots ? 0 : sl + 1;
            __builtin_amdgcn_s_sleep(4);
        }
        sync[L::SYNC_WORDS - 1] = sl;
    }

    // B operands: query row ug, factors 16*s + 8*h .. + 7, as two bf16 planes; and the margins of the query
    tk_u32x4 qh[KS], qm[KS];
    float marg, cmarg;
    {
        const float* qr = Q + (uv ? ug : 0) * F + 8 * h;
        float n2 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f32x4 a = *reinterpret_cast<const f32x4*>(qr + 16 * s);
            f32x4 b = *reinterpret_cast<const f32x4*>(qr + 16 * s + 4);
            if (!uv) {
                a = f32x4{0.f, 0.f, 0.f, 0.f};
                b = a;
            }
            n2 += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3] + b[0] * b[0] + b[1] * b[1] + b[2] * b[2] +
                  b[3] * b[3];
            tk_split8(a, b, qh[s], qm[s]);
        }
        const tk_pair_u32 pn = tk_pair(__float_as_uint(n2));
        const float qn = sqrtf(__uint_as_float(pn.x) + __uint_as_float(pn.y));
        const float ymax = __uint_as_float(*ymax_bits);
        // the second term: at least the absolute error that products in the denormal range can add
        marg = (qn > 0.f && ymax > 0.f) ? 2.f * topk_eps_coef<F>() * qn * ymax + (float)(8 * F + 64) * 2.4e-38f
                                        : qn * ymax;
        // per unit of a tile's largest row norm: what the two products with a mid plane can add to hi * hi
        cmarg = topk_coarse_coef<F>() * qn;
    }
    // A zero margin means all scores are exactly zero on both pipes and only the order of the ids decides; a
    // margin that is not finite covers nothing.  Both go to the fp32 kernel.
    const bool unfit = !(marg > 0.f && marg < INFINITY);
    // Entry threshold of the query: a score enters the buffer only if it beats it.  +inf: nothing ever does.
    float thr = (uv && !unfit) ? -INFINITY : INFINITY;
    bool ovf = false;  // the margin set outgrew the buffer

    // Liked-item filter: the lane of the first half-wave walks the (ascending) mask row of its query.  The row's
    // upcoming entries sit in a small LDS ring, [cur, tail) of the row, refilled four at a time by loads that are
    // issued when the walk has consumed entries and committed the next time it has something to do -- the walk
    // itself never waits for memory unless one query has more liked items in a tile than the ring delivers.
    // A tile in which no query of the wavefront has a liked item costs one compare.
    int* my_ring = rings + q * RING;
    const int32_t* mrow = mask_indices;
    int cur = 0, tail = 0, mlen = 0, pend_base = -1;
    int nxt = 0x7fffffff, nxt2 = 0x7fffffff;  // the row's entries cur and cur + 1 (0x7fffffff: none, or not read yet)
    int rf[4] = {0, 0, 0, 0};
    const bool walker = !BLOCKS && mask_indptr != nullptr && h == 0 && uv;
    if (walker) {
        const int64_t b = mask_indptr[ug];
        mrow = mask_indices + b;
        mlen = (int)(mask_indptr[ug + 1] - b);
        tail = mlen < RING ? mlen : RING;
        for (int e = 0; e < tail; ++e) my_ring[e] = mrow[e];
        // read back from the ring, not from memory: a register that may still be waiting for a memory load
        // would put a wait for all loads in flight (the refill's among them) into the walk
        __builtin_amdgcn_wave_barrier();
        if (mlen > 0) nxt = my_ring[0];
        if (mlen > 1) nxt2 = my_ring[1];
    }
    auto mask_refill_issue = [&]() __attribute__((always_inline)) {
        // room for four more and something left to fetch; loads are unconditional (clamped) inside the branch
        if (walker && pend_base < 0 && tail < mlen && tail - cur <= RING - 4) {
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
    // Liked bits of the items [i0, i0 + 32) for this lane's query (walker lanes; bit b: item i0 + b).
    auto walk_tile = [&](int i0) __attribute__((always_inline)) -> unsigned {
        unsigned mw = 0u;
        if (walker) {
            const int lim = i0 + BI;
            bool dry = false;
            // The common case -- at most one liked item of a query in a tile -- without a loop and without
            // waiting for LDS: the entry after the one consumed is already in a register, the one after that is
            // requested now and not looked at before the next tile.
            bool more = nxt == 0x7fffffff && cur < mlen;  // not delivered when it was looked for: it may be by now
            if (nxt < lim) {
                const int bit = nxt - i0;
                if (bit >= 0) mw |= 1u << bit;
                ++cur;
                nxt = nxt2;
                nxt2 = cur + 1 < tail ? my_ring[(cur + 1) & (RING - 1)] : 0x7fffffff;
                more = nxt < lim || (nxt == 0x7fffffff && cur < mlen);
            }
            if (__ballot(more) != 0ull) {
                if (more && nxt == 0x7fffffff && cur < mlen) {
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
        return mw;
    };

    // The filter as a list per wavefront (BLOCKS): lane j holds entry 64 * chunk + j of the list -- its position
    // and row set -- and the chunk after that waits in a second pair of registers; `cons` entries of the current
    // chunk lie in tiles already swept.  Everything about the walk is the same for all lanes: a tile without
    // a liked item costs one compare, and the loads of a chunk are coalesced and needed 64 entries later.
    int bpos0 = 0x7fffffff, bpos1 = 0x7fffffff;
    unsigned brm0 = 0u, brm1 = 0u;
    int cons = 0, bnext = 0, blen = 0;
    const int2* bent = nullptr;
    unsigned* mwl = reinterpret_cast<unsigned*>(rings) + w * 32;  // liked bits of the wavefront's queries for one tile
    auto block_load = [&](int first, int& pos, unsigned& rm) __attribute__((always_inline)) {
        const int idx = first + lane;
        const int2 e = bent[idx < blen ? idx : (blen > 0 ? blen - 1 : 0)];
        pos = idx < blen ? e.x : 0x7fffffff;
        rm = idx < blen ? (unsigned)e.y : 0u;
    };
    if constexpr (BLOCKS) {
        if (mask_indptr != nullptr) {
            const int64_t row0 = (int64_t)blockIdx.x * NQ + w * 32;  // first query of the wavefront
            if (row0 < nq) {
                bent = reinterpret_cast<const int2*>(mask_indices) + mask_indptr[row0];
                blen = block_len[row0 >> 5];
                if (blen > 0) {
                    block_load(0, bpos0, brm0);
                    block_load(64, bpos1, brm1);
                }
                bnext = 128;
            }
        }
    }
    // Liked bits of the tile [i0, i0 + 32) for this lane's query.  Entries in front of the tile belong to tiles that
    // were not worked: they are passed over.
    auto walk_blocks = [&](int i0) __attribute__((always_inline)) -> unsigned {
        const int lim = i0 + BI;
        bool any = false;
        for (;;) {
            const int nlt = __builtin_popcountll(__ballot(bpos0 < lim));  // the list is sorted: a prefix
            if (nlt > cons) {
                if (__ballot(lane >= cons && lane < nlt && bpos0 >= i0) != 0ull) {
                    if (!any) {
                        if (lane < 32) mwl[lane] = 0u;
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (lane >= cons && lane < nlt && bpos0 >= i0) {
                        const unsigned bit = 1u << (bpos0 - i0);
                        unsigned rm = brm0;
                        while (rm != 0u) {
                            atomicOr(mwl + __builtin_ctz(rm), bit);
                            rm &= rm - 1u;
                        }
                    }
                    any = true;
                }
                cons = nlt;
            }
            if (nlt < 64) break;
            bpos0 = bpos1;
            brm0 = brm1;
            cons = 0;
            block_load(bnext, bpos1, brm1);
            bnext += 64;
        }
        unsigned mw = 0u;
        if (any) {
            __builtin_amdgcn_wave_barrier();
            mw = mwl[c];
        }
        return mw;
    };

    // Item images: the hi plane of tile t goes to LDS slot t % PD by loads that write LDS directly, D tiles ahead of
    // the products; every wavefront issues its share of the pieces of an image (piece p by wavefront p % NW).  The
    // bytes of a piece are consecutive in LDS; each lane computes where its 16 of them come from (a chunk of the
    // row padding repeats the row's last chunk).
    constexpr int TILE_G = TOPK_TILE_SECTIONS * BI * F8;  // 16-byte chunks of one tile in the workspace
    // source of this lane's 16 bytes of a piece, as a byte offset into the tile's hi plane (a chunk of the row
    // padding repeats the row's last chunk)
    int voff[L::NPW];
    const bool plane_lane = lane < L::PLANES;
#pragma unroll
    for (int j = 0; j < L::NPW; ++j) {
        const int piece = (sweeper ? 0 : w - NS) + NS * j;
        const int row = piece * L::PROWS + lane / (F8 + 1);
        int ch = lane % (F8 + 1);
        ch = ch < F8 ? ch : F8 - 1;
        voff[j] = plane_lane ? (row * F8 + ch) * 16 : 0;
    }
    // (Issued as an assembly statement: a load that the compiler knows to write LDS makes it wait for all loads in
    // flight in front of every later LDS access, the flag reads of the next tile included, which would leave no
    // image on its way while another is multiplied.  What orders the readers behind these loads is the wait on
    // the load counter before the progress word is advanced.)
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ybuf;
    auto dma_piece = [&](const tk_u32x4* src, unsigned dst, int voffset) __attribute__((always_inline)) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voffset), "s"(src), "s"(dst)
                     : "memory");
    };
    // the slot of the next request, as a byte offset that wraps around the ring
    int dma_slot = 0;
    auto dma = [&](int tv) __attribute__((always_inline)) {
        const int t = __builtin_amdgcn_readfirstlane(tv);  // the same in all lanes: base address and slot are scalars
        const tk_u32x4* src = reinterpret_cast<const tk_u32x4*>(planes) + (size_t)t * TILE_G;
        const unsigned slot = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)dma_slot));
        dma_slot = dma_slot + L::IMGB == PD * L::IMGB ? 0 : dma_slot + L::IMGB;
        if (plane_lane) {
#pragma unroll
            for (int j = 0; j < L::NPW; ++j)
                dma_piece(src, slot + (unsigned)((w - NS + NS * j) * L::PIECEB), voff[j]);
        }
    };
    // The loading wavefronts' side of the ring: request the images of the tiles below `lim`, each when every sweeping
    // wavefront has taken its operands of the tile whose slot it overwrites, D tiles on their way at a time.  The
    // progress word says what has landed (loads complete in order: all but the newest D - 1 requests, after the wait).
    int dma_next = t_early;  // the images of the tiles below this number have been requested (or are not needed)
    const int* done_word = sync + 8 + (lane < NS ? lane : 0);
    auto load_until = [&](int lim) __attribute__((always_inline)) {
        int flying = 0;
        while (dma_next < lim) {
            const int need = dma_next - PD + 1;  // tiles below this number must have been read
            if (need > 0) {
                while (__ballot(lane < NS && tk_load_acquire(done_word) < need) != 0ull) __builtin_amdgcn_s_sleep(1);
            }
            dma(dma_next++);
            if (flying == D - 1) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * L::NPW) : "memory");
                if (lane == 0) __hip_atomic_store(sync + (w - NS), dma_next - (D - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                ++flying;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(sync + (w - NS), dma_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    __syncthreads();  // the slot number is up
    // the hi plane of the wavefront's queries, for the sweeping wavefront NS below: k-step s of lane l at chunk
    // 64 s + l of the wavefront's part of the slot (a load or store of a wavefront is 1 KiB in a row)
    tk_u32x4* my_stash = stash + ((size_t)sync[L::SYNC_WORDS - 1] * NS + (sweeper ? w : w - NS)) * (64 * KS) + lane;
    if (!sweeper) {
#pragma unroll
        for (int s = 0; s < KS; ++s) my_stash[64 * s] = qh[s];
        load_until(t_early + PD - 1 < ntile ? t_early + PD - 1 : ntile);
    }
    if (h == 0) xcm[q] = cmarg;
    // the first tiles are staged, the fill counts and coarse margins are set, the hi planes are where the sweeping
    // wavefronts find them (a barrier of the language: it orders the stores in front of the loads behind it)
    __syncthreads();

    // One candidate: reserve a slot of the query's buffer and fill it; false: the buffer is full.
    auto append = [&](float s, int item) __attribute__((always_inline)) -> bool {
        const int pos = atomicAdd(&cnt[q], 1);
        if (pos < CAP) {
            sel_s[pos * NQ + q] = s;
            sel_i[pos * NQ + q] = item;
        }
        return pos < CAP;
    };
    auto compact = [&]() __attribute__((always_inline)) {
        bool over;
        tk_compact_all<CAP, NQ, L::LIMIT>(sel_s, sel_i, cnt, q, h, k, marg, thr, over);
        if (over) {
            ovf = true;
            thr = INFINITY;
        }
    };

    // lane 2 l looks at the images landed of wavefront l, lane 2 l + 1 at the images read of wavefront l
    // lane l of a sweeping wavefront looks at the images landed of loading wavefront l % NS
    const int* my_word = sync + lane % NS;

    // The sweep multiplies the hi planes only (a third of the products) and looks at the largest of a lane's
    // sixteen sums: what the two products with a mid plane can add is bounded by cmarg times the tile's largest
    // row norm, so where sum + bound stays at or below the entry threshold for every query of a group of 32, no
    // score of the tile can enter a buffer, now or later (thresholds only rise), and the tile is done for the group.
    // A tile that fails the test is only marked.  The marked tiles of a stretch of the sweep are worked off when the
    // stretch ends: each wavefront takes both planes of its marked tiles straight from the workspace (which the
    // cache holds), forms all three products and runs the selection.  Stretches double from one tile to SEG tiles
    // (the thresholds move fastest at the start) and stay there.
    //
    // A swept tile is bound by the number of instructions beside its products (operand reads, staging, progress
    // words: about a hundred, against eight products), and they are the same whether the operands meet 32 queries
    // or 64.  So half of the wavefronts sweep -- one per SIMD, each for its own group of queries and for the group
    // of the wavefront NS above it, sixteen products per tile and operand read -- while the other half waits at the
    // barrier; the marks of the partner's group go back through LDS, and everything else (marked tiles,
    // compactions, the final stage) is done by all wavefronts, each for its own group, as the latencies of that
    // work want it.  A sweeping wavefront reads its partner's entry thresholds and coarse margins from LDS and
    // the partner's hi plane from the query matrix at the start of every stretch.
    constexpr int SEG = MIREC_TOPK_STRETCH;
    unsigned long long marks = 0ull;   // bit r: tile r of the stretch (the same in all lanes: a pair of scalar registers)
    unsigned long long marks1 = 0ull;  // the same for the partner's group (sweeping wavefronts)
    static_assert(SEG <= 64, "a mark per tile of the stretch in one 64-bit scalar");
    // Wait until the image of tile t is complete.  `seen`: this lane's progress word as read earlier.
    auto ready = [&](int t, int seen) __attribute__((always_inline)) {
        const int need = t + 1;
        if (__ballot(seen < need) != 0ull) {
            while (__ballot(tk_load_acquire(my_word) < need) != 0ull) __builtin_amdgcn_s_sleep(1);
        }
    };
    // the sixteen sums of a lane against the entry threshold: true = the tile has to be worked
    auto test = [&](const f32x16& a, float tn, float cm, float th) __attribute__((always_inline)) -> bool {
        float best = fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(a[4], a[5]), fmaxf(a[6], a[7])));
        best = fmaxf(best, fmaxf(fmaxf(fmaxf(a[8], a[9]), fmaxf(a[10], a[11])), fmaxf(fmaxf(a[12], a[13]), fmaxf(a[14], a[15]))));
        return __ballot(fmaf(cm, tn, best) + (float)(8 * F + 64) * 2.4e-38f > th) != 0ull;
    };
    // the same in eight instructions (a maximum written in the language costs an extra instruction per input,
    // which quiets signalling NaNs; v_max3_f32 returns the other operands where one is a NaN, as fmaxf does).
    // Only where the sums are old -- at least four products issued since the last one into them: the assembler
    // statement hides from the compiler that they come from the matrix pipe, whose results must not be read too
    // early.
    auto test_fast = [&](const f32x16& a, float tn, float cm, float th) __attribute__((always_inline)) -> bool {
        auto max3 = [](float x, float y, float z) __attribute__((always_inline)) -> float {
            float r;
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
            return r;
        };
        const float m0 = max3(a[0], a[1], a[2]), m1 = max3(a[3], a[4], a[5]), m2 = max3(a[6], a[7], a[8]);
        const float m3 = max3(a[9], a[10], a[11]), m4 = max3(a[12], a[13], a[14]);
        const float best = max3(max3(m0, m1, m2), max3(m3, m4, a[15]), -INFINITY);
        return __ballot(fmaf(cm, tn, best) + (float)(8 * F + 64) * 2.4e-38f > th) != 0ull;
    };
    auto test_old = [&](const f32x16& a, float tn, float cm, float th) __attribute__((always_inline)) -> bool {
        if constexpr (KS >= 5)
            return test_fast(a, tn, cm, th);
        else
            return test(a, tn, cm, th);
    };
    // One stretch [a, b) of the sweep, software-pipelined by hand: the products of tile t for the own group, then
    // those for the partner's; between them the wavefront issues the operand reads of tile t + 1 (the operand of
    // k-step s into the registers that the second product of that k-step has just taken its operand from: one set
    // of operand registers, each read a whole tile ahead of its use), the read of its progress word, and the test
    // of the sums that are at rest -- the partner's of tile t - 1 under the own products, the own of tile t under
    // the partner's.  A matrix instruction keeps the pipe for 32 cycles and the issue port for 8, so what fits into
    // the gaps is free, and nothing in the loop waits for a latency it has just started; there are no branches in
    // a tile but the rare ones out of it.
    tk_u32x4 ya[KS];
    f32x16 acc0, acc1 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // lane l: the largest row norm of tile l of the stretch (a scalar load in the loop would make every wait for an
    // operand a wait for everything in flight)
    constexpr int NTN = 1;
    float tnv[NTN];
    auto tnorm = [&](int rel) __attribute__((always_inline)) -> float {
        return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(tnv[0]), rel & 63));
    };
    const char* ylane = ybuf + c * L::ROWB + 16 * h;  // this lane's row and half in slot 0
    int fetch_slot = 0;                               // the slot of the tile whose operands are read next
    auto fetch_part = [&](int s0, int s1) __attribute__((always_inline)) {
        const char* yb = ylane + fetch_slot;
#pragma unroll
        for (int s = s0; s < s1; ++s) ya[s] = *reinterpret_cast<const tk_u32x4*>(yb + 32 * s);
    };
    auto fetch_advance = [&]() __attribute__((always_inline)) {
        fetch_slot = fetch_slot + L::IMGB == PD * L::IMGB ? 0 : fetch_slot + L::IMGB;
    };
    int* my_progress = sync + 8 + w;
    unsigned char* my_flags = xflag + 128 * (sweeper ? w : 0);
    // q1, thr1, cm1: the partner group's hi plane, entry thresholds and coarse margins; NEXT: tile t + 1 belongs to
    // the stretch.  A wavefront alone on its SIMD issues in order, and a matrix instruction that finds the pipe busy
    // holds up everything behind it: what stands between two products runs under the first of them only as far as it
    // fits into its 32 cycles, so the tile's other work is cut into pieces of a handful of instructions, one or two to
    // a gap -- a test in four (five maxima of three, the last three, bound and compare, the mark).
    auto max3 = [](float x, float y, float z) __attribute__((always_inline)) -> float {
        float r;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
        return r;
    };
    auto swept_tile = [&](int t, int a, int& seen, const tk_u32x4 (&q1)[KS], float thr1, float cm1, auto has_next)
                          __attribute__((always_inline)) {
        constexpr bool NEXT = decltype(has_next)::value;
        constexpr bool PIECES = KS >= 8;  // (narrower operands: the test in one piece, written in the language)
        constexpr int S_TEST = KS > 4 ? 4 : KS - 1, S_READY = KS > 6 ? 6 : KS - 1;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        bool work0 = false, work1 = false;
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, best = 0.f;
        // (a failed test sets a byte, lane by lane: finding out whether any lane failed, and keeping a mask of the
        // tiles in scalar registers, took a dozen scalar instructions per tile and group)
        unsigned char* fl0 = my_flags + (t - a);       // tile t, own group
        unsigned char* fl1 = my_flags + 64 + (t - a);  // tile t - 1, partner's group: byte 64 + (t - 1 - a) + 1
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            acc0 = tk_mma(ya[s], qh[s], s == 0 ? zero : acc0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NEXT)
                if (s == 0) seen = __hip_atomic_load(my_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if constexpr (PIECES) {
                // the partner group's sums of tile t - 1 (of nothing, at the first tile of a stretch)
                if (s == 2) {
                    m0 = max3(acc1[0], acc1[1], acc1[2]), m1 = max3(acc1[3], acc1[4], acc1[5]);
                    m2 = max3(acc1[6], acc1[7], acc1[8]), m3 = max3(acc1[9], acc1[10], acc1[11]);
                    m4 = max3(acc1[12], acc1[13], acc1[14]);
                }
                if (s == 3) best = max3(max3(m0, m1, m2), max3(m3, m4, acc1[15]), -INFINITY);
                if (s == 4) work1 = fmaf(cm1, tnorm(t - 1 - a), best) + (float)(8 * F + 64) * 2.4e-38f > thr1;
                if (s == 5)
                    if (work1) *fl1 = 1;
            } else {
                if (s == S_TEST) {
                    if (t > a) work1 = test_old(acc1, tnorm(t - 1 - a), cm1, thr1);
                    if (work1) marks1 |= 1ull << (t - 1 - a);
                }
            }
            if constexpr (NEXT)
                if (s == S_READY) ready(t + 1, seen);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            acc1 = tk_mma(ya[s], q1[s], s == 0 ? zero : acc1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NEXT) fetch_part(s, s + 1);
            // the operands of tile t are in registers (the last product of the first eight has waited for them);
            // every lane stores the same word: cheaper than singling one out
            if (s == 0) __hip_atomic_store(my_progress, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if constexpr (PIECES) {
                if (s == 2) {
                    m0 = max3(acc0[0], acc0[1], acc0[2]), m1 = max3(acc0[3], acc0[4], acc0[5]);
                    m2 = max3(acc0[6], acc0[7], acc0[8]), m3 = max3(acc0[9], acc0[10], acc0[11]);
                    m4 = max3(acc0[12], acc0[13], acc0[14]);
                }
                if (s == 3) best = max3(max3(m0, m1, m2), max3(m3, m4, acc0[15]), -INFINITY);
                if (s == 4) work0 = fmaf(cmarg, tnorm(t - a), best) + (float)(8 * F + 64) * 2.4e-38f > thr;
                if (s == 5)
                    if (work0) *fl0 = 1;
            } else {
                if (s == S_TEST) {
                    work0 = test_old(acc0, tnorm(t - a), cmarg, thr);
                    if (work0) marks |= 1ull << (t - a);
                }
            }
            if constexpr (NEXT)
                if (s == KS - 1) fetch_advance();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using T = std::true_type;
    using N = std::false_type;
    auto sweep_stretch = [&](int a, int b) __attribute__((always_inline)) {
        // the partner's group (query slot q + 32 NS): its hi plane, as the partner left it
        tk_u32x4 q1[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) q1[s] = my_stash[64 * s];
        float thr1 = xthr[q + 32 * NS], cm1 = xcm[q + 32 * NS];
        int seen = __hip_atomic_load(my_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ready(a, seen);
        fetch_part(0, KS);
        fetch_advance();
#pragma unroll
        for (int j = 0; j < NTN; ++j) {
            const int tj = a + 64 * j + lane;
            tnv[j] = tile_norm[tj < ntile ? tj : ntile - 1];
        }
        // used here once, so that the wait for these loads stands here and not in front of their first use inside
        // the loop, where it would be a wait for every image on its way, tile after tile
#pragma unroll
        for (int j = 0; j < NTN; ++j) asm volatile("" : "+v"(tnv[j]));
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(q1[s]));
        asm volatile("" : "+v"(thr1), "+v"(cm1));
        if (lane < 32) reinterpret_cast<unsigned*>(my_flags)[lane] = 0u;
        int t = a;
        for (; t + 1 < b; ++t) swept_tile(t, a, seen, q1, thr1, cm1, T{});
        swept_tile(t, a, seen, q1, thr1, cm1, N{});
        if (test(acc1, tnorm(b - 1 - a), cm1, thr1)) marks1 |= 1ull << (b - 1 - a);
        // byte 64 is the partner group's sums of before the stretch
        marks |= __ballot(my_flags[lane] != 0);
        marks1 |= __ballot(lane < 63 && my_flags[65 + (lane < 63 ? lane : 0)] != 0);
    };
    // (The operands of a marked tile come straight from the workspace, from the sections laid out by chunks: the
    // lanes of a load read 1 KiB in a row.  By rows -- each lane its own row, 32 cache lines per instruction -- the
    // lookups, not the bytes, made the sixteen loads take as long as everything else that is done for the tile.)
    auto work_tile = [&](int t) __attribute__((always_inline)) {
        const int i0 = t * BI;
        // item c of the tile, both planes, factors 16 s + 8 h .. + 7: chunk 2 s + h
        const tk_u32x4* tp = reinterpret_cast<const tk_u32x4*>(planes) + (size_t)t * TILE_G + BI * F8 + h * BI + c;
        tk_u32x4 yh[KS], ym[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) yh[s] = tp[2 * s * BI];
#pragma unroll
        for (int s = 0; s < KS; ++s) ym[s] = tp[BI * F8 + 2 * s * BI];
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 acc = zero;
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = tk_mma(yh[s], qh[s], acc);  // hi hi
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = tk_mma(yh[s], qm[s], acc);  // hi mid
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = tk_mma(ym[s], qh[s], acc);  // mid hi
        // mid * mid is below 2^-16 of |q| |y| and left to the margin
        float sc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[i] = acc[i];
        if (t == ntile - 1) {  // rows past the last item
            const int left = (int)(ni - i0) - 4 * h;
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[i] = tk_rowpos(i) < left ? sc[i] : -INFINITY;
        }
        // liked bits of this lane's 16 items: bit tk_rowpos(i)
        unsigned liked = 0u;
        if constexpr (BLOCKS) {
            // the first entry not yet passed sits in lane `cons`
            if (__builtin_amdgcn_readlane(bpos0, cons) < i0 + BI) liked = walk_blocks(i0) >> (4 * h);
        } else {
            unsigned mw = 0u;
            if (__ballot(nxt < i0 + BI || (nxt == 0x7fffffff && cur < mlen)) != 0ull) {
                mask_refill_commit();
                mw = walk_tile(i0);
                mask_refill_issue();
            }
            if (mask_indptr != nullptr) liked = (unsigned)__builtin_amdgcn_ds_bpermute(c << 2, (int)mw) >> (4 * h);
        }
        unsigned fail = 0u;
        // candidates are few: a group of four scores is looked at only where its best beats the threshold
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float best4 = fmaxf(fmaxf(sc[4 * g], sc[4 * g + 1]), fmaxf(sc[4 * g + 2], sc[4 * g + 3]));
            if (best4 > thr) {
#pragma unroll
                for (int i = 4 * g; i < 4 * g + 4; ++i) {
                    if (sc[i] > thr && !((liked >> tk_rowpos(i)) & 1u)) {
                        if (!append(sc[i], i0 + tk_rowpos(i) + 4 * h)) fail |= 1u << i;
                    }
                }
            }
        }
        // Buffers that were full: compact (every query of the wavefront, which raises all thresholds) and offer
        // the candidates that were turned away again.  A second compaction covers a tile whose candidates alone
        // overfill the space a compaction leaves; after that the query is given up.
        int offers = 0;
        while (__ballot(fail != 0u) != 0ull) {
            if (offers == 2) {
                if (fail != 0u) {
                    ovf = true;
                    thr = INFINITY;
                }
                break;
            }
            ++offers;
            compact();
            const unsigned again = fail;
            fail = 0u;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (((again >> i) & 1u) && sc[i] > thr) {
                    if (!append(sc[i], i0 + tk_rowpos(i) + 4 * h)) fail |= 1u << i;
                }
            }
        }
    };
    for (int seg_a = 0; seg_a < ntile;) {
        int seg_b = seg_a == 0 ? 1 : (seg_a < SEG ? 2 * seg_a : seg_a + SEG);
        seg_b = seg_b < ntile ? seg_b : ntile;
        if (seg_a < t_early) {
            // The first tiles hold the items that most queries score highest (that is what the order of the sweep is
            // for) and the thresholds start from nothing: nearly every one of them would be marked by every group.
            // They are not swept: every wavefront works them, stretch by stretch as below, without a barrier.
            seg_b = seg_b < t_early ? seg_b : t_early;
            marks = seg_b - seg_a >= 64 ? ~0ull : (1ull << (seg_b - seg_a)) - 1ull;
        } else {
            if (h == 0) xthr[q] = thr;
            __syncthreads();  // everybody's thresholds are up; the marks of the stretch before have been taken
            if (sweeper) {
                sweep_stretch(seg_a, seg_b);
                if (lane == 0) xmark[w + NS] = marks1;
                marks1 = 0ull;
            } else {
                load_until(seg_b + PD - 1 < ntile ? seg_b + PD - 1 : ntile);
            }
            __syncthreads();
            if (!sweeper) {
                const unsigned long long m = xmark[w];
                marks = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)m) |
                        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(m >> 32)) << 32);
            }
        }
        // the marked tiles in ascending order (the filter is walked forward only)
        if (marks != 0ull) {
            while (marks != 0ull) {
                const int b = __builtin_ctzll(marks);
                marks &= marks - 1ull;
                work_tile(seg_a + b);
            }
            // thresholds only rise at a compaction: what the stretch has added is counted in before the next
            compact();
        }
        seg_a = seg_b;
    }

    // (every load from the slot has come back: the barrier behind the last sweep waited for them)
    if (tid == 0) atomicExch(stash_flags + sync[L::SYNC_WORDS - 1], 0);
    // ---- margin set -> exact scores -> the k best, for the 32 queries of the wavefront
    bool over;
    int nfin = tk_compact_all<CAP, NQ, L::LIMIT>(sel_s, sel_i, cnt, q, h, k, marg, thr, over);
    ovf = ovf || over;
    {
        const tk_pair_u32 po = tk_pair(ovf ? 1u : 0u);
        ovf = (po.x | po.y) != 0u;
    }
    const bool hand_over = uv && (unfit || ovf || (mask_indptr != nullptr && nfin < k));
    const bool answer = uv && !hand_over;
    nfin = answer ? nfin : 0;
    if (h == 0) {
        if (hand_over) qlist[atomicAdd(qcount, 1)] = (int32_t)ug;
        cnt[q] = answer ? nfin : -1;
    }
    // exact score of every survivor: the fp32 chain s = fma(q[k], y[k], s), ascending k, from +0.  The lane in
    // half h takes the entries 2j + h of its query, whose row it keeps in registers.  The item rows are read by
    // four lanes each: a lane that fetched its own 512-byte row 16 bytes at a time would touch 64 cache lines per
    // load instruction of the wavefront, and the lookups, not the bytes, set the pace.  The four lanes of a quad
    // take their four rows in turn -- each lane loads a quarter of every 64-byte piece of the row on turn -- and run
    // the chain with the row's values passed around the quad by the instruction's own lane select; the lane whose
    // row it is keeps the sum (the other three multiply their own query with a row that is not theirs, for nothing).
    {
        f32x4 qv[F / 4];
        const float* qr = Q + (uv ? ug : 0) * F;
#pragma unroll
        for (int k4 = 0; k4 < F / 4; ++k4) qv[k4] = *reinterpret_cast<const f32x4*>(qr + 4 * k4);
        constexpr int NSEG = F / 16;  // 64-byte pieces of a row
        const int r4 = lane & 3;
        for (int j = 0;; ++j) {
            const int e = 2 * j + h;
            const bool act = e < nfin;
            if (__ballot(act) == 0ull) break;
            int id = act ? sel_i[e * NQ + q] : 0;
            if (order != nullptr) {  // the buffers hold positions of the sweep: from here on, items
                id = order[id];
                id = id < 0 ? 0 : (id < ni ? id : (int)ni - 1);
                if (act) sel_i[e * NQ + q] = id;
            }
            float mine = 0.f;
            // the four rows of the quad, by turn
            const int id0 = __builtin_amdgcn_mov_dpp(id, 0x00, 0xf, 0xf, true);
            const int id1 = __builtin_amdgcn_mov_dpp(id, 0x55, 0xf, 0xf, true);
            const int id2 = __builtin_amdgcn_mov_dpp(id, 0xaa, 0xf, 0xf, true);
            const int id3 = __builtin_amdgcn_mov_dpp(id, 0xff, 0xf, 0xf, true);
            f32x4 ya[NSEG], yb[NSEG];
            auto load_row = [&](int item, f32x4 (&yv)[NSEG]) __attribute__((always_inline)) {
                const float* yr = Y + (int64_t)item * F + 4 * r4;
#pragma unroll
                for (int g = 0; g < NSEG; ++g) yv[g] = *reinterpret_cast<const f32x4*>(yr + 16 * g);
            };
            // s = fma(q, y of lane `src` of the quad, s): the multiply-add takes its first factor through the quad
            // permutation, so passing the row around costs no instruction of its own
            auto chain = [&](const f32x4 (&yv)[NSEG]) __attribute__((always_inline)) -> float {
                float s = 0.f;
#pragma unroll
                for (int g = 0; g < NSEG; ++g) {
                    // elements 16 g .. 16 g + 15: four from each lane of the quad, in lane order
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf"
                            : "+v"(s) : "v"(yv[g][x]), "v"(qv[4 * g][x]));
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf"
                            : "+v"(s) : "v"(yv[g][x]), "v"(qv[4 * g + 1][x]));
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf"
                            : "+v"(s) : "v"(yv[g][x]), "v"(qv[4 * g + 2][x]));
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
                            : "+v"(s) : "v"(yv[g][x]), "v"(qv[4 * g + 3][x]));
                }
                return s;
            };
            load_row(id0, ya);
            load_row(id1, yb);
            __builtin_amdgcn_sched_barrier(0);
            float s = chain(ya);
            mine = r4 == 0 ? s : mine;
            __builtin_amdgcn_sched_barrier(0);
            load_row(id2, ya);
            __builtin_amdgcn_sched_barrier(0);
            s = chain(yb);
            mine = r4 == 1 ? s : mine;
            __builtin_amdgcn_sched_barrier(0);
            load_row(id3, yb);
            __builtin_amdgcn_sched_barrier(0);
            s = chain(ya);
            mine = r4 == 2 ? s : mine;
            __builtin_amdgcn_sched_barrier(0);
            s = chain(yb);
            mine = r4 == 3 ? s : mine;
            if (act) sel_s[e * NQ + q] = mine;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // Order and write: two queries at a time, one per half-wave, one survivor per lane (at most LIMIT <= 32):
    // its rank under (score desc, item asc) is the number of survivors that are better.
    for (int p = 0; p < 16; ++p) {
        const int qq = w * 32 + 2 * p + h;
        const int n2 = cnt[qq];  // -1: not answered here
        const int nmax = max(__builtin_amdgcn_readlane(n2, 0), __builtin_amdgcn_readlane(n2, 32));
        const bool mine = c < n2;
        const float ms = mine ? sel_s[c * NQ + qq] : -INFINITY;
        const int mi = mine ? sel_i[c * NQ + qq] : 0x7fffffff;
        int rank = 0;
        for (int j = 0; j < nmax; ++j) {
            const float sj = sel_s[j * NQ + qq];
            const int ij = sel_i[j * NQ + qq];
            rank += (j < n2 && better(sj, ij, ms, mi)) ? 1 : 0;
        }
        const int64_t uq = (int64_t)blockIdx.x * NQ + qq;
        if (mine && rank < k) {
            out_ids[uq * k + rank] = mi;
            out_scores[uq * k + rank] = ms;
        }
        if (n2 >= 0 && c >= n2 && c < k) {  // fewer than k survivors: the tail is empty
            out_ids[uq * k + c] = -1;
            out_scores[uq * k + c] = -FLT_MAX;
        }
    }
}

template <int F, int NW, int CAP, int PD>
static int launch_topk_planes_cfg(const float* Q, int64_t nq, const float* Y, int64_t ni, const int64_t* mask_indptr,
                                  const int32_t* mask_indices, const int32_t* item_order, const int32_t* mask_blocks,
                                  const int32_t* block_len, int k, int32_t* out_ids, float* out_scores, void* ws,
                                  hipStream_t stream) {
    using L = TopkPlanesLds<F, NW, CAP, PD>;
    static_assert(L::BYTES <= 160 * 1024, "top-K plane kernel: LDS image too large");
    static OncePerDevice once;  // idempotent; a race only repeats the call
    bool* attr_done = once.slot();
    MIREC_REQUIRE(attr_done != nullptr, MIREC_E_HIP, "cannot read the current device ordinal (or it is >= %d)", kMaxDevices);
    if (!*attr_done) {
        MIREC_HIP(hipFuncSetAttribute((const void*)topk_planes_kernel<F, NW, CAP, PD, false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES));
        MIREC_HIP(hipFuncSetAttribute((const void*)topk_planes_kernel<F, NW, CAP, PD, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES));
        *attr_done = true;
    }
    const TopkWorkspace lay = topk_workspace_layout(nq, ni, F);
    char* base = static_cast<char*>(ws);
    uint32_t* ymax_bits = reinterpret_cast<uint32_t*>(base);
    int32_t* qcount = reinterpret_cast<int32_t*>(base + 4);
    int32_t* qlist = reinterpret_cast<int32_t*>(base + lay.qlist_off);
    float* tile_norm = reinterpret_cast<float*>(base + lay.tnorm_off);
    uint32_t* planes = reinterpret_cast<uint32_t*>(base + lay.planes_off);
    int32_t* stash_flags = reinterpret_cast<int32_t*>(base + lay.flags_off);
    tk_u32x4* stash = reinterpret_cast<tk_u32x4*>(base + lay.stash_off);
    const int ntile = (int)((ni + TOPK_BI - 1) / TOPK_BI);

    MIREC_HIP(hipMemsetAsync(base, 0, 256, stream));
    MIREC_HIP(hipMemsetAsync(stash_flags, 0, (size_t)lay.stash_slots * 4, stream));
    if (item_order == nullptr && mask_indptr == nullptr) {
        int32_t* bins = reinterpret_cast<int32_t*>(base + lay.bins_off);
        int32_t* made = reinterpret_cast<int32_t*>(base + lay.order_off);
        MIREC_HIP(hipMemsetAsync(bins, 0, TOPK_ORDER_BINS * 4, stream));
        const unsigned nb = (unsigned)((ni + 255) / 256);
        hipLaunchKernelGGL((topk_order_count_kernel<F>), dim3(nb), dim3(256), 0, stream, Y, ni, bins);
        hipLaunchKernelGGL((topk_order_scan_kernel<F>), dim3(1), dim3(1024), 0, stream, bins);
        hipLaunchKernelGGL((topk_order_place_kernel<F>), dim3(nb), dim3(256), 0, stream, Y, ni, bins, made);
        MIREC_LAUNCH_CHECK();
        item_order = made;
    }
    hipLaunchKernelGGL((topk_split_items_kernel<F>), dim3((unsigned)ntile), dim3(32 * (F / 8)), 0, stream, Y, ni,
                       item_order, planes, tile_norm, ymax_bits);
    MIREC_LAUNCH_CHECK();
    const int64_t blocks = (nq + L::NQ - 1) / L::NQ;
    if (item_order != nullptr) {
        // the sweep meets positions of the order; its filter is the lists per 32 queries, in positions
        hipLaunchKernelGGL((topk_planes_kernel<F, NW, CAP, PD, true>), dim3((unsigned)blocks), dim3(64 * NW), L::BYTES, stream,
                           Q, nq, Y, ni, planes, tile_norm, ymax_bits, mask_indptr, mask_blocks, block_len, item_order, k,
                           out_ids, out_scores, qlist, qcount, stash_flags, lay.stash_slots, stash);
    } else {
        hipLaunchKernelGGL((topk_planes_kernel<F, NW, CAP, PD, false>), dim3((unsigned)blocks), dim3(64 * NW), L::BYTES, stream,
                           Q, nq, Y, ni, planes, tile_norm, ymax_bits, mask_indptr, mask_indices, nullptr, nullptr, k, out_ids,
                           out_scores, qlist, qcount, stash_flags, lay.stash_slots, stash);
    }
    MIREC_LAUNCH_CHECK();
    // the queries handed over (none, as a rule): the fp32 kernel in list mode, sized for the worst case
    return launch_topk_fused<F>(Q, nq, Y, ni, mask_indptr, mask_indices, nullptr, nullptr, k, out_ids, out_scores, qlist,
                                qcount, stream);
}

// Entry of a width.  The bf16-plane kernel needs the caller's workspace; without one, for other widths, for
// k > 24, for the similar-items mode (per-score divisions) and for tiny item sets the fp32 kernel runs alone.
template <int F>
int launch_topk(const float* Q, int64_t nq, const float* Y, int64_t ni, const int64_t* mask_indptr,
                const int32_t* mask_indices, const int32_t* item_order, const int32_t* mask_blocks,
                const int32_t* block_len, const float* y_norm, const float* q_norm, int k, int32_t* out_ids,
                float* out_scores, void* ws, size_t ws_bytes, hipStream_t stream) {
    if constexpr (F % 16 == 0 && F <= 128) {
        if (ws != nullptr && y_norm == nullptr && q_norm == nullptr && topk_planes_supported(F, k) && ni >= 2 * TOPK_BI) {
            MIREC_REQUIRE(ws_bytes >= topk_workspace_layout(nq, ni, F).bytes, MIREC_E_INVALID,
                          "mirec_topk: workspace of %zu bytes given, mirec_topk_workspace_bytes asks for %zu", ws_bytes,
                          topk_workspace_layout(nq, ni, F).bytes);
            MIREC_REQUIRE(((uintptr_t)ws & 255) == 0, MIREC_E_INVALID, "mirec_topk: workspace must be 256-byte aligned");
            return launch_topk_planes_cfg<F, 8, 48, 6>(Q, nq, Y, ni, mask_indptr, mask_indices, item_order, mask_blocks,
                                                       block_len, k, out_ids, out_scores, ws, stream);
        }
    }
    return launch_topk_fused<F>(Q, nq, Y, ni, mask_indptr, mask_indices, y_norm, q_norm, k, out_ids, out_scores, nullptr,
                                nullptr, stream);
}

}  // namespace mirec
