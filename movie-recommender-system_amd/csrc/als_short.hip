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
// Two kernels share the passes over the gathered rows and differ in where the
// dense part, G v, is evaluated: als_short_kernel (every width) on the vector
// ALU from a copy of G in LDS, als_short_mfma_kernel (F = 64 and 128, the
// widths of the reference's configurations) on the matrix cores, by wavefronts
// set aside for it -- see the comment above that kernel.
//
// Layout: persistent workgroups of 16 wavefronts (8 beyond F = 128); G (F x F)
// sits in LDS, shared by all of them; the rest of the LDS is split into one
// cache per wavefront that keeps the first CAP gathered rows of the current
// CSR row, so short rows are fetched once and re-read on chip.  Within a
// wavefront the two 32-lane halves take alternate gathered rows (16-byte
// loads at F = 128: lane c owns columns [c*NT, c*NT+NT)), GV rows each per
// step; their GV dot products are reduced across the 32 lanes by one
// reduce-scatter butterfly (about 2 shuffles per row instead of 5), after
// which a group of 32/GV adjacent lanes holds row j's dot.  A CSR row whose last
// step has at most GV entries left ends with a half step (GV/2 gathered rows per
// half-wave): half the gathers and arithmetic for the same bits (1 M x 27 k
// user half-sweep: 8.06 -> 7.89 ms).
#include "als_internal.h"

namespace mirec {

// Wavefronts per workgroup and gathered rows per half-wave per step.  The kernel is bound by vector-instruction
// issue and latency, not by registers well spent: measured on the 1 M x 27 k user half-sweep at F = 128,
// (waves, rows/step) = (8,16) 12.8 ms, (8,8) 14.3, (12,8) 10.7, (16,8) 10.5 -- so up to F = 128 the kernel runs
// 16 waves (4 per SIMD, 128 VGPRs); wider rows keep 8 waves.
#ifndef MIREC_SYMV_UNROLL
#define MIREC_SYMV_UNROLL 8
#endif
constexpr int GV = 8;          // gathered rows per half-wave per step
constexpr int STEP = 2 * GV;   // stored entries per step (both halves)
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

// Cross-lane moves that stay in the vector ALU: DPP row operations (16-lane rows) and the lane-swap
// instructions; none of them touches the LDS pipeline that the G reads and the broadcasts already load.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_ROR8 = 0x128;        // rotate within a row by 8 lanes (= lane ^ 8)
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7 - i within groups of 8

// Lanes of the first row of each half-wave get a(self) + a(lane ^ 16), lanes of the second row b(self) + b(lane ^ 16).
__device__ __forceinline__ float swap16_sum(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// a(self) + a(lane ^ 32)
__device__ __forceinline__ float swap32_sum(float a) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Sum over the 32 lanes of a half-wave; all lanes get the result, and the SAME BITS: every stage pairs two lanes
// that add each other's value, so all lanes build one tree.  (A rotation by four lanes in place of the mirror gives
// the quads of a row two different trees, sums that differ in the last bit from lane to lane, and with them lanes
// that disagree about a threshold the sum is compared with -- the exponent of a vector's norm, once in ~1e8 rows.)
__device__ __forceinline__ float half_sum(float v) {
    v += dpp_mov<DPP_XOR1>(v);
    v += dpp_mov<DPP_XOR2>(v);
    v += dpp_mov<DPP_HALF_MIRROR>(v);  // quads are uniform by now: lane 7 - i stands for lane i ^ 4
    v += dpp_mov<DPP_ROR8>(v);
    return swap16_sum(v, v);
}

// GV = 8 values per lane -> lane c of each half ends with the 32-lane sum of value c >> 2.  Three halving
// stages (lane bits 4, 3 and the mirror pairing that flips bits 0-2), then the quad is summed.
__device__ __forceinline__ float reduce_scatter(const float (&p)[GV], int c) {
    float q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = swap16_sum(p[i], p[i + 4]);
    const bool b3 = (c & 8) != 0;
    float r[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b3 ? q[i + 2] : q[i];
        const float send = b3 ? q[i] : q[i + 2];
        r[i] = keep + dpp_mov<DPP_ROR8>(send);
    }
    const bool b2 = (c & 4) != 0;
    const float keep = b2 ? r[1] : r[0];
    const float send = b2 ? r[0] : r[1];
    float d = keep + dpp_mov<DPP_HALF_MIRROR>(send);
    d += dpp_mov<DPP_XOR1>(d);
    d += dpp_mov<DPP_XOR2>(d);
    return d;
}

// The same for 4 values per lane: lane c of each half ends with the 32-lane sum of value c >> 3.  The additions pair
// the lanes in the order of reduce_scatter (bit 4, bit 3, the mirror, bits 0 and 1), so an entry's dot product has
// the same bits whichever of the two reduces it.
__device__ __forceinline__ float reduce_scatter4(const float (&p)[4], int c) {
    float q[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) q[i] = swap16_sum(p[i], p[i + 2]);
    const bool b3 = (c & 8) != 0;
    const float keep = b3 ? q[1] : q[0];
    const float send = b3 ? q[0] : q[1];
    float d = keep + dpp_mov<DPP_ROR8>(send);
    d += dpp_mov<DPP_HALF_MIRROR>(d);
    d += dpp_mov<DPP_XOR1>(d);
    d += dpp_mov<DPP_XOR2>(d);
    return d;
}

// Address of this lane's NT columns of gathered row `sel`.  Tables below 4 GiB carry `sel` as a byte offset
// (one 32-bit add per gather, the base stays in scalar registers); larger ones as a row index.
template <int NT, bool BIG>
__device__ __forceinline__ const float* gather_ptr(const float* Y, int sel, int c) {
    if constexpr (BIG) {
        return Y + (int64_t)sel * (32 * NT) + c * NT;
    } else {
        return reinterpret_cast<const float*>(reinterpret_cast<const char*>(Y) +
                                              ((uint32_t)sel + (uint32_t)(c * NT * 4)));
    }
}

// The arithmetic of a step on the gathered rows yv[0 .. GVN): the dot products with v, their reduction, the
// coefficients and the update of acc.
template <int NT, bool FIRST, int GVN>
__device__ __forceinline__ void step_math(const float (&yv)[GV][NT], const float (&v)[NT], float (&acc)[NT], int g0,
                                          int n, float cur_dat, int coefbase, int h, int c) {
    float part[GVN];
#pragma unroll
    for (int j = 0; j < GVN; ++j) {
        float s0 = yv[j][0] * v[0];
        float s1 = NT > 1 ? yv[j][NT > 1 ? 1 : 0] * v[NT > 1 ? 1 : 0] : 0.f;
#pragma unroll
        for (int t = 2; t < NT; ++t) {
            if (t & 1) s1 = fmaf(yv[j][t], v[t], s1);
            else s0 = fmaf(yv[j][t], v[t], s0);
        }
        part[j] = s0 + s1;
    }
    float d;
    if constexpr (GVN == GV) d = reduce_scatter(part, c);
    else d = reduce_scatter4(part, c);
    constexpr int LANES = 32 / GVN;  // adjacent lanes of a half that end with the same entry's dot product
    const int kkm = g0 + 2 * (c / LANES) + h;
    const float w = fabsf(cur_dat) - 1.f;
    float coef = FIRST ? ((cur_dat > 0.f ? cur_dat : 0.f) - w * d) : (w * d);
    coef = kkm < n ? coef : 0.f;
#pragma unroll
    for (int j = 0; j < GVN; ++j) {
        const float cj = __int_as_float(__builtin_amdgcn_ds_bpermute(coefbase + 4 * LANES * j, __float_as_int(coef)));
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = fmaf(cj, yv[j][t], acc[t]);
    }
}

// One step of a pass: stored entries [g0, g0 + 16), half h taking entry g0 + 2j + h as its j-th.  Entries
// past the end of the row repeat the row's last entry (always loadable) and get coefficient zero, so no
// load is conditional.  MODE 0: every entry slot of the step lies inside this wave's on-chip cache,
// 1: the step straddles the end of the cache (gc is then a compile-time constant), 2: it lies past it.
// gc = position of the step in the cache (the cache may start at a later entry than the row).
// GVN = 8: a full step; GVN = 4: the half step that ends a row with at most 8 entries left (entries
// [g0, g0 + 8), the same assignment to halves and cache slots, half the gathers and arithmetic).
template <int NT, int CAP, bool FIRST, bool BIG, int MODE, int GVN = GV>
__device__ __forceinline__ void sparse_step(const float (&v)[NT], float (&acc)[NT], const float* __restrict__ Y,
                                            float* cache, int g0, int n, int cur_sel, float cur_dat, int selbase,
                                            int coefbase, int h, int c, int gc) {
    constexpr int F = 32 * NT;
    constexpr int G0C = (CAP / STEP) * STEP;
    float yv[GV][NT];
    int sel[GVN];
    float* slot = cache + (gc + h) * F + c * NT;
#pragma unroll
    for (int j = 0; j < GVN; ++j) {
        const bool inside = MODE == 0 || (MODE == 1 && G0C + 2 * j < CAP);
        if (FIRST || !inside) sel[j] = __builtin_amdgcn_ds_bpermute(selbase + 8 * j, cur_sel);
    }
#pragma unroll
    for (int j = 0; j < GVN; ++j) {
        const bool inside = MODE == 0 || (MODE == 1 && G0C + 2 * j < CAP);
        if (FIRST || !inside) {
            load_n<NT>(yv[j], gather_ptr<NT, BIG>(Y, sel[j], c));
        } else {
            load_n<NT>(yv[j], slot + 2 * j * F);
        }
    }
    if constexpr (FIRST) {
#pragma unroll
        for (int j = 0; j < GVN; ++j) {
            const bool inside = MODE == 0 || (MODE == 1 && G0C + 2 * j < CAP);
            if (inside) store_n<NT>(slot + 2 * j * F, yv[j]);
        }
    }
    step_math<NT, FIRST, GVN>(yv, v, acc, g0, n, cur_dat, coefbase, h, c);
}

// The same step on rows that stay in registers for all passes of a CSR row: the first pass gathers them into
// R, the later ones only do the arithmetic.
template <int NT, bool FIRST, bool BIG, int GVN>
__device__ __forceinline__ void reg_step(float (&R)[GV][NT], const float (&v)[NT], float (&acc)[NT],
                                         const float* __restrict__ Y, int g0, int n, int cur_sel, float cur_dat,
                                         int selbase, int coefbase, int h, int c) {
    if constexpr (FIRST) {
        int sel[GVN];
#pragma unroll
        for (int j = 0; j < GVN; ++j) sel[j] = __builtin_amdgcn_ds_bpermute(selbase + 8 * j, cur_sel);
#pragma unroll
        for (int j = 0; j < GVN; ++j) load_n<NT>(R[j], gather_ptr<NT, BIG>(Y, sel[j], c));
        // every register of R is written in the first pass of a CSR row (the later passes read the same ones,
        // which the compiler cannot see: left unwritten they would count as carried over from the previous row)
#pragma unroll
        for (int j = GVN; j < GV; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) R[j][t] = 0.f;
    }
    step_math<NT, FIRST, GVN>(R, v, acc, g0, n, cur_dat, coefbase, h, c);
}

// acc (in every lane: the sum over all the row's stored entries) of
//   FIRST:  (max(c,0) - (|c|-1)(y.v)) y        (start residual, v = x)
//   else :  (|c|-1)(y.v) y                     (operator, v = p)
// Lane l keeps, one step ahead, the index of entry g0 + (l & 15) and the confidence of the entry whose dot
// product the reduction leaves in it.
// RS: the first RS steps of the row work on rows held in registers (R, RS * 8 rows per half-wave); the cache
// then starts at entry RS * 16 of the row.
template <int NT, int CAP, bool FIRST, bool BIG, int RS = 0>
__device__ __forceinline__ void sparse_pass(const float (&v)[NT], float (&acc)[NT], const int32_t* __restrict__ indices,
                                            const float* __restrict__ data, const float* __restrict__ Y, float* cache,
                                            int n, int lane, int h, int c, float (*R)[GV][NT] = nullptr) {
    constexpr int F = 32 * NT;
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = 0.f;
    const int e_idx = lane & 15;
    const int e_dat = 2 * (c >> 2) + h;   // full step: lanes 4j .. 4j+3 of a half end with entry 2j + h
    const int e_dat4 = 2 * (c >> 3) + h;  // half step: lanes 8j .. 8j+7
    const int selbase = ((lane & 32) | h) << 2;
    const int coefbase = (lane & 32) << 2;
    const int last = n - 1;
    auto load_sel = [&](int g0) __attribute__((always_inline)) {
        const int k = g0 + e_idx;
        return indices[k < last ? k : last];
    };
    // (a step that starts with at most GV entries left is a half step)
    auto load_dat = [&](int g0) __attribute__((always_inline)) {
        const int k = g0 + (n - g0 <= GV ? e_dat4 : e_dat);
        return data[k < last ? k : last];
    };
    int nxt_sel = load_sel(0);
    float nxt_dat = load_dat(0);
    constexpr int GR = RS * STEP;  // entries of the row that live in registers
#pragma unroll
    for (int sidx = 0; sidx < RS; ++sidx) {
        const int g0 = sidx * STEP;
        if (g0 < n) {
            const int cur_sel = BIG ? nxt_sel : (int)((uint32_t)nxt_sel * (uint32_t)(F * 4));
            const float cur_dat = nxt_dat;
            if (g0 + STEP < n) {
                nxt_sel = load_sel(g0 + STEP);
                nxt_dat = load_dat(g0 + STEP);
            }
            if (n - g0 <= GV) reg_step<NT, FIRST, BIG, GV / 2>(R[sidx], v, acc, Y, g0, n, cur_sel, cur_dat, selbase, coefbase, h, c);
            else reg_step<NT, FIRST, BIG, GV>(R[sidx], v, acc, Y, g0, n, cur_sel, cur_dat, selbase, coefbase, h, c);
        } else if constexpr (FIRST) {
#pragma unroll
            for (int j = 0; j < GV; ++j)
#pragma unroll
                for (int t = 0; t < NT; ++t) R[sidx][j][t] = 0.f;
        }
    }
    for (int g0 = GR; g0 < n; g0 += STEP) {
        // scaled to a byte offset only now, so that the load above stays in flight through the previous step
        const int cur_sel = BIG ? nxt_sel : (int)((uint32_t)nxt_sel * (uint32_t)(F * 4));
        const float cur_dat = nxt_dat;
        if (g0 + STEP < n) {
            nxt_sel = load_sel(g0 + STEP);
            nxt_dat = load_dat(g0 + STEP);
        }
        const int gc = g0 - GR;
        if (n - g0 <= GV) {
            if (gc + GV <= CAP) {
                sparse_step<NT, CAP, FIRST, BIG, 0, GV / 2>(v, acc, Y, cache, g0, n, cur_sel, cur_dat, selbase, coefbase, h, c, gc);
            } else if (CAP % STEP == 0 || gc >= CAP) {
                sparse_step<NT, CAP, FIRST, BIG, 2, GV / 2>(v, acc, Y, cache, g0, n, cur_sel, cur_dat, selbase, coefbase, h, c, gc);
            } else {
                if constexpr (CAP % STEP != 0)
                    sparse_step<NT, CAP, FIRST, BIG, 1, GV / 2>(v, acc, Y, cache, g0, n, cur_sel, cur_dat, selbase,
                                                                coefbase, h, c, (CAP / STEP) * STEP);
            }
        } else if (gc + STEP <= CAP) {
            sparse_step<NT, CAP, FIRST, BIG, 0>(v, acc, Y, cache, g0, n, cur_sel, cur_dat, selbase, coefbase, h, c, gc);
        } else if (CAP % STEP == 0 || gc >= CAP) {
            sparse_step<NT, CAP, FIRST, BIG, 2>(v, acc, Y, cache, g0, n, cur_sel, cur_dat, selbase, coefbase, h, c, gc);
        } else {
            if constexpr (CAP % STEP != 0)
                sparse_step<NT, CAP, FIRST, BIG, 1>(v, acc, Y, cache, g0, n, cur_sel, cur_dat, selbase, coefbase, h, c,
                                                    (CAP / STEP) * STEP);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = swap32_sum(acc[t]);
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
#pragma unroll MIREC_SYMV_UNROLL
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
    for (int t = 0; t < NT; ++t) out[t] = swap32_sum(out[t]);
}

template <int NT, bool BIG>
__global__ __launch_bounds__(64 * ShortWaves<NT>::SW) void als_short_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ data,
    int64_t row_begin, int64_t nrows, float* X, const float* __restrict__ Y, const float* __restrict__ G,
    int32_t short_max, int cg_steps) {
    constexpr int F = 32 * NT;
    using L = ShortLds<NT>;
    constexpr int CAP = L::CAP;
    constexpr int SW = L::SW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // LDS: the waves' staging vectors first (their reads then address with immediates), G, the row caches
    float* Gs = smem + SW * F;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int h = lane >> 5;
    const int c = lane & 31;
    float* cache = smem + SW * F + F * F + w * (CAP * F);
    float* vec = smem + w * F;

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

        sparse_pass<NT, CAP, true, BIG>(x, acc, ind, dat, Y, cache, n, lane, h, c);
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
            sparse_pass<NT, CAP, false, BIG>(p, acc, ind, dat, Y, cache, n, lane, h, c);
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

// ---------------------------------------------------------------------------------------------------
// The same solver with the four G v products of a row on the matrix cores (F = 64 and 128).
//
// On the vector ALU a product costs F*F/64 fused multiply-adds per lane and a read of all of G from LDS, per
// row; with the gathers hidden that was a third of the kernel.  A matrix-vector product does not fill a matrix
// instruction, several of them do.  The workgroup therefore has two kinds of wavefronts:
//   NS solvers        each with a row of its own, through the same passes as above, but where the vector
//                     kernel multiplies by G they PUBLISH the vector -- scaled by a power of two and split
//                     into two fp16 planes (v = hi + mid to 22 bits, as the explicit-Gram kernel splits its
//                     rows) into their row of an LDS image, then a sequence number -- run the pass over the
//                     gathered rows (the product is not needed before its end) and COLLECT the product;
//   NSRV servers      each holding its share of G as scaled fp16 fragments in registers.  A server polls the
//                     sequence numbers, and once the vectors wait evaluates its bands of G times all of
//                     them with v_mfma_f32_16x16x32_f16 (three plane products per k step, the smallest
//                     first), stores the columns that were waiting and counts them done.
// Nobody waits at a barrier: a solver waits only for its own product, which the servers evaluate while it is
// busy with its pass, so solvers keep hiding each other's memory latency as before.  A column of a product
// depends on its vector only, so a row's result does not depend on what else was in flight.
// Width 128: ten solvers and two servers, twelve wavefronts of 168 registers; width 64: fourteen solvers and two
// servers, sixteen wavefronts of 128 registers.  With G out of LDS a solver keeps the first 16 * RS gathered rows of
// its CSR row in registers and the next CAP in LDS for all four passes: rows of up to 16 * RS + CAP entries are
// gathered once.  Measured on the 1 M x 27 k user half-sweep (F = 128): RS = 1 6.17 ms, RS = 2 7.0 ms (the
// second set of registers spills), sixteen wavefronts of 128 registers with bf16 planes and no rows in
// registers 6.34 ms.
typedef __attribute__((ext_vector_type(8))) _Float16 sm_f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 sm_f16x2;
typedef __attribute__((ext_vector_type(2))) float sm_f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t sm_u32x4;

__device__ __forceinline__ uint32_t sm_cvt_pk(float a, float b) {
    const sm_f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, sm_f16x2));
}

// (a, b) -> packed fp16 pairs (a in the low half) with hi + mid == value to 22 bits: round to nearest at both
// levels, the remainder exact in fp32.
__device__ __forceinline__ void sm_split2(float a, float b, uint32_t& hi, uint32_t& mid) {
    hi = sm_cvt_pk(a, b);
    const sm_f16x2 hv = __builtin_bit_cast(sm_f16x2, hi);
    a -= (float)hv[0];
    b -= (float)hv[1];
    mid = sm_cvt_pk(a, b);
}

__device__ __forceinline__ f32x4 sm_mma(const sm_u32x4& a, const sm_u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sm_f16x8, a), __builtin_bit_cast(sm_f16x8, b), c,
                                                  0, 0, 0);
}

// Power of two 2^s with bound * 2^s < 2^15 for any finite bound (|values| <= bound then fit fp16 and those that
// matter are normal numbers), as (2^s, 2^-s).  s is kept within [-113, 73] so that both are normal fp32 numbers.
__device__ __forceinline__ sm_f32x2 sm_scale_for(float bound) {
    int e = (int)((__float_as_uint(bound) >> 23) & 0xffu);   // biased exponent: bound < 2^(e - 126)
    int se = 268 - e;                                         // biased exponent of 2^(14 - (e - 127))
    se = se > 200 ? 200 : se;
    return sm_f32x2{__uint_as_float((uint32_t)se << 23), __uint_as_float((uint32_t)(254 - se) << 23)};
}

// Steps (of 16 stored entries) of a CSR row whose gathered rows stay in registers.
#ifndef MIREC_SHORT_RS
#define MIREC_SHORT_RS 1
#endif
// Wavefronts of the workgroup and how many of them are servers, for width 128.  The kernel's rate follows the
// number of solvers, not the number of gathered rows they can keep on chip: 8 wavefronts of 256 registers with
// 60 / 92 / 108 rows on chip (RS = 1 / 3 / 4) ran the 1 M x 27 k user half-sweep in 7.9 / 7.5 / 7.7 ms, 10
// wavefronts in 6.5 ms, 12 in 6.2 ms.  Sixteen (four per SIMD, 128 registers: four servers with a quarter of G
// each) have no room for the register-resident rows: 8.1 ms with them spilled, 6.2 ms without them.
#ifndef MIREC_SHORT_NW
#define MIREC_SHORT_NW 12
#endif
#ifndef MIREC_SHORT_NSRV
#define MIREC_SHORT_NSRV 2
#endif
// The same for width 64, where a solver needs 116 registers and half of G is 32: sixteen wavefronts ran the
// 1 M x 27 k user half-sweep in 4.01 ms with twelve solvers and four servers, 3.95 ms with fourteen and two, against
// the 4.37 ms of twelve wavefronts (ten solvers); same bits.
#ifndef MIREC_SHORT_NW64
#define MIREC_SHORT_NW64 16
#endif
#ifndef MIREC_SHORT_NSRV64
#define MIREC_SHORT_NSRV64 2
#endif
// A server sweeps when this many vectors wait (or every solver that still has rows has published).  Smaller
// batches shorten the first publisher's wait and lengthen everybody else's: 3 / 5 / 7 / 9 gave 6.5 / 6.9 / 7.0 /
// 7.0 ms against 6.2 ms for "all".
#ifndef MIREC_SRV_MIN_BATCH
#define MIREC_SRV_MIN_BATCH 16
#endif
// A server starts a sweep when every solver that still has rows has published, or after this many polls with at
// least one vector waiting (a bound, so that the loop ends whatever the solvers do; with it reached the sweep
// serves those that wait -- any split of the vectors over sweeps gives the same results).
// Units of a server whose sums are built side by side: 1 / 2 / 4 ran the 1 M x 27 k user half-sweep in 6.27 / 6.20 /
// 6.37 ms (four keep so many sums that fragments of G spill); same bits.
#ifndef MIREC_SRV_CHAINS
#define MIREC_SRV_CHAINS 2
#endif
#ifndef MIREC_SRV_MAX_POLLS
#define MIREC_SRV_MAX_POLLS 100000
#endif

template <int NT>
struct MfmaShort {
    static_assert(NT == 2 || NT == 4, "matrix-core G v: widths 64 and 128");
    static constexpr int F = 32 * NT;
    static constexpr int NW = NT == 2 ? MIREC_SHORT_NW64 : MIREC_SHORT_NW;        // wavefronts of the workgroup
    static constexpr int NSRV = NT == 2 ? MIREC_SHORT_NSRV64 : MIREC_SHORT_NSRV;  // servers (the last wavefronts)
    static_assert(NW % 4 == 0 && NW <= 16, "whole wavefronts per SIMD, at most 1024 lanes");
    static constexpr int NS = NW - NSRV;        // solvers
    static constexpr int COLS = 16;             // columns of a matrix instruction (NS of them used)
    static constexpr int RS = MIREC_SHORT_RS;   // steps of a CSR row whose gathered rows stay in registers
    static constexpr int TILES = F / 16;        // 16-row bands of G
    static constexpr int KP = 2;                // parts of the k range
    static constexpr int KSU = (F / 32) / KP;   // 32-wide k steps per part
    static constexpr int UNITS = TILES * KP;    // (band, part) pairs
    static constexpr int UPS = UNITS / NSRV;    // per server
    static constexpr int CH = F / 8;            // 16-byte chunks of a plane row
    static constexpr int ROWB = 2 * F;          // bytes of a plane row
    static constexpr int PLANE_FLOATS = COLS * ROWB / 4;
    static constexpr int PART_ROW = F + 4;      // floats; the odd number of 16-byte chunks spreads the stores over the banks
    // LDS map (floats): flags; two planes of the vectors; the partial products [KP][COLS][PART_ROW]; row caches
    static constexpr int FLAGS = 0;             // int pub[16], int done[16], int finished
    static constexpr int VPL = 64;
    static constexpr int PART = VPL + 2 * PLANE_FLOATS;
    static constexpr int CACHE = PART + KP * COLS * PART_ROW;
    static constexpr int TOTAL = 160 * 1024 / 4;
    static constexpr int CAP_RAW = (TOTAL - CACHE) / (NS * F);
    static constexpr int CAP = CAP_RAW >= 64 ? 64 : (CAP_RAW & ~1);
    static constexpr int FLOATS = CACHE + NS * CAP * F;
    static constexpr size_t BYTES = (size_t)FLOATS * 4;
    // chunk kc of row n sits at position kc ^ swz(n): the sixteen lanes of every ds_read_b128 group then read
    // sixteen different 16-byte bank groups
    __device__ static __forceinline__ int swz(int n) { return (n / (16 / CH)) % CH; }
};

__device__ __forceinline__ int lds_load_acquire(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store_release(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Largest diagonal entry of G (it bounds every |G_ij| of the positive semi-definite G), in every lane.
template <int F>
__device__ __forceinline__ float sm_gmax(const float* __restrict__ G, int lane) {
    float g = 0.f;
#pragma unroll
    for (int i = lane; i < F; i += 64) g = fmaxf(g, G[i * F + i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) g = fmaxf(g, __shfl_xor(g, off, 64));
    return g;
}

// Solver k: vector v, every |v_i| <= bound -> row k of the two planes (scaled by the power of two the bound gives),
// then the sequence number `seq` of this product.  Returns the factor that takes the collected sums back.
template <int NT>
__device__ __forceinline__ float symv_publish(float* smem, const float (&v)[NT], float bound, int k, int seq, int lane,
                                              int h, int c) {
    using M = MfmaShort<NT>;
    char* vpl = reinterpret_cast<char*>(smem + M::VPL);
    constexpr int PLANEB = M::PLANE_FLOATS * 4;
    // one scale per vector: the servers' sums mix all columns, so every lane must scale by the same power of two
    const sm_f32x2 sc = sm_scale_for(__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(bound))));
    // lane c holds columns c*NT .. c*NT + NT-1; the two halves hold the same values
    if (h == 0) {
        if constexpr (NT == 4) {
            uint32_t hi[2], mid[2];
            sm_split2(v[0] * sc[0], v[1] * sc[0], hi[0], mid[0]);
            sm_split2(v[2] * sc[0], v[3] * sc[0], hi[1], mid[1]);
            char* dst = vpl + k * M::ROWB + (((c >> 1) ^ M::swz(k)) << 4) + ((c & 1) << 3);
            *reinterpret_cast<uint2*>(dst) = make_uint2(hi[0], hi[1]);
            *reinterpret_cast<uint2*>(dst + PLANEB) = make_uint2(mid[0], mid[1]);
        } else {
            uint32_t hi, mid;
            sm_split2(v[0] * sc[0], v[1] * sc[0], hi, mid);
            char* dst = vpl + k * M::ROWB + (((c >> 2) ^ M::swz(k)) << 4) + ((c & 3) << 2);
            *reinterpret_cast<uint32_t*>(dst) = hi;
            *reinterpret_cast<uint32_t*>(dst + PLANEB) = mid;
        }
    }
    // the release orders the plane stores of the whole wavefront (one instruction stream) before the number
    int* pub = reinterpret_cast<int*>(smem + M::FLAGS);
    if (lane == 0) lds_store_release(pub + k, seq);
    return sc[1];
}

// Solver k: wait until all servers have counted product `seq` of this solver done, then out = G v
// (unscale = the factor symv_publish returned times the inverse of the scale of G).
template <int NT>
__device__ __forceinline__ void symv_collect(float* smem, const float (&v)[NT], float (&out)[NT], float unscale, int k,
                                             int seq, int c) {
    using M = MfmaShort<NT>;
    const int* done = reinterpret_cast<const int*>(smem + M::FLAGS) + 16;
    const int want = seq * M::NSRV;
    while (lds_load_acquire(done + k) != want) __builtin_amdgcn_s_sleep(1);
    const float* p0 = smem + M::PART + k * M::PART_ROW + c * NT;
    float a0[NT], a1[NT];
    load_n<NT>(a0, p0);
    load_n<NT>(a1, p0 + M::COLS * M::PART_ROW);
#pragma unroll
    for (int t = 0; t < NT; ++t) out[t] = (a0[t] + a1[t]) * unscale;
}

// Server `srv`: until every solver has finished, evaluate its bands of G times the vectors that wait.
template <int NT>
__device__ __forceinline__ void symv_server(float* smem, const float* __restrict__ G, float gscale, int srv, int lane) {
    using M = MfmaShort<NT>;
    constexpr int F = M::F;
    constexpr int PLANEB = M::PLANE_FLOATS * 4;
    const int n = lane & 15, q = lane >> 4;   // operand lane: column n (a solver's vector), k group q
    // fragments of G: unit u = (band t, k part kp); rows 16t + (lane & 15), columns 32 (kp KSU + s) + 8 q .. + 7
    sm_u32x4 Ag[M::UPS][2][M::KSU];
#pragma unroll
    for (int j = 0; j < M::UPS; ++j) {
        const int u = srv * M::UPS + j;
        const int t = u % M::TILES, kp = u / M::TILES;
#pragma unroll
        for (int s = 0; s < M::KSU; ++s) {
            const float* g = G + (16 * t + n) * F + 32 * (kp * M::KSU + s) + 8 * q;
            const float4 g0 = *reinterpret_cast<const float4*>(g);
            const float4 g1 = *reinterpret_cast<const float4*>(g + 4);
            uint32_t hi[4], mid[4];
            sm_split2(g0.x * gscale, g0.y * gscale, hi[0], mid[0]);
            sm_split2(g0.z * gscale, g0.w * gscale, hi[1], mid[1]);
            sm_split2(g1.x * gscale, g1.y * gscale, hi[2], mid[2]);
            sm_split2(g1.z * gscale, g1.w * gscale, hi[3], mid[3]);
            Ag[j][0][s] = sm_u32x4{hi[0], hi[1], hi[2], hi[3]};
            Ag[j][1][s] = sm_u32x4{mid[0], mid[1], mid[2], mid[3]};
        }
    }
    int* flags = reinterpret_cast<int*>(smem + M::FLAGS);
    const int* pub = flags;
    int* done = flags + 16;
    const int* finished = flags + 32;
    const char* vpl = reinterpret_cast<const char*>(smem + M::VPL);
    float* part = smem + M::PART;
    int seen = 0;    // products of solver n this server has evaluated
    int polls = 0;
    for (;;) {
        // (finished is read first: a solver publishes nothing after it has counted itself finished)
        const int nfin = lds_load_acquire(finished);
        const int pubv = n < M::NS ? lds_load_acquire(pub + n) : 0;
        const bool pend = pubv != seen;
        const unsigned waiting = (unsigned)__builtin_amdgcn_ballot_w64(pend) & 0xffffu;
        if (waiting == 0u) {
            if (nfin == M::NS) break;
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        // The solvers work on rows of equal length (the plan lists them longest first) and so publish at about
        // the same time: wait until every solver that still has rows has published, and one sweep serves them
        // all.  Sweeps for one or two vectors each would cost the matrix pipe as much as a full one, and there
        // would be six times as many (measured on 1 M x 27 k users: 7.5 ms against 6.6 ms for the half-sweep).
        const int active = M::NS - nfin;
        if (__builtin_popcount(waiting) < (active < MIREC_SRV_MIN_BATCH ? active : MIREC_SRV_MIN_BATCH) &&
            ++polls <= MIREC_SRV_MAX_POLLS) {
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        polls = 0;
        // the units of a server share their k part: its vector operands are read once per sweep
        static_assert(M::UPS <= M::TILES && M::TILES % M::UPS == 0, "the units of a server lie in one k part");
        const int kp = (srv * M::UPS) / M::TILES;
        sm_u32x4 b[2][M::KSU];
#pragma unroll
        for (int s = 0; s < M::KSU; ++s) {
            const int kc = 4 * (kp * M::KSU + s) + q;
            const char* src = vpl + n * M::ROWB + ((kc ^ M::swz(n)) << 4);
            b[0][s] = *reinterpret_cast<const sm_u32x4*>(src);
            b[1][s] = *reinterpret_cast<const sm_u32x4*>(src + PLANEB);
        }
        // (the sums of MIREC_SRV_CHAINS units are built side by side: a matrix instruction that adds to the result of
        // the one just before it waits for that result, the next unit's does not)
        constexpr int NCH = MIREC_SRV_CHAINS;
        static_assert(M::UPS % NCH == 0, "units of a server in groups");
#pragma unroll
        for (int j = 0; j < M::UPS; j += NCH) {
            f32x4 acc[NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < M::KSU; ++s) {
                // smallest products first; mid*mid is below 2^-22 of the term and dropped
#pragma unroll
                for (int i = 0; i < NCH; ++i) acc[i] = sm_mma(Ag[j + i][1][s], b[0][s], acc[i]);
#pragma unroll
                for (int i = 0; i < NCH; ++i) acc[i] = sm_mma(Ag[j + i][0][s], b[1][s], acc[i]);
#pragma unroll
                for (int i = 0; i < NCH; ++i) acc[i] = sm_mma(Ag[j + i][0][s], b[0][s], acc[i]);
            }
            // lane (n, q) holds rows 16t + 4q .. + 3 of G times vector n; only columns that were waiting count
            // (the others were read while their solver may have been writing them)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int t = (srv * M::UPS + j + i) % M::TILES;
                if (pend) *reinterpret_cast<f32x4*>(part + (kp * M::COLS + n) * M::PART_ROW + 16 * t + 4 * q) = acc[i];
            }
        }
        seen = pubv;
        // the release orders this wavefront's stores before the counts
        if (pend && q == 0) __hip_atomic_fetch_add(done + n, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}


template <int NT, bool BIG>
__global__ __launch_bounds__(64 * MfmaShort<NT>::NW) void als_short_mfma_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ data,
    int64_t row_begin, float* X, const float* __restrict__ Y, const float* __restrict__ G, int cg_steps,
    const int32_t* __restrict__ order, int64_t n_short) {
    using M = MfmaShort<NT>;
    constexpr int F = M::F;
    constexpr int CAP = M::CAP;
    constexpr int RS = M::RS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    int* flags = reinterpret_cast<int*>(smem + M::FLAGS);
    if (tid < 64) flags[tid] = 0;
    __syncthreads();
    // G enters the matrix instructions scaled by a power of two that its largest (diagonal) entry fixes
    const sm_f32x2 gsc = sm_scale_for(sm_gmax<F>(G, lane));
    if (w >= M::NS) {
        // the servers' few instructions go ahead of the solvers' on their SIMDs: every solver of the workgroup
        // waits for them (measured: 6.27 -> 6.23 ms on the 1 M x 27 k user half-sweep, same bits)
        __builtin_amdgcn_s_setprio(3);
        symv_server<NT>(smem, G, gsc[0], w - M::NS, lane);
        return;
    }
    const int h = lane >> 5;
    const int c = lane & 31;
    float* cache = smem + M::CACHE + w * (CAP * F);
    int seq = 0;  // products this solver has published

    // the solvers of a workgroup take consecutive rows of the plan's list (rows of about equal length)
    const int64_t stride = (int64_t)gridDim.x * M::NS;
    for (int64_t r = (int64_t)blockIdx.x * M::NS + w; r < n_short; r += stride) {
        const int64_t row = row_begin + order[r];
        const int64_t rs = indptr[row];
        const int n = (int)(indptr[row + 1] - rs);
        float* xg = X + row * F + c * NT;
        float x[NT], rr[NT], p[NT], Ap[NT], acc[NT];
        float R[RS > 0 ? RS : 1][GV][NT];  // the first RS * 16 gathered rows of this CSR row (8 per half-wave and step)
        if (n == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) x[t] = 0.f;
            if (h == 0) store_n<NT>(xg, x);
            continue;
        }
        load_n<NT>(x, xg);
        const int32_t* ind = indices + rs;
        const float* dat = data + rs;

        // |x_i| <= |x|: the norm bounds the entries (a reduction per row; the CG vectors come with bounds for free)
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) part = fmaf(x[t], x[t], part);
        float unscale = symv_publish<NT>(smem, x, sqrtf(half_sum(part)), w, ++seq, lane, h, c) * gsc[1];
        sparse_pass<NT, CAP, true, BIG, RS>(x, acc, ind, dat, Y, cache, n, lane, h, c, R);
        symv_collect<NT>(smem, x, Ap, unscale, w, seq, c);
        part = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            rr[t] = acc[t] - Ap[t];
            p[t] = rr[t];
            part = fmaf(rr[t], rr[t], part);
        }
        float rsold = half_sum(part);
        if (rsold < 1e-20f) continue;
        float pbound = sqrtf(rsold) * 1.0001f;  // |p| for p = r; afterwards |r + beta p| <= |r| + beta |p|
        for (int it = 0; it < cg_steps; ++it) {
            unscale = symv_publish<NT>(smem, p, pbound, w, ++seq, lane, h, c) * gsc[1];
            sparse_pass<NT, CAP, false, BIG, RS>(p, acc, ind, dat, Y, cache, n, lane, h, c, R);
            symv_collect<NT>(smem, p, Ap, unscale, w, seq, c);
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
            pbound = (sqrtf(rsnew) + fabsf(beta) * pbound) * 1.0001f;
            rsold = rsnew;
        }
        if (h == 0) store_n<NT>(xg, x);
    }
    // every product this solver published has been collected: the servers may stop once all solvers are here
    if (lane == 0) __hip_atomic_fetch_add(flags + 32, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int NT, bool BIG>
static int launch_short_mfma(const int64_t* indptr, const int32_t* indices, const float* data, int64_t row_begin,
                             float* X, const float* Y, const float* G, int cg_steps, const int32_t* order,
                             int64_t n_short, hipStream_t stream) {
    using M = MfmaShort<NT>;
    static_assert(M::CAP >= 2, "no room for a row cache");
    static OncePerDevice once;  // idempotent; a race only repeats the call
    static int cus_of[kMaxDevices] = {};
    bool* attr_done = once.slot();
    MIREC_REQUIRE(attr_done != nullptr, MIREC_E_HIP, "cannot read the current device ordinal (or it is >= %d)", kMaxDevices);
    int& cus = cus_of[attr_done - once.done];
    if (!*attr_done) {
        MIREC_HIP(hipFuncSetAttribute((const void*)als_short_mfma_kernel<NT, BIG>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)M::BYTES));
        int dev = 0;
        MIREC_HIP(hipGetDevice(&dev));
        MIREC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        *attr_done = true;
    }
    int64_t blocks = (n_short + M::NS - 1) / M::NS;
    if (blocks > cus) blocks = cus;
    if (blocks < 1) return MIREC_OK;
    hipLaunchKernelGGL((als_short_mfma_kernel<NT, BIG>), dim3((unsigned)blocks), dim3(64 * M::NW), M::BYTES, stream,
                       indptr, indices, data, row_begin, X, Y, G, cg_steps, order, n_short);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}

template <int NT, bool BIG>
static int launch_short(const int64_t* indptr, const int32_t* indices, const float* data, int64_t row_begin,
                        int64_t nrows, float* X, const float* Y, const float* G, int32_t short_max, int cg_steps,
                        hipStream_t stream) {
    using L = ShortLds<NT>;
    constexpr int SW = L::SW;
    static OncePerDevice once;  // idempotent; a race only repeats the call
    static int cus_of[kMaxDevices] = {};
    bool* attr_done = once.slot();
    MIREC_REQUIRE(attr_done != nullptr, MIREC_E_HIP, "cannot read the current device ordinal (or it is >= %d)", kMaxDevices);
    int& cus = cus_of[attr_done - once.done];
    if (!*attr_done) {
        MIREC_HIP(hipFuncSetAttribute((const void*)als_short_kernel<NT, BIG>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES));
        int dev = 0;
        MIREC_HIP(hipGetDevice(&dev));
        MIREC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        *attr_done = true;
    }
    int64_t blocks = (nrows + SW - 1) / SW;
    if (blocks > cus) blocks = cus;
    if (blocks < 1) return MIREC_OK;
    hipLaunchKernelGGL((als_short_kernel<NT, BIG>), dim3((unsigned)blocks), dim3(64 * SW), L::BYTES, stream, indptr,
                       indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}

template <int NT>
static int launch_short_nt(int64_t y_rows, const int64_t* indptr, const int32_t* indices, const float* data,
                           int64_t row_begin, int64_t nrows, float* X, const float* Y, const float* G,
                           int32_t short_max, int cg_steps, const int32_t* order, int64_t n_short,
                           hipStream_t stream) {
    // byte offsets into the gathered table are kept in 32 bits while the whole table is addressable that way
    const bool big = y_rows <= 0 || (uint64_t)y_rows * (32u * NT * 4u) > (1ull << 32);
    if constexpr (NT == 2 || NT == 4) {
        if (order != nullptr)
            return big ? launch_short_mfma<NT, true>(indptr, indices, data, row_begin, X, Y, G, cg_steps, order, n_short, stream)
                       : launch_short_mfma<NT, false>(indptr, indices, data, row_begin, X, Y, G, cg_steps, order, n_short, stream);
    }
    return big ? launch_short<NT, true>(indptr, indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps, stream)
               : launch_short<NT, false>(indptr, indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps, stream);
}

int launch_als_short(int nt, int64_t y_rows, const int64_t* indptr, const int32_t* indices, const float* data,
                     int64_t row_begin, int64_t nrows, float* X, const float* Y, const float* G, int32_t short_max,
                     int cg_steps, const int32_t* order, int64_t n_short, hipStream_t stream) {
    switch (nt) {
        case 1: return launch_short_nt<1>(y_rows, indptr, indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps, order, n_short, stream);
        case 2: return launch_short_nt<2>(y_rows, indptr, indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps, order, n_short, stream);
        case 3: return launch_short_nt<3>(y_rows, indptr, indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps, order, n_short, stream);
        case 4: return launch_short_nt<4>(y_rows, indptr, indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps, order, n_short, stream);
        case 5: return launch_short_nt<5>(y_rows, indptr, indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps, order, n_short, stream);
        case 6: return launch_short_nt<6>(y_rows, indptr, indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps, order, n_short, stream);
    }
    set_error("launch_als_short: unsupported width");
    return MIREC_E_UNSUPPORTED;
}

int als_short_cache_rows(int nt) {
    switch (nt) {
        case 1: return ShortLds<1>::CAP;
        case 2: return MfmaShort<2>::CAP + MfmaShort<2>::RS * STEP;
        case 4: return MfmaShort<4>::CAP + MfmaShort<4>::RS * STEP;
        case 3: return ShortLds<3>::CAP;
        case 5: return ShortLds<5>::CAP;
        case 6: return ShortLds<6>::CAP;
    }
    return 0;
}

}  // namespace mirec

