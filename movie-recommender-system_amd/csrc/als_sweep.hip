// ALS half-sweep, explicit form: per-row confidence-weighted Gram on the matrix
// cores, then a warm-started conjugate-gradient solve (3 steps by default).
//
// Work unit = one "chunk": a run of stored entries of one CSR row (rows longer
// than chunk_nnz are cut into up to 32 equal chunks).  A workgroup of NT
// wavefronts (F = 32*NT factor columns) gathers the chunk's factor rows 32 at a
// time through a double-buffered LDS tile -- the next tile's gathers and the
// indices of the one after are in flight while the current one is consumed --
// and accumulates
//     A += sum_k (|c_k|-1) y_k y_k^T      (only the 16x16 tiles on or above the
//                                          diagonal)
//     b += sum_k max(c_k,0) y_k           (fp32 vector ALU)
// The Gram runs on v_mfma_f32_16x16x32_f16 at fp32 accuracy: each scaled row
// z_k = 2^s sqrt(||c_k|-1|) y_k is split into two fp16 planes z = hi + mid
// (11 + 11 significant bits; the remainder z - hi is exact in fp32) and a tile
// accumulates the three plane products mid*hi, hi*mid, hi*hi; the dropped
// mid*mid is below 2^-22 of the term, the rounding of mid likewise -- measured,
// an entry of A lands as close to its float64 value as with an fp32 fma chain
// or closer -- at 3/16 of the matrix-core time of the fp32 instruction.  The
// power of two 2^s (one per launch, from the largest diagonal entry of G and
// the largest confidence) puts the largest |z| just below 2^15, so that no hi
// overflows fp16 and every mid that matters is a normal number; the sums are
// scaled back by 2^-2s, exactly, when they leave the accumulators.  An entry
// with |c_k| < 1 has a negative weight: its sign is applied to the left operand.
// Rows that fit one chunk are finished in the same workgroup: A + G is written
// to LDS (aliasing the gather tiles) and wave 0 runs the CG steps on it.  Rows
// split over several chunks store their partial (A, b) to a workspace slot; a
// second kernel sums the slots in chunk order (fixed, so results are
// reproducible) and runs the same CG.
#include <type_traits>

#include "als_internal.h"

namespace mirec {

constexpr int TR = 32;               // gathered rows per LDS tile = one k-step of the bf16 MFMA
constexpr int MAX_ROW_CHUNKS = 32;   // a row is split into at most this many chunks

// ---------------------------------------------------------------- plan -----
// Device layout of a plan for `nrows` rows and at most `cap` chunks:
//   int32 chunk_ptr[nrows+1]   exclusive scan of chunks per row
//   int32 multi_ptr[nrows+1]   exclusive scan of (chunks per row > 1)
//   int32 mslot_ptr[nrows+1]   exclusive scan of (chunks if > 1 else 0)
//   int32 multi_row[nrows+1]   row of every split row (first n_multi used)
//   int32 scan scratch
//   int32 counter              rows left to the matrix-free kernel
//   int32 short_order[nrows]   those rows, longest first (the first n_short entries count)
//   sort scratch               used while the plan is built
//   int32 chunk_row[cap]       row of every chunk
constexpr int SCAN_TILE = 2048;  // 256 threads x 8 items

struct PlanLayout {
    size_t chunk_ptr, multi_ptr, mslot_ptr, chunk_row, multi_row, scratch, counter, short_order, sort_tmp, total;
    int64_t cap;
    int scan_blocks;
};

static inline int32_t auto_chunk(int64_t nnz) {
    int64_t c = (nnz / 16384 + 31) / 32 * 32;
    if (c < 4096) c = 4096;
    if (c > (1 << 20)) c = 1 << 20;
    return (int32_t)c;
}

static inline PlanLayout plan_layout(int64_t nrows, int64_t nnz, int32_t chunk_nnz) {
    PlanLayout L;
    if (chunk_nnz <= 0) chunk_nnz = auto_chunk(nnz);
    L.cap = nrows + nnz / chunk_nnz + 1;
    const size_t arr = align_up((size_t)(nrows + 1) * 4, 256);
    L.scan_blocks = (int)((nrows + 1 + SCAN_TILE - 1) / SCAN_TILE);
    size_t off = 0;
    // chunk_row comes last so that no other offset depends on the chunk capacity
    L.chunk_ptr = off; off += arr;
    L.multi_ptr = off; off += arr;
    L.mslot_ptr = off; off += arr;
    L.multi_row = off; off += arr;
    L.scratch = off;   off += align_up((size_t)L.scan_blocks * 4, 256);
    L.counter = off;   off += 256;
    L.short_order = off; off += arr;
    L.sort_tmp = off;  off += align_up(short_order_tmp_bytes(nrows), 256);
    L.chunk_row = off; off += align_up((size_t)L.cap * 4, 256);
    L.total = off;
    return L;
}

// Rows with at most short_max entries (empty rows included) belong to the matrix-free kernel: no chunks.
__global__ void plan_count_kernel(const int64_t* __restrict__ indptr, int64_t row_begin, int64_t nrows,
                                  int32_t chunk_nnz, int32_t short_max, int32_t* chunk_ptr, int32_t* multi_ptr,
                                  int32_t* mslot_ptr, int32_t* n_short) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nrows) return;
    int32_t nch = 0;
    if (r < nrows) {
        const int64_t nnz = indptr[row_begin + r + 1] - indptr[row_begin + r];
        nch = nnz <= (int64_t)short_max ? 0 : (int32_t)((nnz + chunk_nnz - 1) / chunk_nnz);
        if (nch > MAX_ROW_CHUNKS) nch = MAX_ROW_CHUNKS;
    }
    if (r < nrows && nch == 0) atomicAdd(n_short, 1);
    chunk_ptr[r] = nch;
    multi_ptr[r] = nch > 1 ? 1 : 0;
    mslot_ptr[r] = nch > 1 ? nch : 0;
}

// In-place exclusive scan of one 2048-element tile per block; block totals out.
__global__ __launch_bounds__(256) void scan_tile_kernel(int32_t* data, int64_t n, int32_t* block_sums) {
    __shared__ int32_t wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)tid * 8;
    int32_t v[8];
    int32_t run = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int32_t t = (base + i < n) ? data[base + i] : 0;
        v[i] = run;
        run += t;
    }
    int32_t incl = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t y = __shfl_up(incl, off, 64);
        if (lane >= off) incl += y;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int32_t woff = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < wave) woff += wsum[i];
        total += wsum[i];
    }
    const int32_t toff = woff + incl - run;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (base + i < n) data[base + i] = v[i] + toff;
    if (tid == 0) block_sums[blockIdx.x] = total;
}

__global__ void scan_sums_kernel(int32_t* block_sums, int nblk) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int32_t run = 0;
        for (int b = 0; b < nblk; ++b) {
            const int32_t t = block_sums[b];
            block_sums[b] = run;
            run += t;
        }
    }
}

__global__ void scan_add_kernel(int32_t* data, int64_t n, const int32_t* __restrict__ block_sums) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] += block_sums[i / SCAN_TILE];
}

__global__ void plan_fill_kernel(int64_t nrows, const int32_t* __restrict__ chunk_ptr,
                                 const int32_t* __restrict__ multi_ptr, int32_t* chunk_row, int32_t* multi_row) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const int32_t c0 = chunk_ptr[r], c1 = chunk_ptr[r + 1];
    for (int32_t c = c0; c < c1; ++c) chunk_row[c] = (int32_t)r;
    if (c1 - c0 > 1) multi_row[multi_ptr[r]] = (int32_t)r;
}

static int exclusive_scan(int32_t* data, int64_t n, int32_t* scratch, hipStream_t stream) {
    const int nblk = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL(scan_tile_kernel, dim3(nblk), dim3(256), 0, stream, data, n, scratch);
    MIREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(64), 0, stream, scratch, nblk);
    MIREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, data, n, scratch);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}

// Conjugate-gradient steps of implicit's solver on the explicit system (A, b), run by all NW waves of the
// workgroup (every thread must call it).  Wave 0 owns the CG state (lane holds columns lane + 64*m); each
// product A v is split over the waves by k range, the partial vectors meet in LDS (`part`: NW*F + 4 floats)
// and wave 0 sums them in wave order, so the result does not depend on timing.  x is read from and written
// back to global memory.
template <int F, int NW>
__device__ void cg_solve_block(const float* A, const float* b, float* vec, float* part, float* xg, int cg_steps,
                               int tid) {
    constexpr int NC = (F + 63) / 64;
    constexpr int KW = F / NW;  // k range of one wave
    static_assert(F % NW == 0 && KW % 4 == 0, "k ranges must be whole float4 groups");
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    int* ctrl = reinterpret_cast<int*>(part + NW * F);
    int colc[NC];
#pragma unroll
    for (int m = 0; m < NC; ++m) colc[m] = (lane + 64 * m < F) ? lane + 64 * m : F - 1;

    // out = A v (valid in wave 0); returns false in every wave when wave 0 passed go == false
    auto symv_all = [&](const float (&v)[NC], float (&out)[NC], bool go) __attribute__((always_inline)) {
        if (w == 0) {
#pragma unroll
            for (int m = 0; m < NC; ++m)
                if (lane + 64 * m < F) vec[lane + 64 * m] = v[m];
            if (lane == 0) *ctrl = go ? 1 : 0;
        }
        __syncthreads();
        if (*ctrl == 0) return false;
        float o[NC];
#pragma unroll
        for (int m = 0; m < NC; ++m) o[m] = 0.f;
#pragma unroll
        for (int k = w * KW; k < w * KW + KW; k += 4) {
            const float4 pk = *reinterpret_cast<const float4*>(vec + k);
#pragma unroll
            for (int m = 0; m < NC; ++m) {
                o[m] = fmaf(A[(k + 0) * F + colc[m]], pk.x, o[m]);
                o[m] = fmaf(A[(k + 1) * F + colc[m]], pk.y, o[m]);
                o[m] = fmaf(A[(k + 2) * F + colc[m]], pk.z, o[m]);
                o[m] = fmaf(A[(k + 3) * F + colc[m]], pk.w, o[m]);
            }
        }
#pragma unroll
        for (int m = 0; m < NC; ++m)
            if (lane + 64 * m < F) part[w * F + lane + 64 * m] = o[m];
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int m = 0; m < NC; ++m) {
                float s = part[colc[m]];
#pragma unroll
                for (int ww = 1; ww < NW; ++ww) s += part[ww * F + colc[m]];
                out[m] = s;
            }
        }
        return true;
    };

    float x[NC], r[NC], p[NC], Ap[NC];
#pragma unroll
    for (int m = 0; m < NC; ++m) {
        const int col = lane + 64 * m;
        x[m] = (w == 0 && col < F) ? xg[col] : 0.f;
        Ap[m] = 0.f;
    }
    symv_all(x, Ap, true);
    float partial = 0.f;
#pragma unroll
    for (int m = 0; m < NC; ++m) {
        const int col = lane + 64 * m;
        r[m] = col < F ? b[col] - Ap[m] : 0.f;
        p[m] = r[m];
        partial = fmaf(r[m], r[m], partial);
    }
    float rsold = wave_sum(partial);
    bool alive = !(rsold < 1e-20f);
    for (int it = 0; it < cg_steps; ++it) {
        if (!symv_all(p, Ap, alive)) break;
        partial = 0.f;
#pragma unroll
        for (int m = 0; m < NC; ++m) {
            if (lane + 64 * m >= F) Ap[m] = 0.f;
            partial = fmaf(p[m], Ap[m], partial);
        }
        const float alpha = rsold / wave_sum(partial);
        partial = 0.f;
#pragma unroll
        for (int m = 0; m < NC; ++m) {
            x[m] = fmaf(alpha, p[m], x[m]);
            r[m] = fmaf(-alpha, Ap[m], r[m]);
            partial = fmaf(r[m], r[m], partial);
        }
        const float rsnew = wave_sum(partial);
        alive = !(rsnew < 1e-20f);
        const float beta = rsnew / rsold;
#pragma unroll
        for (int m = 0; m < NC; ++m) p[m] = fmaf(beta, p[m], r[m]);
        rsold = rsnew;
    }
    if (w == 0) {
#pragma unroll
        for (int m = 0; m < NC; ++m) {
            const int col = lane + 64 * m;
            if (col < F) xg[col] = x[m];
        }
    }
}

// Exact solve of the explicit system (implicit's use_cg=False path, cpu/_als.pyx `_least_squares`: LAPACK posv),
// run by all NTHR threads of the workgroup (every thread must call it): A x = b by a Cholesky factorisation
// A = L L^T, x written to global memory.  The warm start plays no part.
//   A     LDS, row stride F + 1 (so that a column is as free of bank conflicts as a row), lower triangle read
//   pan   LDS scratch of 2 * 4 * (F + 4) floats, 16-byte aligned (it may overlay b: b is in registers by then)
// The lower triangle is cut into 4 x 4 blocks that live in registers, one or a few per thread; b rides along as one
// more row of the matrix (row F: after the factorisation it holds z = L^-1 b, so the forward substitution costs
// nothing).  Right-looking by block columns, one workgroup barrier per block column: the owners of block column c
// publish their blocks as they stand after the updates of the columns before (a panel of F + 4 rows x 4); after
// the barrier every thread factors the panel's 4 x 4 diagonal block itself (a few dozen operations, cheaper than
// a second barrier), turns the panel rows of its block row and block column into rows of L by a 4 x 4 triangular
// solve, and takes their product off its blocks.  Two panel buffers alternate, so a thread that is one block
// column ahead never overwrites what a slower one still reads.  Blocks are numbered block column by block column,
// so the threads of a wavefront own neighbouring block columns and the whole wavefront falls idle once the
// factorisation has passed them.  The back substitution L^T x = z runs on wavefront 0 from L in LDS, one column
// per step.  A pivot that is not positive (A not positive definite, possible only with confidences below 1)
// yields NaN in that row; LAPACK's LU fallback is not reproduced.
__device__ __forceinline__ float chol_rsqrt(float d) {
    // 1 / sqrt(d): the hardware estimate and one Newton step (zero or negative d: not a number)
    float inv = __builtin_amdgcn_rsqf(d);
    return inv * fmaf(-0.5f * d * inv, inv, 1.5f);
}

// Row r of a panel block times the inverse transpose of the factored diagonal block: x L_D^T = r.
__device__ __forceinline__ void chol_trsm_row(const float4 r, const float (&l)[6], const float (&iv)[4], float (&x)[4]) {
    x[0] = r.x * iv[0];
    x[1] = fmaf(-x[0], l[0], r.y) * iv[1];
    x[2] = fmaf(-x[1], l[2], fmaf(-x[0], l[1], r.z)) * iv[2];
    x[3] = fmaf(-x[2], l[5], fmaf(-x[1], l[4], fmaf(-x[0], l[3], r.w))) * iv[3];
}

template <int F, int NTHR>
__device__ void chol_solve_block(float* A, const float* b, float* pan, float* xg, int tid) {
    constexpr int LDA = F + 1;
    constexpr int NBLK = F / 4;
    constexpr int NB = NBLK * (NBLK + 1) / 2 + NBLK;  // lower-triangle blocks + those of the row that carries b
    // Threads that own blocks: all of them where that gives every thread two blocks or more (F >= 128), else the
    // first OWN, a whole number of wavefronts with two blocks each (F = 64: 128 of 256 threads, F = 96: 192 of 384).
    // With ONE block per thread and the blocks spread over three or more wavefronts (F = 64, 96) about 3 in 10^4
    // solves came out different from run to run whenever other workgroups shared the compute unit (none with a
    // workgroup per unit, none when the routine runs alone, A and b verified equal at entry, extra barriers, scalar
    // panel stores and panels in LDS of their own changed nothing: tools/probes/exact_repro.py, chol_standalone.hip);
    // the cause was not established.  With two blocks per owner: 0 differences in 6 x 10^5 solves per width.
    constexpr int OWN = ((NB + NTHR - 1) / NTHR >= 2) ? NTHR : (NB / 2 + 63) / 64 * 64;
    static_assert(OWN <= NTHR && OWN % 64 == 0, "owners are whole wavefronts of the workgroup");
    constexpr int NBK = (NB + OWN - 1) / OWN;         // blocks per owning thread
    constexpr int PB = 4 * (F + 4);                   // floats of one panel buffer
    float a[NBK][4][4];
    int bi[NBK], bj[NBK];  // block row / block column; bi == NBLK: the b row; bj == -1: no block
#pragma unroll
    for (int s = 0; s < NBK; ++s) {
        const int id = tid + s * OWN;
        bi[s] = 0;
        bj[s] = -1;
        if (id < NB && tid < OWN) {
            // first id of block column c: c (NBLK + 1) - c (c - 1) / 2  (rows c .. NBLK)
            const float h = (float)NBLK + 1.5f;
            int c = (int)(h - sqrtf(h * h - 2.f * (float)id));
            c = c < 0 ? 0 : (c > NBLK - 1 ? NBLK - 1 : c);
            while (c > 0 && c * (NBLK + 1) - c * (c - 1) / 2 > id) --c;
            while (c < NBLK - 1 && (c + 1) * (NBLK + 1) - (c + 1) * c / 2 <= id) ++c;
            bj[s] = c;
            bi[s] = c + (id - (c * (NBLK + 1) - c * (c - 1) / 2));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = 0.f;
                if (bj[s] >= 0) {
                    if (bi[s] < NBLK) v = A[(4 * bi[s] + r) * LDA + 4 * bj[s] + c];
                    else if (r == 0) v = b[4 * bj[s] + c];
                }
                a[s][r][c] = v;
            }
    }
    __syncthreads();  // b is in registers: the panel buffers may overlay it
    for (int cj = 0; cj < NBLK; ++cj) {
        float* pb = pan + (cj & 1) * PB;
#pragma unroll
        for (int s = 0; s < NBK; ++s)
            if (bj[s] == cj) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<float4*>(pb + 16 * bi[s] + 4 * r) = make_float4(a[s][r][0], a[s][r][1], a[s][r][2], a[s][r][3]);
            }
        __syncthreads();
        bool any = false;
#pragma unroll
        for (int s = 0; s < NBK; ++s) any = any || bj[s] >= cj;
        if (!any) continue;
        // the diagonal block D = L_D L_D^T: l = (l10, l20, l21, l30, l31, l32), iv = 1 / diagonal of L_D
        float l[6], iv[4];
        {
            const float4 d0 = *reinterpret_cast<const float4*>(pb + 16 * cj);
            const float4 d1 = *reinterpret_cast<const float4*>(pb + 16 * cj + 4);
            const float4 d2 = *reinterpret_cast<const float4*>(pb + 16 * cj + 8);
            const float4 d3 = *reinterpret_cast<const float4*>(pb + 16 * cj + 12);
            iv[0] = chol_rsqrt(d0.x);
            l[0] = d1.x * iv[0];
            l[1] = d2.x * iv[0];
            l[3] = d3.x * iv[0];
            iv[1] = chol_rsqrt(fmaf(-l[0], l[0], d1.y));
            l[2] = fmaf(-l[1], l[0], d2.y) * iv[1];
            l[4] = fmaf(-l[3], l[0], d3.y) * iv[1];
            iv[2] = chol_rsqrt(fmaf(-l[2], l[2], fmaf(-l[1], l[1], d2.z)));
            l[5] = fmaf(-l[4], l[2], fmaf(-l[3], l[1], d3.z)) * iv[2];
            iv[3] = chol_rsqrt(fmaf(-l[5], l[5], fmaf(-l[4], l[4], fmaf(-l[3], l[3], d3.w))));
        }
#pragma unroll
        for (int s = 0; s < NBK; ++s) {
            if (bj[s] < cj) continue;
            if (bj[s] == cj && bi[s] == cj) {
                // the diagonal block itself becomes L_D (what lies above its diagonal is never read)
                a[s][0][0] = 1.0f / iv[0];
                a[s][1][0] = l[0]; a[s][1][1] = 1.0f / iv[1];
                a[s][2][0] = l[1]; a[s][2][1] = l[2]; a[s][2][2] = 1.0f / iv[2];
                a[s][3][0] = l[3]; a[s][3][1] = l[4]; a[s][3][2] = l[5]; a[s][3][3] = 1.0f / iv[3];
                continue;
            }
            float pi[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                chol_trsm_row(*reinterpret_cast<const float4*>(pb + 16 * bi[s] + 4 * r), l, iv, pi[r]);
            if (bj[s] == cj) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) a[s][r][c] = pi[r][c];
                continue;
            }
            // one row of L of the block column's rows at a time (four registers live instead of sixteen)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float pk[4];
                chol_trsm_row(*reinterpret_cast<const float4*>(pb + 16 * bj[s] + 4 * c), l, iv, pk);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int m = 0; m < 4; ++m) a[s][r][c] = fmaf(-pi[r][m], pk[m], a[s][r][c]);
            }
        }
    }
    // L (the rows below the diagonal are what the back substitution reads), z and 1 / diagonal to LDS
    __syncthreads();  // the last panel is read by all before z and the inverses overwrite the buffers
    float* zv = pan;
    float* dinv = pan + PB;
#pragma unroll
    for (int s = 0; s < NBK; ++s) {
        if (bj[s] < 0) continue;
        if (bi[s] < NBLK) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) A[(4 * bi[s] + r) * LDA + 4 * bj[s] + c] = a[s][r][c];
            if (bi[s] == bj[s]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dinv[4 * bi[s] + r] = 1.0f / a[s][r][r];
            }
        } else {
            *reinterpret_cast<float4*>(zv + 4 * bj[s]) = make_float4(a[s][0][0], a[s][0][1], a[s][0][2], a[s][0][3]);
        }
    }
    __syncthreads();
    if (tid < 64) {
        constexpr int NC = (F + 63) / 64;
        const int lane = tid;
        float z[NC];
#pragma unroll
        for (int m = 0; m < NC; ++m) z[m] = (lane + 64 * m < F) ? zv[lane + 64 * m] : 0.f;
        for (int i = F - 1; i >= 0; --i) {
            float zi = z[0];
#pragma unroll
            for (int m = 1; m < NC; ++m) zi = (i >> 6) == m ? z[m] : zi;
            const float xi = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(zi), i & 63)) * dinv[i];
#pragma unroll
            for (int m = 0; m < NC; ++m) {
                const int jc = lane + 64 * m;
                const float lv = A[i * LDA + (jc < F ? jc : F - 1)];
                z[m] = jc < i ? fmaf(-lv, xi, z[m]) : (jc == i ? xi : z[m]);
            }
        }
#pragma unroll
        for (int m = 0; m < NC; ++m)
            if (lane + 64 * m < F) xg[lane + 64 * m] = z[m];
    }
}


// ------------------------------------------------------------- kernels -----
// Geometry of the explicit-Gram kernel at width F = 32*NT:
//   A_u is cut into 16x16 tiles; only the NB(NB+1)/2 tiles on or above the
//   diagonal are accumulated (NB = F/16 bands).  Wave w of the NT waves owns
//   band-rows w and NB-1-w: (NB-w) + (w+1) = NB+1 tiles each, so the matrix
//   pipes of the four SIMDs are evenly loaded.
//   A tile of 32 gathered rows sits in LDS as two fp16 planes [column][32
//   entries]: 64-byte rows of four 16-byte chunks; column c has row
//   c ^ ((c >> 2) & 1) and its chunk q sits at position q ^ swz(c).  Lane (m, q)
//   of an MFMA operand reads the eight entries 8q..8q+7 of column 16*band + m
//   with one ds_read_b128, a producer thread stores the eight entries of one
//   column with one ds_write_b128; with this placement neither has a bank
//   conflict (the 16 lanes of a read group cover 16 different 16-byte bank
//   groups, the 8 lanes of a store group 8), and the same image serves as left
//   and right operand.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;  // one MFMA operand (8 bf16) = one LDS chunk

template <int NT>
struct GramCfg {
    static constexpr int F = 32 * NT;
    static constexpr int NB = 2 * NT;
    static constexpr int NTHR = 64 * NT;
    static constexpr int NTILES = NB * (NB + 1) / 2;
    static constexpr int NPLANES = 2;
    static constexpr int PLANE_BYTES = F * 64;
    static constexpr int TILE_BYTES = NPLANES * PLANE_BYTES;
    static constexpr int TILE_FLOATS = TILE_BYTES / 4;
    static constexpr int LDX = F + 1;  // row stride of A for the exact solver (columns free of bank conflicts)
    static constexpr int BIGN = (F * LDX > 2 * TILE_FLOATS) ? F * LDX : 2 * TILE_FLOATS;
    static constexpr int PART_STRIDE = NTILES * 256 + F;  // floats of one partial-result slot
    // LDS map (floats): [0, BIGN) gather tiles / A; sign masks of the two tiles; b partials of the four entry
    // groups; b; the CG work vector; the CG partial products
    // row of column c in a plane, and the swizzle of its chunks (both depend on c & 15 only)
    __host__ __device__ static constexpr int col_row(int c) { return c ^ ((c >> 2) & 1); }
    __host__ __device__ static constexpr int col_swz(int c) { return ((c >> 1) & 1) | (((c >> 3) & 1) << 1); }
    static constexpr int SMASK = BIGN;
    static constexpr int BRED = SMASK + 2 * 16;
    static constexpr int BVEC = BRED + 4 * F;
    static constexpr int VEC = BVEC + F;
    static constexpr int CGP = VEC + F;  // partial products of the block CG: one vector per wave + a control word
    static constexpr int FLOATS = CGP + 2 * NT * F + 4;
    static constexpr size_t BYTES = (size_t)FLOATS * 4;
    // The exact solver's two panel buffers (2 * 4 * (F + 4) floats) overlay everything from the b partials on
    // -- all of it dead once b is in registers -- or, at F = 32, the part of the tile area that A leaves free.
    static constexpr int CHOL_PANEL = (FLOATS - BRED >= 8 * (F + 4)) ? BRED : F * LDX;
    static_assert(CHOL_PANEL % 4 == 0 && CHOL_PANEL + 8 * (F + 4) <= (CHOL_PANEL == F * LDX ? BIGN : FLOATS),
                  "no room for the panels of the exact solver");
    // first tile id of band-row b in the row-major enumeration of the upper triangle
    static constexpr int row_start(int b) { return b * NB - b * (b - 1) / 2; }
};

// Accumulators of wave W: band-row W (tiles (W, W..NB-1)) and band-row NB-1-W (tiles (R1, R1..NB-1)).
template <int NT, int W>
struct WaveAcc {
    static constexpr int NB = 2 * NT;
    static constexpr int R1 = NB - 1 - W;
    static constexpr int N0 = NB - W;
    static constexpr int N1 = W + 1;
    f32x4 a0[N0];
    f32x4 a1[N1];
};

__device__ __forceinline__ uint32_t cvt_pk_f16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}

// Two entries (a, b) x two columns (x, y) -> for each column two words of packed fp16 pairs (entry a in the
// low half) with hi + mid == value to 22 bits: round to nearest at both levels, the remainder exact in fp32.
// The remainders are formed per entry on column pairs (packed fp32 ops on adjacent registers).
__device__ __forceinline__ void split2x2(f32x2 a, f32x2 b, uint32_t (&hi)[2], uint32_t (&mid)[2]) {
    hi[0] = cvt_pk_f16(a[0], b[0]);
    hi[1] = cvt_pk_f16(a[1], b[1]);
    const f16x2 h0 = __builtin_bit_cast(f16x2, hi[0]), h1 = __builtin_bit_cast(f16x2, hi[1]);
    a -= f32x2{(float)h0[0], (float)h1[0]};
    b -= f32x2{(float)h0[1], (float)h1[1]};
    mid[0] = cvt_pk_f16(a[0], b[0]);
    mid[1] = cvt_pk_f16(a[1], b[1]);
}

__device__ __forceinline__ f32x4 mma_f16(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                  0);
}

// acc += l^T r over the 32 entries of the tile, planes {hi, mid}; smallest products first.
__device__ __forceinline__ f32x4 mma3(f32x4 acc, const u32x4 (&l)[2], const u32x4 (&r)[2]) {
    acc = mma_f16(l[1], r[0], acc);
    acc = mma_f16(l[0], r[1], acc);
    acc = mma_f16(l[0], r[0], acc);
    return acc;
}

// Power of two that puts the largest scaled entry z = sqrt(||c|-1|) y just below 2^15: |y| <= ymax (the largest
// |entry| of the gathered table where the caller has it, else sqrt of the largest diagonal entry of
// G = Y^T Y + reg I, which bounds it but overshoots by up to the square root of the table's row count),
// ||c|-1| <= max(cmax - 1, 1).  Clamped to 2^+-40 so that its inverse square stays a normal fp32 number.
// Returns (scale, 1 / scale^2).  An entry keeps its 22 bits while |z| >= 2^-3, i.e. while
// |y| sqrt(||c|-1|) is within 2^-18 of ymax sqrt(cmax - 1); below that the fp16 planes run out of exponent and
// it keeps 2^-25 of the scale absolutely.
__device__ __forceinline__ f32x2 gram_scale(float ymax, float cmax) {
    const float wmax = fmaxf(cmax - 1.f, 1.f);
    const float bound = ymax * sqrtf(wmax);
    int e = 0;
    (void)frexpf(bound, &e);  // bound = m 2^e, 0.5 <= m < 1 (e = 0 for bound = 0)
    int sh = 15 - e;
    if (!(bound > 0.f) || sh > 40) sh = 40;
    if (sh < -40) sh = -40;
    return f32x2{ldexpf(1.f, sh), ldexpf(1.f, -2 * sh)};
}

// One LDS tile of 32 gathered rows: the wave's NB+1 tiles, three MFMAs each.  The left operands are the right
// ones with the sign of every negative-weight entry flipped (a wave-uniform branch skips that for tiles
// without such entries).
template <int NT, int W>
__device__ __forceinline__ void gram_tile(const char* tl, const uint32_t* sm, WaveAcc<NT, W>& A, int lane) {
    using C = GramCfg<NT>;
    using WA = WaveAcc<NT, W>;
    constexpr int N0 = WA::N0, R1 = WA::R1;
    const int m = lane & 15, q = lane >> 4;
    const char* base = tl + C::col_row(m) * 64 + ((q ^ C::col_swz(m)) << 4);
    auto ld = [&](int band, int plane) __attribute__((always_inline)) {
        return *reinterpret_cast<const u32x4*>(base + plane * C::PLANE_BYTES + band * 1024);
    };
    u32x4 r0[2], r1[2], l0[2], l1[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        r0[p] = ld(W, p);
        r1[p] = ld(R1, p);
    }
    // sign words of this lane's eight entries (bit 15 / 31 set where the entry's weight is negative)
    const u32x4 sg = *reinterpret_cast<const u32x4*>(sm + 4 * q);
    const bool flip = __builtin_amdgcn_ballot_w64((sg[0] | sg[1] | sg[2] | sg[3]) != 0u) != 0ull;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        l0[p] = r0[p];
        l1[p] = r1[p];
    }
    if (flip) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            l0[p] ^= sg;
            l1[p] ^= sg;
        }
    }
#pragma unroll
    for (int j = 0; j < N0; ++j) {
        const int band = W + j;
        u32x4 rb[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) rb[p] = (j == 0) ? r0[p] : (band == R1 ? r1[p] : ld(band, p));
        A.a0[j] = mma3(A.a0[j], l0, rb);
        if (band >= R1) A.a1[band - R1] = mma3(A.a1[band - R1], l1, rb);
    }
}

// The workgroup of a chunk has 2*NT waves in two roles that meet at one barrier per tile:
//   producers (NT waves)  gather tile t+1, split it into the bf16 planes and store it into LDS buffer (t+1)&1,
//   consumers (NT waves)  run the MFMAs of tile t from buffer t&1,
// so that conversion and matrix-core work of consecutive tiles overlap on every SIMD.
//
// Producer thread p: columns 2*c2, 2*c2+1 of the eight entries 8*eg .. 8*eg+7 of a tile (a column's eight
// entries make one 16-byte LDS chunk).  While tile t+1 is converted, the gathers of tiles t+2 and t+3 and the
// indices of tile t+4 are in flight (two register sets each).  Indices are requested before the gathers of
// the tile ahead of them, so that waiting for them (the memory counter retires in order) leaves those gathers
// outstanding.  The steady-state loop is free of branches, which lets every wait name the exact number of
// younger loads it may leave in flight.  Entries past the end of the chunk repeat its last index and get
// confidence -1: weight 0 and no contribution to b.
// BIG: the gathered table spans 4 GiB or more, so rows are addressed by 64-bit arithmetic instead of 32-bit
// byte offsets.
template <int NT, bool BIG>
__device__ __forceinline__ void gram_producer(const int32_t* __restrict__ indices, const float* __restrict__ data,
                                              const float* __restrict__ Y, int64_t ks, int64_t ke, float* smem,
                                              int ptid, float scale2) {
    using C = GramCfg<NT>;
    constexpr int F = C::F;
    char* tiles = reinterpret_cast<char*>(smem);
    uint32_t* smask = reinterpret_cast<uint32_t*>(smem + C::SMASK);
    float* bred = smem + C::BRED;

    const int ntiles = (int)((ke - ks + TR - 1) / TR);
    const int nent = (int)(ke - ks);
    const int c2 = ptid % (F / 2);
    const int eg = ptid / (F / 2);
    int idx2[2][8];
    float dd2[2][8];
    f32x2 yv2[2][8];
    float bacc0 = 0.f, bacc1 = 0.f;
    const int32_t* ip = indices + ks + 8 * eg;
    const float* dp = data + ks + 8 * eg;
    int left_i = nent - 8 * eg;  // entries of the chunk from ip on (may be <= 0 in the last tile)
    int left_d = left_i;
    // `full`: the tile lies wholly inside the chunk (wave-uniform)
    auto load_idx = [&](int (&idx)[8], bool full) __attribute__((always_inline)) {
        if (full) {
#pragma unroll
            for (int e = 0; e < 8; ++e) idx[e] = ip[e];
        } else {
            const int last = left_i - 1;
#pragma unroll
            for (int e = 0; e < 8; ++e) idx[e] = ip[e < last ? e : last];
        }
        ip += TR;
        left_i -= TR;
    };
    const uint32_t lane_off = (uint32_t)c2 * 8u;
    auto load_rows = [&](const int (&idx)[8], f32x2 (&yv)[8], float (&dd)[8], bool full)
                         __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (BIG) {
                yv[e] = *reinterpret_cast<const f32x2*>(Y + (int64_t)idx[e] * F + 2 * c2);
            } else {
                yv[e] = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(Y) +
                                                        ((uint32_t)idx[e] * (uint32_t)(F * 4) + lane_off));
            }
        }
        if (full) {
#pragma unroll
            for (int e = 0; e < 8; ++e) dd[e] = dp[e];
        } else {
            const int last = left_d - 1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = dp[e < last ? e : last];
                dd[e] = e <= last ? v : -1.f;
            }
        }
        dp += TR;
        left_d -= TR;
    };
    // columns 2*c2 + t: rows 2*c2 + (t ^ ((c2 >> 1) & 1)) of a plane, chunk eg at position eg ^ swz
    const int wswap = (c2 >> 1) & 1;
    const int wbase = c2 * 128 + ((eg ^ C::col_swz(2 * c2)) << 4);
    auto store_tile = [&](const f32x2 (&yv)[8], const float (&dd)[8], int buf) __attribute__((always_inline)) {
        char* tile = tiles + buf * C::TILE_BYTES;
        uint32_t* sm = smask + buf * 16;
        float sw[8], w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            w[e] = fabsf(dd[e]) - 1.f;
            sw[e] = __builtin_amdgcn_sqrtf(fabsf(w[e]) * scale2);  // = 2^s sqrt(|w|): scale2 = 4^s
            // max(c, 0) on the bit pattern: positive floats order like integers, everything else is <= 0
            const int di = __float_as_int(dd[e]);
            const float cp = __int_as_float(di > 0 ? di : 0);
            bacc0 = fmaf(cp, yv[e][0], bacc0);
            bacc1 = fmaf(cp, yv[e][1], bacc1);
        }
        // b is final for this tile before the next gathers overwrite yv
        asm volatile("" : "+v"(bacc0), "+v"(bacc1));
        u32x4 hi[2], mid[2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t h[2], m[2];
            split2x2(yv[2 * q] * sw[2 * q], yv[2 * q + 1] * sw[2 * q + 1], h, m);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                hi[t][q] = h[t];
                mid[t][q] = m[t];
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            char* dst = tile + wbase + ((t ^ wswap) << 6);
            *reinterpret_cast<u32x4*>(dst) = hi[t];
            *reinterpret_cast<u32x4*>(dst + C::PLANE_BYTES) = mid[t];
        }
        if (c2 == 0) {
            // sign bits of the weights of entries (2q, 2q+1) at bits 15 and 31 of word q
            u32x4 sgn;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                sgn[q] = __builtin_amdgcn_perm(__float_as_uint(w[2 * q + 1]), __float_as_uint(w[2 * q]), 0x07060302u) &
                         0x80008000u;
            *reinterpret_cast<u32x4*>(sm + 4 * eg) = sgn;
        }
    };
    auto is_full = [&](int tile) __attribute__((always_inline)) { return (tile + 1) * TR <= nent; };

    // Before phase t: tile t is in LDS buffer t & 1; the rows of tiles t+1 and t+2 are requested into register
    // sets (t+1) & 1 and t & 1; the indices of tile t+3 are requested into index set (t+1) & 1.
    auto phase = [&](auto set, auto steady, int tile) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
        constexpr bool STEADY = decltype(steady)::value;
        if (STEADY || tile + 1 < ntiles) {
            store_tile(yv2[S ^ 1], dd2[S ^ 1], S ^ 1);
            if (STEADY || tile + 4 < ntiles) load_idx(idx2[S], STEADY || is_full(tile + 4));
            if (STEADY || tile + 3 < ntiles) load_rows(idx2[S ^ 1], yv2[S ^ 1], dd2[S ^ 1], STEADY || is_full(tile + 3));
        }
        __syncthreads();  // tile + 1 is in LDS; the consumers are done with tile
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;

    load_idx(idx2[0], is_full(0));
    if (ntiles > 1) load_idx(idx2[1], is_full(1));
    load_rows(idx2[0], yv2[0], dd2[0], is_full(0));
    if (ntiles > 2) load_idx(idx2[0], is_full(2));
    if (ntiles > 1) load_rows(idx2[1], yv2[1], dd2[1], is_full(1));
    if (ntiles > 3) load_idx(idx2[1], is_full(3));
    store_tile(yv2[0], dd2[0], 0);
    if (ntiles > 2) load_rows(idx2[0], yv2[0], dd2[0], is_full(2));
    __syncthreads();  // tile 0 is in LDS
    int tile = 0;
    for (; tile + 6 < ntiles; tile += 2) {  // every tile touched below is a full one that exists
        phase(T0{}, std::true_type{}, tile);
        phase(T1{}, std::true_type{}, tile + 1);
    }
    for (; tile < ntiles; ++tile) {
        if (tile & 1) phase(T1{}, std::false_type{}, tile);
        else phase(T0{}, std::false_type{}, tile);
    }
    bred[eg * F + 2 * c2] = bacc0;
    bred[eg * F + 2 * c2 + 1] = bacc1;
}

// Consumer wave W: accumulates its tiles over the chunk, then stores them (executes the same barriers as the
// producers).
template <int NT, int W, int lda>
__device__ __forceinline__ void gram_consumer(const float* __restrict__ G, int ntiles, float* smem, bool single,
                                              float* P, int lane, float unscale) {
    using C = GramCfg<NT>;
    using WA = WaveAcc<NT, W>;
    constexpr int F = C::F;
    float* big = smem;
    const char* tiles = reinterpret_cast<const char*>(smem);
    const uint32_t* smask = reinterpret_cast<const uint32_t*>(smem + C::SMASK);

    WA A;
#pragma unroll
    for (int j = 0; j < WA::N0; ++j) A.a0[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < WA::N1; ++j) A.a1[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        gram_tile<NT, W>(tiles + (tile & 1) * C::TILE_BYTES, smask + (tile & 1) * 16, A, lane);
        __syncthreads();
    }
    // every consumer is past its last tile, so the tile buffers that alias `big` are free

    const int q = lane >> 4, i = lane & 15;
    if (single) {
        // A + G into LDS, both triangles
#pragma unroll
        for (int j = 0; j < WA::N0; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int rr = W * 16 + q * 4 + e, cc = (W + j) * 16 + i;
                const float v = A.a0[j][e] * unscale;
                big[rr * lda + cc] = v + G[rr * F + cc];
                if (j > 0) big[cc * lda + rr] = v + G[cc * F + rr];
            }
#pragma unroll
        for (int j = 0; j < WA::N1; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int rr = WA::R1 * 16 + q * 4 + e, cc = (WA::R1 + j) * 16 + i;
                const float v = A.a1[j][e] * unscale;
                big[rr * lda + cc] = v + G[rr * F + cc];
                if (j > 0) big[cc * lda + rr] = v + G[cc * F + rr];
            }
    } else {
        // upper tiles only, tile-major: slot[tile][row][col]
#pragma unroll
        for (int j = 0; j < WA::N0; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) P[(C::row_start(W) + j) * 256 + (q * 4 + e) * 16 + i] = A.a0[j][e] * unscale;
#pragma unroll
        for (int j = 0; j < WA::N1; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) P[(C::row_start(WA::R1) + j) * 256 + (q * 4 + e) * 16 + i] = A.a1[j][e] * unscale;
    }
}

// Chunk c of a row with n entries split into nch chunks starts at c * chunk_len(n, nch).
__device__ __forceinline__ int64_t chunk_len(int64_t n, int nch) {
    const int64_t per = (n + nch - 1) / nch;
    return (per + TR - 1) / TR * TR;
}

// EXACT: the instantiation that ends a row with the Cholesky solve instead of conjugate gradients (cg_steps is
// then ignored); kept apart so that the registers and the LDS stride of the iterative kernel stay what they are.
template <int NT, bool BIG, bool EXACT>
__global__ __launch_bounds__(128 * NT, (NT <= 4 ? 4 : 1)) void als_gram_cg_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ data,
    int64_t row_begin, float* X, const float* __restrict__ Y, const float* __restrict__ G,
    const int32_t* __restrict__ chunk_ptr, const int32_t* __restrict__ mslot_ptr,
    const int32_t* __restrict__ chunk_row, int cg_steps, float* partials, float conf_max,
    const float* __restrict__ conf_max_dev, const float* __restrict__ y_absmax) {
    using C = GramCfg<NT>;
    constexpr int F = C::F;
    constexpr int NTHR = 128 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* big = smem;
    float* bred = smem + C::BRED;
    float* bvec = smem + C::BVEC;
    float* vec = smem + C::VEC;

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    const int chunk = blockIdx.x;
    const int r = chunk_row[chunk];
    const int cbase = chunk_ptr[r];
    const int nch = chunk_ptr[r + 1] - cbase;
    const int lc = chunk - cbase;
    const int64_t row = row_begin + r;
    const int64_t rs = indptr[row], re = indptr[row + 1];
    float* xrow = X + row * F;
    if (rs == re) {
        for (int i = tid; i < F; i += NTHR) xrow[i] = 0.f;
        return;
    }
    const int64_t clen = chunk_len(re - rs, nch);
    const int64_t ks = rs + (int64_t)lc * clen;
    const int64_t ke = (ks + clen < re) ? ks + clen : re;
    const bool single = nch == 1;
    float* P = single ? nullptr : partials + (size_t)(mslot_ptr[r] + lc) * C::PART_STRIDE;
    if (ks >= ke) {
        // rounding the chunk length up can leave a trailing chunk empty: its slot must still read as zero
        for (int i = tid; i < C::PART_STRIDE; i += NTHR) P[i] = 0.f;
        return;
    }
    const int ntiles = (int)((ke - ks + TR - 1) / TR);

    // the power of two that scales the fp16 operands: the same in every workgroup of every launch of a fit
    // (it depends on the diagonal of G and on the largest confidence only)
    f32x2 sc;
    {
        float gm = tid < F ? G[tid * F + tid] : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) gm = fmaxf(gm, __shfl_xor(gm, off, 64));
        if (lane == 0) vec[w] = gm;
        __syncthreads();
        float gmax = vec[0];
#pragma unroll
        for (int ww = 1; ww < 2 * NT; ++ww) gmax = fmaxf(gmax, vec[ww]);
        sc = gram_scale(y_absmax != nullptr ? *y_absmax : sqrtf(gmax), conf_max > 0.f ? conf_max : *conf_max_dev);
    }
    const float scale2 = sc[0] * sc[0], unscale = sc[1];
    constexpr int lda = EXACT ? C::LDX : F;

    switch (w) {
        case 0: gram_consumer<NT, 0, lda>(G, ntiles, smem, single, P, lane, unscale); break;
        case 1: if constexpr (NT > 1) gram_consumer<NT, 1, lda>(G, ntiles, smem, single, P, lane, unscale); break;
        case 2: if constexpr (NT > 2) gram_consumer<NT, 2, lda>(G, ntiles, smem, single, P, lane, unscale); break;
        case 3: if constexpr (NT > 3) gram_consumer<NT, 3, lda>(G, ntiles, smem, single, P, lane, unscale); break;
        case 4: if constexpr (NT > 4) gram_consumer<NT, 4, lda>(G, ntiles, smem, single, P, lane, unscale); break;
        case 5: if constexpr (NT > 5) gram_consumer<NT, 5, lda>(G, ntiles, smem, single, P, lane, unscale); break;
    }
    if (w >= NT) gram_producer<NT, BIG>(indices, data, Y, ks, ke, smem, tid - 64 * NT, scale2);
    __syncthreads();  // A (or the partial slot) and the b partials of the four entry groups are written
    if (tid < F) {
        const float b = (bred[tid] + bred[F + tid]) + (bred[2 * F + tid] + bred[3 * F + tid]);
        if (single) bvec[tid] = b;
        else P[C::NTILES * 256 + tid] = b;
    }
    if (single) {
        __syncthreads();
        if constexpr (EXACT) chol_solve_block<F, NTHR>(big, bvec, smem + C::CHOL_PANEL, xrow, tid);
        else cg_solve_block<F, 2 * NT>(big, bvec, vec, smem + C::CGP, xrow, cg_steps, tid);
    }
}

// Split rows: sum the chunk partials in chunk order, add G, mirror the upper triangle, solve.
template <int NT, bool EXACT>
__global__ __launch_bounds__(64 * NT) void als_finalize_kernel(
    int64_t row_begin, float* X, const float* __restrict__ G, const int32_t* __restrict__ chunk_ptr,
    const int32_t* __restrict__ mslot_ptr, const int32_t* __restrict__ multi_row, int cg_steps,
    const float* __restrict__ partials) {
    using C = GramCfg<NT>;
    constexpr int F = C::F;
    constexpr int NB = C::NB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* big = smem;
    float* bvec = smem + C::BVEC;
    float* vec = smem + C::VEC;

    const int tid = threadIdx.x;
    constexpr int lda = EXACT ? C::LDX : F;
    const int r = multi_row[blockIdx.x];
    const int nch = chunk_ptr[r + 1] - chunk_ptr[r];
    const float* P = partials + (size_t)mslot_ptr[r] * C::PART_STRIDE;
    // every thread sums its elements tid, tid + NTHR, ... over the chunks in chunk order; the loads of one chunk
    // (one per element) are independent, so a whole slot row is in flight per thread
    constexpr int NE = (C::PART_STRIDE + C::NTHR - 1) / C::NTHR;
    float acc[NE];
#pragma unroll
    for (int it = 0; it < NE; ++it) acc[it] = 0.f;
    for (int ch = 0; ch < nch; ++ch) {
        const float* Pc = P + (size_t)ch * C::PART_STRIDE;
#pragma unroll
        for (int it = 0; it < NE; ++it) {
            const int e = tid + it * C::NTHR;
            const float v = Pc[e < C::PART_STRIDE ? e : C::PART_STRIDE - 1];
            acc[it] += e < C::PART_STRIDE ? v : 0.f;
        }
    }
#pragma unroll
    for (int it = 0; it < NE; ++it) {
        const int e = tid + it * C::NTHR;
        if (e >= C::PART_STRIDE) continue;
        const float s = acc[it];
        if (e >= C::NTILES * 256) {
            bvec[e - C::NTILES * 256] = s;
        } else {
            const int t = e >> 8, rr16 = (e >> 4) & 15, cc16 = e & 15;
            int bi = 0, t0 = 0;  // band-row of tile t
            while (t0 + (NB - bi) <= t) {
                t0 += NB - bi;
                ++bi;
            }
            const int bj = bi + (t - t0);
            const int rr = bi * 16 + rr16, cc = bj * 16 + cc16;
            big[rr * lda + cc] = s + G[rr * F + cc];
            if (bj > bi) big[cc * lda + rr] = s + G[cc * F + rr];
        }
    }
    __syncthreads();
    if constexpr (EXACT) chol_solve_block<F, C::NTHR>(big, bvec, smem + C::CHOL_PANEL, X + (row_begin + r) * F, tid);
    else cg_solve_block<F, NT>(big, bvec, vec, smem + C::CGP, X + (row_begin + r) * F, cg_steps, tid);
}

// Largest |confidence| over the stored entries of rows [row_begin, row_end) (non-negative floats order like their
// bit patterns).
__global__ __launch_bounds__(256) void conf_absmax_kernel(const float* __restrict__ data,
                                                          const int64_t* __restrict__ indptr, int64_t row_begin,
                                                          int64_t row_end, uint32_t* out) {
    __shared__ uint32_t wmax[4];
    uint32_t m = 0u;
    const int64_t n = indptr[row_end];
    for (int64_t i = indptr[row_begin] + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t b = __float_as_uint(data[i]) & 0x7fffffffu;
        m = b > m ? b : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) m = wmax[i] > m ? wmax[i] : m;
        atomicMax(out, m);
    }
}

template <int NT, bool EXACT>
static int launch_sweep_kind(const int64_t* indptr, const int32_t* indices, const float* data, int64_t row_begin,
                             float* X, const float* Y, const float* G, int cg_steps, const char* plan,
                             const PlanLayout& L, const int64_t counts[3], float* partials, int64_t y_rows,
                             float conf_max, const float* conf_max_dev, const float* y_absmax, hipStream_t stream) {
    using S = GramCfg<NT>;
    const int32_t* chunk_ptr = (const int32_t*)(plan + L.chunk_ptr);
    const int32_t* mslot_ptr = (const int32_t*)(plan + L.mslot_ptr);
    const int32_t* chunk_row = (const int32_t*)(plan + L.chunk_row);
    const int32_t* multi_row = (const int32_t*)(plan + L.multi_row);
    static OncePerDevice once;  // idempotent; a race only repeats the call
    bool* attr_done = once.slot();
    MIREC_REQUIRE(attr_done != nullptr, MIREC_E_HIP, "cannot read the current device ordinal (or it is >= %d)", kMaxDevices);
    if (!*attr_done) {
        MIREC_HIP(hipFuncSetAttribute((const void*)als_gram_cg_kernel<NT, false, EXACT>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::BYTES));
        MIREC_HIP(hipFuncSetAttribute((const void*)als_gram_cg_kernel<NT, true, EXACT>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::BYTES));
        MIREC_HIP(hipFuncSetAttribute((const void*)als_finalize_kernel<NT, EXACT>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::BYTES));
        *attr_done = true;
    }
    if (counts[0] > 0) {
        // byte offsets into the gathered table are kept in 32 bits while the whole table is addressable that way
        const bool big = y_rows <= 0 || (uint64_t)y_rows * (uint64_t)(S::F * 4) > (1ull << 32);
        if (big)
            hipLaunchKernelGGL((als_gram_cg_kernel<NT, true, EXACT>), dim3((unsigned)counts[0]), dim3(128 * NT), S::BYTES,
                               stream, indptr, indices, data, row_begin, X, Y, G, chunk_ptr, mslot_ptr, chunk_row,
                               cg_steps, partials, conf_max, conf_max_dev, y_absmax);
        else
            hipLaunchKernelGGL((als_gram_cg_kernel<NT, false, EXACT>), dim3((unsigned)counts[0]), dim3(128 * NT), S::BYTES,
                               stream, indptr, indices, data, row_begin, X, Y, G, chunk_ptr, mslot_ptr, chunk_row,
                               cg_steps, partials, conf_max, conf_max_dev, y_absmax);
        MIREC_LAUNCH_CHECK();
    }
    if (counts[1] > 0) {
        hipLaunchKernelGGL((als_finalize_kernel<NT, EXACT>), dim3((unsigned)counts[1]), dim3(64 * NT), S::BYTES, stream,
                           row_begin, X, G, chunk_ptr, mslot_ptr, multi_row, cg_steps, partials);
        MIREC_LAUNCH_CHECK();
    }
    return MIREC_OK;
}

template <int NT>
static int launch_sweep(const int64_t* indptr, const int32_t* indices, const float* data, int64_t row_begin,
                        float* X, const float* Y, const float* G, int cg_steps, const char* plan,
                        const PlanLayout& L, int32_t chunk_nnz, const int64_t counts[3], float* partials,
                        int64_t y_rows, float conf_max, const float* conf_max_dev, const float* y_absmax,
                        hipStream_t stream) {
    (void)chunk_nnz;
    if (cg_steps < 0)
        return launch_sweep_kind<NT, true>(indptr, indices, data, row_begin, X, Y, G, cg_steps, plan, L, counts,
                                           partials, y_rows, conf_max, conf_max_dev, y_absmax, stream);
    return launch_sweep_kind<NT, false>(indptr, indices, data, row_begin, X, Y, G, cg_steps, plan, L, counts, partials,
                                        y_rows, conf_max, conf_max_dev, y_absmax, stream);
}

}  // namespace mirec

using namespace mirec;


extern "C" size_t mirec_als_plan_bytes(int64_t nrows, int64_t nnz, int32_t chunk_nnz) {
    if (nrows < 0 || nnz < 0) return 0;
    return plan_layout(nrows, nnz, chunk_nnz).total;
}

extern "C" int mirec_als_short_cache_rows(int f) {
    if (f % 32 != 0 || f < 32 || f > 192) return -1;
    return als_short_cache_rows(f / 32);
}

extern "C" int mirec_als_plan_build(const int64_t* indptr, int64_t row_begin, int64_t row_end, int64_t nnz, int f,
                                    void* plan_, size_t plan_bytes, int32_t* chunk_nnz, int32_t short_max,
                                    int64_t host_counts[4], mirec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    (void)f;
    MIREC_REQUIRE(indptr && plan_ && chunk_nnz && host_counts, MIREC_E_INVALID, "mirec_als_plan_build: null argument");
    MIREC_REQUIRE(row_end >= row_begin && nnz >= 0, MIREC_E_INVALID, "mirec_als_plan_build: bad row range");
    MIREC_REQUIRE(short_max >= 0, MIREC_E_INVALID, "mirec_als_plan_build: short_max must be >= 0");
    const int64_t nrows = row_end - row_begin;
    MIREC_REQUIRE(nrows < (int64_t)1 << 31, MIREC_E_UNSUPPORTED, "mirec_als_plan_build: more than 2^31 rows in one range");
    int32_t ch = *chunk_nnz;
    if (ch <= 0) ch = auto_chunk(nnz);
    MIREC_REQUIRE(ch > 0, MIREC_E_INVALID, "mirec_als_plan_build: chunk_nnz must be positive");
    const PlanLayout L = plan_layout(nrows, nnz, ch);
    MIREC_REQUIRE(plan_bytes >= L.total, MIREC_E_WORKSPACE, "mirec_als_plan_build: plan needs %zu bytes, %zu given",
                  L.total, plan_bytes);
    char* plan = (char*)plan_;
    int32_t* chunk_ptr = (int32_t*)(plan + L.chunk_ptr);
    int32_t* multi_ptr = (int32_t*)(plan + L.multi_ptr);
    int32_t* mslot_ptr = (int32_t*)(plan + L.mslot_ptr);
    int32_t* scratch = (int32_t*)(plan + L.scratch);
    const int64_t n1 = nrows + 1;
    int32_t* n_short = (int32_t*)(plan + L.counter);
    MIREC_HIP(hipMemsetAsync(n_short, 0, 256, stream));
    hipLaunchKernelGGL(plan_count_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, stream, indptr, row_begin,
                       nrows, ch, short_max, chunk_ptr, multi_ptr, mslot_ptr, n_short);
    MIREC_LAUNCH_CHECK();
    int rc;
    if ((rc = exclusive_scan(chunk_ptr, n1, scratch, stream)) != MIREC_OK) return rc;
    if ((rc = exclusive_scan(multi_ptr, n1, scratch, stream)) != MIREC_OK) return rc;
    if ((rc = exclusive_scan(mslot_ptr, n1, scratch, stream)) != MIREC_OK) return rc;
    // The totals come back before anything is written through them: a too small `nnz` argument must end as an
    // error, not as chunk rows stored past the plan's capacity.
    int32_t totals[4] = {0, 0, 0, 0};
    MIREC_HIP(hipMemcpyAsync(&totals[0], chunk_ptr + nrows, 4, hipMemcpyDeviceToHost, stream));
    MIREC_HIP(hipMemcpyAsync(&totals[1], multi_ptr + nrows, 4, hipMemcpyDeviceToHost, stream));
    MIREC_HIP(hipMemcpyAsync(&totals[2], mslot_ptr + nrows, 4, hipMemcpyDeviceToHost, stream));
    MIREC_HIP(hipMemcpyAsync(&totals[3], n_short, 4, hipMemcpyDeviceToHost, stream));
    MIREC_HIP(hipStreamSynchronize(stream));
    MIREC_REQUIRE(totals[0] <= L.cap, MIREC_E_INVALID,
                  "mirec_als_plan_build: %d chunks exceed the plan capacity %lld (nnz argument too small?)", totals[0],
                  (long long)L.cap);
    if (nrows > 0) {
        hipLaunchKernelGGL(plan_fill_kernel, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, stream, nrows,
                           chunk_ptr, multi_ptr, (int32_t*)(plan + L.chunk_row), (int32_t*)(plan + L.multi_row));
        MIREC_LAUNCH_CHECK();
        if (totals[3] > 0 &&
            (rc = build_short_order(indptr, row_begin, nrows, short_max, (int32_t*)(plan + L.short_order),
                                    plan + L.sort_tmp, stream)) != MIREC_OK)
            return rc;
    }
    host_counts[0] = totals[0];
    host_counts[1] = totals[1];
    host_counts[2] = totals[2];
    host_counts[3] = totals[3];
    *chunk_nnz = ch;
    return MIREC_OK;
}

constexpr size_t SWEEP_WS_HEADER = 256;  // the largest |confidence|, when the sweep has to find it itself

extern "C" size_t mirec_als_sweep_workspace_bytes(const int64_t host_counts[4], int f) {
    if (!host_counts || f <= 0) return 0;
    const size_t slots = host_counts[2] > 0 ? (size_t)host_counts[2] : 0;
    const size_t nb = (size_t)f / 16;  // upper-triangle 16x16 tiles + the b vector (GramCfg::PART_STRIDE)
    return SWEEP_WS_HEADER + slots * (nb * (nb + 1) / 2 * 256 + f) * sizeof(float);
}

extern "C" int mirec_als_half_sweep(const int64_t* indptr, const int32_t* indices, const float* data,
                                    int64_t row_begin, int64_t row_end, int64_t nnz, float* X, const float* Y,
                                    int64_t y_rows, const float* G, const float* y_absmax, int f, int cg_steps,
                                    const void* plan, int32_t chunk_nnz, int32_t short_max,
                                    const int64_t host_counts[4], float conf_max, void* ws, size_t ws_bytes,
                                    mirec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (row_end == row_begin) return MIREC_OK;  // a shard without rows (its X block may be a null pointer)
    MIREC_REQUIRE(indptr && X && Y && G && plan && host_counts, MIREC_E_INVALID, "mirec_als_half_sweep: null argument");
    MIREC_REQUIRE(nnz == 0 || (indices && data), MIREC_E_INVALID, "mirec_als_half_sweep: null indices/data");
    MIREC_REQUIRE(row_end >= row_begin && cg_steps >= MIREC_SOLVE_EXACT, MIREC_E_INVALID,
                  "mirec_als_half_sweep: bad range or cg_steps");
    // the exact solver lives in the explicit-Gram kernels only: the plan must leave nothing but empty rows to the other
    MIREC_REQUIRE(cg_steps >= 0 || short_max <= 0, MIREC_E_INVALID,
                  "mirec_als_half_sweep: cg_steps = MIREC_SOLVE_EXACT needs a plan built with short_max = 0 (got %d)",
                  short_max);
    MIREC_REQUIRE(f % 32 == 0 && f >= 32 && f <= 192, MIREC_E_UNSUPPORTED,
                  "mirec_als_half_sweep: factor width %d is not a multiple of 32 in [32,192]; zero-pad the columns", f);
    MIREC_REQUIRE(chunk_nnz > 0, MIREC_E_INVALID, "mirec_als_half_sweep: bad chunk_nnz %d", chunk_nnz);
    MIREC_REQUIRE(((uintptr_t)Y & 15) == 0 && ((uintptr_t)X & 15) == 0, MIREC_E_INVALID,
                  "mirec_als_half_sweep: factor matrices must be 16-byte aligned");
    MIREC_REQUIRE(host_counts[0] == 0 || (ws && ws_bytes >= mirec_als_sweep_workspace_bytes(host_counts, f)),
                  MIREC_E_WORKSPACE, "mirec_als_half_sweep: workspace of %zu bytes needed, %zu given",
                  mirec_als_sweep_workspace_bytes(host_counts, f), ws_bytes);
    MIREC_REQUIRE(conf_max >= 0.f || conf_max != conf_max, MIREC_E_INVALID,
                  "mirec_als_half_sweep: conf_max must be the largest |confidence| (or 0: not known)");
    const int64_t nrows = row_end - row_begin;
    if (nrows == 0) return MIREC_OK;
    const PlanLayout L = plan_layout(nrows, nnz, chunk_nnz);
    const char* p = (const char*)plan;
    float* partials = ws ? (float*)((char*)ws + SWEEP_WS_HEADER) : nullptr;
    const float* conf_max_dev = (const float*)ws;
    if (host_counts[0] > 0 && !(conf_max > 0.f)) {
        // not given: one pass over the stored values of the row range (their largest magnitude bounds the
        // fp16 operands of the explicit-Gram kernel)
        conf_max = 0.f;
        MIREC_HIP(hipMemsetAsync(ws, 0, SWEEP_WS_HEADER, stream));
        if (nnz > 0) {
            hipLaunchKernelGGL(conf_absmax_kernel, dim3(1024), dim3(256), 0, stream, data, indptr, row_begin, row_end,
                               (uint32_t*)ws);
            MIREC_LAUNCH_CHECK();
        }
    }
    if (short_max >= 0 && host_counts[3] > 0) {
        const int rc = launch_als_short(f / 32, y_rows, indptr, indices, data, row_begin, nrows, X, Y, G, short_max, cg_steps,
                                        (const int32_t*)(p + L.short_order), host_counts[3], stream);
        if (rc != MIREC_OK) return rc;
    }
    if (host_counts[0] == 0 && host_counts[1] == 0) return MIREC_OK;
    switch (f / 32) {
        case 1: return launch_sweep<1>(indptr, indices, data, row_begin, X, Y, G, cg_steps, p, L, chunk_nnz, host_counts, partials, y_rows, conf_max, conf_max_dev, y_absmax, stream);
        case 2: return launch_sweep<2>(indptr, indices, data, row_begin, X, Y, G, cg_steps, p, L, chunk_nnz, host_counts, partials, y_rows, conf_max, conf_max_dev, y_absmax, stream);
        case 3: return launch_sweep<3>(indptr, indices, data, row_begin, X, Y, G, cg_steps, p, L, chunk_nnz, host_counts, partials, y_rows, conf_max, conf_max_dev, y_absmax, stream);
        case 4: return launch_sweep<4>(indptr, indices, data, row_begin, X, Y, G, cg_steps, p, L, chunk_nnz, host_counts, partials, y_rows, conf_max, conf_max_dev, y_absmax, stream);
        case 5: return launch_sweep<5>(indptr, indices, data, row_begin, X, Y, G, cg_steps, p, L, chunk_nnz, host_counts, partials, y_rows, conf_max, conf_max_dev, y_absmax, stream);
        case 6: return launch_sweep<6>(indptr, indices, data, row_begin, X, Y, G, cg_steps, p, L, chunk_nnz, host_counts, partials, y_rows, conf_max, conf_max_dev, y_absmax, stream);
    }
    return MIREC_E_UNSUPPORTED;
}
