// Shared Gram matrix  G = Y^T Y + reg*I  on the fp32 matrix cores.
//
// Row blocks of Y are contracted with v_mfma_f32_32x32x2_f32 (an exact fp32
// fma chain over the rows).  Only the 32x32 tiles on or above the diagonal are
// formed: a wave owns band-rows w and NT-1-w (NT+1 tiles), two groups of such
// waves split the rows of a workgroup, and the loads of the next eight rows are
// in flight while the current ones are multiplied.  Block partials are then
// summed in a fixed order (fp64 accumulation, one rounding) and mirrored, so
// the result is symmetric to the bit and does not depend on scheduling.
#include <mutex>

#include "common.h"

namespace mirec {

constexpr int GRAM_GROUPS = 2;  // wave groups per workgroup, each with its own partial slot

// Rows handled by one workgroup: at most ~512 workgroups, at least 512 rows.
static inline int gram_rows_per_block(int64_t n) {
    int64_t r = (n + 511) / 512;
    if (r < 512) r = 512;
    r = (r + 15) / 16 * 16;
    return (int)r;
}

template <int NT>
struct GramWaves {
    static constexpr int ROLES = (NT + 1) / 2;  // waves per group
    static constexpr int NW = ROLES * GRAM_GROUPS;
};

// Wave role W of a group: band-rows W and R1 = NT-1-W (the same one when NT is odd and W is the middle).
// (role 0 loads every column band of its rows, so it also keeps the largest |entry| it sees: `amax`)
template <int NT, int W>
__device__ __forceinline__ void gram_rows_wave(const float* __restrict__ Y, int64_t r0, int64_t r1, float* out,
                                               int lane, float* amax) {
    constexpr int F = 32 * NT;
    constexpr int R1 = NT - 1 - W;
    constexpr int N0 = NT - W;                  // tiles (W, W..NT-1)
    constexpr int N1 = (R1 > W) ? (W + 1) : 0;  // tiles (R1, R1..NT-1)
    const int h = lane >> 5;  // which of the two rows of a k-step this lane feeds
    const int c = lane & 31;

    f32x16 acc0[N0];
    f32x16 acc1[N1 > 0 ? N1 : 1];
#pragma unroll
    for (int t = 0; t < N0; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc0[t][i] = 0.f;
#pragma unroll
    for (int t = 0; t < N1; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[t][i] = 0.f;

    // b[p][t]: column 32*(W+t) + c of row r + 2p + h; rows past the end read row r0 and are zeroed
    float cur[4][N0], nxt[4][N0];
    auto load = [&](float (&b)[4][N0], int64_t r) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int64_t row = r + 2 * p + h;
            const bool valid = row < r1;
            const float* yr = Y + (valid ? row : r0) * F + W * 32 + c;
#pragma unroll
            for (int t = 0; t < N0; ++t) {
                const float v = yr[t * 32];
                b[p][t] = valid ? v : 0.f;
            }
        }
    };
    if (r0 < r1) load(cur, r0);
    float big = 0.f;
    for (int64_t r = r0; r < r1; r += 8) {
        if (r + 8 < r1) load(nxt, r + 8);
        if constexpr (W == 0) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int t = 0; t < N0; ++t) big = fmaxf(big, fabsf(cur[p][t]));  // (fmaxf drops a NaN operand)
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int t = 0; t < N0; ++t)
                acc0[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[p][0], cur[p][t], acc0[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < N1; ++t)
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[p][R1 - W], cur[p][R1 - W + t], acc1[t], 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int t = 0; t < N0; ++t) cur[p][t] = nxt[p][t];
    }

    if constexpr (W == 0) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) big = fmaxf(big, __shfl_xor(big, off, 64));
        if (lane == 0) *amax = big;
    }
#pragma unroll
    for (int t = 0; t < N0; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            out[(size_t)(W * 32 + row) * F + (W + t) * 32 + c] = acc0[t][i];
        }
#pragma unroll
    for (int t = 0; t < N1; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            out[(size_t)(R1 * 32 + row) * F + (R1 + t) * 32 + c] = acc1[t][i];
        }
}

// Partial slot (blockIdx, group): F x F floats of which only the tiles on or above the diagonal are written.
template <int NT>
__global__ __launch_bounds__(64 * GramWaves<NT>::NW) void gram_partial_kernel(const float* __restrict__ Y, int64_t n,
                                                                              int rows_per_block,
                                                                              float* __restrict__ partials,
                                                                              float* __restrict__ slot_max) {
    constexpr int F = 32 * NT;
    constexpr int ROLES = GramWaves<NT>::ROLES;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int group = w / ROLES, role = w % ROLES;
    const int half = rows_per_block / GRAM_GROUPS;  // a multiple of 8

    int64_t r0 = (int64_t)blockIdx.x * rows_per_block + (int64_t)group * half;
    int64_t r1 = r0 + half;
    if (r0 > n) r0 = n;
    if (r1 > n) r1 = n;
    float* out = partials + ((size_t)blockIdx.x * GRAM_GROUPS + group) * F * F;
    float* amax = slot_max + (size_t)blockIdx.x * GRAM_GROUPS + group;
    switch (role) {
        case 0: gram_rows_wave<NT, 0>(Y, r0, r1, out, lane, amax); break;
        case 1: if constexpr (ROLES > 1) gram_rows_wave<NT, 1>(Y, r0, r1, out, lane, amax); break;
        case 2: if constexpr (ROLES > 2) gram_rows_wave<NT, 2>(Y, r0, r1, out, lane, amax); break;
    }
}

// One workgroup per 64 elements of the upper block triangle; four threads share the slots of an element
// (slot s goes to thread s & 3) and their fp64 sums are combined in a fixed order.
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* __restrict__ partials, int nslots, int f,
                                                          float reg, float* __restrict__ G,
                                                          const float* __restrict__ slot_max,
                                                          float* __restrict__ y_absmax) {
    __shared__ double part[4][64];
    if (y_absmax != nullptr && blockIdx.x == 0 && threadIdx.x < 64) {
        // the largest |entry| of the matrix: the maximum of the slots' (wavefront 0 of the first workgroup)
        float m = 0.f;
        for (int b = threadIdx.x; b < nslots; b += 64) m = fmaxf(m, slot_max[b]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if (threadIdx.x == 0) *y_absmax = m;
    }
    const int ff = f * f;
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int q = threadIdx.x >> 6;
    const int i = e / f, j = e % f;
    const bool upper = e < ff && (i >> 5) <= (j >> 5);
    double s = 0.0;
    if (upper)
        for (int b = q; b < nslots; b += 4) s += (double)partials[(size_t)b * ff + e];
    part[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && upper) {
        const int t = threadIdx.x;
        double tot = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
        if (i == j) tot += (double)reg;
        const float g = (float)tot;
        G[e] = g;
        if ((i >> 5) < (j >> 5)) G[j * f + i] = g;
    }
}

template <int NT>
static int launch_gram(const float* Y, int64_t n, float reg, float* G, float* y_absmax, float* partials,
                       hipStream_t stream) {
    constexpr int F = 32 * NT;
    const int rpb = gram_rows_per_block(n);
    const int nblk = (int)((n + rpb - 1) / rpb);
    // the slots' largest magnitudes follow the slots in the workspace
    float* slot_max = partials + (size_t)(nblk > 0 ? nblk : 1) * GRAM_GROUPS * F * F;
    if (nblk > 0) {
        hipLaunchKernelGGL(gram_partial_kernel<NT>, dim3(nblk), dim3(64 * GramWaves<NT>::NW), 0, stream, Y, n, rpb,
                           partials, slot_max);
        MIREC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((F * F + 63) / 64), dim3(256), 0, stream, partials,
                       nblk * GRAM_GROUPS, F, reg, G, slot_max, y_absmax);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}

// Block partials of mirec_gram_static: device memory that belongs to the library, one copy per GPU, held from
// the moment the code object is loaded on a device until the process ends.  Host code never passes this name
// to a kernel: it is a host-side stand-in; the array's address on a device comes from hipGetSymbolAddress with
// that device current (static_scratch below).
constexpr size_t GRAM_STATIC_BYTES = MIREC_GRAM_STATIC_BYTES;
static __device__ __align__(256) float scratch[GRAM_STATIC_BYTES / sizeof(float)];

static std::mutex scratch_mutex;
static float* scratch_of[kMaxDevices] = {};

// Device address of `scratch` on the current device (looked up once per device), or nullptr with the error set.
static float* static_scratch() {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) {
        set_error("mirec_gram_static: cannot read the current device ordinal (or it is >= %d)", kMaxDevices);
        return nullptr;
    }
    std::lock_guard<std::mutex> guard(scratch_mutex);
    if (scratch_of[d] == nullptr) {
        float* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || ptr == nullptr) {
            set_error("mirec_gram_static: hipGetSymbolAddress failed on device %d: %s", d, hipGetErrorString(e));
            return nullptr;
        }
        scratch_of[d] = ptr;
    }
    return scratch_of[d];
}

static int gram_dispatch(const float* Y, int64_t n, int f, float reg, float* G, float* y_absmax, float* partials,
                         hipStream_t stream) {
    switch (f / 32) {
        case 1: return launch_gram<1>(Y, n, reg, G, y_absmax, partials, stream);
        case 2: return launch_gram<2>(Y, n, reg, G, y_absmax, partials, stream);
        case 3: return launch_gram<3>(Y, n, reg, G, y_absmax, partials, stream);
        case 4: return launch_gram<4>(Y, n, reg, G, y_absmax, partials, stream);
        case 5: return launch_gram<5>(Y, n, reg, G, y_absmax, partials, stream);
        case 6: return launch_gram<6>(Y, n, reg, G, y_absmax, partials, stream);
    }
    return MIREC_E_UNSUPPORTED;
}

}  // namespace mirec

extern "C" size_t mirec_gram_workspace_bytes(int64_t n, int f) {
    if (n < 0 || f <= 0) return 0;
    const int rpb = mirec::gram_rows_per_block(n);
    const int64_t nblk = (n + rpb - 1) / rpb;
    // per slot: F x F partial sums and one float, the slot's largest |entry|
    return (size_t)(nblk > 0 ? nblk : 1) * mirec::GRAM_GROUPS * ((size_t)f * f + 1) * sizeof(float);
}

extern "C" int mirec_gram(const float* Y, int64_t n, int f, float reg, float* G, float* y_absmax, void* ws,
                          size_t ws_bytes, mirec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MIREC_REQUIRE(G != nullptr && n >= 0, MIREC_E_INVALID, "mirec_gram: null output or negative n");
    MIREC_REQUIRE(n == 0 || Y != nullptr, MIREC_E_INVALID, "mirec_gram: null input");
    MIREC_REQUIRE(f % 32 == 0 && f >= 32 && f <= 192, MIREC_E_UNSUPPORTED,
                  "mirec_gram: factor width %d is not a multiple of 32 in [32,192]; zero-pad the columns", f);
    MIREC_REQUIRE(ws != nullptr && ws_bytes >= mirec_gram_workspace_bytes(n, f), MIREC_E_WORKSPACE,
                  "mirec_gram: workspace of %zu bytes needed, %zu given", mirec_gram_workspace_bytes(n, f), ws_bytes);
    return mirec::gram_dispatch(Y, n, f, reg, G, y_absmax, (float*)ws, stream);
}

extern "C" size_t mirec_gram_static_max_bytes(void) { return mirec::GRAM_STATIC_BYTES; }

extern "C" int mirec_gram_static(const float* Y, int64_t n, int f, float reg, float* G, float* y_absmax,
                                 mirec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MIREC_REQUIRE(G != nullptr && n >= 0, MIREC_E_INVALID, "mirec_gram_static: null output or negative n");
    MIREC_REQUIRE(n == 0 || Y != nullptr, MIREC_E_INVALID, "mirec_gram_static: null input");
    MIREC_REQUIRE(f % 32 == 0 && f >= 32 && f <= 192, MIREC_E_UNSUPPORTED,
                  "mirec_gram_static: factor width %d is not a multiple of 32 in [32,192]; zero-pad the columns", f);
    const size_t need = mirec_gram_workspace_bytes(n, f);
    MIREC_REQUIRE(need <= mirec::GRAM_STATIC_BYTES, MIREC_E_WORKSPACE,
                  "mirec_gram_static: %lld rows of width %d need %zu bytes of block partials, the library's static "
                  "buffer holds %zu; use mirec_gram with a caller workspace", (long long)n, f, need,
                  mirec::GRAM_STATIC_BYTES);
    float* partials = mirec::static_scratch();
    if (partials == nullptr) return MIREC_E_HIP;
    // Nothing has to start at zero: the partial kernel writes every element of the slots (and every slot maximum)
    // the reduction reads.
    return mirec::gram_dispatch(Y, n, f, reg, G, y_absmax, partials, stream);
}
