// LightFM WARP training: one epoch of `fit_partial` (lightfm 1.17 `fit_warp` with the adagrad schedule, as
// reached from hybrid_docker/utils.py:100-107).
//
// The library runs its examples Hogwild: OpenMP threads update shared parameters without locks.  Here a group of
// G lanes (G = 16 for up to 15 components, the reference's setting) owns one example at a time: lane c holds
// component c of the three representations and updates column c of every feature row the example touches, lane
// nc does the same for the biases.  Groups run concurrently on parameters that other groups are changing -- the
// library's contract, at a few thousand examples in flight instead of four -- and add their steps with
// floating-point atomics.  With one group (`serial`) every score is the sequential chain bias_u + bias_i, then
// fma over the components in ascending order, the updates are the library's read-modify-write expressions, and an
// epoch reproduces the sequential statement bit for bit.  Negatives come from a counter-based generator keyed by (seed, example).
// Parameters are float32, the update expressions double, as in the library's Cython.
#include <math.h>

#include "common.h"

namespace mirec {

__device__ __forceinline__ uint32_t warp_hash(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

struct WarpCsr {
    const int64_t* indptr;
    const int32_t* indices;
    const float* data;
};

struct WarpParams {
    float* emb;
    float* emb_grad;
    float* bias;
    float* bias_grad;
};

// lane c < nc: component c; lane nc: the bias; other lanes: 0
__device__ __forceinline__ float warp_repr(const WarpCsr& m, int64_t row, const float* emb, const float* bias,
                                           double scale, int nc, int c) {
    float acc = 0.f;
    const int64_t e = m.indptr[row + 1];
    for (int64_t k = m.indptr[row]; k < e; ++k) {
        const int32_t f = m.indices[k];
        const float w = (float)(m.data[k] * scale);
        if (c < nc)
            acc = fmaf(w, emb[(int64_t)f * nc + c], acc);
        else if (c == nc)
            acc = fmaf(w, bias[f], acc);
    }
    return acc;
}

// SERIAL: the library's chain (bias_u + bias_i, then one fma per component in ascending order), evaluated alike by
// every lane of the group.  Concurrent: the same terms summed as a tree over the lanes -- a rounding-level
// difference that a run with parameters in flux cannot see.
template <int G, bool SERIAL>
__device__ __forceinline__ float warp_score(float u, float it, int nc, int c) {
    if (SERIAL) {
        float s = __shfl(u, nc, G) + __shfl(it, nc, G);
        for (int j = 0; j < nc; ++j) s = fmaf(__shfl(u, j, G), __shfl(it, j, G), s);
        return s;
    }
    float t = c < nc ? u * it : (c == nc ? u + it : 0.f);
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) t += __shfl_xor(t, off, G);
    return t;
}

// one adagrad step on column `c` (or the bias) of every feature of `row`; returns the sum of the learning rates.
// SERIAL: the library's read-modify-write expressions, exactly.  Concurrent: the same step and the same
// regularisation factor folded into one floating-point atomic add per parameter and one per accumulator, so that
// the steps of examples in flight on a shared feature (a popular item, a gender token) add up instead of
// overwriting each other -- with thousands of examples in flight, lost updates would otherwise stall learning.
template <bool SERIAL>
__device__ __forceinline__ double warp_update_row(const WarpCsr& m, int64_t row, float* param, float* grad, int stride,
                                                  int col, double gradient, double lr, double alpha) {
    double sum_lr = 0.0;
    const int64_t e = m.indptr[row + 1];
    for (int64_t k = m.indptr[row]; k < e; ++k) {
        const int64_t at = (int64_t)m.indices[k] * stride + col;
        const double fw = m.data[k];
        const double local = lr / sqrt((double)grad[at]);
        if (SERIAL) {
            param[at] = (float)(param[at] - local * fw * gradient);
            grad[at] = (float)(grad[at] + (gradient * fw) * (gradient * fw));
            param[at] = (float)(param[at] * (1.0 + alpha * local));
        } else {
            const double p = param[at];
            const double stepped = (p - local * fw * gradient) * (1.0 + alpha * local);
            // (plain stores here run three times as fast and do not reach the sequential statement's AUC: measured)
            unsafeAtomicAdd(&param[at], (float)(stepped - p));
            unsafeAtomicAdd(&grad[at], (float)((gradient * fw) * (gradient * fw)));
        }
        sum_lr += local;
    }
    return sum_lr;
}

template <int G>
__device__ __forceinline__ double shfl_double(double v, int src) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__shfl((int)(unsigned)b, src, G);
    const unsigned hi = (unsigned)__shfl((int)(unsigned)(b >> 32), src, G);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int G, bool SERIAL>
__global__ __launch_bounds__(256) void lightfm_warp_epoch_kernel(
    WarpCsr itf, int64_t n_items, WarpCsr uf, const int64_t* __restrict__ pos_indptr,
    const int32_t* __restrict__ pos_indices, const int32_t* __restrict__ user_ids, const int32_t* __restrict__ item_ids,
    const float* __restrict__ y, const float* __restrict__ sample_weight, const int32_t* __restrict__ order,
    int64_t n_examples, WarpParams ip, WarpParams up, int nc, double learning_rate, double item_alpha, double user_alpha,
    int max_sampled, uint32_t seed, const double* __restrict__ log_table, double* scales) {
    const int c = threadIdx.x % G;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t e = group; e < n_examples; e += ngroups) {
        const int32_t row = order[e];
        if (!(y[row] > 0.f)) continue;
        const int32_t u = user_ids[row], pi = item_ids[row];
        const double weight = sample_weight ? (double)sample_weight[row] : 1.0;
        double item_scale = scales[0], user_scale = scales[1];
        const float ur = warp_repr(uf, u, up.emb, up.bias, user_scale, nc, c);
        const float pr = warp_repr(itf, pi, ip.emb, ip.bias, item_scale, nc, c);
        const float ppred = warp_score<G, SERIAL>(ur, pr, nc, c);
        uint32_t rng = warp_hash(seed ^ warp_hash((uint32_t)row + 0x9e3779b9u));
        int sampled = 0;
        while (sampled < max_sampled) {
            ++sampled;
            rng = warp_hash(rng + 0x6d2b79f5u);
            const int32_t neg = (int32_t)(rng % (uint32_t)n_items);
            const float nr = warp_repr(itf, neg, ip.emb, ip.bias, item_scale, nc, c);
            const float npred = warp_score<G, SERIAL>(ur, nr, nc, c);
            if (!(npred > ppred - 1.f)) continue;
            // a sampled negative that is a positive of this user is drawn again
            int64_t a = pos_indptr[u], b = pos_indptr[u + 1];
            const int64_t end = b;
            while (a < b) {
                const int64_t m = a + (b - a) / 2;
                if (pos_indices[m] < neg)
                    a = m + 1;
                else
                    b = m;
            }
            if (a < end && pos_indices[a] == neg) continue;
            double loss = weight * log_table[sampled];
            if (loss > 10.0) loss = 10.0;
            double sp = 0.0, sn = 0.0, su = 0.0;
            if (c == nc) {
                sp = warp_update_row<SERIAL>(itf, pi, ip.bias, ip.bias_grad, 1, 0, -loss, learning_rate, item_alpha);
                sn = warp_update_row<SERIAL>(itf, neg, ip.bias, ip.bias_grad, 1, 0, loss, learning_rate, item_alpha);
            } else if (c < nc) {
                const double uc = ur, pc = pr, ncp = nr;
                sp = warp_update_row<SERIAL>(itf, pi, ip.emb, ip.emb_grad, nc, c, -loss * uc, learning_rate, item_alpha);
                sn = warp_update_row<SERIAL>(itf, neg, ip.emb, ip.emb_grad, nc, c, loss * uc, learning_rate, item_alpha);
                su = warp_update_row<SERIAL>(uf, u, up.emb, up.emb_grad, nc, c, loss * (ncp - pc), learning_rate, user_alpha);
            }
            if (item_alpha != 0.0 || user_alpha != 0.0) {
                // the mean learning rate, summed in the order of the sequential statement
                double avg = shfl_double<G>(sp, nc);
                avg += shfl_double<G>(sn, nc);
                for (int j = 0; j < nc; ++j) {
                    avg += shfl_double<G>(sp, j);
                    avg += shfl_double<G>(sn, j);
                    avg += shfl_double<G>(su, j);
                }
                const double denom = (double)(nc + 1) * (double)((uf.indptr[u + 1] - uf.indptr[u]) +
                                                                 (itf.indptr[pi + 1] - itf.indptr[pi]) +
                                                                 (itf.indptr[neg + 1] - itf.indptr[neg]));
                avg /= denom;
                // unsynchronised read-modify-write, like the library's
                if (c == 0) {
                    scales[0] = scales[0] * (1.0 + item_alpha * avg);
                    scales[1] = scales[1] * (1.0 + user_alpha * avg);
                }
            }
            break;
        }
    }
}

// `regularize`: fold the lazy-regularisation scale into the parameters
__global__ __launch_bounds__(256) void lightfm_fold_scale_kernel(float* emb, float* bias, int64_t n_features, int nc,
                                                                 const double* scale) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double s = *scale;
    if (gid < n_features * nc) emb[gid] = (float)(emb[gid] / s);
    if (gid < n_features) bias[gid] = (float)(bias[gid] / s);
}

__global__ void lightfm_reset_scales_kernel(double* scales) {
    scales[0] = 1.0;
    scales[1] = 1.0;
}

template <int G>
static void launch_warp_epoch(bool serial, int blocks, int threads, hipStream_t stream, WarpCsr itf, int64_t n_items,
                              WarpCsr uf, const int64_t* pos_indptr, const int32_t* pos_indices, const int32_t* user_ids,
                              const int32_t* item_ids, const float* y, const float* sample_weight, const int32_t* order,
                              int64_t n_examples, WarpParams ip, WarpParams up, int nc, double learning_rate,
                              double item_alpha, double user_alpha, int max_sampled, uint32_t seed,
                              const double* log_table, double* scales) {
    if (serial)
        hipLaunchKernelGGL((lightfm_warp_epoch_kernel<G, true>), dim3(blocks), dim3(threads), 0, stream, itf, n_items, uf,
                           pos_indptr, pos_indices, user_ids, item_ids, y, sample_weight, order, n_examples, ip, up, nc,
                           learning_rate, item_alpha, user_alpha, max_sampled, seed, log_table, scales);
    else
        hipLaunchKernelGGL((lightfm_warp_epoch_kernel<G, false>), dim3(blocks), dim3(threads), 0, stream, itf, n_items, uf,
                           pos_indptr, pos_indices, user_ids, item_ids, y, sample_weight, order, n_examples, ip, up, nc,
                           learning_rate, item_alpha, user_alpha, max_sampled, seed, log_table, scales);
}

}  // namespace mirec

extern "C" size_t mirec_lightfm_warp_workspace_bytes(int max_sampled) {
    return max_sampled < 1 ? 0 : mirec::align_up(sizeof(double) * (size_t)(max_sampled + 1) + 2 * sizeof(double), 256);
}

extern "C" int mirec_lightfm_warp_epoch(
    const int64_t* if_indptr, const int32_t* if_indices, const float* if_data, int64_t n_items, int64_t n_item_features,
    const int64_t* uf_indptr, const int32_t* uf_indices, const float* uf_data, int64_t n_user_features,
    const int64_t* pos_indptr, const int32_t* pos_indices, const int32_t* user_ids, const int32_t* item_ids,
    const float* y, const float* sample_weight, const int32_t* order, int64_t n_examples, float* item_emb,
    float* item_emb_grad, float* item_bias, float* item_bias_grad, float* user_emb, float* user_emb_grad,
    float* user_bias, float* user_bias_grad, int nc, double learning_rate, double item_alpha, double user_alpha,
    int max_sampled, uint32_t seed, int serial, int64_t examples_seen, void* ws, size_t ws_bytes,
    mirec_stream_t stream_) {
    using namespace mirec;
    hipStream_t stream = (hipStream_t)stream_;
    MIREC_REQUIRE(nc >= 1 && nc <= 63, MIREC_E_UNSUPPORTED, "mirec_lightfm_warp_epoch: no_components must be 1..63 (got %d)", nc);
    MIREC_REQUIRE(n_items >= 1 && n_items < ((int64_t)1 << 31) && n_examples >= 0 && max_sampled >= 1, MIREC_E_INVALID,
                  "mirec_lightfm_warp_epoch: bad sizes");
    MIREC_REQUIRE(examples_seen >= 0, MIREC_E_INVALID, "mirec_lightfm_warp_epoch: examples_seen must be >= 0");
    if (n_examples == 0) return MIREC_OK;  // an epoch over no interactions changes nothing
    MIREC_REQUIRE(if_indptr && if_indices && if_data && uf_indptr && uf_indices && uf_data && pos_indptr && user_ids &&
                      item_ids && y && order && item_emb && item_emb_grad && item_bias && item_bias_grad && user_emb &&
                      user_emb_grad && user_bias && user_bias_grad && ws,
                  MIREC_E_INVALID, "mirec_lightfm_warp_epoch: null argument");
    MIREC_REQUIRE(ws_bytes >= mirec_lightfm_warp_workspace_bytes(max_sampled) && ((uintptr_t)ws & 7) == 0, MIREC_E_INVALID,
                  "mirec_lightfm_warp_epoch: workspace too small or misaligned");
    // workspace: [item_scale, user_scale], then log(max(1, floor((n_items - 1) / sampled))) for sampled = 0..max
    double* scales = static_cast<double*>(ws);
    double* log_table = scales + 2;
    {
        double host[66];
        MIREC_REQUIRE(max_sampled <= 63, MIREC_E_UNSUPPORTED, "mirec_lightfm_warp_epoch: max_sampled above 63");
        host[0] = host[1] = 1.0;
        host[2] = 0.0;
        for (int s = 1; s <= max_sampled; ++s) host[2 + s] = log(fmax(1.0, floor((double)(n_items - 1) / s)));
        MIREC_HIP(hipMemcpyAsync(scales, host, sizeof(double) * (size_t)(max_sampled + 3), hipMemcpyHostToDevice, stream));
        // the copy above reads a stack buffer: it must have left before this function returns
        MIREC_HIP(hipStreamSynchronize(stream));
    }
    if (n_examples > 0) {
        WarpCsr itf{if_indptr, if_indices, if_data}, uf{uf_indptr, uf_indices, uf_data};
        WarpParams ip{item_emb, item_emb_grad, item_bias, item_bias_grad}, up{user_emb, user_emb_grad, user_bias, user_bias_grad};
        const int G = nc < 16 ? 16 : (nc < 32 ? 32 : 64);
        // Concurrency.  Examples that run side by side read parameters and adagrad accumulators that do not yet
        // hold each other's updates.  That is harmless once a row's accumulator is large against what the
        // concurrent examples add to it, i.e. when the examples in flight are few against the examples already
        // worked through -- and ruinous at the start, where every accumulator is one and thousands of full-size
        // steps would land on the rows that all examples share.  So the epoch runs in phases: one workgroup for
        // the first examples of a fresh model, then as many groups as a KAPPA-th of the examples the model has
        // worked through (in earlier calls, as the caller reports them, and in this one), the phase length
        // doubling with it (each phase costs about KAPPA examples' time, whatever its size).
        // Measured on planted-cluster problems (tools/probes/warp_quality.py; held-out AUC after 1 / 3 epochs):
        // 20 k users x 2 k items, 0.4 M examples -- sequential 0.692 / 0.906; all examples in flight at n/64
        // groups (the first version of this function) 0.522 / 0.664; KAPPA 64 0.677 / 0.907 in 0.09 s per epoch,
        // 256 0.679 / 0.907 in 0.20 s, 1024 0.668 / 0.904 in 0.55 s.  100 k users x 27 k items, 4 M examples --
        // sequential 0.539 after one epoch (131 s); n/64 groups 0.500 / 0.501 (nothing learnt); KAPPA 64
        // 0.529 / 0.762 in 0.48 s, 256 0.532 / 0.774 in 0.58 s, 1024 0.535 / 0.792 in 1.0 s.
        constexpr int64_t kappa = 256;
        constexpr int64_t cap = 256 * 8;  // at most eight 256-thread workgroups per CU
        const int64_t per_block = serial ? 1 : 256 / G;
        int64_t done = 0;
        while (done < n_examples) {
            int blocks = 1, threads = serial ? G : 256;
            int64_t len = n_examples;
            if (!serial) {
                const int64_t seen = examples_seen + done;
                int64_t nb = seen / kappa / per_block;
                nb = nb < 1 ? 1 : (nb < cap ? nb : cap);
                blocks = (int)nb;
                len = nb >= cap ? n_examples - done : (seen < per_block * kappa ? per_block * kappa : seen);
                if (len > n_examples - done) len = n_examples - done;
            }
            const int32_t* ord = order + done;
#define MIREC_WARP_LAUNCH(GG)                                                                                           \
    launch_warp_epoch<GG>(serial != 0, blocks, threads, stream, itf, n_items, uf, pos_indptr, pos_indices, user_ids, item_ids, y,  \
                          sample_weight, ord, len, ip, up, nc, learning_rate, item_alpha, user_alpha,                 \
                          max_sampled, seed, log_table, scales)
            if (G == 16)
                MIREC_WARP_LAUNCH(16);
            else if (G == 32)
                MIREC_WARP_LAUNCH(32);
            else
                MIREC_WARP_LAUNCH(64);
#undef MIREC_WARP_LAUNCH
            MIREC_LAUNCH_CHECK();
            done += len;
        }
    }
    const int64_t ni_el = n_item_features * nc, nu_el = n_user_features * nc;
    hipLaunchKernelGGL(lightfm_fold_scale_kernel, dim3((unsigned)((ni_el + 255) / 256)), dim3(256), 0, stream, item_emb,
                       item_bias, n_item_features, nc, scales);
    MIREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(lightfm_fold_scale_kernel, dim3((unsigned)((nu_el + 255) / 256)), dim3(256), 0, stream, user_emb,
                       user_bias, n_user_features, nc, scales + 1);
    MIREC_LAUNCH_CHECK();
    return MIREC_OK;
}
