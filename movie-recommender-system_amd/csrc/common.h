// Shared helpers of the mirec HIP sources (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "mirec.h"

namespace mirec {

void set_error(const char* fmt, ...);

#define MIREC_REQUIRE(cond, code, ...)      \
    do {                                    \
        if (!(cond)) {                      \
            mirec::set_error(__VA_ARGS__);  \
            return (code);                  \
        }                                   \
    } while (0)

#define MIREC_HIP(call)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            mirec::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                             __LINE__);                                                        \
            return MIREC_E_HIP;                                                                \
        }                                                                                      \
    } while (0)

// Checks the launch just made (launch errors surface through hipGetLastError).
#define MIREC_LAUNCH_CHECK()                                                                   \
    do {                                                                                       \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess) {                                                                \
            mirec::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_),       \
                             __FILE__, __LINE__);                                              \
            return MIREC_E_HIP;                                                                \
        }                                                                                      \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

// Kernel attributes (dynamic LDS size) are set once per device: a process may drive several GPUs.
constexpr int kMaxDevices = 64;
struct OncePerDevice {
    bool done[kMaxDevices] = {};
    // Slot of the current device, or nullptr if its ordinal is out of range / cannot be read.
    bool* slot() {
        int d = -1;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return nullptr;
        return &done[d];
    }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace mirec
