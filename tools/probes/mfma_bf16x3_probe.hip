// Probe: v_mfma_f32_16x16x32_bf16 operand layout and the accuracy of a 3-way bf16 split (6 products)
// against an fp32 fma chain.  One wavefront; prints the maximum relative errors.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ inline unsigned short bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ inline float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

union Pack { bf16x8 v; unsigned short s[8]; };

__global__ void probe(const float* A, const float* B, float* D, int terms) {
    // A: [16][32] row-major (m, k); B: [32][16] (k, n)
    const int l = threadIdx.x, r = l & 15, kg = l >> 4;
    Pack a[3], b[3];
    for (int e = 0; e < 8; ++e) {
        float x = A[r * 32 + kg * 8 + e];
        for (int p = 0; p < 3; ++p) { unsigned short h = bf16_rne(x); a[p].s[e] = h; x -= bf16_f(h); }
        float y = B[(kg * 8 + e) * 16 + r];
        for (int p = 0; p < 3; ++p) { unsigned short h = bf16_rne(y); b[p].s[e] = h; y -= bf16_f(h); }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (terms >= 6) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2].v, b[0].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[2].v, acc, 0, 0, 0);
    }
    if (terms >= 4) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, b[1].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, b[0].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[1].v, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[0].v, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * kg + i) * 16 + r] = acc[i];
}

int main() {
    std::vector<float> A(16 * 32), B(32 * 16), D(256);
    srand(7);
    for (auto& x : A) x = (float)rand() / RAND_MAX - 0.3f;
    for (auto& x : B) x = ((float)rand() / RAND_MAX - 0.5f) * 3.f;
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int terms : {1, 4, 6}) {
        probe<<<1, 64>>>(dA, dB, dD, terms);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        double worst = 0, worst32 = 0, scale = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double ref = 0, mag = 0; float f = 0.f;
                for (int k = 0; k < 32; ++k) { ref += (double)A[m * 32 + k] * B[k * 16 + n]; mag += fabs((double)A[m * 32 + k] * B[k * 16 + n]); f = fmaf(A[m * 32 + k], B[k * 16 + n], f); }
                worst = fmax(worst, fabs(D[m * 16 + n] - ref) / mag);
                worst32 = fmax(worst32, fabs((double)f - ref) / mag);
                scale = fmax(scale, mag);
            }
        printf("terms=%d  max |err|/sum|prod| = %.3e   (fp32 fma chain: %.3e)\n", terms, worst, worst32);
    }
    return 0;
}
