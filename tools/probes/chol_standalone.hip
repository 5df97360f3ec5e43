// Hammer chol_solve_block in isolation: every workgroup solves the SAME system; all answers must have the same bits.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Imovie-recommender-system_amd/csrc \
//        tools/probes/chol_standalone.hip -o gpurun_out/chol_standalone
#include "../../movie-recommender-system_amd/csrc/als_sweep.hip"
#include <vector>
#include <random>
#include <cstring>

template <int F, int NTHR, int NTHR_CHOL>
__global__ __launch_bounds__(NTHR) void hammer(const float* __restrict__ A0, const float* __restrict__ b0, float* out) {
    constexpr int LDA = F + 1;
    __shared__ __attribute__((aligned(16))) float A[F * LDA];
    __shared__ __attribute__((aligned(16))) float bpan[8 * (F + 4)];
    const int tid = threadIdx.x;
    // junk that differs from workgroup to workgroup everywhere the solver may look: a read of anything the solver
    // has not written itself shows up as differing answers
    for (int i = tid; i < F * LDA; i += NTHR) A[i] = __uint_as_float(0x3f000000u + 977u * blockIdx.x + i);
    for (int i = tid; i < 8 * (F + 4); i += NTHR) bpan[i] = __uint_as_float(0x40000000u + 131u * blockIdx.x + i);
    __syncthreads();
    for (int i = tid; i < F * F; i += NTHR) A[(i / F) * LDA + (i % F)] = A0[i];
    for (int i = tid; i < F; i += NTHR) bpan[i] = b0[i];
    __syncthreads();
    mirec::chol_solve_block<F, NTHR_CHOL>(A, bpan, bpan, out + (size_t)blockIdx.x * F, tid);
}

template <int F, int NTHR, int NTHR_CHOL>
int run(int blocks, int reps) {
    std::mt19937 rng(7);
    std::normal_distribution<float> nd;
    std::vector<float> Y(200 * F), A(F * F, 0.f), b(F);
    for (auto& v : Y) v = nd(rng) * 0.1f;
    for (int k = 0; k < 200; ++k)
        for (int i = 0; i < F; ++i)
            for (int j = 0; j < F; ++j) A[i * F + j] += 39.f * Y[k * F + i] * Y[k * F + j];
    for (int i = 0; i < F; ++i) { A[i * F + i] += 0.01f; b[i] = nd(rng); }
    float *dA, *db, *dout;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dout, (size_t)blocks * F * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> out((size_t)blocks * F);
    long bad = 0;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL((hammer<F, NTHR, NTHR_CHOL>), dim3(blocks), dim3(NTHR), 0, 0, dA, db, dout);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
        for (int k = 1; k < blocks; ++k)
            if (memcmp(&out[(size_t)k * F], &out[0], F * 4) != 0) ++bad;
    }
    printf("F=%d threads=%d (blocks owned by the first %d): %ld of %ld solves differ from the first\n", F, NTHR, NTHR_CHOL, bad,
           (long)reps * (blocks - 1));
    hipFree(dA); hipFree(db); hipFree(dout);
    return 0;
}

int main() {
    run<64, 256, 256>(20000, 10);
    run<64, 128, 128>(20000, 10);
    run<128, 512, 512>(8000, 5);
    run<32, 128, 128>(20000, 10);
    return 0;
}
