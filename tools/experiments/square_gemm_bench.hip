// Standalone timing + self-check of the 64×64-tile fp64 MFMA GEMM (OUT = A·B, A [M×K], B [K×N]) on the dense
// metric's shape (2048 chains × 1024 × 1024) and the logistic η product (1024 × 256 × 100032).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../dynamichmc.jl_amd/csrc/gemm_f64_mfma.hpp"
static void run(int M, int K, int N, bool check) {
    std::vector<double> hA((size_t)M * K), hB((size_t)K * N), hO((size_t)M * N);
    srand(1);
    for (auto& x : hA) x = (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 20) - 10);
    for (auto& x : hB) x = (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 20) - 10);
    double *A, *B, *O;
    (void)hipMalloc(&A, hA.size() * 8); (void)hipMalloc(&B, hB.size() * 8); (void)hipMalloc(&O, hO.size() * 8);
    (void)hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(B, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        dhmc::launch_gemm(A, K, B, N, O, N, M, K, N, 0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%d x %d x %d: %.3f ms  (%.1f TFLOP/s)", M, K, N, best, 2.0 * M * K * N / best / 1e9);
    if (check) {
        (void)hipMemcpy(hO.data(), O, hO.size() * 8, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < M; i += 37)
            for (int j = 0; j < N; ++j) {
                double acc = 0;
                for (int k = 0; k < K; ++k) acc = fma(hA[(size_t)i * K + k], hB[(size_t)k * N + j], acc);
                bad += acc != hO[(size_t)i * N + j];
            }
        printf("  mismatches vs k-ordered fma chain (sampled rows): %ld", bad);
    }
    printf("\n");
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(O);
}
int main() {
    run(2048, 1024, 1024, true);
    run(4096, 1024, 1024, false);
    run(1024, 256, 100032, false);
    run(1024, 100032, 256, true);
    return 0;
}
