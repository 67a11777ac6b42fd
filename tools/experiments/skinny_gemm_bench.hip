// Standalone timing of gemm_skinny_f64_kernel (R·X of the logistic gradient: 1024 × 100032 × 256) with parts
// of the K-step compiled out (-DDHMC_SK_NO_LOADS / -DDHMC_SK_NO_MFMA) to see what bounds it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../dynamichmc.jl_amd/csrc/gemm_f64_mfma.hpp"
int main() {
    const int M = 1024, K = 100032, N = 256;
    double *A, *B, *O;
    (void)hipMalloc(&A, (size_t)M * K * 8); (void)hipMalloc(&B, (size_t)K * N * 8); (void)hipMalloc(&O, (size_t)M * N * 8);
    (void)hipMemset(A, 0, (size_t)M * K * 8); (void)hipMemset(B, 0, (size_t)K * N * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        dhmc::launch_gemm(A, K, B, N, O, N, M, K, N, 0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms  (%.1f TFLOP/s)\n", VARIANT, ms, 2.0 * M * K * N / ms / 1e9);
    }
    return 0;
}
