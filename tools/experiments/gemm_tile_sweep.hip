// Tile-shape sweep of gemm_rows_f64_kernel<WT, FR, TK> on the shapes the round engines multiply: the dense metric's
// half batch (2048 x 1024 x 1024), the logistic Q'·Xᵀ over ~half the rows (512 x 256 x 100032) and one block of the split-K
// R·X (512 x 2048 x 256, 49 blocks at once).  Every variant is checked against the k-ordered fma chain on sampled rows.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../dynamichmc.jl_amd/csrc/gemm_f64_mfma.hpp"
#include "gemm_f64_pc64.hpp"
using namespace dhmc;
template <int WT, int FR, int TK, bool BLK>
static void one(const char* name, int M, int K, int N, int kblk, const double* A, const double* B, double* O, const std::vector<double>& hA,
                const std::vector<double>& hB) {
    constexpr int T = 16 * FR * WT;
    if (N % T) { printf("  %-14s n/a (N %% %d)\n", name, T); return; }
    const int nz = kblk ? (K + kblk - 1) / kblk : 1;
    dim3 grid(N / T, (M + T - 1) / T, nz);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((gemm_rows_f64_kernel<WT, FR, TK, BLK>), grid, dim3(64 * WT * WT), 0, 0, A, K, B, N, O, N, K, M, nullptr, nullptr, kblk,
                           (size_t)M * N);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    std::vector<double> hO((size_t)M * N);
    (void)hipMemcpy(hO.data(), O, hO.size() * 8, hipMemcpyDeviceToHost);      // block z = 0
    const int Ke = kblk ? kblk : K;
    long bad = 0;
    for (int i = 0; i < M; i += 97)
        for (int j = 0; j < N; j += 13) {
            double acc = 0;
            for (int k = 0; k < Ke; ++k) acc = fma(hA[(size_t)i * K + k], hB[(size_t)k * N + j], acc);
            bad += acc != hO[(size_t)i * N + j];
        }
    printf("  %-14s grid %5d x %3d x %2d: %7.3f ms  %5.1f TFLOP/s  mismatches %ld\n", name, grid.x, grid.y, grid.z, best, 2.0 * M * K * N / best / 1e9, bad);
}
static void pc64(int M, int K, int N, int kblk, const double* A, const double* B, double* O, const std::vector<double>& hA, const std::vector<double>& hB) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    (void)hipMemset(O, 0xff, (size_t)M * N * 8);
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        launch_gemm_pc64(A, K, B, N, O, N, M, K, N, nullptr, nullptr, kblk, (size_t)M * N, 0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    hipError_t err = hipGetLastError();
    std::vector<double> hO((size_t)M * N);
    (void)hipMemcpy(hO.data(), O, hO.size() * 8, hipMemcpyDeviceToHost);
    const int Ke = kblk ? kblk : K;
    long bad = 0, n = 0;
    for (int i = 0; i < M; i += 31)
        for (int j = 0; j < N; j += 7) {
            double acc = 0;
            for (int k = 0; k < Ke; ++k) acc = fma(hA[(size_t)i * K + k], hB[(size_t)k * N + j], acc);
            bad += acc != hO[(size_t)i * N + j]; ++n;
        }
    printf("  %-14s                    : %7.3f ms  %5.1f TFLOP/s  mismatches %ld of %ld  (%s)\n", "pc64 4x4x4", best, 2.0 * M * K * N / best / 1e9, bad, n, hipGetErrorString(err));
}
static void shape(int M, int K, int N, int kblk) {
    std::vector<double> hA((size_t)M * K), hB((size_t)K * N);
    srand(1);
    for (auto& x : hA) x = (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 20) - 10);
    for (auto& x : hB) x = (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 20) - 10);
    const int nz = kblk ? (K + kblk - 1) / kblk : 1;
    double *A, *B, *O;
    (void)hipMalloc(&A, hA.size() * 8); (void)hipMalloc(&B, hB.size() * 8); (void)hipMalloc(&O, (size_t)M * N * 8 * nz);
    (void)hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(B, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
    printf("%d x %d x %d%s\n", M, K, N, kblk ? " (split-K blocks of 2048)" : "");
    one<2, 2, 16, false>("<2,2,16>", M, K, N, kblk, A, B, O, hA, hB);
    one<2, 2, 16, true>("<2,2,16,BLK>", M, K, N, kblk, A, B, O, hA, hB);
    one<2, 2, 32, false>("<2,2,32>", M, K, N, kblk, A, B, O, hA, hB);
    one<2, 4, 16, false>("<2,4,16>", M, K, N, kblk, A, B, O, hA, hB);
    one<2, 4, 16, true>("<2,4,16,BLK>", M, K, N, kblk, A, B, O, hA, hB);
    one<4, 2, 16, false>("<4,2,16>", M, K, N, kblk, A, B, O, hA, hB);
    one<4, 2, 16, true>("<4,2,16,BLK>", M, K, N, kblk, A, B, O, hA, hB);
    one<2, 1, 64, false>("<2,1,64>", M, K, N, kblk, A, B, O, hA, hB);
    pc64(M, K, N, kblk, A, B, O, hA, hB);
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(O);
}
int main(int argc, char** argv) {
    if (argc > 1) {                                  // quick mode: the new kernel's correctness on small and ragged shapes
        shape(64, 64, 64, 0); shape(100, 256, 128, 0); shape(200, 4096 + 32, 64, 2048);
    }
    shape(2048, 1024, 1024, 0);
    shape(4096, 1024, 1024, 0);
    shape(512, 256, 100032, 0);
    shape(512, 100032, 256, 2048);
    return 0;
}
