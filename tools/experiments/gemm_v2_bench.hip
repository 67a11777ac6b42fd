// gemm_rows_f64_v2_kernel (gemm_f64_v2.hpp) against gemm_rows_f64_kernel<2,2,16> on the round engines' product shapes; every variant
// checked against the k-ordered fma chain on sampled outputs.   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off gemm_v2_bench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_f64_v2.hpp"
using namespace dhmc;

static long check(const std::vector<double>& hA, const std::vector<double>& hB, const double* O, int M, int K, int N, int kblk) {
    std::vector<double> hO((size_t)M * N);
    (void)hipMemcpy(hO.data(), O, hO.size() * 8, hipMemcpyDeviceToHost);
    const int Ke = kblk ? kblk : K;
    long bad = 0;
    for (int i = 0; i < M; i += 53)
        for (int j = 0; j < N; j += 11) {
            double acc = 0;
            for (int k = 0; k < Ke; ++k) acc = fma(hA[(size_t)i * K + k], hB[(size_t)k * N + j], acc);
            bad += acc != hO[(size_t)i * N + j];
        }
    return bad;
}
template <class F>
static float time_it(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}
template <int WM, int WN, int FM, int FN, int TK>
static void v2(const char* name, int M, int K, int N, int kblk, const double* A, const double* B, double* O, const std::vector<double>& hA,
               const std::vector<double>& hB) {
    constexpr int TM = 16 * FM * WM, TN = 16 * FN * WN;
    if (N % TN) { printf("  %-22s n/a\n", name); return; }
    const int nz = kblk ? (K + kblk - 1) / kblk : 1;
    dim3 grid(N / TN, (M + TM - 1) / TM, nz);
    const size_t lds = gemm_v2_lds_bytes<WM, WN, FM, FN, TK>();
    (void)hipFuncSetAttribute((const void*)gemm_rows_f64_v2_kernel<WM, WN, FM, FN, TK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipMemset(O, 0xff, (size_t)M * N * 8);
    const float ms = time_it([&] {
        hipLaunchKernelGGL((gemm_rows_f64_v2_kernel<WM, WN, FM, FN, TK>), grid, dim3(64 * WM * WN), lds, 0, A, K, B, N, O, N, K, M, nullptr, nullptr, kblk,
                           (size_t)M * N);
    });
    const hipError_t e = hipGetLastError();
    printf("  %-22s grid %5d x %3d x %2d  lds %6zu: %7.3f ms  %5.1f TFLOP/s  mismatches %ld  (%s)\n", name, grid.x, grid.y, grid.z, lds, ms,
           2.0 * M * K * N / ms / 1e9, check(hA, hB, O, M, K, N, kblk), hipGetErrorString(e));
}
template <int WT, int FR, int TK>
static void base_t(const char* name, int M, int K, int N, int kblk, const double* A, const double* B, double* O, const std::vector<double>& hA, const std::vector<double>& hB) {
    const int nz = kblk ? (K + kblk - 1) / kblk : 1;
    constexpr int T = 16 * FR * WT;
    if (N % T) { printf("  %-22s n/a\n", name); return; }
    dim3 grid(N / T, (M + T - 1) / T, nz);
    const float ms = time_it([&] {
        hipLaunchKernelGGL((gemm_rows_f64_kernel<WT, FR, TK, false>), grid, dim3(64 * WT * WT), 0, 0, A, K, B, N, O, N, K, M, nullptr, nullptr, kblk, (size_t)M * N);
    });
    printf("  %-22s grid %5d x %3d x %2d            : %7.3f ms  %5.1f TFLOP/s  mismatches %ld\n", name, grid.x, grid.y, grid.z, ms,
           2.0 * M * K * N / ms / 1e9, check(hA, hB, O, M, K, N, kblk));
}
static void base(int M, int K, int N, int kblk, const double* A, const double* B, double* O, const std::vector<double>& hA, const std::vector<double>& hB) {
    base_t<2, 2, 16>("library <2,2,16>", M, K, N, kblk, A, B, O, hA, hB);
    if (getenv("GEMM_BENCH_LIBRARY_TK")) {
        base_t<2, 2, 32>("library <2,2,32>", M, K, N, kblk, A, B, O, hA, hB);
        base_t<2, 2, 64>("library <2,2,64>", M, K, N, kblk, A, B, O, hA, hB);
        base_t<2, 2, 8>("library <2,2,8>", M, K, N, kblk, A, B, O, hA, hB);
    }
}
int main() {
    struct Shape { int M, K, N, kblk; } shapes[] = {{4096, 1024, 1024, 0}, {2048, 1024, 1024, 0}, {1024, 1024, 1024, 0}, {300, 1024, 1024, 0},
                                                    {1024, 256, 100352, 0}, {512, 256, 100352, 0}, {1024, 100352, 256, 2048}, {512, 100352, 256, 2048},
                                                    {8192, 8192, 8192, 0}};
    for (auto s : shapes) {
        const int nz = s.kblk ? (s.K + s.kblk - 1) / s.kblk : 1;
        std::vector<double> hA((size_t)s.M * s.K), hB((size_t)s.K * s.N);
        srand(1);
        for (auto& x : hA) x = (rand() % 2001 - 1000) / 1000.0;
        for (auto& x : hB) x = (rand() % 2001 - 1000) / 1000.0;
        double *A, *B, *O;
        (void)hipMalloc(&A, hA.size() * 8); (void)hipMalloc(&B, hB.size() * 8); (void)hipMalloc(&O, (size_t)s.M * s.N * 8 * nz);
        (void)hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(B, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
        printf("%d x %d x %d%s\n", s.M, s.K, s.N, s.kblk ? " (split-K blocks of 2048)" : "");
        base(s.M, s.K, s.N, s.kblk, A, B, O, hA, hB);
        v2<2, 2, 4, 4, 16>("v2 128x128 (4x4) tk16", s.M, s.K, s.N, s.kblk, A, B, O, hA, hB);
        v2<2, 2, 4, 4, 8>("v2 128x128 (4x4) tk8", s.M, s.K, s.N, s.kblk, A, B, O, hA, hB);
        v2<2, 2, 4, 2, 16>("v2 128x64 (4x2) tk16", s.M, s.K, s.N, s.kblk, A, B, O, hA, hB);
        v2<2, 2, 2, 4, 16>("v2 64x128 (2x4) tk16", s.M, s.K, s.N, s.kblk, A, B, O, hA, hB);
        v2<2, 2, 2, 2, 16>("v2 64x64 (2x2) tk16", s.M, s.K, s.N, s.kblk, A, B, O, hA, hB);
        v2<2, 2, 2, 2, 32>("v2 64x64 (2x2) tk32", s.M, s.K, s.N, s.kblk, A, B, O, hA, hB);
        (void)hipFree(A); (void)hipFree(B); (void)hipFree(O);
    }
    return 0;
}
