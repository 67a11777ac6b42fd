// Is v_mfma_f64_16x16x4_f64 a k-ordered chain of correctly rounded fma's?  (The HIP guide states it
// for the f32 forms; the dense-metric GEMM's bit-exact parity with the oracle depends on it for f64.)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k(const double* A, const double* B, double* C, int K) {   // A [16][K], B [K][16], C [16][16]
    int l = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 4) {
        double a = A[(l & 15) * K + k0 + (l >> 4)];
        double b = B[(k0 + (l >> 4)) * 16 + (l & 15)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];   // row = (lane>>4) + 4*reg, col = lane&15
}
int main() {
    const int K = 64;
    std::vector<double> A(16 * K), B(K * 16), C(256), R(256);
    srand(1);
    for (auto& x : A) x = (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 40) - 20);
    for (auto& x : B) x = (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 40) - 20);
    double *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, 256 * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
    hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost);
    int bad_chain = 0, bad_naive = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double acc = 0, acc2 = 0;
            for (int kk = 0; kk < K; ++kk) { acc = fma(A[i * K + kk], B[kk * 16 + j], acc); acc2 = acc2 + A[i * K + kk] * B[kk * 16 + j]; }
            bad_chain += (acc != C[i * 16 + j]);
            bad_naive += (acc2 != C[i * 16 + j]);
        }
    printf("mismatches vs k-ordered fma chain: %d / 256   (vs mul+add chain: %d / 256)\n", bad_chain, bad_naive);
    return 0;
}
