// Operand layout and summation order of v_mfma_f64_4x4x4_f64 on gfx950, found empirically:
//  (1) one-hot A and B lanes -> which D lanes receive the product;
//  (2) random wide-range operands -> which ordering of the 4 products of an output, as a chain of fma's from
//      the accumulator, reproduces the instruction bit for bit.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void onehot(double* D) {   // grid (64, 64): block (la, lb)
    int l = threadIdx.x, la = blockIdx.x, lb = blockIdx.y;
    double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    D[((size_t)la * 64 + lb) * 64 + l] = d;
}
__global__ void rnd(const double* A, const double* B, const double* C, double* D) {
    int l = threadIdx.x;
    D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], C[l], 0, 0, 0);
}
int main() {
    double* dD; (void)hipMalloc(&dD, 64 * 64 * 64 * 8);
    hipLaunchKernelGGL(onehot, dim3(64, 64), dim3(64), 0, 0, dD);
    std::vector<double> D(64 * 64 * 64);
    (void)hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);
    std::vector<std::vector<std::pair<int, int>>> src(64);
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) for (int l = 0; l < 64; ++l)
        if (D[((size_t)la * 64 + lb) * 64 + l] != 0) src[l].push_back({la, lb});
    for (int l = 0; l < 64; ++l) {
        printf("D lane %2d <-", l);
        for (auto& p : src[l]) printf(" (A%d,B%d)", p.first, p.second);
        printf("\n");
    }
    // (2) order
    srand(3);
    int perm[4] = {0, 1, 2, 3}, ok_perm[24] = {0}, np = 0, trials = 200;
    std::vector<double> A(64), B(64), C(64), R(64);
    double *dA, *dB, *dC, *dR; (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dC, 512); (void)hipMalloc(&dR, 512);
    std::vector<std::vector<int>> perms;
    do { perms.push_back({perm[0], perm[1], perm[2], perm[3]}); } while (std::next_permutation(perm, perm + 4));
    np = perms.size();
    for (int t = 0; t < trials; ++t) {
        for (int i = 0; i < 64; ++i) {
            A[i] = (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 30) - 15);
            B[i] = (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 30) - 15);
            C[i] = (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 30) - 15);
        }
        (void)hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
        (void)hipMemcpy(dC, C.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(rnd, dim3(1), dim3(64), 0, 0, dA, dB, dC, dR);
        (void)hipMemcpy(R.data(), dR, 512, hipMemcpyDeviceToHost);
        for (int pi = 0; pi < np; ++pi) {
            bool all = true;
            for (int l = 0; l < 64 && all; ++l) {
                double acc = C[l];
                for (int s = 0; s < 4; ++s) { auto& pr = src[l][perms[pi][s]]; acc = fma(A[pr.first], B[pr.second], acc); }
                all = acc == R[l];
            }
            ok_perm[pi] += all;
        }
    }
    for (int pi = 0; pi < np; ++pi)
        if (ok_perm[pi]) printf("order of the listed pairs %d%d%d%d reproduces all 64 outputs in %d / %d trials\n",
                                perms[pi][0], perms[pi][1], perms[pi][2], perms[pi][3], ok_perm[pi], trials);
    return 0;
}
