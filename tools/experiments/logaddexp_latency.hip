// Latency of one logaddexp_pair (the two deterministic logaddexp's of a tree merge, evaluated in even/odd lanes)
// as a dependent chain, one wave per SIMD — the scalar chain on the critical path of every merge.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../dynamichmc.jl_amd/csrc/nuts_kernels.hpp"
__global__ __launch_bounds__(64) void k(double* out, int n, double a0) {
    double a = a0, b = a0 - 0.3, c = -1.0, d = -2.5;
    const int lane = threadIdx.x;
    for (int i = 0; i < n; ++i) {
        double r1, r2;
        dhmc::logaddexp_pair(a, c, b, d, lane, r1, r2);
        a = r1 - 0.7; b = r2 - 0.6;      // dependent
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b;
}
int main() {
    double* out; (void)hipMalloc(&out, 1024 * 64 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int n = 20000;
    hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, out, 10, -0.1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, out, n, -0.1);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("logaddexp_pair: %.1f ns per call (one wave per SIMD)\n", ms * 1e6 / n);
    return 0;
}
