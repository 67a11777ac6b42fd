// Rate of the "16x16x4 step as four 4x4x4 block MFMAs" inner loop in isolation (operands in registers):
// with and without the three DPP rotations of B per step.  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../dynamichmc.jl_amd/csrc/gemm_f64_mfma.hpp"
template <int DPP>
__global__ __launch_bounds__(256, 1) void k(double* out, int n, const double* in) {
    double a[16], b[16], acc[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; ++i) { a[i] = in[threadIdx.x + 256 * i]; b[i] = in[threadIdx.x + 256 * (16 + i)]; }
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (DPP) { for (int s = 0; s < 4; ++s) acc[s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[(i + (s >> 1)) & 15], b[(i + (s & 1)) & 15], acc[s], 0, 0, 0); }
            else {
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[(i + s) & 15], acc[s], 0, 0, 0);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int DPP> void run(double* out, const double* in) {
    const int n = 1563;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<DPP>, dim3(256), dim3(256), 0, 0, out, 10, in);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<DPP>, dim3(256), dim3(256), 0, 0, out, n, in);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("DPP=%d: %.3f ms for %d K-tiles of 64 MFMAs -> %.1f ns per K-tile, %.1f TFLOP/s\n", DPP, ms, n, ms * 1e6 / n,
           (double)n * 64 * 512 * 1024 / ms / 1e9);
}
int main() {
    double *out, *in; (void)hipMalloc(&out, 256 * 256 * 8); (void)hipMalloc(&in, 256 * 32 * 8); (void)hipMemset(in, 0, 256 * 32 * 8);
    run<0>(out, in); run<1>(out, in);
    return 0;
}
