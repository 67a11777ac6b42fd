// How fast is a DEPENDENT chain of fp64 MFMAs (same accumulator) on gfx950?
// NACC independent accumulators per wave; WPS waves per SIMD (256 CUs x 4 SIMDs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, int KIND>
__global__ __launch_bounds__(256) void chain(double* out, int n, double a0, double b0) {
    d4 acc[NACC]; double acs[NACC];
    for (int j = 0; j < NACC; ++j) { acc[j] = d4{0, 0, 0, 0}; acs[j] = 0; }
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            if (KIND == 16) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
            else acs[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acs[j], 0, 0, 0);
        }
    double s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3] + acs[j];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int KIND>
void run(double* out, int n, int wps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((chain<NACC, KIND>), dim3(256 * wps), dim3(256), 0, 0, out, 16, 1.0, 1.0);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((chain<NACC, KIND>), dim3(256 * wps), dim3(256), 0, 0, out, n, 1.0, 1.0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)n * NACC, fl = KIND == 16 ? 2048 : 512;
    printf("mfma_%dx%dx4 NACC=%d waves/SIMD=%d: %.1f ns per MFMA per wave, %.1f TFLOP/s\n", KIND, KIND, NACC, wps,
           ms * 1e6 / mf, mf * 1024 * wps * fl / (ms * 1e-3) / 1e12);
}
int main() {
    double* out; (void)hipMalloc(&out, (size_t)256 * 8 * 256 * 8);
    for (int wps : {1, 2, 4}) { run<1, 16>(out, 40000, wps); run<2, 16>(out, 20000, wps); run<4, 16>(out, 10000, wps); run<8, 16>(out, 5000, wps); run<16, 16>(out, 2500, wps); }
    for (int wps : {1, 2, 4, 8}) { run<1, 4>(out, 100000, wps); run<4, 4>(out, 25000, wps); }
    return 0;
}
