// How fast is a DEPENDENT chain of fp64 MFMAs (same accumulator) on gfx950?
// NACC independent accumulators per wave; WPS waves per SIMD (256 CUs x 4 SIMDs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, int KIND>
__global__ __launch_bounds__(256) void chain(double* out, int n, double a0, double b0, unsigned long long* clk) {
    d4 acc[NACC]; double acs[NACC];
    for (int j = 0; j < NACC; ++j) { acc[j] = d4{0, 0, 0, 0}; acs[j] = 0; }
    double a = a0 + threadIdx.x * 1e-9, b = b0, side = a0;
    double av[NACC], bv[NACC];
    for (int j = 0; j < NACC; ++j) { av[j] = a + 1e-3 * j; bv[j] = b - 1e-3 * j; asm volatile("" : "+v"(av[j]), "+v"(bv[j])); }
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            if (KIND == 16) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
            else if (KIND == 20) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(av[j]), "v"(bv[j]));     // accumulators in architectural VGPRs
            else if (KIND == 18) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j], bv[j], acc[j], 0, 0, 0);          // every accumulator its own operand registers
            else if (KIND == 19) { acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j], bv[j], acc[j], 0, 0, 0); asm volatile("v_add_f64 %0, %0, 1.0" : "+v"(side)); }   // … and an independent vector instruction after each
            else acs[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acs[j], 0, 0, 0);
        }
    double s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3] + acs[j];
    s += side;
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - r0; }
}
static unsigned long long* g_clk;
template <int NACC, int KIND>
void run(double* out, int n, int wps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((chain<NACC, KIND>), dim3(256 * wps), dim3(256), 0, 0, out, 16, 1.0, 1.0, (unsigned long long*)nullptr);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((chain<NACC, KIND>), dim3(256 * wps), dim3(256), 0, 0, out, n, 1.0, 1.0, g_clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)n * NACC, fl = KIND >= 16 ? 2048 : 512;
    unsigned long long h[2]; (void)hipMemcpy(h, g_clk, sizeof h, hipMemcpyDeviceToHost);
    printf("mfma_%dx%dx4 [kind %d] NACC=%d waves/SIMD=%d: %.1f ns per MFMA per wave, %.1f TFLOP/s; shader clock %.0f MHz (cycle counter against the 100 MHz wall clock), %.1f ms\n", KIND >= 16 ? 16 : 4, KIND >= 16 ? 16 : 4, KIND, NACC, wps,
           ms * 1e6 / mf, mf * 1024 * wps * fl / (ms * 1e-3) / 1e12, (double)h[0] / (double)h[1] * 100.0, ms);
}
int main() {
    double* out; (void)hipMalloc(&out, (size_t)256 * 8 * 256 * 8);
    (void)hipMalloc(&g_clk, 16);
    for (int wps : {1, 2, 4}) { run<1, 16>(out, 40000, wps); run<2, 16>(out, 20000, wps); run<4, 16>(out, 10000, wps); run<8, 16>(out, 5000, wps); run<16, 16>(out, 2500, wps); }
    for (int wps : {1, 4}) run<4, 16>(out, 400000, wps);      // 40 x longer: the clock under a sustained load
    for (int wps : {1, 2, 4}) { run<1, 20>(out, 40000, wps); run<2, 20>(out, 20000, wps); run<4, 20>(out, 10000, wps); run<8, 20>(out, 5000, wps); run<16, 20>(out, 2500, wps); }
    for (int wps : {1, 2, 4}) { run<4, 18>(out, 10000, wps); run<8, 18>(out, 5000, wps); run<16, 18>(out, 2500, wps); run<4, 19>(out, 10000, wps); run<8, 19>(out, 5000, wps); }
    for (int wps : {1, 2, 4, 8}) { run<1, 4>(out, 100000, wps); run<4, 4>(out, 25000, wps); }
    return 0;
}
