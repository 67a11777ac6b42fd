// Do v_mfma_f64_16x16x4_f64 (VGPR accumulators) and fp64 vector FMAs of ANOTHER wave on the same SIMD overlap on gfx950?
// Workgroups of 512 threads = 2 waves per SIMD: waves 0-3 run a chain of matrix instructions (4 accumulators), waves 4-7 a chain of
// v_fma_f64 (8 independent accumulators); each role also runs alone (the other role's waves leave at once).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void mix(double* out, int n_mfma, int n_valu, int mode) {   // mode 1: matrix only, 2: vector only, 3: both, 4: all eight waves vector
    const int wv = threadIdx.x >> 6;
    double s = 0;
    if (wv < 4 && mode != 4) {
        if (!(mode & 1)) return;
        d4 acc[4];
        for (int j = 0; j < 4; ++j) acc[j] = d4{0, 0, 0, 0};
        double a = 1.0 + threadIdx.x * 1e-9, b = 1.0;
        for (int i = 0; i < n_mfma; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
        for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else {
        if (!(mode & 2) && mode != 4) return;
        double x[8];
        for (int j = 0; j < 8; ++j) x[j] = 1.0 + j + threadIdx.x * 1e-9;
        const double c = 0.999999, d = 1e-9;
        for (int i = 0; i < n_valu; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[j]) : "v"(c), "v"(d));
        for (int j = 0; j < 8; ++j) s += x[j];
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}
static float run(double* out, int nm, int nv, int mode) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mix, dim3(256), dim3(512), 0, 0, out, 16, 16, mode);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(mix, dim3(256), dim3(512), 0, 0, out, nm, nv, mode);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    double* out; (void)hipMalloc(&out, (size_t)256 * 512 * 8);
    const int nm = 20000, nv = 80000;          // 80 000 matrix instructions and 640 000 fma's per wave
    const float m = run(out, nm, nv, 1), v = run(out, nm, nv, 2), b = run(out, nm, nv, 3);
    printf("matrix waves alone  %.2f ms  (%.1f TFLOP/s)\n", m, 4.0 * nm * 2048 * 1024 / (m * 1e-3) / 1e12);
    printf("vector waves alone  %.2f ms  (%.1f TFLOP/s)\n", v, 8.0 * nv * 128 * 1024 / (v * 1e-3) / 1e12);
    printf("both together       %.2f ms  (sum of the two alone: %.2f ms, the longer one: %.2f ms)\n", b, m + v, m > v ? m : v);
    const float v2 = run(out, nm, nv, 4);
    printf("vector waves, two per SIMD  %.2f ms  (%.1f TFLOP/s; one per SIMD: %.1f)\n", v2, 2 * 8.0 * nv * 128 * 1024 / (v2 * 1e-3) / 1e12, 8.0 * nv * 128 * 1024 / (v * 1e-3) / 1e12);
    return 0;
}
