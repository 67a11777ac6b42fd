// Round 6: the permuted-wave reduction (wave.hpp xl_reduce) against wave_allreduce, bit for bit, on random partial sums.
// build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o _v/xl_test xl_reduce_test.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "../../dynamichmc.jl_amd/csrc/wave.hpp"
using namespace dhmc;
// in: [N][64] logical-lane partial sums; out: [2][N] totals (old, xl) + flags
template <int N>
__global__ void k(const double* in, double* out, int* flags) {
    int p = threadIdx.x;
    int l = xl_logical_lane(p);
    double a[N], b[N];
    for (int i = 0; i < N; ++i) { a[i] = in[i * 64 + p]; b[i] = in[i * 64 + l]; }
    wave_allreduce<N>(a);
    double u[(N + 3) / 4];
    xl_reduce<N>(b, u);
    bool neg = xl_any_negative<N>(u);
    double c[N];
    for (int i = 0; i < N; ++i) c[i] = xl_value<N>(u, i);
    if (p == 0) { for (int i = 0; i < N; ++i) { out[i] = a[i]; out[N + i] = c[i]; } flags[0] = neg; }
}
template <int N> int run() {
    double* in; double* out; int* fl;
    (void)hipHostMalloc((void**)&in, sizeof(double) * N * 64); (void)hipHostMalloc((void**)&out, sizeof(double) * 2 * N); (void)hipHostMalloc((void**)&fl, 4);
    int bad = 0;
    for (int rep = 0; rep < 2000; ++rep) {
        for (int i = 0; i < N * 64; ++i) { double x = (double)rand() / RAND_MAX - 0.5; int e = rand() % 40 - 20; in[i] = ldexp(x, e); if (rep % 7 == 3 && rand() % 50 == 0) in[i] = -0.0; }
        if (rep % 5 == 0) for (int i = 0; i < 64; ++i) in[(rep % N) * 64 + i] = fabs(in[(rep % N) * 64 + i]);
        hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, in, out, fl);
        (void)hipDeviceSynchronize();
        bool neg = false;
        for (int i = 0; i < N; ++i) { if (memcmp(&out[i], &out[N + i], 8)) bad++; if (out[i] < 0) neg = true; }
        if ((int)neg != fl[0]) bad++;
    }
    printf("N=%d bad=%d\n", N, bad);
    return bad;
}
int main() { int b = run<1>() + run<2>() + run<3>() + run<4>() + run<6>(); printf(b ? "FAIL\n" : "XL REDUCE OK\n"); return b != 0; }
