// Where does gemm_rows_f64_kernel<2,2,16> lose against 78.6 TFLOP/s?  (round 6, second session)
//   1. fixed cost against steady state: M = 4096, N = 1024, K = 256 ... 16384 — time = t0 + K / rate;
//   2. ablations of the k-loop (same tiles, same launch): MODE 1 no global loads inside the loop (the first tile is multiplied again and again),
//      MODE 2 no LDS traffic either (operands stay in registers: the matrix instructions alone), MODE 3 loads and LDS traffic without the
//      matrix instructions.
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form gemm_ramp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void gemm_probe(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, double* __restrict__ OUT, int ldo,
                                                  int K) {
    constexpr int TK = 16, LS = 80, PER = 4;
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    __shared__ double As[TK * LS];
    __shared__ double Bs[TK * LS];
    const int a_row = t / 4, a_k = (t % 4) * PER;
    const double* a_src = A + (size_t)(row0 + a_row) * lda + a_k;
    const int b_k = t / 16, b_c = (t % 16) * PER;
    const double* b_src = B + (size_t)b_k * ldb + col0 + b_c;
    d4 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = d4{0, 0, 0, 0};
    double av[PER], bv[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) { av[i] = a_src[i]; bv[i] = b_src[i]; }
    double ra[2] = {av[0], av[1]}, rb[2] = {bv[0], bv[1]};
    for (int k0 = 0; k0 < K; k0 += TK) {
        if (MODE != 2) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PER; ++i) { As[(a_k + i) * LS + a_row] = av[i]; Bs[b_k * LS + b_c + i] = bv[i]; }
            __syncthreads();
        }
        if (MODE == 0 || MODE == 3) {
            if (k0 + TK < K) {
#pragma unroll
                for (int i = 0; i < PER; ++i) { av[i] = a_src[k0 + TK + i]; bv[i] = b_src[(size_t)(k0 + TK) * ldb + i]; }
            }
        }
#pragma unroll
        for (int kk = 0; kk < TK; kk += 4) {
            const int kr = (kk + (lane >> 4)) * LS;
            double a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (MODE != 2) { a[i] = As[kr + wr * 32 + 16 * i + (lane & 15)]; b[i] = Bs[kr + wc * 32 + 16 * i + (lane & 15)]; }
                else { a[i] = ra[i]; b[i] = rb[i]; }
            }
            if (MODE != 3) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            } else {
                acc[0][0][0] += a[0] + b[0]; acc[1][1][0] += a[1] + b[1];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* o = OUT + (size_t)(row0 + wr * 32 + i * 16 + (lane >> 4) + 4 * r) * ldo + col0 + wc * 32 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 2; ++j) o[16 * j] = acc[i][j][r];
        }
}

template <int MODE>
static float run(int M, int K, int N, const double* A, const double* B, double* O) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(gemm_probe<MODE>, dim3(N / 64, M / 64), dim3(256), 0, 0, A, K, B, N, O, N, K);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}
int main() {
    const int Kmax = 16384, Mmax = 16384, N = 1024;
    std::vector<double> h((size_t)Mmax * 1024);
    srand(1);
    for (auto& x : h) x = (rand() % 2001 - 1000) / 1000.0;
    double *A, *B, *O;
    (void)hipMalloc(&A, (size_t)4096 * Kmax * 8); (void)hipMalloc(&B, (size_t)Kmax * N * 8); (void)hipMalloc(&O, (size_t)Mmax * N * 8);
    for (size_t off = 0; off < (size_t)4096 * Kmax; off += h.size()) (void)hipMemcpy(A + off, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, h.data(), (size_t)Kmax * N * 8, hipMemcpyHostToDevice);
    printf("M = 4096, N = 1024: time against K (full kernel)\n");
    for (int K = 256; K <= Kmax; K *= 2) {
        const float ms = run<0>(4096, K, N, A, B, O);
        printf("  K %6d: %8.3f ms  %5.1f TFLOP/s\n", K, ms, 2.0 * 4096 * K * N / ms / 1e9);
    }
    printf("K = 1024, N = 1024: time against M (full kernel; A re-used beyond its 4096 rows is out of range: M <= 4096 * 16 only through lda)\n");
    for (int M = 512; M <= 4096; M *= 2) {
        const float ms = run<0>(M, 1024, N, A, B, O);
        printf("  M %6d: %8.3f ms  %5.1f TFLOP/s\n", M, ms, 2.0 * M * 1024 * N / ms / 1e9);
    }
    const char* names[4] = {"full", "no global loads in the loop", "matrix instructions from registers only", "loads + LDS, no matrix instructions"};
    for (int K : {1024, 8192}) {
        printf("M = 4096, N = 1024, K = %d: ablations\n", K);
        float ms[4] = {run<0>(4096, K, N, A, B, O), run<1>(4096, K, N, A, B, O), run<2>(4096, K, N, A, B, O), run<3>(4096, K, N, A, B, O)};
        for (int m = 0; m < 4; ++m) printf("  %-44s %8.3f ms  (%5.1f TFLOP/s equivalent)\n", names[m], ms[m], 2.0 * 4096 * K * N / ms[m] / 1e9);
    }
    return 0;
}
