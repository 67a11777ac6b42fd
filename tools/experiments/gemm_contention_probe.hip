// Does the product kernel's single k-tile of prefetch survive a neighbour that saturates HBM?  (round 6, second session)
// In the dense round engine a half-batch's product runs beside the OTHER half's tree kernel (rounds_k3b_kernel: HBM streaming at ~5 TB/s), and both
// then take twice their stand-alone time (profiles/r06_config3_alternation_experiment.txt).  Here: the library-form kernel (rotated A columns) with
// PD = 1, 2, 3 k-tiles of global loads in flight, timed alone and beside a streaming kernel on a second stream.
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form gemm_contention_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));

template <int PD>
__global__ __launch_bounds__(256) void gemm_pd(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, double* __restrict__ OUT, int ldo, int K) {
    constexpr int TK = 16, LS = 80, PER = 4;
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    __shared__ __attribute__((aligned(16))) double As[TK * LS];
    __shared__ __attribute__((aligned(16))) double Bs[TK * LS];
    const int a_row = t / 4, a_k = (t % 4) * PER;
    const double* a_src = A + (size_t)(row0 + a_row) * lda + a_k;
    const int b_k = t / 16, b_c = (t % 16) * PER;
    const double* b_src = B + (size_t)b_k * ldb + col0 + b_c;
    const int a_col = (a_row + 2 * a_k) & 63;
    d4 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = d4{0, 0, 0, 0};
    double av[PD][PER], bv[PD][PER];
    const int nt = K / TK;
    auto issue = [&](int tile, double (&a)[PER], double (&b)[PER]) {
        const int tt = tile < nt ? tile : nt - 1;
#pragma unroll
        for (int i = 0; i < PER; ++i) { a[i] = a_src[tt * TK + i]; b[i] = b_src[(size_t)(tt * TK) * ldb + i]; }
    };
#pragma unroll
    for (int s = 0; s < PD; ++s) issue(s, av[s], bv[s]);
    auto step = [&](int tile, int s) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) As[(a_k + i) * LS + a_col] = av[s][i];
        *reinterpret_cast<d2v*>(&Bs[b_k * LS + b_c]) = d2v{bv[s][0], bv[s][1]};
        *reinterpret_cast<d2v*>(&Bs[b_k * LS + b_c + 2]) = d2v{bv[s][2], bv[s][3]};
        __syncthreads();
        issue(tile + PD, av[s], bv[s]);
#pragma unroll
        for (int kk = 0; kk < TK; kk += 4) {
            const int kr = (kk + (lane >> 4)) * LS;
            double a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = As[kr + ((wr * 32 + 16 * i + (lane & 15) + 2 * kk) & 63)];
                b[i] = Bs[kr + wc * 32 + 16 * i + (lane & 15)];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    const int nfull = nt - nt % PD;
    for (int t0 = 0; t0 < nfull; t0 += PD) {
#pragma unroll
        for (int s = 0; s < PD; ++s) step(t0 + s, s);
    }
#pragma unroll
    for (int s = 0; s < PD; ++s)
        if (nfull + s < nt) step(nfull + s, s);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* o = OUT + (size_t)(row0 + wr * 32 + i * 16 + (lane >> 4) + 4 * r) * ldo + col0 + wc * 32 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 2; ++j) o[16 * j] = acc[i][j][r];
        }
}

// the neighbour: a grid-stride stream over n doubles (read + write), `waves` resident waves per SIMD worth of blocks
__global__ __launch_bounds__(256) void hog(const double* __restrict__ in, double* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * 2;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i + 1 < n; i += stride) {
        const d2v v = *reinterpret_cast<const d2v*>(in + i);
        *reinterpret_cast<d2v*>(out + i) = d2v{v[0] + 1.0, v[1] + 1.0};
    }
}

template <int PD>
static void run(const char* name, int M, int K, int N, const double* A, const double* B, double* O, const double* hin, double* hout, size_t hn, int hog_blocks,
                hipStream_t s1, hipStream_t s2) {
    hipEvent_t e0, e1, h0, h1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&h0); (void)hipEventCreate(&h1);
    float alone = 1e9, beside = 1e9, hog_ms = 0, hog_alone = 1e9, span_ms = 0, first_ms = 0;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0, s1);
        hipLaunchKernelGGL(gemm_pd<PD>, dim3(N / 64, M / 64), dim3(256), 0, s1, A, K, B, N, O, N, K);
        (void)hipEventRecord(e1, s1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); alone = ms < alone ? ms : alone;
    }
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(h0, s2);
        hipLaunchKernelGGL(hog, dim3(hog_blocks), dim3(256), 0, s2, hin, hout, hn);
        (void)hipEventRecord(h1, s2); (void)hipEventSynchronize(h1);
        float ms; (void)hipEventElapsedTime(&ms, h0, h1); hog_alone = ms < hog_alone ? ms : hog_alone;
    }
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(h0, s2);
        hipLaunchKernelGGL(hog, dim3(hog_blocks), dim3(256), 0, s2, hin, hout, hn);
        (void)hipEventRecord(h1, s2);
        // a few products back to back inside the neighbour's run; the fastest of them after the first (which overlaps the neighbour's start)
        float best = 1e9;
        hipEvent_t ev[5];
        for (int q = 0; q < 5; ++q) (void)hipEventCreate(&ev[q]);
        (void)hipEventRecord(ev[0], s1);
        for (int q = 1; q < 5; ++q) {
            hipLaunchKernelGGL(gemm_pd<PD>, dim3(N / 64, M / 64), dim3(256), 0, s1, A, K, B, N, O, N, K);
            (void)hipEventRecord(ev[q], s1);
        }
        (void)hipDeviceSynchronize();
        for (int q = 2; q < 4; ++q) { float ms; (void)hipEventElapsedTime(&ms, ev[q - 1], ev[q]); best = ms < best ? ms : best; }
        beside = best < beside ? best : beside;
        (void)hipEventElapsedTime(&hog_ms, h0, h1);
        (void)hipEventElapsedTime(&span_ms, h0, ev[4]);
        (void)hipEventElapsedTime(&first_ms, h0, ev[1]);
    }
    printf("  %-28s alone %7.3f ms (%5.1f TFLOP/s)   beside the stream %7.3f ms (%5.1f)   [stream alone %6.3f ms = %4.2f TB/s, with 4 products inside %6.3f ms; stream start -> first product's end %6.3f, -> fourth product's end %6.3f ms]\n", name, alone,
           2.0 * M * K * N / alone / 1e9, beside, 2.0 * M * K * N / beside / 1e9, hog_alone, 16.0 * hn / hog_alone / 1e9, hog_ms, first_ms, span_ms);
}

// a neighbour shaped like the tree kernel: 2048 short-lived workgroups of 256 threads, ~100 VGPRs (40 doubles held per thread), 13 KB of LDS, each
// streaming ~140 KB (17 rows of 8 KB in, 8 out) with two barriers
__global__ __launch_bounds__(256) void treeish(const double* __restrict__ in, double* __restrict__ out) {
    __shared__ double xch[6][4][64];
    const size_t base = (size_t)blockIdx.x * 17 * 1024;
    double v[17][4];
#pragma unroll
    for (int r = 0; r < 17; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[r][k] = in[base + r * 1024 + threadIdx.x + 256 * k];
    double acc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[r % 6] = __builtin_fma(v[r][k], v[r + 1][k], acc[r % 6]);
#pragma unroll
    for (int n = 0; n < 6; ++n) xch[n][threadIdx.x >> 6][threadIdx.x & 63] = acc[n];
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int n = 0; n < 6; ++n)
#pragma unroll
        for (int w = 0; w < 4; ++w) t += xch[n][w][threadIdx.x & 63];
    __syncthreads();
    const size_t ob = (size_t)blockIdx.x * 8 * 1024;
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) out[ob + r * 1024 + threadIdx.x + 256 * k] = v[r][k] + v[r + 8][k] + t;
}

static void tree_run(const char* name, int M, int K, int N, const double* A, const double* B, double* O, const double* hin, double* hout, hipStream_t sg, hipStream_t st) {
    hipEvent_t e[8], h0, h1; for (auto& x : e) (void)hipEventCreate(&x); (void)hipEventCreate(&h0); (void)hipEventCreate(&h1);
    float g_alone = 1e9, t_alone = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e[0], sg); hipLaunchKernelGGL(gemm_pd<2>, dim3(N / 64, M / 64), dim3(256), 0, sg, A, K, B, N, O, N, K); (void)hipEventRecord(e[1], sg);
        (void)hipEventSynchronize(e[1]); float ms; (void)hipEventElapsedTime(&ms, e[0], e[1]); g_alone = ms < g_alone ? ms : g_alone;
        (void)hipEventRecord(h0, st); hipLaunchKernelGGL(treeish, dim3(2048), dim3(256), 0, st, hin, hout); (void)hipEventRecord(h1, st);
        (void)hipEventSynchronize(h1); (void)hipEventElapsedTime(&ms, h0, h1); t_alone = ms < t_alone ? ms : t_alone;
    }
    // six tree-like launches back to back on one stream; three products on the other, the first enqueued once the neighbour is running
    float best_span = 1e9, g_in[3] = {0, 0, 0}, t_total = 0;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(h0, st);
        for (int q = 0; q < 6; ++q) hipLaunchKernelGGL(treeish, dim3(2048), dim3(256), 0, st, hin, hout);
        (void)hipEventRecord(h1, st);
        (void)hipEventRecord(e[0], sg);
        for (int q = 0; q < 3; ++q) { hipLaunchKernelGGL(gemm_pd<2>, dim3(N / 64, M / 64), dim3(256), 0, sg, A, K, B, N, O, N, K); (void)hipEventRecord(e[q + 1], sg); }
        (void)hipDeviceSynchronize();
        float span; (void)hipEventElapsedTime(&span, h0, e[3]);
        float tt; (void)hipEventElapsedTime(&tt, h0, h1);
        if (span < best_span) { best_span = span; t_total = tt; for (int q = 0; q < 3; ++q) (void)hipEventElapsedTime(&g_in[q], e[q], e[q + 1]); }
    }
    printf("  %-34s product alone %6.3f ms, tree-like launch alone %6.3f ms;  six tree-like launches with three products beside them: %6.3f ms, the products %6.3f / %6.3f / %6.3f ms, all done after %6.3f ms  (serial: %6.3f)\n",
           name, g_alone, t_alone, t_total, g_in[0], g_in[1], g_in[2], best_span, 6 * t_alone + 3 * g_alone);
}

int main() {
    const int K = 1024, N = 1024, Mmax = 4096;
    std::vector<double> h((size_t)Mmax * K);
    srand(1);
    for (auto& x : h) x = (rand() % 2001 - 1000) / 1000.0;
    double *A, *B, *O, *hin, *hout;
    const size_t hn = (size_t)1 << 27;      // 1 GiB in + 1 GiB out
    (void)hipMalloc(&A, h.size() * 8); (void)hipMalloc(&B, (size_t)K * N * 8); (void)hipMalloc(&O, (size_t)Mmax * N * 8);
    (void)hipMalloc(&hin, hn * 8); (void)hipMalloc(&hout, hn * 8);
    (void)hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(B, h.data(), (size_t)K * N * 8, hipMemcpyHostToDevice);
    (void)hipMemset(hin, 0, hn * 8);
    hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
    {
        int lo, hi; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        hipStream_t s_hi, s_lo; (void)hipStreamCreateWithPriority(&s_hi, hipStreamNonBlocking, hi); (void)hipStreamCreateWithPriority(&s_lo, hipStreamNonBlocking, lo);
        printf("a neighbour shaped like the tree kernel (2048 short-lived workgroups per launch); stream priorities %d (low) .. %d (high)\n", lo, hi);
        tree_run("both streams at default priority", 2048, K, N, A, B, O, hin, hout, s1, s2);
        tree_run("products on a HIGH-priority stream", 2048, K, N, A, B, O, hin, hout, s_hi, s2);
        tree_run("... and the neighbour on a LOW one", 2048, K, N, A, B, O, hin, hout, s_hi, s_lo);
        tree_run("both streams at default priority", 2048, K, N, A, B, O, hin, hout, s1, s2);
    }
    for (int hog_blocks : {512}) {
        for (int M : {2048}) {
            printf("M = %d, K = N = 1024, neighbour of %d blocks:\n", M, hog_blocks);
            run<1>("1 k-tile of loads in flight", M, K, N, A, B, O, hin, hout, hn, hog_blocks, s1, s2);
            run<2>("2 k-tiles", M, K, N, A, B, O, hin, hout, hn, hog_blocks, s1, s2);
        }
    }
    return 0;
}
