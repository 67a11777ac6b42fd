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


// VARIANT kernels: V = 1 the A tile's LDS columns rotated by 8 (k >> 2) (the four k-groups a write instruction covers land on four bank ranges
// instead of one: no 4-way conflict), B written as two 16-byte stores; V = 2 the same with two LDS buffers and ONE barrier per k-tile.
typedef double d2v __attribute__((ext_vector_type(2)));
template <int V>
__global__ __launch_bounds__(256) void gemm_var(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, double* __restrict__ OUT, int ldo,
                                                int K) {
    constexpr int TK = 16, LS = 80, PER = 4, NB = V == 2 ? 2 : 1;
    int bx = blockIdx.x, by = blockIdx.y;
    if (V == 3) {      // consecutive workgroup ids go round the 8 XCDs: give every XCD whole row blocks (A fetched by one L2 only)
        const int L = blockIdx.x + gridDim.x * blockIdx.y, total = gridDim.x * gridDim.y, per = (total + 7) / 8;
        const int tix = (L & 7) * per + (L >> 3);
        if (tix >= total) return;
        bx = tix % gridDim.x; by = tix / gridDim.x;
    }
    const int row0 = by * 64, col0 = bx * 64;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    __shared__ double As[NB][TK * LS];
    __shared__ double Bs[NB][TK * LS];
    const int a_row = t / 4, a_k = (t % 4) * PER;
    const double* a_src = A + (size_t)(row0 + a_row) * lda + a_k;
    const int b_k = t / 16, b_c = (t % 16) * PER;
    const double* b_src = B + (size_t)b_k * ldb + col0 + b_c;
    const int a_col = (a_row + 8 * (a_k >> 2)) & 63;       // the thread's four k share k >> 2
    d4 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = d4{0, 0, 0, 0};
    double av[PER], bv[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) { av[i] = a_src[i]; bv[i] = b_src[i]; }
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += TK) {
        if (V != 2) __syncthreads();
        double* as = As[buf];
        double* bs = Bs[buf];
#pragma unroll
        for (int i = 0; i < PER; ++i) as[(a_k + i) * LS + a_col] = av[i];
        *reinterpret_cast<d2v*>(&bs[b_k * LS + b_c]) = d2v{bv[0], bv[1]};
        *reinterpret_cast<d2v*>(&bs[b_k * LS + b_c + 2]) = d2v{bv[2], bv[3]};
        __syncthreads();
        if (k0 + TK < K) {
#pragma unroll
            for (int i = 0; i < PER; ++i) { av[i] = a_src[k0 + TK + i]; bv[i] = b_src[(size_t)(k0 + TK) * ldb + i]; }
        }
#pragma unroll
        for (int kk = 0; kk < TK; kk += 4) {
            const int kr = (kk + (lane >> 4)) * LS;
            double a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = as[kr + ((wr * 32 + 16 * i + (lane & 15) + 2 * kk) & 63)];      // 8 (k >> 2) = 2 kk for the step's four k
                b[i] = bs[kr + wc * 32 + 16 * i + (lane & 15)];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (V == 2) buf ^= 1;
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
template <int V>
static float runv(int M, int K, int N, const double* A, const double* B, double* O) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(gemm_var<V>, dim3(N / 64, M / 64), dim3(256), 0, 0, A, K, B, N, O, N, K);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

// V4: tiles arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), two LDS buffers, ONE barrier per k-tile, XCD-aware tile
// order.  The DMA writes lane-linearly (wave base + 16 lane), so the bank swizzle goes on the per-lane SOURCE address and on the fragment reads:
//   A buffer [64 rows][8 slots of 2 k]: slot p of row r holds k-pair p ^ ((r >> 1) & 7)   (a fragment read: 16 rows x 2 k-halves -> 32 bank pairs)
//   B buffer [16 k][32 slots of 2 cols]: slot p of row k holds column pair p ^ (8 (k & 1))   (k and k+1 of a read land on opposite bank halves)
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
__global__ __launch_bounds__(256) void gemm_dma(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, double* __restrict__ OUT, int ldo,
                                                int K) {
    constexpr int TK = 16;
    int bx, by;
    {
        const int L = blockIdx.x + gridDim.x * blockIdx.y, total = gridDim.x * gridDim.y, per = (total + 7) / 8;
        const int tix = (L & 7) * per + (L >> 3);
        if (tix >= total) return;
        bx = tix % gridDim.x; by = tix / gridDim.x;
    }
    const int row0 = by * 64, col0 = bx * 64;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    __shared__ __attribute__((aligned(16))) double As[2][64 * TK];
    __shared__ __attribute__((aligned(16))) double Bs[2][TK * 64];
    // wave w stages A rows 16 w .. 16 w + 15 (two DMA instructions of 8 rows) and B rows 4 w .. 4 w + 3 (two of 2 rows)
    const double* asrc[2];
    const double* bsrc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = 16 * w + 8 * h + (lane >> 3), p = lane & 7;
        asrc[h] = A + (size_t)(row0 + r) * lda + 2 * (p ^ ((r >> 1) & 7));
        const int k = 4 * w + 2 * h + (lane >> 5), q = lane & 31;
        bsrc[h] = B + (size_t)k * ldb + col0 + 2 * (q ^ (8 * (k & 1)));
    }
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_global_load_lds((glb_void*)(asrc[h] + k0), (lds_void*)&As[buf][(16 * w + 8 * h) * TK], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(bsrc[h] + (size_t)k0 * ldb), (lds_void*)&Bs[buf][(4 * w + 2 * h) * 64], 16, 0, 0);
        }
    };
    d4 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = d4{0, 0, 0, 0};
    // fragment read offsets (doubles): A row ar[i], B column bc[j]; per step s the k-pair 2 s + (kk >> 1), half kk & 1
    const int kk = lane >> 4, r16 = lane & 15;
    int aoff[2], afx[2], bcol[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wr * 32 + 16 * i + r16;
        aoff[i] = r * TK + (kk & 1);
        afx[i] = (r >> 1) & 7;
        const int c = wc * 32 + 16 * i + r16;
        bcol[i] = 2 * ((c >> 1) ^ (8 * (kk & 1))) + (c & 1);
    }
    stage(0, 0);
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += TK) {
        __syncthreads();                       // tile k0 has landed (the compiler drains vmcnt before the barrier); buffer buf ^ 1 is free
        if (k0 + TK < K) stage(buf ^ 1, k0 + TK);
        const double* as = As[buf];
        const double* bs = Bs[buf];
#pragma unroll
        for (int s = 0; s < TK / 4; ++s) {
            double a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = as[aoff[i] + 2 * ((2 * s + (kk >> 1)) ^ afx[i])];
                b[i] = bs[(4 * s + kk) * 64 + bcol[i]];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        buf ^= 1;
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
static float rund(int M, int K, int N, const double* A, const double* B, double* O) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(gemm_dma, dim3(N / 64, M / 64), dim3(256), 0, 0, A, K, B, N, O, N, K);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

static long diff(const double* O, const double* O2, size_t n) {
    std::vector<double> a(n), b(n);
    (void)hipMemcpy(a.data(), O, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), O2, n * 8, hipMemcpyDeviceToHost);
    long bad = 0;
    for (size_t i = 0; i < n; ++i) bad += a[i] != b[i];
    return bad;
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
    double* O2; (void)hipMalloc(&O2, (size_t)4096 * N * 8);
    for (int K : {1024, 8192}) {
        const float m0 = run<0>(4096, K, N, A, B, O);
        const float m1 = runv<1>(4096, K, N, A, B, O2); const long d1 = diff(O, O2, (size_t)4096 * N);
        const float m2 = runv<2>(4096, K, N, A, B, O2); const long d2 = diff(O, O2, (size_t)4096 * N);
        const float m4 = rund(4096, K, N, A, B, O2); const long d4_ = diff(O, O2, (size_t)4096 * N);
        printf("M = 4096, N = 1024, K = %d: LDS-DMA tiles, one barrier per k-tile, XCD-aware %8.3f ms (%5.1f, %ld differ)\n", K, m4, 2.0 * 4096 * K * N / m4 / 1e9, d4_);
        const float m3 = runv<3>(4096, K, N, A, B, O2); const long d3 = diff(O, O2, (size_t)4096 * N);
        printf("M = 4096, N = 1024, K = %d: rotated + XCD-aware tile order %8.3f ms (%5.1f, %ld differ)\n", K, m3, 2.0 * 4096 * K * N / m3 / 1e9, d3);
        printf("M = 4096, N = 1024, K = %d: variants\n  library form %8.3f ms (%5.1f)   rotated A columns + 16-byte B stores %8.3f ms (%5.1f, %ld differ)   + two buffers, one barrier %8.3f ms (%5.1f, %ld differ)\n",
               K, m0, 2.0 * 4096 * K * N / m0 / 1e9, m1, 2.0 * 4096 * K * N / m1 / 1e9, d1, m2, 2.0 * 4096 * K * N / m2 / 1e9, d2);
    }
    for (int M : {1024, 2048}) {
        const float m0 = run<0>(M, 1024, N, A, B, O), m1 = runv<1>(M, 1024, N, A, B, O2), m2 = runv<2>(M, 1024, N, A, B, O2), m3 = runv<3>(M, 1024, N, A, B, O2), m4 = rund(M, 1024, N, A, B, O2);
        printf("M = %d, N = 1024, K = 1024: %8.3f / %8.3f / %8.3f / %8.3f / dma %8.3f ms\n", M, m0, m1, m2, m3, m4);
    }
    return 0;
}
