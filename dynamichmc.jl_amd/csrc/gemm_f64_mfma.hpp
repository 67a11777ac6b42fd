// fp64 MFMA GEMM for the dense kinetic energy: OUT[r][:] = A[r][:] · B for a (possibly gathered) set
// of chain rows r, with A, OUT row-major [C][ld] and B row-major [ld][ld] (ld = Dpad).
//
// This is the p·M⁻¹ contraction of the reference's dense GaussianKineticEnergy
// (calculate_p♯ / ∇kinetic_energy, src/hamiltonian.jl:110,117, and rand_p's W·z, :124) for ALL chains
// at once: one chain is one row of A, so M⁻¹ (B) is streamed once per 64-row tile instead of once per
// chain.  v_mfma_f64_16x16x4_f64 is an exact k-ordered chain of fused multiply-adds (checked on gfx950,
// tools/experiments/mfma_f64_numerics.hip), and this kernel walks k in ascending order without split-K,
// so OUT[r][i] = fma(A[r][K-1], B[K-1][i], … fma(A[r][0], B[0][i], 0)) — bit for bit the oracle's chain.
//
// Tiling: 256-thread workgroup (4 waves) -> 64 rows × 64 columns (or one wave -> 32×32 for skinny
// products); each wave a 32×32 block as 2×2 MFMA
// tiles (16 accumulator VGPR pairs); K advances 16 at a time through LDS (A tile stored k-major so both
// operand fragments are conflict-free ds_read_b64: row stride 80 doubles puts the four k-rows of one
// fragment on disjoint bank halves).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dhmc {

typedef double mfma_d4 __attribute__((ext_vector_type(4)));

constexpr int GEMM_TK = 16, GEMM_LDS_STRIDE = 80;   // metric_dense_adapt.hpp's covariance kernel uses these

// OUT[r][0..N) = Σ_k A[r][k] · B[k][0..N), k = 0..K-1 ascending, for rows r = 0..nrows-1 or the gathered rows
// row_list[0..*row_count-1].  A is [*][lda], B is [K][ldb], OUT is [*][ldo]; K a multiple of TK, N a multiple of
// the column tile.  WT = waves per side, FR = 16×16 MFMA tiles per wave side: <2,2> -> 64×64 tile (4 waves,
// each 32×32); <2,1> -> 32×32 tile (4 waves, each one MFMA tile) for skinny products.
template <int WT, int FR, int TK>
__global__ __launch_bounds__(64 * WT * WT) void gemm_rows_f64_kernel(const double* __restrict__ A, int lda,
                                                                   const double* __restrict__ B, int ldb,
                                                                   double* __restrict__ OUT, int ldo, int K, int nrows,
                                                                   const int* __restrict__ row_list,
                                                                   const int* __restrict__ row_count) {
    constexpr int WS = 16 * FR;                       // rows/cols per wave
    constexpr int TM = WS * WT, TN = WS * WT, NT = 64 * WT * WT;
    constexpr int LS = TN + 16;                       // LDS row stride: the 4 k-rows of a fragment land on disjoint bank halves
    const int count = row_list ? *row_count : nrows;
    const int row0 = blockIdx.y * TM;
    if (row0 >= count) return;
    const int col0 = blockIdx.x * TN;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w / WT, wc = w % WT;

    __shared__ double As[TK * LS];   // As[k][row]
    __shared__ double Bs[TK * LS];   // Bs[k][col]

    // global -> LDS assignment: A tile TM×TK and B tile TK×TN, (TM*TK)/NT doubles per thread each
    constexpr int PER = (TM * TK) / NT;
    constexpr int A_TPR = TK / PER;                     // threads per A row
    const int a_row = t / A_TPR, a_k = (t % A_TPR) * PER;
    int a_grow = row0 + a_row;
    a_grow = a_grow < count ? a_grow : count - 1;       // clamp (clamped rows are not stored)
    if (row_list) a_grow = row_list[a_grow];
    const double* a_src = A + (size_t)a_grow * lda + a_k;
    constexpr int B_TPR = TN / PER;                     // threads per B row (one k)
    const int b_k = t / B_TPR, b_c = (t % B_TPR) * PER;
    const double* b_src = B + (size_t)b_k * ldb + col0 + b_c;

    mfma_d4 acc[FR][FR];
#pragma unroll
    for (int i = 0; i < FR; ++i)
#pragma unroll
        for (int j = 0; j < FR; ++j) acc[i][j] = mfma_d4{0.0, 0.0, 0.0, 0.0};

    // software pipeline: the global loads of K-tile t+1 are in flight while tile t is multiplied
    double av[PER], bv[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) { av[i] = a_src[i]; bv[i] = b_src[i]; }
    for (int k0 = 0; k0 < K; k0 += TK) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            As[(a_k + i) * LS + a_row] = av[i];
            Bs[b_k * LS + b_c + i] = bv[i];
        }
        __syncthreads();
        if (k0 + TK < K) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                av[i] = a_src[k0 + TK + i];
                bv[i] = b_src[(size_t)(k0 + TK) * ldb + i];
            }
        }
#pragma unroll
        for (int kk = 0; kk < TK; kk += 4) {
            const int kr = (kk + (lane >> 4)) * LS;
            double a[FR], b[FR];
#pragma unroll
            for (int i = 0; i < FR; ++i) {
                a[i] = As[kr + wr * WS + 16 * i + (lane & 15)];
                b[i] = Bs[kr + wc * WS + 16 * i + (lane & 15)];
            }
#pragma unroll
            for (int i = 0; i < FR; ++i)
#pragma unroll
                for (int j = 0; j < FR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int i = 0; i < FR; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lrow = row0 + wr * WS + i * 16 + (lane >> 4) + 4 * r;
            if (lrow < count) {
                const int grow = row_list ? row_list[lrow] : lrow;
                double* o = OUT + (size_t)grow * ldo + col0 + wc * WS + (lane & 15);
#pragma unroll
                for (int j = 0; j < FR; ++j) o[16 * j] = acc[i][j][r];
            }
        }
}

// host launchers.  Square metric products: OUT rows <- A rows · B with K = N = ld = Dpad.
inline void launch_gemm_rows(const double* A, const double* B, double* OUT, int ld, int nrows, const int* row_list,
                             const int* row_count, hipStream_t s) {
    dim3 grid(ld / 64, (nrows + 63) / 64);
    hipLaunchKernelGGL((gemm_rows_f64_kernel<2, 2, 16>), grid, dim3(256), 0, s, A, ld, B, ld, OUT, ld, ld, nrows, row_list, row_count);
}
// General product OUT[M][N] = A[M][K] · B[K][N] (N a multiple of 32, K of 16).  64×64 tiles (4 waves × 32×32)
// when that grid fills the chip; otherwise 32×32 tiles worked by 4 waves of one 16×16 MFMA tile each, so a
// skinny product (e.g. R·X: 1024 × 10⁵ × 256) still puts a wave on every SIMD.
inline void launch_gemm(const double* A, int lda, const double* B, int ldb, double* OUT, int ldo, int M, int K, int N,
                        hipStream_t s) {
    const long tiles64 = (long)((M + 63) / 64) * (N / 64);
    if (N % 64 == 0 && tiles64 >= 512) {
        dim3 grid(N / 64, (M + 63) / 64);
        hipLaunchKernelGGL((gemm_rows_f64_kernel<2, 2, 16>), grid, dim3(256), 0, s, A, lda, B, ldb, OUT, ldo, K, M, nullptr, nullptr);
    } else {
        dim3 grid(N / 32, (M + 31) / 32);
        hipLaunchKernelGGL((gemm_rows_f64_kernel<2, 1, 64>), grid, dim3(256), 0, s, A, lda, B, ldb, OUT, ldo, K, M, nullptr, nullptr);
    }
}

}  // namespace dhmc
