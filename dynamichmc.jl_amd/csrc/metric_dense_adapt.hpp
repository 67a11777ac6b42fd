// End-of-stage dense metric estimate: κ := GaussianKineticEnergy(regularize_M⁻¹(sample_M⁻¹(Symmetric, pm), λ))
// (reference src/mcmc.jl:210 `Symmetric(cov(posterior_matrix; dims=2))`, :218-222 `(1-λ)Σ + λ Diagonal(diag Σ)`,
// :281-284).  The reference estimates one matrix per chain from that chain's own draws; here M⁻¹ is shared by the
// chains of a context, so the estimate pools the draws of all chains: J = C·N rows in chain-major
// order.  With one chain it is the reference's estimator.  Order fixed by the ABI (Statistics.cov delegates to
// BLAS, unpinned): mean_i = (Σ_j x_ji)/J sequentially in j; S_ik = Σ_j fma(x_ji - mean_i, x_jk - mean_k, ·)
// sequentially in j — an fp64 MFMA product over the pooled draws; Σ = S/(J-1).
#pragma once
#include "gemm_f64_mfma.hpp"

namespace dhmc {

// (blockIdx.z: batch element — one chain of a per-chain dense context — at strides x_bs / mean_bs / out_bs doubles)
// (sums_only: leave Σ_j x_ji in `mean` — the cross-rank estimate adds the ranks' sums before dividing by the job's row count)
__global__ void pooled_mean_kernel(int D, int64_t J, const double* __restrict__ X, double* __restrict__ mean, size_t x_bs = 0, size_t mean_bs = 0,
                                   int sums_only = 0) {
    X += blockIdx.z * x_bs; mean += blockIdx.z * mean_bs;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    double s = 0.0;
    int64_t j = 0;
    for (; j + 8 <= J; j += 8) {                       // eight rows in flight; the sum itself stays sequential in j
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = X[(size_t)(j + u) * D + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s = s + x[u];
    }
    for (; j < J; ++j) s = s + X[(size_t)j * D + i];
    mean[i] = sums_only ? s : s / (double)J;
}
// the job-wide mean from the all-reduced sums: buf[0..D) = Σ over ranks of the column sums, buf[D] = Σ over ranks of the row counts
__global__ void pooled_mean_finish_kernel(int D, double* __restrict__ buf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D) buf[i] = buf[i] / buf[D];
}

// OUT[i][k] (ld = ldo, i,k < Dpad) = Σ_j (X[j][i]-mean[i])·(X[j][k]-mean[k]); X is [J][D] unpadded; columns >= D give 0.
__global__ __launch_bounds__(256) void pooled_cov_kernel(int D, int64_t J, const double* __restrict__ X,
                                                         const double* __restrict__ mean, double* __restrict__ OUT, int ldo,
                                                         size_t x_bs = 0, size_t mean_bs = 0, size_t out_bs = 0) {
    X += blockIdx.z * x_bs; mean += blockIdx.z * mean_bs; OUT += blockIdx.z * out_bs;
    const int i0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    __shared__ double As[GEMM_TK * GEMM_LDS_STRIDE];   // As[draw][i]
    __shared__ double Bs[GEMM_TK * GEMM_LDS_STRIDE];   // Bs[draw][k]
    const int lk = t >> 4, lc = (t & 15) * 4;          // draw within the tile, 4 consecutive columns
    double ma[4], mb[4];
    bool va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        va[u] = i0 + lc + u < D; vb[u] = k0 + lc + u < D;
        ma[u] = va[u] ? mean[i0 + lc + u] : 0.0;
        mb[u] = vb[u] ? mean[k0 + lc + u] : 0.0;
    }
    mfma_d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma_d4{0.0, 0.0, 0.0, 0.0};
    for (int64_t j0 = 0; j0 < J; j0 += GEMM_TK) {
        const int64_t j = j0 + lk;
        double av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {   // draws beyond J contribute (0)(0)
            av[u] = (j < J && va[u]) ? X[(size_t)j * D + i0 + lc + u] - ma[u] : 0.0;
            bv[u] = (j < J && vb[u]) ? X[(size_t)j * D + k0 + lc + u] - mb[u] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            As[lk * GEMM_LDS_STRIDE + lc + u] = av[u];
            Bs[lk * GEMM_LDS_STRIDE + lc + u] = bv[u];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GEMM_TK; kk += 4) {
            const int kr = (kk + (lane >> 4)) * GEMM_LDS_STRIDE;
            const double a0 = As[kr + wr * 32 + (lane & 15)], a1 = As[kr + wr * 32 + 16 + (lane & 15)];
            const double b0 = Bs[kr + wc * 32 + (lane & 15)], b1 = Bs[kr + wc * 32 + 16 + (lane & 15)];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* o = OUT + (size_t)(i0 + wr * 32 + a * 16 + (lane >> 4) + 4 * r) * ldo + k0 + wc * 32 + (lane & 15);
            o[0] = acc[a][0][r];
            o[16] = acc[a][1][r];
        }
}

// Σ = S/(J-1), then regularize_M⁻¹(Σ, λ) = (1-λ)Σ + λ Diagonal(diag Σ)   (mcmc.jl:218-222), in place.
__global__ void cov_regularize_kernel(int D, int ld, int64_t J, double lambda, double* __restrict__ S, size_t bs = 0) {
    S += blockIdx.z * bs;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)D * D) return;
    const int i = (int)(idx / D), k = (int)(idx % D);
    const double s = S[(size_t)i * ld + k] / (double)(J - 1);
    double v = (1 - lambda) * s;
    if (i == k) v = v + lambda * s;
    S[(size_t)i * ld + k] = v;
}

}  // namespace dhmc
