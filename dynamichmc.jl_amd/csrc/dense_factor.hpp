// Device-side construction of the dense Gaussian kinetic energy (reference src/hamiltonian.jl:73:
// GaussianKineticEnergy(M⁻¹) = GaussianKineticEnergy(M⁻¹, cholesky(inv(M⁻¹)).L)) — SURVEY.md §8 f-2: the step right
// after every adaptation window (src/mcmc.jl:281-284), on the GPU, no O(D³) host loop and no round trip of the matrix.
//
// Julia calls LAPACK there (blocked, summation order unpinned).  The ABI's definition (oracle/metric.hpp, unchanged
// since round 1) is element-wise: every entry of a factor is ONE fma chain over k ascending —
//     chol:  L_ij = (A_ij − Σ_{k<j} L_ik L_jk) / L_jj          s = A_ij;  s = fma(−L_ik, L_jk, s), k = 0..j−1
//     inv:   X_ic = (δ_ic − Σ_{c≤k<i} L_ik X_kc) / L_ii         s = δ_ic;  s = fma(−L_ik, X_kc, s), k = c..i−1
//     M = XᵀX:  M_ij = Σ_{r≥max(i,j)} X_ri X_rj                 s = 0;     s = fma(X_ri, X_rj, s),  r ascending
//     S = Symmetric(M⁻¹) from the upper triangle, L₁ = chol(S), X = L₁⁻¹, M = XᵀX, W = chol(M).
// A chain over k ascending is exactly what a RIGHT-LOOKING sweep applies to an element (step k subtracts the k-th
// product from everything still unfinished), so the blocked right-looking forms below produce the same bits as the
// unblocked left-looking loops of the oracle, with all the parallelism of the trailing update:
//     for each block of NB = 32 steps:  factor the diagonal block (one workgroup) — finish the block's columns for the
//     rows below (a thread per row) — apply the block's NB products, in order, to every trailing element (tiled).
// Terms that the triangular structure makes zero (X_kc for k < c, X_ri for r < i) contribute fma(±0·x, s) = s, so the
// full-range products of the trailing kernels and of the MFMA GEMM (XᵀX) leave the chains' bits unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dhmc_detmath.h"
#include "gemm_f64_mfma.hpp"

namespace dhmc {

constexpr int DF_NB = 32;

// S[i][j] (ld-padded, zero outside D×D) = Symmetric(src) read from the upper triangle of src (row stride lsrc);
// *bad |= any non-finite entry
// Batches (per-chain metrics, several chains per launch): every kernel below takes the matrices of batch element blockIdx.z at
// stride `bs` doubles (flags at stride 2 ints); bs = 0 and a grid of depth 1 is the single-matrix form.
__global__ void df_symmetrize_kernel(const double* __restrict__ src, int lsrc, int D, double* __restrict__ S, int ld, int* __restrict__ bad,
                                     size_t src_bs = 0, size_t bs = 0) {
    src += blockIdx.z * src_bs; S += blockIdx.z * bs; bad += 2 * blockIdx.z;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)ld * ld) return;
    const int i = (int)(idx / ld), j = (int)(idx % ld);
    double v = 0.0;
    if (i < D && j < D) {
        v = (i <= j) ? src[(size_t)i * lsrc + j] : src[(size_t)j * lsrc + i];
        if (!dm_isfinite(v)) *bad = 1;
    }
    S[idx] = v;
}
__global__ void df_copy_kernel(const double* __restrict__ src, double* __restrict__ dst, size_t n, size_t bs = 0) {
    src += blockIdx.z * bs; dst += blockIdx.z * bs;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) dst[idx] = src[idx];
}
__global__ void df_identity_kernel(double* __restrict__ X, int D, int ld, size_t bs = 0) {
    X += blockIdx.z * bs;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)ld * ld) return;
    const int i = (int)(idx / ld), j = (int)(idx % ld);
    X[idx] = (i == j && i < D) ? 1.0 : 0.0;
}
// dst = srcᵀ (both [ld][ld]); 32×32 tiles through LDS
__global__ void df_transpose_kernel(const double* __restrict__ src, double* __restrict__ dst, int ld, size_t src_bs = 0, size_t dst_bs = 0) {
    src += blockIdx.z * src_bs; dst += blockIdx.z * dst_bs;
    __shared__ double t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) t[r][threadIdx.x] = src[(size_t)(by + r) * ld + bx + threadIdx.x];
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) dst[(size_t)(bx + r) * ld + by + threadIdx.x] = t[threadIdx.x][r];
}
__global__ void df_zero_upper_kernel(double* __restrict__ A, int D, int ld, size_t bs = 0) {
    A += blockIdx.z * bs;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)D * D) return;
    const int i = (int)(idx / D), j = (int)(idx % D);
    if (i < j) A[(size_t)i * ld + j] = 0.0;
}

// ---- Cholesky, block J0..J0+nb: (1) the diagonal block, one workgroup ------------------------------------------
__global__ __launch_bounds__(64) void df_chol_diag_kernel(double* __restrict__ A, int ld, int J0, int nb, int* __restrict__ notpd, size_t bs = 0) {
    A += blockIdx.z * bs; notpd += 2 * blockIdx.z;
    __shared__ double a[DF_NB][DF_NB + 1];
    const int l = threadIdx.x;                       // row of the block
    if (l < nb)
        for (int j = 0; j <= l; ++j) a[l][j] = A[(size_t)(J0 + l) * ld + J0 + j];
    __syncthreads();
    for (int k = 0; k < nb; ++k) {
        if (l == k) {
            const double d = a[k][k];
            if (!(d > 0)) *notpd = 1;                // not positive definite (the reference's cholesky throws)
            a[k][k] = __builtin_sqrt(d);
        }
        __syncthreads();
        if (l > k && l < nb) a[l][k] = a[l][k] / a[k][k];
        __syncthreads();
        if (l > k && l < nb)
            for (int j = k + 1; j <= l; ++j) a[l][j] = __builtin_fma(-a[l][k], a[j][k], a[l][j]);
        __syncthreads();
    }
    if (l < nb)
        for (int j = 0; j <= l; ++j) A[(size_t)(J0 + l) * ld + J0 + j] = a[l][j];
}
// (2) the block's columns for the rows below it: a thread per row
__global__ __launch_bounds__(128) void df_chol_panel_kernel(double* __restrict__ A, int ld, int D, int J0, int nb, size_t bs = 0) {
    A += blockIdx.z * bs;
    __shared__ double Ld[DF_NB][DF_NB + 1];
    for (int t = threadIdx.x; t < nb * nb; t += blockDim.x) Ld[t / nb][t % nb] = A[(size_t)(J0 + t / nb) * ld + J0 + t % nb];
    __syncthreads();
    const int i = J0 + nb + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    double a[DF_NB];
#pragma unroll
    for (int j = 0; j < DF_NB; ++j) a[j] = j < nb ? A[(size_t)i * ld + J0 + j] : 0.0;
#pragma unroll
    for (int k = 0; k < DF_NB; ++k) {
        if (k < nb) {
            a[k] = a[k] / Ld[k][k];
#pragma unroll
            for (int j = k + 1; j < DF_NB; ++j)
                if (j < nb) a[j] = __builtin_fma(-a[k], Ld[j][k], a[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < DF_NB; ++j)
        if (j < nb) A[(size_t)i * ld + J0 + j] = a[j];
}
// (3) trailing update: A_ij <- chain over the block's columns k ascending of fma(−A_ik, A_jk, ·), J0+nb <= j <= i < D.
// 32×32 output tile per workgroup of 256 threads (4 outputs each); tiles above the diagonal exit at once.
__global__ __launch_bounds__(256) void df_chol_trailing_kernel(double* __restrict__ A, int ld, int D, int J0, int nb, size_t bs = 0) {
    A += blockIdx.z * bs;
    const int T0 = J0 + nb;
    const int ti = T0 + blockIdx.y * 32, tj = T0 + blockIdx.x * 32;
    if (tj > ti) return;
    __shared__ double Li[32][DF_NB + 1], Lj[32][DF_NB + 1];
    for (int t = threadIdx.x; t < 32 * DF_NB; t += 256) {
        const int r = t / DF_NB, k = t % DF_NB;
        Li[r][k] = (ti + r < D && k < nb) ? A[(size_t)(ti + r) * ld + J0 + k] : 0.0;
        Lj[r][k] = (tj + r < D && k < nb) ? A[(size_t)(tj + r) * ld + J0 + k] : 0.0;
    }
    __syncthreads();
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;       // column of the tile, first of 4 rows (r0, r0+8, ...)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = r0 + 8 * u;
        const int i = ti + r, j = tj + c;
        if (i < D && j <= i) {
            double s = A[(size_t)i * ld + j];
#pragma unroll
            for (int k = 0; k < DF_NB; ++k)
                if (k < nb) s = __builtin_fma(-Li[r][k], Lj[c][k], s);
            A[(size_t)i * ld + j] = s;
        }
    }
}

// ---- X = L⁻¹ (lower triangular), block of rows K0..K0+nb; X starts as the identity -------------------------------
// (1) the block's rows for every column c < K0+nb: a thread per column
__global__ __launch_bounds__(128) void df_inv_rows_kernel(const double* __restrict__ L, double* __restrict__ X, int ld, int K0, int nb, size_t bs = 0) {
    L += blockIdx.z * bs; X += blockIdx.z * bs;
    __shared__ double Ld[DF_NB][DF_NB + 1];
    for (int t = threadIdx.x; t < nb * nb; t += blockDim.x) Ld[t / nb][t % nb] = L[(size_t)(K0 + t / nb) * ld + K0 + t % nb];
    __syncthreads();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= K0 + nb) return;
    double s[DF_NB];
#pragma unroll
    for (int r = 0; r < DF_NB; ++r) s[r] = r < nb ? X[(size_t)(K0 + r) * ld + c] : 0.0;
#pragma unroll
    for (int k = 0; k < DF_NB; ++k) {
        if (k < nb) {
            // row K0+k of X is final for c <= K0+k; for c > K0+k it is the zero of the upper triangle (s[k] = 0 there, and stays 0)
            s[k] = (c <= K0 + k) ? s[k] / Ld[k][k] : s[k];
#pragma unroll
            for (int r = k + 1; r < DF_NB; ++r)
                if (r < nb) s[r] = __builtin_fma(-Ld[r][k], s[k], s[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < DF_NB; ++r)
        if (r < nb) X[(size_t)(K0 + r) * ld + c] = s[r];
}
// (2) trailing update: X_ic <- chain over the block's rows k ascending of fma(−L_i,K0+k, X_K0+k,c, ·), i >= K0+nb, c < K0+nb
__global__ __launch_bounds__(256) void df_inv_trailing_kernel(const double* __restrict__ L, double* __restrict__ X, int ld, int D, int K0, int nb,
                                                              size_t bs = 0) {
    L += blockIdx.z * bs; X += blockIdx.z * bs;
    const int ti = K0 + nb + blockIdx.y * 32, tc = blockIdx.x * 32;
    __shared__ double Li[32][DF_NB + 1], Xk[DF_NB][33];
    for (int t = threadIdx.x; t < 32 * DF_NB; t += 256) {
        const int r = t / DF_NB, k = t % DF_NB;
        Li[r][k] = (ti + r < D && k < nb) ? L[(size_t)(ti + r) * ld + K0 + k] : 0.0;
    }
    for (int t = threadIdx.x; t < DF_NB * 32; t += 256) {
        const int k = t / 32, c = t % 32;
        Xk[k][c] = (k < nb && tc + c < K0 + nb) ? X[(size_t)(K0 + k) * ld + tc + c] : 0.0;
    }
    __syncthreads();
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = r0 + 8 * u;
        const int i = ti + r, col = tc + c;
        if (i < D && col < K0 + nb) {
            double s = X[(size_t)i * ld + col];
#pragma unroll
            for (int k = 0; k < DF_NB; ++k)
                if (k < nb) s = __builtin_fma(-Li[r][k], Xk[k][c], s);
            X[(size_t)i * ld + col] = s;
        }
    }
}

// M = XᵀX for a batch of matrices: M_ij = fma chain over r = 0 .. ld−1 ascending from 0 — the operations of the MFMA product
// launch_gemm(Xᵀ, X) of the single-matrix path, one output per thread (32×32 tiles, X tiles through LDS).
__global__ __launch_bounds__(256) void df_xtx_kernel(const double* __restrict__ X, double* __restrict__ M, int ld, size_t bs) {
    X += blockIdx.z * bs; M += blockIdx.z * bs;
    __shared__ double Xi[32][33], Xj[32][33];
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < ld; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + 8 * u;
            Xi[r][c] = X[(size_t)(k0 + r) * ld + i0 + c];
            Xj[r][c] = X[(size_t)(k0 + r) * ld + j0 + c];
        }
        __syncthreads();
        for (int k = 0; k < 32; ++k)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_fma(Xi[k][r0 + 8 * u], Xj[k][c], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) M[(size_t)(i0 + r0 + 8 * u) * ld + j0 + c] = acc[u];
}
// where the flags of a batch element are clear: its S and Wᵀ become the chain's metric (per-chain dense contexts)
__global__ void df_commit_kernel(const double* __restrict__ S, const double* __restrict__ WT, const int* __restrict__ flags, double* __restrict__ Minv,
                                 double* __restrict__ WTout, size_t n) {
    const int b = blockIdx.z;
    if (flags[2 * b] || flags[2 * b + 1]) return;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    Minv[(size_t)b * n + idx] = S[(size_t)b * n + idx];
    WTout[(size_t)b * n + idx] = WT[(size_t)b * n + idx];
}

// In-place lower Cholesky of the symmetric [D×D] matrix in A (ld-padded); the strict upper triangle is zeroed.
// *notpd (device int, zeroed by the caller) is set if a pivot is not positive.  B matrices at stride bs (flags at stride 2).
inline void df_cholesky(double* A, int D, int ld, int* notpd, hipStream_t s, int B = 1, size_t bs = 0) {
    const unsigned Bz = (unsigned)B;
    for (int J0 = 0; J0 < D; J0 += DF_NB) {
        const int nb = D - J0 < DF_NB ? D - J0 : DF_NB;
        hipLaunchKernelGGL(df_chol_diag_kernel, dim3(1, 1, Bz), dim3(64), 0, s, A, ld, J0, nb, notpd, bs);
        const int below = D - J0 - nb;
        if (below > 0) {
            hipLaunchKernelGGL(df_chol_panel_kernel, dim3((below + 127) / 128, 1, Bz), dim3(128), 0, s, A, ld, D, J0, nb, bs);
            const int nt = (below + 31) / 32;
            hipLaunchKernelGGL(df_chol_trailing_kernel, dim3(nt, nt, Bz), dim3(256), 0, s, A, ld, D, J0, nb, bs);
        }
    }
    hipLaunchKernelGGL(df_zero_upper_kernel, dim3((unsigned)(((size_t)D * D + 255) / 256), 1, Bz), dim3(256), 0, s, A, D, ld, bs);
}
// X = L⁻¹ for the lower-triangular [D×D] L; X is [ld][ld]
inline void df_lower_inverse(const double* L, double* X, int D, int ld, hipStream_t s, int B = 1, size_t bs = 0) {
    const unsigned Bz = (unsigned)B;
    hipLaunchKernelGGL(df_identity_kernel, dim3((unsigned)(((size_t)ld * ld + 255) / 256), 1, Bz), dim3(256), 0, s, X, D, ld, bs);
    for (int K0 = 0; K0 < D; K0 += DF_NB) {
        const int nb = D - K0 < DF_NB ? D - K0 : DF_NB;
        hipLaunchKernelGGL(df_inv_rows_kernel, dim3((K0 + nb + 127) / 128, 1, Bz), dim3(128), 0, s, L, X, ld, K0, nb, bs);
        const int below = D - K0 - nb;
        if (below > 0)
            hipLaunchKernelGGL(df_inv_trailing_kernel, dim3((K0 + nb + 31) / 32, (below + 31) / 32, Bz), dim3(256), 0, s, L, X, ld, D, K0, nb, bs);
    }
}

// Minv_out <- Symmetric(S_in) and WT_out <- Wᵀ with W Wᵀ = inv(S) (all [ld][ld], zero padded), from S_in already
// symmetric and padded on the device.  work: 3 buffers of B·ld² doubles.  flags [B][2]: [0] non-finite input (set by the caller's
// symmetrisation), [1] not positive definite.  B > 1: B matrices at stride ld² in S / Minv_out / WT_out, every step one launch
// for the whole batch (M = XᵀX by df_xtx_kernel instead of the MFMA GEMM: the same fma chains).
inline void df_dense_metric(const double* S, double* Minv_out, double* WT_out, int D, int ld, double* work, int* flags, hipStream_t s, int B = 1) {
    const size_t n = (size_t)ld * ld;
    const size_t bs = B > 1 ? n : 0;
    const unsigned Bz = (unsigned)B;
    double* L1 = work;                    // chol(S), later M = XᵀX and its Cholesky factor W
    double* X = work + (size_t)B * n;
    double* XT = work + 2 * (size_t)B * n;
    const unsigned g = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(df_copy_kernel, dim3(g, 1, Bz), dim3(256), 0, s, S, L1, n, bs);
    df_cholesky(L1, D, ld, flags + 1, s, B, bs);                            // L₁ = chol(S)
    df_lower_inverse(L1, X, D, ld, s, B, bs);                               // X = L₁⁻¹
    if (B == 1) {
        hipLaunchKernelGGL(df_transpose_kernel, dim3(ld / 32, ld / 32), dim3(32, 8), 0, s, X, XT, ld, (size_t)0, (size_t)0);
        launch_gemm(XT, ld, X, ld, L1, ld, ld, ld, ld, s);                  // M = XᵀX: k-ascending fma chains (fp64 MFMA)
    } else {
        hipLaunchKernelGGL(df_xtx_kernel, dim3(ld / 32, ld / 32, Bz), dim3(256), 0, s, X, L1, ld, bs);
    }
    df_cholesky(L1, D, ld, flags + 1, s, B, bs);                            // W = chol(M)
    hipLaunchKernelGGL(df_transpose_kernel, dim3(ld / 32, ld / 32, Bz), dim3(32, 8), 0, s, L1, WT_out, ld, bs, bs);
    if (Minv_out != S) hipLaunchKernelGGL(df_copy_kernel, dim3(g, 1, Bz), dim3(256), 0, s, S, Minv_out, n, bs);
}

}  // namespace dhmc
