// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Dense Gaussian kinetic energy: GaussianKineticEnergy(M⁻¹::AbstractMatrix) =
// GaussianKineticEnergy(M⁻¹, cholesky(inv(M⁻¹)).L)  (src/hamiltonian.jl:73).
// Julia delegates inv/cholesky to LAPACK, whose blocked summation order is unpinned; the ABI
// fixes plain unblocked algorithms with explicit order so that library and oracle agree bit for
// bit (the library's host code in dynamichmc.jl_amd/csrc/dense_metric.hpp restates them):
//   S  = Symmetric(M⁻¹) from the upper triangle (as Julia's Symmetric wrapper reads it),
//   L₁ = chol(S) (lower),  X = L₁⁻¹ (forward substitution),  M = Xᵀ X,  W = chol(M) (lower).
#pragma once
#include <cmath>
#include <vector>
#include "hamiltonian.hpp"

namespace oracle {

// In-place lower Cholesky of the symmetric row-major n×n matrix A (upper part ignored and
// zeroed).  Returns false if A is not positive definite.
inline bool cholesky_lower(std::vector<double>& A, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d = __builtin_fma(-A[(size_t)j * n + k], A[(size_t)j * n + k], d);
        if (!(d > 0)) return false;
        d = std::sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s = __builtin_fma(-A[(size_t)i * n + k], A[(size_t)j * n + k], s);
            A[(size_t)i * n + j] = s / d;
        }
        for (int i = 0; i < j; ++i) A[(size_t)i * n + j] = 0.0;
    }
    return true;
}

// X = L⁻¹ for lower-triangular L (row-major), by forward substitution column by column.
inline std::vector<double> lower_inverse(const std::vector<double>& L, int n) {
    std::vector<double> X((size_t)n * n, 0.0);
    for (int c = 0; c < n; ++c) {
        for (int i = c; i < n; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) s = __builtin_fma(-L[(size_t)i * n + k], X[(size_t)k * n + c], s);
            X[(size_t)i * n + c] = s / L[(size_t)i * n + i];
        }
    }
    return X;
}

inline GaussianKineticEnergy GaussianKineticEnergy::dense_from(const double* minv, int D) {
    GaussianKineticEnergy k;
    k.D = D;
    k.dense = true;
    k.Minv.resize((size_t)D * D);
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) k.Minv[(size_t)i * D + j] = (i <= j) ? minv[(size_t)i * D + j] : minv[(size_t)j * D + i];
    std::vector<double> L1 = k.Minv;
    if (!cholesky_lower(L1, D)) { k.D = -1; return k; }
    std::vector<double> X = lower_inverse(L1, D);
    std::vector<double> M((size_t)D * D, 0.0);
    for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j) {                 // M = Xᵀ X; X is lower triangular: rows k >= max(i,j)
            double s = 0.0;
            for (int r = i; r < D; ++r) s = __builtin_fma(X[(size_t)r * D + i], X[(size_t)r * D + j], s);
            M[(size_t)i * D + j] = s;
            M[(size_t)j * D + i] = s;
        }
    if (!cholesky_lower(M, D)) { k.D = -1; return k; }
    k.W = M;
    return k;
}

// Pooled dense metric estimate (see dynamichmc.jl_amd/csrc/metric_dense_adapt.hpp): draws [J][D], J = C·N rows.
// sample_M⁻¹(Symmetric, ·) (mcmc.jl:210) then regularize_M⁻¹ (mcmc.jl:218-222).  Returns D×D row-major.
inline std::vector<double> pooled_regularized_cov(const double* X, int64_t J, int D, double lambda) {
    std::vector<double> mean(D), S((size_t)D * D);
    for (int i = 0; i < D; ++i) {
        double s = 0.0;
        for (int64_t j = 0; j < J; ++j) s = s + X[(size_t)j * D + i];
        mean[i] = s / (double)J;
    }
    for (int i = 0; i < D; ++i)
        for (int k = 0; k < D; ++k) {
            double acc = 0.0;
            for (int64_t j = 0; j < J; ++j) acc = __builtin_fma(X[(size_t)j * D + i] - mean[i], X[(size_t)j * D + k] - mean[k], acc);
            double s = acc / (double)(J - 1);
            double v = (1 - lambda) * s;
            if (i == k) v = v + lambda * s;
            S[(size_t)i * D + k] = v;
        }
    return S;
}

}  // namespace oracle
