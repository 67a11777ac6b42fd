// Host-side construction of the dense Gaussian kinetic energy (reference src/hamiltonian.jl:73:
// GaussianKineticEnergy(M⁻¹) = GaussianKineticEnergy(M⁻¹, cholesky(inv(M⁻¹)).L)).
// Julia calls LAPACK there (summation order unpinned); the ABI fixes plain unblocked algorithms:
//   S = Symmetric(M⁻¹) read from the upper triangle, L₁ = chol(S), X = L₁⁻¹, M = XᵀX, W = chol(M).
// Runs once per metric update on the host (O(D³)); outputs are uploaded padded to [Dpad][Dpad].
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

namespace dhmc {

inline bool host_cholesky_lower(std::vector<double>& A, int n) {
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

// Fills S (symmetrised M⁻¹) and W (lower triangular, W Wᵀ = inv(S)); false if S is not positive definite.
inline bool host_dense_metric(const double* minv, int D, std::vector<double>& S, std::vector<double>& W) {
    S.assign((size_t)D * D, 0.0);
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) S[(size_t)i * D + j] = (i <= j) ? minv[(size_t)i * D + j] : minv[(size_t)j * D + i];
    std::vector<double> L1 = S;
    if (!host_cholesky_lower(L1, D)) return false;
    std::vector<double> X((size_t)D * D, 0.0);
    for (int c = 0; c < D; ++c)
        for (int i = c; i < D; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) s = __builtin_fma(-L1[(size_t)i * D + k], X[(size_t)k * D + c], s);
            X[(size_t)i * D + c] = s / L1[(size_t)i * D + i];
        }
    W.assign((size_t)D * D, 0.0);
    for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0.0;
            for (int r = i; r < D; ++r) s = __builtin_fma(X[(size_t)r * D + i], X[(size_t)r * D + j], s);
            W[(size_t)i * D + j] = s;
            W[(size_t)j * D + i] = s;
        }
    return host_cholesky_lower(W, D);
}

}  // namespace dhmc
