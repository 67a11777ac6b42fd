// Family-independent helper kernels of the C ABI (the capi_*.hip translation units: `static`, every unit has its own host stubs).
#pragma once
#include "nuts_kernels.hpp"

namespace dhmc {

// Set M⁻¹ from a user array (GaussianKineticEnergy(Diagonal), hamiltonian.jl:80)
static __global__ void set_metric_diag_kernel(int D, int Dpad, int C, const double* __restrict__ src, int per_chain,
                                       double* __restrict__ minv, double* __restrict__ W) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)C * Dpad) return;
    int c = (int)(idx / Dpad), e = (int)(idx % Dpad);
    if (e < D) {
        double m = src[per_chain ? (size_t)c * D + e : (size_t)e];
        minv[idx] = m;
        W[idx] = __builtin_sqrt(1.0 / m);
    } else {
        minv[idx] = 1.0;
        W[idx] = 0.0;
    }
}

// the metric window's variance m2 / (n - 1) becomes the chain's diagonal M⁻¹ (mcmc.jl:209,282; hamiltonian.jl:80): the end of
// dhmc_update_metric_diag_window, the same assignments as metric_diag_kernel's last step
static __global__ void window_finish_kernel(int D, int Dpad, int C, const double* __restrict__ m2, int64_t n, double* __restrict__ minv,
                                            double* __restrict__ W) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)C * Dpad) return;
    int e = (int)(idx % Dpad);
    if (e < D) {
        double var = m2[idx] / (double)(n - 1);
        minv[idx] = var;
        W[idx] = __builtin_sqrt(1.0 / var);
    } else {
        minv[idx] = 1.0;
        W[idx] = 0.0;
    }
}

// flag := 1 if any element is not a finite positive number (the @argcheck of GaussianKineticEnergy, hamiltonian.jl:63)
static __global__ void check_positive_finite_kernel(const double* __restrict__ v, size_t n, int* __restrict__ flag) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n && (!(v[idx] > 0) || !dm_isfinite(v[idx]))) *flag = 1;
}

// padded [C][Dpad] <-> unpadded [C][D]
static __global__ void unpad_kernel(int D, int Dpad, int C, const double* __restrict__ src, double* __restrict__ dst) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)C * D) return;
    int c = (int)(idx / D), e = (int)(idx % D);
    dst[idx] = src[(size_t)c * Dpad + e];
}

}  // namespace dhmc
