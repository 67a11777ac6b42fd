// Effective sample size and R-hat of selected coordinates of the draws [C][N][D] where they lie in HBM
// (the reference's tests call MCMCDiagnosticTools.ess_rhat, test/sample-correctness_utilities.jl:40-43,
// which is not vendored: the estimator here is the one of dynamichmc.jl_amd/diagnostics.py ess_rhat — multi-chain
// autocorrelation, Geyer's initial monotone positive sequence, no rank normalisation).
//
//   ess_acov_kernel   one workgroup per (coordinate, chain): the chain's series (stride D in HBM) into LDS, its mean,
//                     and the biased autocovariances acov[t] = (1/N) Σ_n x̃[n] x̃[n+t] for every lag (fixed order)
//   ess_finish_kernel one workgroup per coordinate: W, B, var⁺, the chain-averaged autocorrelations (chains summed in
//                     ascending order: deterministic), Geyer truncation, ESS and R-hat
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dhmc {

constexpr int ESS_THREADS = 256;

__device__ __forceinline__ double ess_block_sum(double v, double* red) {
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int s = ESS_THREADS / 2; s > 0; s >>= 1) {
        if (t < s) red[t] = red[t] + red[t + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// grid (ncoords, C); dynamic LDS: N doubles
__global__ __launch_bounds__(ESS_THREADS) void ess_acov_kernel(const double* __restrict__ draws, int64_t N, int64_t D,
                                                              const int32_t* __restrict__ coords, int64_t C,
                                                              double* __restrict__ acov, double* __restrict__ means) {
    extern __shared__ double x[];
    __shared__ double red[ESS_THREADS];
    const int j = blockIdx.x;
    const int64_t c = blockIdx.y;
    const double* src = draws + (size_t)c * N * D + coords[j];
    double s = 0.0;
    for (int64_t n = threadIdx.x; n < N; n += ESS_THREADS) {
        const double v = src[(size_t)n * D];
        x[n] = v;
        s = s + v;
    }
    const double mean = ess_block_sum(s, red) / (double)N;
    for (int64_t n = threadIdx.x; n < N; n += ESS_THREADS) x[n] = x[n] - mean;
    __syncthreads();
    double* out = acov + ((size_t)j * C + c) * N;
    for (int64_t t = threadIdx.x; t < N; t += ESS_THREADS) {
        double a = 0.0;
        for (int64_t n = 0; n + t < N; ++n) a = __builtin_fma(x[n], x[n + t], a);
        out[t] = a / (double)N;
    }
    if (threadIdx.x == 0) means[(size_t)j * C + c] = mean;
}

// grid (ncoords); dynamic LDS: N doubles (the chain-averaged autocovariance)
__global__ __launch_bounds__(ESS_THREADS) void ess_finish_kernel(const double* __restrict__ acov, const double* __restrict__ means,
                                                                int64_t N, int64_t C, double* __restrict__ ess,
                                                                double* __restrict__ rhat) {
    extern __shared__ double macov[];
    __shared__ double red[ESS_THREADS];
    const int j = blockIdx.x;
    const double* a = acov + (size_t)j * C * N;
    for (int64_t t = threadIdx.x; t < N; t += ESS_THREADS) {
        double s = 0.0;
        for (int64_t c = 0; c < C; ++c) s = s + a[(size_t)c * N + t];
        macov[t] = s / (double)C;
    }
    // mean and (ddof = 1) variance of the chain means
    double s = 0.0;
    for (int64_t c = threadIdx.x; c < C; c += ESS_THREADS) s = s + means[(size_t)j * C + c];
    const double mm = ess_block_sum(s, red) / (double)C;
    double q = 0.0;
    for (int64_t c = threadIdx.x; c < C; c += ESS_THREADS) {
        const double d = means[(size_t)j * C + c] - mm;
        q = __builtin_fma(d, d, q);
    }
    const double ssq = ess_block_sum(q, red);
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double Nd = (double)N, Cd = (double)C;
    const double W = macov[0] * Nd / (Nd - 1.0);
    const double B = C > 1 ? Nd * ssq / (Cd - 1.0) : 0.0;
    const double var_plus = W * (Nd - 1.0) / Nd + B / Nd;
    auto rho = [&](int64_t t) { return t == 0 ? 1.0 : 1.0 - (W - macov[t]) / var_plus; };
    double sum = 0.0, prev = 1.0e300;
    for (int64_t k = 0; 2 * k + 1 < N; ++k) {        // Geyer: pairs rho[2k] + rho[2k+1], initial positive, monotone
        double pk = rho(2 * k) + rho(2 * k + 1);
        if (!(pk > 0.0)) break;
        pk = pk < prev ? pk : prev;
        prev = pk;
        sum = sum + pk;
    }
    double tau = -1.0 + 2.0 * sum;
    const double floor_tau = 1.0 / log10(Cd * Nd);
    tau = tau > floor_tau ? tau : floor_tau;
    ess[j] = Cd * Nd / tau;
    rhat[j] = sqrt(var_plus / W);
}

}  // namespace dhmc
