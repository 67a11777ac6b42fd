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

// Σ_c p[c·stride] over the chains in ascending order (eight loads in flight; the sum itself stays sequential)
__device__ __forceinline__ double ess_chain_sum(const double* __restrict__ p, size_t stride, int64_t C) {
    double s = 0.0;
    int64_t c = 0;
    for (; c + 8 <= C; c += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = p[(size_t)(c + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s = s + x[u];
    }
    for (; c < C; ++c) s = s + p[(size_t)c * stride];
    return s;
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
        macov[t] = ess_chain_sum(a + t, (size_t)N, C) / (double)C;
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

// ---- series longer than the LDS holds (N > ESS_LDS_MAX_N): the same estimator, the same arithmetic order (mean: strided partial
// sums + the block tree; acov[t]: one fma chain over n ascending; chains averaged in ascending order), with the centred series
// in HBM and the autocovariances computed a CHUNK of lags at a time — Geyer's truncation stops at the first non-positive pair,
// a few dozen lags for NUTS draws, so the host stops after the chunk in which every coordinate has truncated. ------------------
constexpr int ESS_LDS_MAX_N = 7680;
constexpr int ESS_LAG_CHUNK = 1024;    // lags per chunk (even)
constexpr int ESS_SEG = 2048;          // series elements staged per pass

// grid (ncoords, C): centred series xs[(j C + c) N + n] and the chain means
__global__ __launch_bounds__(ESS_THREADS) void ess_center_kernel(const double* __restrict__ draws, int64_t N, int64_t D,
                                                                const int32_t* __restrict__ coords, int64_t C,
                                                                double* __restrict__ xs, double* __restrict__ means) {
    __shared__ double red[ESS_THREADS];
    const int j = blockIdx.x;
    const int64_t c = blockIdx.y;
    const double* src = draws + (size_t)c * N * D + coords[j];
    double s = 0.0;
    for (int64_t n = threadIdx.x; n < N; n += ESS_THREADS) s = s + src[(size_t)n * D];
    const double mean = ess_block_sum(s, red) / (double)N;
    double* out = xs + ((size_t)j * C + c) * N;
    for (int64_t n = threadIdx.x; n < N; n += ESS_THREADS) out[n] = src[(size_t)n * D] - mean;
    if (threadIdx.x == 0) means[(size_t)j * C + c] = mean;
}

// grid (ncoords·C, ESS_LAG_CHUNK / ESS_THREADS): thread -> lag t = t0 + 256 blockIdx.y + threadIdx.x; acov[(series) Lc + t − t0]
__global__ __launch_bounds__(ESS_THREADS) void ess_acov_lags_kernel(const double* __restrict__ xs, int64_t N, int64_t t0,
                                                                   double* __restrict__ acov) {
    __shared__ double wa[ESS_SEG];
    __shared__ double wb[ESS_SEG + ESS_THREADS];
    const double* x = xs + (size_t)blockIdx.x * N;
    const int64_t tb = t0 + (int64_t)blockIdx.y * ESS_THREADS;      // the block's first lag
    const int64_t t = tb + threadIdx.x;
    if (tb >= N) return;
    double a = 0.0;
    for (int64_t n0 = 0; n0 + tb < N; n0 += ESS_SEG) {
        __syncthreads();
        for (int i = threadIdx.x; i < ESS_SEG; i += ESS_THREADS) wa[i] = n0 + i < N ? x[n0 + i] : 0.0;
        for (int i = threadIdx.x; i < ESS_SEG + ESS_THREADS; i += ESS_THREADS) wb[i] = n0 + tb + i < N ? x[n0 + tb + i] : 0.0;
        __syncthreads();
        const int64_t lim = N - t - n0;                            // pairs (n, n + t) with n + t < N
        const int m = lim < ESS_SEG ? (lim > 0 ? (int)lim : 0) : ESS_SEG;
        for (int i = 0; i < m; ++i) a = __builtin_fma(wa[i], wb[i + threadIdx.x], a);
    }
    if (t < N) acov[(size_t)blockIdx.x * ESS_LAG_CHUNK + (t - t0)] = a / (double)N;
}

struct EssState { double W, var_plus, sum, prev; int64_t k; int32_t done, pad_; };

// grid (ncoords): folds the chunk [t0, t0 + ESS_LAG_CHUNK) of lags into the coordinate's state; writes ess / rhat when done
__global__ __launch_bounds__(ESS_THREADS) void ess_finish_chunk_kernel(const double* __restrict__ acov, const double* __restrict__ means,
                                                                      int64_t N, int64_t C, int64_t t0, EssState* __restrict__ state,
                                                                      double* __restrict__ ess, double* __restrict__ rhat) {
    __shared__ double macov[ESS_LAG_CHUNK];
    __shared__ double red[ESS_THREADS];
    const int j = blockIdx.x;
    EssState& S = state[j];
    if (t0 > 0 && S.done) return;
    const double* a = acov + (size_t)j * C * ESS_LAG_CHUNK;
    const int64_t nl = N - t0 < ESS_LAG_CHUNK ? N - t0 : ESS_LAG_CHUNK;
    for (int64_t t = threadIdx.x; t < nl; t += ESS_THREADS) {
        macov[t] = ess_chain_sum(a + t, (size_t)ESS_LAG_CHUNK, C) / (double)C;
    }
    double mm = 0.0, ssq = 0.0;
    if (t0 == 0) {
        double s = 0.0;
        for (int64_t c = threadIdx.x; c < C; c += ESS_THREADS) s = s + means[(size_t)j * C + c];
        mm = ess_block_sum(s, red) / (double)C;
        double q = 0.0;
        for (int64_t c = threadIdx.x; c < C; c += ESS_THREADS) {
            const double d = means[(size_t)j * C + c] - mm;
            q = __builtin_fma(d, d, q);
        }
        ssq = ess_block_sum(q, red);
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double Nd = (double)N, Cd = (double)C;
    if (t0 == 0) {
        S.W = macov[0] * Nd / (Nd - 1.0);
        const double B = C > 1 ? Nd * ssq / (Cd - 1.0) : 0.0;
        S.var_plus = S.W * (Nd - 1.0) / Nd + B / Nd;
        S.sum = 0.0; S.prev = 1.0e300; S.k = 0; S.done = 0;
    }
    const double W = S.W, var_plus = S.var_plus;
    auto rho = [&](int64_t t) { return t == 0 ? 1.0 : 1.0 - (W - macov[t - t0]) / var_plus; };
    double sum = S.sum, prev = S.prev;
    int64_t k = S.k;
    bool done = false;
    for (;; ++k) {
        if (!(2 * k + 1 < N)) { done = true; break; }
        if (2 * k + 1 >= t0 + nl) break;                 // the next pair lies in the next chunk
        double pk = rho(2 * k) + rho(2 * k + 1);
        if (!(pk > 0.0)) { done = true; break; }
        pk = pk < prev ? pk : prev;
        prev = pk;
        sum = sum + pk;
    }
    S.sum = sum; S.prev = prev; S.k = k; S.done = done ? 1 : 0;
    if (!done) return;
    double tau = -1.0 + 2.0 * sum;
    const double floor_tau = 1.0 / log10(Cd * Nd);
    tau = tau > floor_tau ? tau : floor_tau;
    ess[j] = Cd * Nd / tau;
    rhat[j] = sqrt(var_plus / W);
}

// ---- rank normalisation (bulk ESS of Vehtari, Gelman, Simpson, Carpenter, Bürkner 2021; what MCMCDiagnosticTools'
// ess_rhat computes by default for the reference's tests): each chain is split in two halves, all S = 2C·N' draws of a
// coordinate are replaced by z = Φ⁻¹((rank − 3/8)/(S + 1/4)) with average ranks for ties (NUTS repeats a draw
// whenever the initial point is selected), and the estimator above runs on z laid out as [2C][N'][1]. -------------

// keys[i] = the i-th kept draw of coordinate `coord` (i = c·2N' + n: the odd leftover draw of a chain is dropped)
__global__ void ess_gather_kernel(const double* __restrict__ draws, int64_t N, int64_t D, int32_t coord, int64_t C, int64_t N2,
                                  double* __restrict__ keys, int32_t* __restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * N2) return;
    const int64_t c = i / N2, n = i % N2;
    keys[i] = draws[((size_t)c * N + n) * D + coord];
    idx[i] = (int32_t)i;
}

// sorted keys + their original indices -> z scores at the original places (which ARE the split layout [2C][N']:
// i = c·2N' + n = (2c + n/N')·N' + n%N')
__global__ void ess_rank_kernel(const double* __restrict__ skeys, const int32_t* __restrict__ sidx, int64_t S, double* __restrict__ z) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= S) return;
    const double key = skeys[r];
    int64_t lo = r, hi = r;                                   // the run of equal keys containing r (ties are short)
    while (lo > 0 && skeys[lo - 1] == key) --lo;
    while (hi + 1 < S && skeys[hi + 1] == key) ++hi;
    const double rank = 0.5 * (double)(lo + hi) + 1.0;       // average rank, 1-based
    z[sidx[r]] = normcdfinv((rank - 0.375) / ((double)S + 0.25));
}

// ---- tail ESS (Vehtari et al. 2021, §4.3; MCMCDiagnosticTools ess(kind = :tail)): the smaller of the ESS of the indicators
// I(x <= q₀.₀₅) and I(x >= q₀.₉₅) over the split chains, q the 5 % / 95 % quantile of all S draws of the coordinate (type 7, as
// Statistics.quantile) read from the sorted keys of the rank pass. ----------------------------------------------------------
__global__ void ess_tail_indicator_kernel(const double* __restrict__ skeys, const int32_t* __restrict__ sidx, int64_t S, int upper,
                                          double* __restrict__ z) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= S) return;
    const double h = (double)(S - 1) * (upper ? 0.95 : 0.05);
    const int64_t lo = (int64_t)h;
    const double frac = h - (double)lo;
    const double a = skeys[lo], b = skeys[lo + 1 < S ? lo + 1 : lo];
    const double q = a + frac * (b - a);
    const double key = skeys[r];
    z[sidx[r]] = (upper ? key >= q : key <= q) ? 1.0 : 0.0;
}

}  // namespace dhmc
