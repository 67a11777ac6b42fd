// Post-hoc NUTS diagnostics of DynamicHMC.Diagnostics (reference src/diagnostics.jl:29-106) computed where the SoA tree
// statistics of dhmc_run lie in HBM: EBFMI per chain (:29-32), count_terminations (:65-82),
// count_depths (:87-95), and the mean / quantiles of the acceptance rates of summarize_tree_statistics (:100-106).
//
// Summation orders (Julia's mean / var are pairwise and unpinned; these are the ABI's, restated by
// oracle/diagnostics.hpp): a sum over the n draws of one chain is 64 interleaved partial sums (draw i to partial
// i mod 64, ascending, plain adds) combined by the xor butterfly; the pooled mean sums the per-chain sums the same way
// over the chains.  var is the two-pass form of Statistics.var.  Quantiles are Julia's default (type 7):
// h = (n-1)p, j = min(floor(h), n-2), x[j] + (h-j)(x[j+1]-x[j]) on the sorted pooled values (hipCUB radix sort).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wave.hpp"

namespace dhmc {

constexpr int TS_DEPTH_BINS = 33;   // depth 0 .. MAX_DIRECTIONS_DEPTH (trees.jl:10)

// counts[0..2] = (max_depth, divergence, turning); counts[3 + d] = number of trees of depth d
// grid: chains, block: 64
__global__ __launch_bounds__(64) void treestat_chain_kernel(const double* __restrict__ pi, const double* __restrict__ acc,
                                                           const int64_t* __restrict__ tl, const int64_t* __restrict__ tr,
                                                           const int32_t* __restrict__ depth, int64_t n,
                                                           double* __restrict__ ebfmi, double* __restrict__ asum,
                                                           unsigned long long* __restrict__ counts) {
    __shared__ unsigned int hist[3 + TS_DEPTH_BINS];
    const int64_t c = blockIdx.x;
    const int lane = threadIdx.x;
    if (lane < 3 + TS_DEPTH_BINS) hist[lane] = 0u;
    __syncthreads();
    const double* p = pi + (size_t)c * n;
    // mean(π)
    double s = 0.0;
    for (int64_t i = lane; i < n; i += WAVE) s = s + p[i];
    const double mean = wave_allreduce1(s) / (double)n;
    // Σ (π - mean)², Σ (π[i+1] - π[i])²
    double ss = 0.0, ds = 0.0;
    for (int64_t i = lane; i < n; i += WAVE) {
        const double d = p[i] - mean;
        ss = ss + d * d;
        if (i + 1 < n) {
            const double e = p[i + 1] - p[i];
            ds = ds + e * e;
        }
    }
    double r[2] = {ss, ds};
    wave_allreduce<2>(r);
    const double var = r[0] / (double)(n - 1);                    // Statistics.var
    const double msd = r[1] / (double)(n - 1);                    // mean(abs2, diff(πs)): n-1 differences
    // acceptance rates, terminations, depths
    double as = 0.0;
    for (int64_t i = lane; i < n; i += WAVE) {
        const size_t o = (size_t)c * n + i;
        as = as + acc[o];
        const int64_t l = tl[o], rr = tr[o];
        const int k = (l == 1 && rr == 0) ? 0 : (l == rr ? 1 : 2);   // REACHED_MAX_DEPTH / is_divergent / turning
        atomicAdd(&hist[k], 1u);
        int d = depth[o];
        d = d < 0 ? 0 : (d >= TS_DEPTH_BINS ? TS_DEPTH_BINS - 1 : d);
        atomicAdd(&hist[3 + d], 1u);
    }
    const double atot = wave_allreduce1(as);
    __syncthreads();
    if (lane < 3 + TS_DEPTH_BINS && hist[lane]) atomicAdd(&counts[lane], (unsigned long long)hist[lane]);
    if (lane == 0) {
        if (ebfmi) ebfmi[c] = msd / var;
        asum[c] = atot;
    }
}

// one wave: pooled mean of the acceptance rates and the five quantiles of the sorted values
__global__ __launch_bounds__(64) void treestat_finish_kernel(const double* __restrict__ asum, int64_t chains, int64_t total,
                                                            const double* __restrict__ sorted, double* __restrict__ out6) {
    const int lane = threadIdx.x;
    double s = 0.0;
    for (int64_t c = lane; c < chains; c += WAVE) s = s + asum[c];
    const double mean = wave_allreduce1(s) / (double)total;
    if (lane == 0) out6[0] = mean;
    if (lane < 5) {
        const double P[5] = {0.05, 0.25, 0.5, 0.75, 0.95};        // ACCEPTANCE_QUANTILES (diagnostics.jl:35)
        double q;
        if (total == 1) {
            q = sorted[0];
        } else {
            const double h = (double)(total - 1) * P[lane];
            int64_t j = (int64_t)h;                                // h >= 0: truncation is floor
            if (j > total - 2) j = total - 2;
            const double g = h - (double)j;
            const double a = sorted[j], b = sorted[j + 1];
            q = a + g * (b - a);
        }
        out6[1 + lane] = q;
    }
}

}  // namespace dhmc
