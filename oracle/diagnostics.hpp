// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Restatement of the two functions of src/diagnostics.jl that call the hot path directly:
// leapfrog_trajectory (:214-227, iterator :176-186) and explore_log_acceptance_ratios (:144-152),
// and of the post-hoc tree-statistics diagnostics (:29-106) in the ABI's summation orders
// (dynamichmc.jl_amd/csrc/treestat_kernels.hpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>
#include "mcmc.hpp"

namespace oracle {

// (z, position, Δ) of src/diagnostics.jl:194-197
struct PositionInformation {
    Z z;
    int position;
    double delta;
};

// src/diagnostics.jl:214-227.  `p` null: p = rand_p (stream purpose PURPOSE_PROBE_MOMENTUM, index).
inline std::vector<PositionInformation> leapfrog_trajectory(const Chain& c, const Target& target, const MathOps& M,
                                                            double eps, int A, int B, const double* p_in,
                                                            uint32_t momentum_index, uint32_t* status) {
    Hamiltonian H{&c.kappa, &target, M, status};
    int D = target.D;
    VecP p = p_in ? std::make_shared<const Vec>(p_in, p_in + D)
                  : rand_p(M, c.kappa, c.stream, PURPOSE_PROBE_MOMENTUM, momentum_index);
    Z z0 = make_phasepoint(H, c.Q, p);
    double pi0 = logdensity(H, *z0);                                   // :221
    auto walk = [&](double e, int n, int sign) {                       // Base.iterate, :176-186
        std::vector<PositionInformation> out;
        Z z = z0;
        for (int i = 1; i <= n; ++i) {
            if (!std::isfinite(z->Q.lq)) break;                        // :179
            Z z1 = leapfrog(H, *z, e);                                 // :180
            out.push_back({z1, sign * i, logdensity(H, *z1) - pi0});   // :196
            z = z1;
        }
        return out;
    };
    auto fwd = walk(eps, B, 1);                                        // :223-224
    auto bwd = walk(-eps, -A, -1);                                     // :225
    std::vector<PositionInformation> all(bwd.rbegin(), bwd.rend());    // :226
    all.push_back({z0, 0, logdensity(H, *z0) - pi0});
    all.insert(all.end(), fwd.begin(), fwd.end());
    return all;
}

// src/diagnostics.jl:144-152 with stepsize.jl:75-85; out[m * n_eps + e]
inline void explore_log_acceptance_ratios(const Chain& c, const Target& target, const MathOps& M, const double* eps,
                                          int n_eps, int n_mom, const double* ps_in, uint32_t momentum_index,
                                          double* out, uint32_t* status) {
    Hamiltonian H{&c.kappa, &target, M, status};
    int D = target.D;
    for (int m = 0; m < n_mom; ++m) {
        VecP p = ps_in ? std::make_shared<const Vec>(ps_in + (size_t)m * D, ps_in + (size_t)(m + 1) * D)
                       : rand_p(M, c.kappa, c.stream, PURPOSE_PROBE_MOMENTUM, momentum_index + (uint32_t)m);
        Z z = make_phasepoint(H, c.Q, p);
        double l0 = logdensity(H, *z);                                 // stepsize.jl:76
        if (!std::isfinite(l0)) {                                      // :77-79 throws
            *status |= ST_NONFINITE_START_DENSITY;
            continue;
        }
        for (int e = 0; e < n_eps; ++e) {
            Z z1 = leapfrog(H, *z, eps[e]);                            // :81
            out[(size_t)m * n_eps + e] = logdensity(H, *z1) - l0;      // :82-83
        }
    }
}

// Σ x_i over i in wave order: 64 interleaved partial sums of plain adds, then the butterfly
template <class F>
inline double wave_sum(int64_t n, F term) {
    double partial[64];
    for (int l = 0; l < 64; ++l) partial[l] = 0.0;
    for (int64_t i = 0; i < n; ++i) partial[i % 64] = partial[i % 64] + term(i);
    return wave_tree(partial);
}

// EBFMI (src/diagnostics.jl:29-32): mean(abs2, diff(πs)) / var(πs)
inline double ebfmi(const double* pi, int64_t n) {
    const double mean = wave_sum(n, [&](int64_t i) { return pi[i]; }) / (double)n;
    const double ss = wave_sum(n, [&](int64_t i) { const double d = pi[i] - mean; return d * d; });
    const double ds = wave_sum(n, [&](int64_t i) {
        if (i + 1 >= n) return 0.0;
        const double e = pi[i + 1] - pi[i];
        return e * e;
    });
    return (ds / (double)(n - 1)) / (ss / (double)(n - 1));
}

struct TreeStatisticsSummary {      // src/diagnostics.jl:47-58
    int64_t N;
    double a_mean;
    double a_quantiles[5];
    int64_t max_depth, divergence, turning;
    int64_t depth_counts[33];
};

// summarize_tree_statistics (:100-106) with count_terminations (:65-82) and count_depths (:87-95), chains pooled
inline TreeStatisticsSummary summarize_tree_statistics(const double* acc, const int64_t* tl, const int64_t* tr,
                                                       const int32_t* depth, int64_t chains, int64_t n) {
    TreeStatisticsSummary S{};
    S.N = chains * n;
    std::vector<double> csum(chains);
    for (int64_t c = 0; c < chains; ++c) csum[c] = wave_sum(n, [&](int64_t i) { return acc[c * n + i]; });
    S.a_mean = wave_sum(chains, [&](int64_t c) { return csum[c]; }) / (double)S.N;
    std::vector<double> v(acc, acc + S.N);
    std::sort(v.begin(), v.end());
    const double P[5] = {0.05, 0.25, 0.5, 0.75, 0.95};           // ACCEPTANCE_QUANTILES (:35)
    for (int k = 0; k < 5; ++k) {
        if (S.N == 1) { S.a_quantiles[k] = v[0]; continue; }
        const double h = (double)(S.N - 1) * P[k];
        int64_t j = (int64_t)h;
        if (j > S.N - 2) j = S.N - 2;
        const double g = h - (double)j;
        S.a_quantiles[k] = v[j] + g * (v[j + 1] - v[j]);
    }
    for (int64_t o = 0; o < S.N; ++o) {
        if (tl[o] == 1 && tr[o] == 0) S.max_depth += 1;          // REACHED_MAX_DEPTH (trees.jl:202)
        else if (tl[o] == tr[o]) S.divergence += 1;              // is_divergent (trees.jl:195)
        else S.turning += 1;
        int d = depth[o];
        d = d < 0 ? 0 : (d > 32 ? 32 : d);
        S.depth_counts[d] += 1;
    }
    return S;
}

}  // namespace oracle
