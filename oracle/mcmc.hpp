// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Restatement of the parts of src/mcmc.jl that drive the hot path for ONE chain: warmup state,
// random initial position, initial step size stage, the tuning/inference per-draw loops and
// the end-of-stage metric estimate.  A multi-chain context simply holds C independent chains
// (the reference leaves multi-chain runs to the caller: docs/src/worked_example.md:95-104).
#pragma once
#include <cmath>
#include <memory>
#include <vector>
#include "nuts.hpp"
#include "stepsize.hpp"

namespace oracle {

// WarmupState(Q, κ, ϵ) of src/mcmc.jl:72-79 plus what a resumable chain also needs
struct Chain {
    EvaluatedLogDensity Q;
    GaussianKineticEnergy kappa;
    double eps = NAN;  // `nothing` in the reference (mcmc.jl:130)
    DualAveragingState da;
    uint32_t transition = 0;  // number of transitions this chain has made (RNG counter)
    uint32_t status = 0;
    ChainStream stream;
    bool dense_one_product = false;   // Hamiltonian::one_product for this chain's transitions (not for the search / probes)
    // an open metric window (include/dhmc.h dhmc_metric_window_begin): running moments of the draws instead of the posterior matrix
    int64_t win_n = -1;               // draws in the window, -1: none open
    Vec win_mean, win_m2;
};

// src/mcmc.jl:108  random_position(rng, N) = rand(rng, N) .* 4 .- 2
inline VecP random_position(const ChainStream& s, int D) {
    auto q = std::make_shared<Vec>(D);
    int ncalls = 64 * ((D + 127) / 128);
    for (int j = 0; j < ncalls; ++j) {
        int e0 = (j % 64) + 128 * (j / 64), e1 = e0 + 64;
        if (e0 >= D) continue;
        uint64_t r1, r2;
        s.raw64((uint32_t)j, PURPOSE_INIT_POSITION, 0, r1, r2);
        (*q)[e0] = dhmc::u01_closed_open(r1) * 4 - 2;
        if (e1 < D) (*q)[e1] = dhmc::u01_closed_open(r2) * 4 - 2;
    }
    return q;
}

// src/mcmc.jl:129-132
inline void initialize_warmup_state(Chain& c, const Target& target, const MathOps& M,
                                    const double* q0) {
    int D = target.D;
    VecP q = q0 ? std::make_shared<const Vec>(q0, q0 + D) : random_position(c.stream, D);
    if (!c.kappa.dense) c.kappa = GaussianKineticEnergy::unit(D);   // dense contexts keep their (shared) metric
    c.status = 0;
    c.transition = 0;
    c.win_n = -1;
    Hamiltonian H{&c.kappa, &target, M, &c.status};
    c.Q = evaluate_l(H, q, true);
    c.eps = NAN;
}

// src/mcmc.jl:134-148 -> src/stepsize.jl:46-60,75-85
inline void warmup_stepsize_search(Chain& c, const Target& target, const MathOps& M,
                                   const InitialStepsizeSearch& P) {
    Hamiltonian H{&c.kappa, &target, M, &c.status};
    VecP p = rand_p(M, c.kappa, c.stream, PURPOSE_SEARCH_MOMENTUM, c.transition);
    Z z = make_phasepoint(H, c.Q, p);
    double l0 = logdensity(H, *z);
    if (!std::isfinite(l0)) {  // stepsize.jl:77-79
        c.status |= ST_NONFINITE_START_DENSITY;
        return;
    }
    auto A = [&](double eps) {
        Z z1 = leapfrog(H, *z, eps);
        return logdensity(H, *z1) - l0;
    };
    double eps;
    bool ok = find_initial_stepsize(P, A, eps);
    if (!ok) c.status |= ST_STEPSIZE_SEARCH_FAILED;
    c.eps = eps;
}

// What one transition records (mcmc.jl:272-277, 376-377)
struct DrawSink {
    double* draws = nullptr;  // [N][D] for this chain
    double* logdensities = nullptr;
    double* eps = nullptr;
    double* pi = nullptr;
    double* acceptance_rate = nullptr;
    int64_t* steps = nullptr;
    int64_t* term_left = nullptr;
    int64_t* term_right = nullptr;
    int32_t* depth = nullptr;
    uint32_t* directions = nullptr;
};

// src/mcmc.jl:271-280 (da != nullptr) and :374-379 (da == nullptr)
inline void run_transitions(Chain& c, const Target& target, const MathOps& M, const NUTS& alg,
                            int64_t N, const DualAveraging* da, bool da_init, bool da_finalize,
                            const DrawSink& out) {
    int D = target.D;
    Hamiltonian H{&c.kappa, &target, M, &c.status, c.dense_one_product};
    if (da && da_init) c.da = initial_adaptation_state(M, c.eps);  // mcmc.jl:266
    for (int64_t i = 0; i < N; ++i) {
        double eps = da ? current_eps(M, c.da) : c.eps;  // :272
        TreeStatisticsNUTS st;
        c.Q = sample_tree(alg, H, c.Q, eps, c.stream, c.transition, st);  // :274
        c.transition += 1;
        if (out.draws)
            for (int k = 0; k < D; ++k) out.draws[i * D + k] = (*c.Q.q)[k];  // :275
        if (c.win_n >= 0) {           // posterior_matrix[:, i] = Q.q (:275) goes into the window's moments (dhmc_detmath.h)
            c.win_n += 1;
            const double rn = 1.0 / (double)c.win_n;
            for (int k = 0; k < D; ++k) dhmc::dm_window_update((*c.Q.q)[k], rn, &c.win_mean[k], &c.win_m2[k]);
        }
        if (out.logdensities) out.logdensities[i] = c.Q.lq;                  // :276
        if (out.eps) out.eps[i] = eps;                                       // :273
        if (out.pi) out.pi[i] = st.pi;
        if (out.acceptance_rate) out.acceptance_rate[i] = st.acceptance_rate;
        if (out.steps) out.steps[i] = st.steps;
        if (out.term_left) out.term_left[i] = st.termination.left;
        if (out.term_right) out.term_right[i] = st.termination.right;
        if (out.depth) out.depth[i] = st.depth;
        if (out.directions) out.directions[i] = st.directions;
        if (da) c.da = adapt_stepsize(M, *da, c.da, st.acceptance_rate);     // :278
    }
    if (da && da_finalize) c.eps = final_eps(M, c.da);  // :285
}

// src/mcmc.jl:209 sample_M⁻¹(Diagonal, posterior_matrix) = Diagonal(vec(var(pm; dims=2))):
// Statistics.var with dims reduces sequentially over draws: mean = (Σ x)/n, then
// Σ (x-mean)² / (n-1).  regularize_M⁻¹ is the identity for Diagonal (mcmc.jl:223).
inline void update_metric_diag(Chain& c, const double* draws, int64_t N, int D) {
    Vec var(D);
    for (int k = 0; k < D; ++k) {
        double s = 0.0;
        for (int64_t i = 0; i < N; ++i) s = s + draws[i * D + k];
        double mean = s / (double)N;
        double ss = 0.0;
        for (int64_t i = 0; i < N; ++i) {
            double d = draws[i * D + k] - mean;
            ss = ss + d * d;
        }
        var[k] = ss / (double)(N - 1);
    }
    c.kappa = GaussianKineticEnergy::diagonal(var.data(), D);  // mcmc.jl:282
}

// The same estimate from a metric window's moments (include/dhmc.h dhmc_update_metric_diag_window): var = m2 / (n - 1).
inline void metric_window_begin(Chain& c, int D) {
    c.win_n = 0;
    c.win_mean.assign(D, 0.0);
    c.win_m2.assign(D, 0.0);
}
inline void update_metric_diag_window(Chain& c, int D) {
    Vec var(D);
    for (int k = 0; k < D; ++k) var[k] = c.win_m2[k] / (double)(c.win_n - 1);
    c.kappa = GaussianKineticEnergy::diagonal(var.data(), D);  // mcmc.jl:282
    c.win_n = -1;
}

}  // namespace oracle
