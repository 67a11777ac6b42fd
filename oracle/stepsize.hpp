// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Restatement of src/stepsize.jl: initial step size bracketing search and the dual averaging
// state machine.
#pragma once
#include <cmath>
#include <functional>
#include "hamiltonian.hpp"

namespace oracle {

// src/stepsize.jl:23-36
struct InitialStepsizeSearch {
    double initial_eps = 0.1;
    double log_threshold = -0.22314355131420976;  // log(0.8)
    int maxiter_crossing = 400;
    bool valid() const {
        return std::isfinite(log_threshold) && log_threshold < 0 && std::isfinite(initial_eps) &&
               0 < initial_eps && maxiter_crossing >= 50;
    }
};

// src/stepsize.jl:46-60; returns false where the reference throws (:58)
template <class F>
bool find_initial_stepsize(const InitialStepsizeSearch& P, F&& A, double& eps_out) {
    double eps = P.initial_eps;
    double Ae = A(eps);
    bool dbl = Ae > P.log_threshold;
    for (int it = 0; it < P.maxiter_crossing; ++it) {
        double eps1 = dbl ? 2 * eps : eps / 2;
        double Ae1 = A(eps1);
        if (dbl ? (Ae1 < P.log_threshold) : (Ae1 > P.log_threshold)) {
            eps_out = eps1;
            return true;
        }
        eps = eps1;
    }
    eps_out = eps;
    return false;
}

// src/stepsize.jl:98-118
struct DualAveraging {
    double delta = 0.8, gamma = 0.05, kappa = 0.75;
    int t0 = 10;
    bool valid() const { return 0 < delta && delta < 1 && gamma > 0 && 0.5 < kappa && kappa <= 1 && t0 >= 0; }
};

// src/stepsize.jl:121-127
struct DualAveragingState {
    double mu = 0;
    int64_t m = 1;
    double Hbar = 0, logeps = 0, logeps_bar = 0;
};

// src/stepsize.jl:134-138
inline DualAveragingState initial_adaptation_state(const MathOps& M, double eps) {
    double logeps = M.log(eps);
    return {M.log(10.0) + logeps, 1, 0.0, logeps, 0.0};
}

// src/stepsize.jl:147-156 (m is incremented BEFORE use, so the first update uses m = 2)
inline DualAveragingState adapt_stepsize(const MathOps& M, const DualAveraging& P,
                                         DualAveragingState A, double a) {
    A.m += 1;
    double m = (double)A.m;
    A.Hbar += (P.delta - a - A.Hbar) / (m + P.t0);
    A.logeps = A.mu - std::sqrt(m) / P.gamma * A.Hbar;
    A.logeps_bar += M.pow_pos(m, -P.kappa) * (A.logeps - A.logeps_bar);
    return A;
}
inline double current_eps(const MathOps& M, const DualAveragingState& A) { return M.exp(A.logeps); }   // :163
inline double final_eps(const MathOps& M, const DualAveragingState& A) { return M.exp(A.logeps_bar); } // :170

}  // namespace oracle
