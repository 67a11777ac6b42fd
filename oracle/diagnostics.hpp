// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Restatement of the two functions of src/diagnostics.jl that call the hot path directly:
// leapfrog_trajectory (:214-227, iterator :176-186) and explore_log_acceptance_ratios (:144-152).
#pragma once
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

}  // namespace oracle
