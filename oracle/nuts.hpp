// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Restatement of src/NUTS.jl: the NUTS instantiation of the trajectory interface (leaf,
// generalised U-turn merge, multinomial / biased-progressive proposals, acceptance
// statistic) and the per-transition driver sample_tree.
#pragma once
#include <cmath>
#include <memory>
#include "hamiltonian.hpp"
#include "philox.hpp"
#include "trees.hpp"

namespace oracle {

// The per-transition random source, replacing the caller's AbstractRNG (src/NUTS.jl:232-233)
// by the ABI's counter-based stream (include/dhmc.h).
struct TransitionRng {
    ChainStream stream;
    uint32_t transition;
    uint32_t ndraws = 0;  // Exp(1) draws consumed so far (purpose 2)
    MathOps M;
    double randexp() {  // Random.randexp at src/NUTS.jl:44
        uint64_t r1, r2;
        stream.raw64(ndraws++, PURPOSE_TREE, transition, r1, r2);
        return M.randexp(r1);
    }
};

// src/NUTS.jl:59-67
struct AcceptanceStatistic {
    double log_sum_alpha = -INFINITY;
    int64_t steps = 0;
};

// src/NUTS.jl:107-118; `turning` stands for the reference's `nothing` (:134-137,:142)
struct GeneralizedTurnStatistic {
    VecP p_minus, ps_minus, p_plus, ps_plus, rho;
    bool turning = false;
};

// src/NUTS.jl:43-45
inline bool rand_bool_logprob(TransitionRng& rng, double logprob) {
    return logprob >= 0 || (rng.randexp() > -logprob);
}

// src/NUTS.jl:15-26
struct TrajectoryNUTS {
    using Z = oracle::Z;
    using Zeta = oracle::Z;
    using Tau = GeneralizedTurnStatistic;
    using Visited = AcceptanceStatistic;

    Hamiltonian H;
    double pi0;     // π₀
    double eps;     // ϵ
    double min_delta;

    double logaddexp(double a, double b) const { return H.M.logaddexp(a, b); }

    // src/NUTS.jl:28-31
    Z move(const Z& z, bool fwd) const { return leapfrog(H, *z, fwd ? eps : -eps); }

    // src/NUTS.jl:47-49
    double calculate_logprob2(bool is_doubling, double w1, double w2, double w) const {
        return biased_progressive_logprob2(is_doubling, w1, w2, w);
    }
    // src/NUTS.jl:51-53
    Zeta combine_proposals(TransitionRng& rng, const Zeta& z1, const Zeta& z2, double logprob2,
                           bool) const {
        return rand_bool_logprob(rng, logprob2) ? z2 : z1;
    }
    // src/NUTS.jl:69-71,89
    Visited combine_visited_statistics(const Visited& a, const Visited& b) const {
        return {H.M.logaddexp(a.log_sum_alpha, b.log_sum_alpha), a.steps + b.steps};
    }
    // src/NUTS.jl:130
    bool _is_turning(const Vec& psm, const Vec& psp, const Vec& rho) const {
        int D = (int)rho.size();
        return wave_dot(psm.data(), rho.data(), D) < 0 || wave_dot(psp.data(), rho.data(), D) < 0;
    }
    // src/NUTS.jl:132-139
    Tau combine_turn_statistics(const Tau& x, const Tau& y) const {
        int D = (int)x.rho->size();
        Tau out;
        Vec s(D);
        for (int i = 0; i < D; ++i) s[i] = (*x.rho)[i] + (*y.p_minus)[i];
        if (_is_turning(*x.ps_minus, *y.ps_minus, s)) { out.turning = true; return out; }
        for (int i = 0; i < D; ++i) s[i] = (*x.p_plus)[i] + (*y.rho)[i];
        if (_is_turning(*x.ps_plus, *y.ps_plus, s)) { out.turning = true; return out; }
        auto rho = std::make_shared<Vec>(D);
        for (int i = 0; i < D; ++i) (*rho)[i] = (*x.rho)[i] + (*y.rho)[i];
        if (_is_turning(*x.ps_minus, *y.ps_plus, *rho)) { out.turning = true; return out; }
        out.p_minus = x.p_minus; out.ps_minus = x.ps_minus;
        out.p_plus = y.p_plus;   out.ps_plus = y.ps_plus;
        out.rho = rho;
        return out;
    }
    // src/NUTS.jl:141-142
    bool is_turning(const Tau& t) const { return t.turning; }

    // src/NUTS.jl:148-159 with leaf_acceptance_statistic :78-80 and leaf_turn_statistic :120-123
    bool leaf(const Z& z, bool is_initial, Zeta& zeta, double& omega, Tau& tau, Visited& v) const {
        double delta = is_initial ? 0.0 : logdensity(H, *z) - pi0;
        bool isdiv = delta < min_delta;
        v = is_initial ? Visited{-INFINITY, 0} : Visited{std::fmin(delta, 0.0), 1};
        if (isdiv) return false;
        zeta = z;
        omega = delta;
        tau = Tau{z->p, z->ps, z->p, z->ps, z->p, false};
        return true;
    }
};

// src/NUTS.jl:87
inline double acceptance_rate(const MathOps& M, const AcceptanceStatistic& A) {
    return std::fmin(M.exp(A.log_sum_alpha) / (double)A.steps, 1.0);
}

// src/NUTS.jl:178-195
struct NUTS {
    int max_depth = 10;
    double min_delta = -1000.0;
};

// src/NUTS.jl:208-221
struct TreeStatisticsNUTS {
    double pi;
    int depth;
    InvalidTree termination;
    double acceptance_rate;
    int64_t steps;
    uint32_t directions;
};

// src/hamiltonian.jl:124 with the ABI's stream: call j yields coordinates e0 = (j%64)+128*(j/64)
// and e1 = e0+64.
inline VecP rand_p(const MathOps& M, const GaussianKineticEnergy& kappa, const ChainStream& s,
                   uint32_t purpose, uint32_t transition) {
    int D = kappa.D;
    Vec zn(D, 0.0);
    int ncalls = 64 * ((D + 127) / 128);
    for (int j = 0; j < ncalls; ++j) {
        int e0 = (j % 64) + 128 * (j / 64), e1 = e0 + 64;
        if (e0 >= D) continue;
        uint64_t r1, r2;
        s.raw64((uint32_t)j, purpose, transition, r1, r2);
        double z0, z1;
        M.randn2(r1, r2, &z0, &z1);
        zn[e0] = z0;
        if (e1 < D) zn[e1] = z1;
    }
    auto p = std::make_shared<Vec>(D);
    if (!kappa.dense) {
        for (int i = 0; i < D; ++i) (*p)[i] = kappa.W[i] * zn[i];
    } else {  // lower-triangular W * z, k-ordered fma chain per row
        for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            for (int k = 0; k <= i; ++k) acc = __builtin_fma(kappa.W[(size_t)i * D + k], zn[k], acc);
            (*p)[i] = acc;
        }
    }
    return p;
}

// src/NUTS.jl:232-241
inline EvaluatedLogDensity sample_tree(const NUTS& algorithm, const Hamiltonian& H,
                                       const EvaluatedLogDensity& Q, double eps,
                                       const ChainStream& stream, uint32_t transition,
                                       TreeStatisticsNUTS& stats) {
    VecP p = rand_p(H.M, *H.kappa, stream, PURPOSE_MOMENTUM, transition);
    Directions directions{stream.raw(0, PURPOSE_DIRECTIONS, transition)[0]};
    Z z = make_phasepoint(H, Q, p);
    TrajectoryNUTS trajectory{H, logdensity(H, *z), eps, algorithm.min_delta};
    TransitionRng rng{stream, transition, 0, H.M};
    auto r = sample_trajectory(rng, trajectory, z, algorithm.max_depth, directions);
    stats = TreeStatisticsNUTS{logdensity(H, *r.zeta), r.depth, r.termination,
                               acceptance_rate(H.M, r.v), r.v.steps, directions.flags};
    return r.zeta->Q;
}

}  // namespace oracle
