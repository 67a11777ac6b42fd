// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// The reference's own test double for the tree engine, test/test_trees.jl:28-112
// (DummyTrajectory): integer positions, move = ±1, per-node log-probability vectors so a tree
// returns its whole selection distribution deterministically (no RNG).  Used to replay
// test/test_trees.jl:114-165 (shape tests) and :171-262 (exhaustive detailed balance).
#pragma once
#include <cmath>
#include <set>
#include <vector>
#include "trees.hpp"

namespace oracle {

struct DummyTrajectory {
    using Z = int64_t;
    struct Zeta { int64_t first = 0, last = 0; std::vector<double> logp; };
    struct Tau { bool flag = false; int64_t first = 0, last = 0; };
    struct Visited { double a = 0; int64_t s = 0; };

    std::set<int64_t> turning, divergent;
    double ell_c = 3.0, ell_a = 0.1;   // testℓ(z) = -abs2(z - 3) * 0.1  (test_trees.jl:106)
    std::vector<int64_t> visited;
    int assertion_failures = 0;        // the @test lines inside the double

    double ell(int64_t z) const { double d = (double)z - ell_c; return -(d * d) * ell_a; }
    double logaddexp(double x, double y) const {
        double d = (x == y) ? 0.0 : std::fabs(x - y);
        return std::fmax(x, y) + std::log1p(std::exp(-d));
    }
    Z move(const Z& z, bool fwd) const { return z + (fwd ? 1 : -1); }  // :45
    bool is_turning(const Tau& t) {                                   // :49-53
        if (!(t.last - t.first + 1 > 1)) assertion_failures++;
        return t.flag;
    }
    Tau combine_turn_statistics(const Tau& a, const Tau& b) {         // :55-62
        if (a.last + 1 != b.first) assertion_failures++;
        return {a.flag && b.flag, a.first, b.last};
    }
    Visited combine_visited_statistics(const Visited& a, const Visited& b) const {  // :64-68
        return {a.a + b.a, a.s + b.s};
    }
    static double log1mexp(double x) {  // LogExpFunctions.log1mexp
        return x < -0.6931471805599453 ? std::log1p(-std::exp(x)) : std::log(-std::expm1(x));
    }
    template <class R>
    Zeta combine_proposals(R&, const Zeta& zeta1, const Zeta& zeta2, double logprob2, bool fwd) {  // :70-82
        double lp2 = logprob2 > 0 ? 0.0 : logprob2;
        double lp1 = logprob2 > 0 ? -INFINITY : log1mexp(lp2);
        const Zeta* z1 = &zeta1;
        const Zeta* z2 = &zeta2;
        if (!fwd) { std::swap(z1, z2); std::swap(lp1, lp2); }
        if (z1->last + 1 != z2->first) assertion_failures++;
        Zeta out;
        out.first = z1->first;
        out.last = z2->last;
        for (double p : z1->logp) out.logp.push_back(p + lp1);
        for (double p : z2->logp) out.logp.push_back(p + lp2);
        return out;
    }
    double calculate_logprob2(bool is_doubling, double w1, double w2, double w) const {  // :84-86
        return biased_progressive_logprob2(is_doubling, w1, w2, w);
    }
    bool leaf(const Z& z, bool is_initial, Zeta& zeta, double& omega, Tau& tau, Visited& v) {  // :88-103
        bool d = divergent.count(z) != 0;
        if (is_initial && d) assertion_failures++;
        double delta = ell(z);
        v = is_initial ? Visited{0.0, 0} : Visited{std::fmin(std::exp(delta), 1.0), 1};
        if (!is_initial) visited.push_back(z);
        if (d) return false;
        zeta = Zeta{z, z, {0.0}};
        omega = delta;
        tau = Tau{turning.count(z) != 0, z, z};
        return true;
    }
};

}  // namespace oracle
