// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Restatement of the reference's abstract tree/trajectory engine, src/trees.jl, recursive
// exactly like the original so that the DummyTrajectory known-answer tests and the exhaustive
// detailed-balance test of test/test_trees.jl can be replayed against it.
//
// A trajectory type T provides (the 7-function interface of src/trees.jl:40-121):
//   types  Z, Zeta, Tau, Visited
//   Z        move(const Z&, bool is_forward)
//   bool     is_turning(const Tau&)
//   Tau      combine_turn_statistics(const Tau&, const Tau&)      (time-ordered arguments)
//   Visited  combine_visited_statistics(const Visited&, const Visited&)
//   double   calculate_logprob2(bool is_doubling, double w1, double w2, double w)
//   Zeta     combine_proposals(Rng&, const Zeta&, const Zeta&, double logprob2, bool is_forward)
//   bool     leaf(const Z&, bool is_initial, Zeta&, double& omega, Tau&, Visited&)
//            (returns false for a divergent node, the reference's `nothing`)
//   double   logaddexp(double, double)   (LogExpFunctions.logaddexp at src/trees.jl:145)
#pragma once
#include <cstdint>
#include <utility>

namespace oracle {

// src/trees.jl:10
constexpr int MAX_DIRECTIONS_DEPTH = 32;

// src/trees.jl:19-34: bit k (LSB first) is the direction of doubling k; 1 = forward.
struct Directions {
    uint32_t flags;
};
inline bool next_direction(Directions& d) {
    bool is_forward = (d.flags & 1u) != 0;
    d.flags >>= 1;
    return is_forward;
}

// src/trees.jl:180-202
struct InvalidTree {
    int64_t left, right;
};
inline bool is_divergent(const InvalidTree& t) { return t.left == t.right; }
constexpr InvalidTree REACHED_MAX_DEPTH{1, 0};

// src/trees.jl:159-161
inline double biased_progressive_logprob2(bool bias, double w1, double w2, double w) {
    return w2 - (bias ? w1 : w);
}

template <class T>
struct Subtree {
    bool valid = false;
    InvalidTree invalid{0, 0};
    typename T::Zeta zeta{};
    double omega = 0;
    typename T::Tau tau{};
    typename T::Z zlast{};
    int64_t ilast = 0;
};

// src/trees.jl:135-141
template <class T>
typename T::Tau combine_turn_statistics_in_direction(T& traj, const typename T::Tau& t1,
                                                     const typename T::Tau& t2, bool fwd) {
    return fwd ? traj.combine_turn_statistics(t1, t2) : traj.combine_turn_statistics(t2, t1);
}

// src/trees.jl:143-149
template <class T, class R>
typename T::Zeta combine_proposals_and_logweights(R& rng, T& traj, const typename T::Zeta& z1,
                                                  const typename T::Zeta& z2, double w1,
                                                  double w2, bool fwd, bool is_doubling,
                                                  double& w) {
    w = traj.logaddexp(w1, w2);
    double logprob2 = traj.calculate_logprob2(is_doubling, w1, w2, w);
    return traj.combine_proposals(rng, z1, z2, logprob2, fwd);
}

// src/trees.jl:231-262
template <class T, class R>
Subtree<T> adjacent_tree(R& rng, T& traj, const typename T::Z& z, int64_t i, int depth,
                         bool fwd, typename T::Visited& v) {
    Subtree<T> out;
    int64_t i1 = i + (fwd ? 1 : -1);
    if (depth == 0) {
        typename T::Z z1 = traj.move(z, fwd);
        bool ok = traj.leaf(z1, false, out.zeta, out.omega, out.tau, v);
        if (!ok) {
            out.valid = false;
            out.invalid = {i1, i1};
        } else {
            out.valid = true;
            out.zlast = z1;
            out.ilast = i1;
        }
        return out;
    }
    // "left" tree
    typename T::Visited vm;
    Subtree<T> tm = adjacent_tree(rng, traj, z, i, depth - 1, fwd, vm);
    if (!tm.valid) {
        v = vm;
        return tm;
    }
    // "right" tree — visited information from left is kept even if invalid
    typename T::Visited vp;
    Subtree<T> tp = adjacent_tree(rng, traj, tm.zlast, tm.ilast, depth - 1, fwd, vp);
    v = traj.combine_visited_statistics(vm, vp);
    if (!tp.valid) return tp;
    // turning invalidates
    typename T::Tau tau = combine_turn_statistics_in_direction(traj, tm.tau, tp.tau, fwd);
    if (traj.is_turning(tau)) {
        out.valid = false;
        out.invalid = {i1, tp.ilast};
        return out;
    }
    // valid subtree, combine proposals
    out.valid = true;
    out.zeta = combine_proposals_and_logweights(rng, traj, tm.zeta, tp.zeta, tm.omega,
                                                tp.omega, fwd, false, out.omega);
    out.tau = tau;
    out.zlast = tp.zlast;
    out.ilast = tp.ilast;
    return out;
}

template <class T>
struct Sampled {
    typename T::Zeta zeta;
    typename T::Visited v;
    InvalidTree termination;
    int depth;
};

// src/trees.jl:283-319
template <class T, class R>
Sampled<T> sample_trajectory(R& rng, T& traj, const typename T::Z& z, int max_depth,
                             Directions directions) {
    typename T::Zeta zeta;
    double omega = 0;
    typename T::Tau tau;
    typename T::Visited v;
    traj.leaf(z, true, zeta, omega, tau, v);
    typename T::Z zm = z, zp = z;
    int depth = 0;
    InvalidTree termination = REACHED_MAX_DEPTH;
    int64_t im = 0, ip = 0;
    while (depth < max_depth) {
        bool fwd = next_direction(directions);
        typename T::Visited v1;
        Subtree<T> t1 = adjacent_tree(rng, traj, fwd ? zp : zm, fwd ? ip : im, depth, fwd, v1);
        v = traj.combine_visited_statistics(v, v1);
        // invalid adjacent tree: stop
        if (!t1.valid) {
            termination = t1.invalid;
            break;
        }
        // update edges and combine proposals
        if (fwd) {
            zp = t1.zlast;
            ip = t1.ilast;
        } else {
            zm = t1.zlast;
            im = t1.ilast;
        }
        // tree has doubled successfully
        double w;
        zeta = combine_proposals_and_logweights(rng, traj, zeta, t1.zeta, omega, t1.omega, fwd,
                                                true, w);
        omega = w;
        depth += 1;
        // when the combined tree is turning, stop
        tau = combine_turn_statistics_in_direction(traj, tau, t1.tau, fwd);
        if (traj.is_turning(tau)) {
            termination = {im, ip};
            break;
        }
    }
    return {zeta, v, termination, depth};
}

}  // namespace oracle
