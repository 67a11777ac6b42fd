// The packed small-D engine: SEVERAL CHAINS PER WAVEFRONT (BASELINE config 4's 30-dim funnel; any diagonal-metric chain of at
// most 64 coordinates whose family has a packed evaluator below).
//
// Why.  The wave-per-chain kernel (nuts_kernels.hpp) gives a 30-dim chain 30 of 64 lanes and runs every scalar of the tree logic
// (logaddexp, multinomial picks, the binary counter) wave-uniformly — one lane's worth of work per vector instruction — behind
// v_readlane / s_load round trips.  Here a chain owns a GROUP of L adjacent lanes (L = 1, 2, 4, 8 or 16) with CPL = 4 coordinates
// per lane: lane `sub` of the group holds coordinates 4·sub .. 4·sub + 3 of every D-vector in registers, and EVERY per-chain
// scalar (ω, the visited statistic, π, the leaf counter, the depth, the RNG counters …) is an ordinary per-lane value, replicated
// over the group's lanes — so the scalar code is plain vector code that serves 64 / L chains per instruction, and per-chain
// divergent tree depths are exec masks (the wave's ballots decide how often a merge level or a transition boundary is executed).
//
// What stays pinned (include/dhmc.h): the summation order of a dot product over at most 64 coordinates is the xor
// butterfly over the coordinates — a balanced binary tree over the coordinates in natural order; here the bottom log2(CPL) levels
// are lane-local adds and the top log2(L) levels a DPP butterfly inside the group: the same tree, the same bits (pad coordinates
// contribute +0 exactly as the pad lanes of the wave kernel do).  Random streams, scalar math (dhmc_detmath.h), the order of the
// elementwise operations and the merge order of the iterative adjacent_tree are those of nuts_run_kernel, restated for the
// packed layout; the parity suite compares the two engines and the oracle bit for bit.
//
// This header compiles under hipcc (device) AND under g++: everything that touches the machine — the group's cross-lane
// operations, LDS, atomics — reaches the body (packed_body.inc) through a small environment type, so that tests/hostsim can run
// the very same body on the CPU with L = 1 (one "lane" per chain, 32 or 64 coordinates in it) against the oracle.  That build
// is test infrastructure; nothing in the product library executes on the host.
#pragma once
#include <stdint.h>
#include "../../include/dhmc.h"
#include "../../include/dhmc_detmath.h"
#include "run_params.hpp"

#if defined(__HIPCC__) || defined(__HIP__)
#define PK_FN __device__ __forceinline__
#define PK_UNROLL _Pragma("unroll")
#define PK_LAMBDA_INLINE __attribute__((always_inline))   // a lambda of the body called from two places: its captures must stay in registers
#else
#define PK_FN inline
#define PK_UNROLL
#define PK_LAMBDA_INLINE
#endif

namespace dhmc {
namespace pk {

// purposes of the random stream (include/dhmc.h; csrc/philox_dev.hpp holds the same numbers for the wave kernels)
enum : uint32_t { PK_PURPOSE_MOMENTUM = 0, PK_PURPOSE_DIRECTIONS = 1, PK_PURPOSE_TREE = 2 };

PK_FN void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
    PK_UNROLL
    for (int r = 0; r < 10; ++r) {
        const uint64_t e0 = (uint64_t)0xD2511F53u * c0, e1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(e0 >> 32), lo0 = (uint32_t)e0, hi1 = (uint32_t)(e1 >> 32), lo1 = (uint32_t)e1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// key = (seed lo, global chain index), counter = (index, purpose, transition, seed hi)
PK_FN void stream_raw64(uint32_t k0, uint32_t k1, uint32_t seed_hi, uint32_t index, uint32_t purpose, uint32_t transition,
                        uint64_t& r1, uint64_t& r2) {
    uint32_t w[4];
    philox4x32_10(index, purpose, transition, seed_hi, k0, k1, w);
    r1 = ((uint64_t)w[1] << 32) | w[0];
    r2 = ((uint64_t)w[3] << 32) | w[2];
}

// The lane-local part of the ABI's summation tree: adjacent pairs, N a power of two.
template <int N>
struct Tree {
    static PK_FN double sum(const double* t) { return Tree<N / 2>::sum(t) + Tree<N / 2>::sum(t + N / 2); }
};
template <>
struct Tree<1> {
    static PK_FN double sum(const double* t) { return t[0]; }
};

PK_FN double joint_logdensity(double lq, double K) {  // hamiltonian.jl:251-256
    if (!dm_isfinite(lq)) return -dm_inf();
    return lq - (dm_isfinite(K) ? K : dm_inf());
}
PK_FN double demote_lq(double lq, bool pos_finite, bool grad_finite) {  // hamiltonian.jl:205-216, non-strict
    if (!pos_finite) return -dm_inf();
    const bool ok = (dm_isfinite(lq) && grad_finite) || lq == -dm_inf();
    return ok ? lq : -dm_inf();
}

// The two logaddexp's of a merge (visited statistic, ω) in one basic block when both |x - y| < 16 — the common case: the two
// softplus rows are requested together and the Horner chains interleave; each value by exactly the operations of
// det_logaddexp_t, so the bits of two separate calls (the wave kernels' det_logaddexp_pair_u, csrc/detmath_dev.hpp).
template <class Pol>
PK_FN void logaddexp_pair(double x1, double y1, double x2, double y2, double& r1, double& r2) {
    const double d1 = __builtin_fabs(x1 - y1), d2 = __builtin_fabs(x2 - y2);
    if ((d1 < 16.0) & (d2 < 16.0)) {
        const int i1 = Pol::idx((int)(d1 * 16.0)), i2 = Pol::idx((int)(d2 * 16.0));
        double c1[8], c2[8];
        Pol::row8(DM_SOFTPLUS_TBL[i1], c1);
        Pol::row8(DM_SOFTPLUS_TBL[i2], c2);
        const double t1 = d1 - (double)(2 * i1 + 1) * 0.03125, t2 = d2 - (double)(2 * i2 + 1) * 0.03125;
        const double m1 = Pol::max_nonnan(x1, y1), m2 = Pol::max_nonnan(x2, y2);
        const double s1 = Pol::template horner_row<8>(t1, c1), s2 = Pol::template horner_row<8>(t2, c2);
        r1 = m1 + s1;
        r2 = m2 + s2;
    } else {
        r1 = det_logaddexp_t<Pol>(x1, y1);
        r2 = det_logaddexp_t<Pol>(x2, y2);
    }
}

// combine_turn_statistics (NUTS.jl:132-139) of two adjacent subtrees, time-ordered: x earlier, y later, each (p₋, p₊, ρ) as the
// lane's CPL slots.  Writes ρ of the merge to rho_out (which may alias an input) and returns is_turning.  The arithmetic is
// merge_core's (nuts_kernels.hpp), slot by slot.
template <int CPL, class Grp>
PK_FN bool merge6(const double* xm, const double* xp, const double* xr, const double* ym, const double* yp, const double* yr,
                  const double* m, double* rho_out) {
    double t[6][CPL], rr[CPL];
    PK_UNROLL
    for (int k = 0; k < CPL; ++k) {
        const double s1 = xr[k] + ym[k];      // x.ρ + y.p₋      (:134)
        const double s2 = xp[k] + yr[k];      // x.p₊ + y.ρ      (:135)
        const double r = xr[k] + yr[k];       // ρ               (:136)
        const double a = m[k] * xm[k];        // x.p♯₋
        const double b = m[k] * ym[k];        // y.p♯₋
        const double c = m[k] * xp[k];        // x.p♯₊
        const double d = m[k] * yp[k];        // y.p♯₊
        t[0][k] = __builtin_fma(a, s1, 0.0);
        t[1][k] = __builtin_fma(b, s1, 0.0);
        t[2][k] = __builtin_fma(c, s2, 0.0);
        t[3][k] = __builtin_fma(d, s2, 0.0);
        t[4][k] = __builtin_fma(a, r, 0.0);
        t[5][k] = __builtin_fma(d, r, 0.0);
        rr[k] = r;
    }
    double acc[6];
    PK_UNROLL
    for (int i = 0; i < 6; ++i) acc[i] = Tree<CPL>::sum(t[i]);
    Grp::template sum_n<6>(acc);
    PK_UNROLL
    for (int k = 0; k < CPL; ++k) rho_out[k] = rr[k];
    return acc[0] < 0 || acc[1] < 0 || acc[2] < 0 || acc[3] < 0 || acc[4] < 0 || acc[5] < 0;
}

// … when both subtrees are single leaves with momenta pa (the earlier-built one) and pb: merge_leaf_leaf of nuts_kernels.hpp.
template <int CPL, class Grp>
PK_FN bool merge_leaf_leaf(const double* pa, const double* pb, const double* m, double* rho_out) {
    double t[2][CPL], rr[CPL];
    PK_UNROLL
    for (int k = 0; k < CPL; ++k) {
        const double r = pa[k] + pb[k];
        t[0][k] = __builtin_fma(m[k] * pa[k], r, 0.0);
        t[1][k] = __builtin_fma(m[k] * pb[k], r, 0.0);
        rr[k] = r;
    }
    double acc[2] = {Tree<CPL>::sum(t[0]), Tree<CPL>::sum(t[1])};
    Grp::template sum_n<2>(acc);
    PK_UNROLL
    for (int k = 0; k < CPL; ++k) rho_out[k] = rr[k];
    return acc[0] < 0 || acc[1] < 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Packed evaluators of the built-in families: ℓ(q) (complete: reduced over the group and finished) and g = ∇ℓ(q) for the lane's
// CPL slots, whose first coordinate is e0.  Same operations in the same order as the functors of targets.hpp.
// ---------------------------------------------------------------------------------------------------------------------------
template <int TARGET>
struct PackedTarget;   // only the specialisations below exist: dhmc_run keeps the wave-per-chain kernel for every other family

template <>
struct PackedTarget<DHMC_TARGET_STD_NORMAL> {
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    PK_FN explicit PackedTarget(const TargetParams&) {}
    template <int CPL, class Grp, class Pol>
    PK_FN double eval(const double (&q)[CPL], double (&g)[CPL], int, int) const {
        double t[CPL];
        PK_UNROLL
        for (int k = 0; k < CPL; ++k) {
            t[k] = __builtin_fma(q[k], q[k], 0.0);
            g[k] = -q[k];
        }
        return -0.5 * Grp::sum(Tree<CPL>::sum(t));
    }
};

template <>
struct PackedTarget<DHMC_TARGET_DIAG_NORMAL> {
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    const double* mu;
    const double* prec;
    PK_FN explicit PackedTarget(const TargetParams& p) : mu(p.a), prec(p.b) {}
    template <int CPL, class Grp, class Pol>
    PK_FN double eval(const double (&q)[CPL], double (&g)[CPL], int e0, int) const {
        double t[CPL];
        PK_UNROLL
        for (int k = 0; k < CPL; ++k) {
            const double d = q[k] - mu[e0 + k];
            const double w = prec[e0 + k] * d;
            t[k] = __builtin_fma(d, w, 0.0);
            g[k] = -w;
        }
        return -0.5 * Grp::sum(Tree<CPL>::sum(t));
    }
};

// Neal's funnel (targets.hpp FunnelT): v = q_0, ℓ = -v²/18 - 1/2 e^{-v} Σ_{i>=1} q_i² - (D-1)/2 v
template <>
struct PackedTarget<DHMC_TARGET_FUNNEL> {
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    PK_FN explicit PackedTarget(const TargetParams&) {}
    template <int CPL, class Grp, class Pol>
    PK_FN double eval(const double (&q)[CPL], double (&g)[CPL], int e0, int D) const {
        const double v = Grp::first(q[0]);
#ifdef PK_ABL_NO_EXP
        const double ev = 1.0 - v;
#else
        const double ev = det_exp_t<Pol>(-v);
#endif
        double t[CPL];
        PK_UNROLL
        for (int k = 0; k < CPL; ++k) {
            const double x = (e0 + k == 0) ? 0.0 : q[k];
            t[k] = __builtin_fma(x, x, 0.0);
        }
        const double S = Grp::sum(Tree<CPL>::sum(t));
        const double hd = 0.5 * (double)(D - 1);
        const double hes = (0.5 * ev) * S;
        const double lq = (((v * v) * (-1.0 / 18.0)) - hes) - hd * v;
        PK_UNROLL
        for (int k = 0; k < CPL; ++k) g[k] = -(ev * q[k]);
        if (e0 == 0) g[0] = ((v * (-1.0 / 9.0)) + hes) - hd;
        return lq;
    }
};

// ℓ = -1/2 q'Pq, P symmetric tridiagonal (targets.hpp TridiagNormalT): (Pq)_e = diag_e q_e + off_{e-1} q_{e-1} + off_e q_{e+1}.  A lane holds
// CPL CONSECUTIVE coordinates, so only its first and last one look at a neighbour lane of the group (Grp::prev / Grp::next; the
// group's outermost lanes read a lane of another chain there, and the conditions on e discard it exactly where the functor's do).
template <>
struct PackedTarget<DHMC_TARGET_TRIDIAG_NORMAL> {
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    const double* diag;
    const double* off;
    PK_FN explicit PackedTarget(const TargetParams& p) : diag(p.a), off(p.b) {}
    template <int CPL, class Grp, class Pol>
    PK_FN double eval(const double (&q)[CPL], double (&g)[CPL], int e0, int D) const {
        const double left = Grp::prev(q[CPL - 1]), right = Grp::next(q[0]);
        double t[CPL];
        PK_UNROLL
        for (int k = 0; k < CPL; ++k) {
            const int e = e0 + k;
            const double qm = k > 0 ? q[k > 0 ? k - 1 : 0] : left;
            const double qp = k + 1 < CPL ? q[k + 1 < CPL ? k + 1 : k] : right;
            double w = diag[e] * q[k];
            if (e > 0 && e < D) w = w + off[e - 1] * qm;
            if (e < D - 1) w = w + off[e] * qp;
            t[k] = __builtin_fma(q[k], w, 0.0);
            g[k] = -w;
        }
        return -0.5 * Grp::sum(Tree<CPL>::sum(t));
    }
};

// ℓ = -1/2 (q-μ)'P(q-μ), P symmetric and dense, [Dpad][Dpad] zero padded (targets.hpp DenseNormalT): (Pd)_i is ONE fma chain over
// k = 0 … D-1 ascending of P[k][i]·d_k (oracle/targets.hpp DenseNormal).  A lane owns CPL consecutive coordinates i and walks k through
// the group's lanes (d_k is register k mod CPL of lane k / CPL: Grp::pick), every chain of the wave reading the same row of P.
template <>
struct PackedTarget<DHMC_TARGET_DENSE_NORMAL> {
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    const double* mu;
    const double* P;
    int Dpad;
    PK_FN explicit PackedTarget(const TargetParams& p) : mu(p.a), P(p.b), Dpad(p.Dpad) {}
    template <int CPL, class Grp, class Pol>
    PK_FN double eval(const double (&q)[CPL], double (&g)[CPL], int e0, int D) const {
        double d[CPL], Pd[CPL], t[CPL];
        PK_UNROLL
        for (int k = 0; k < CPL; ++k) { d[k] = q[k] - mu[e0 + k]; Pd[k] = 0.0; }
        for (int j = 0; j * CPL < D; ++j) {
            PK_UNROLL
            for (int c = 0; c < CPL; ++c) {
                const int kk = j * CPL + c;
                if (kk < D) {                                   // (the same for every chain: the chain of a padded k is never begun)
                    const double dk = Grp::pick(d[c], j);
                    const double* row = P + (size_t)kk * Dpad + e0;
                    PK_UNROLL
                    for (int i = 0; i < CPL; ++i) Pd[i] = __builtin_fma(row[i], dk, Pd[i]);
                }
            }
        }
        PK_UNROLL
        for (int k = 0; k < CPL; ++k) {
            t[k] = __builtin_fma(d[k], Pd[k], 0.0);
            g[k] = -Pd[k];
        }
        return -0.5 * Grp::sum(Tree<CPL>::sum(t));
    }
};

// the reference's AlwaysDivergentTest (test/test_NUTS.jl:58-73; targets.hpp AlwaysDivergentT)
template <>
struct PackedTarget<DHMC_TARGET_ALWAYS_DIVERGENT> {
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    PK_FN explicit PackedTarget(const TargetParams&) {}
    template <int CPL, class Grp, class Pol>
    PK_FN double eval(const double (&q)[CPL], double (&g)[CPL], int e0, int D) const {
        bool zero = true;
        PK_UNROLL
        for (int k = 0; k < CPL; ++k) {
            g[k] = (e0 + k < D) ? 1.0 : 0.0;
            zero = zero && (q[k] == 0.0);
        }
        return Grp::all(zero) ? 0.0 : -dm_inf();
    }
};

// Lanes per chain for a chain of D coordinates at `cpl` (2 or 4) coordinates per lane; 0: not served.  Two coordinates per lane
// (16 lanes for the 30-dim funnel, 4 chains per wave) halve the vector work and the register rows of a trip — the shorter trip is
// what a launch waits for when a few chains with deep trees hold it open; four (8 lanes, 8 chains per wave) serve more chains per
// instruction — the better choice once there are more chains than the chip has SIMDs to give a wave each.
inline int lanes_per_chain(int D, int cpl) {
    const int L = (D + cpl - 1) / cpl;
    return L <= 1 ? 1 : L <= 2 ? 2 : L <= 4 ? 4 : L <= 8 ? 8 : L <= 16 ? 16 : 0;
}
inline bool dim_is_packed(int D) { return D >= 1 && D <= 64; }
inline bool family_is_packed(int target) {
    return target == DHMC_TARGET_STD_NORMAL || target == DHMC_TARGET_DIAG_NORMAL || target == DHMC_TARGET_TRIDIAG_NORMAL ||
           target == DHMC_TARGET_DENSE_NORMAL || target == DHMC_TARGET_FUNNEL || target == DHMC_TARGET_ALWAYS_DIVERGENT;
}
// LDS of one wave (bytes): six rows per chain that are touched once per doubling (64·CPL doubles per row set of the wave's 64 / L
// chains), `levels` suspended levels (1 .. levels: first, last, ρ, proposal), and four scalars per level and chain
constexpr size_t kMaxLdsPerWave = 64 * 1024;
inline size_t lds_bytes_per_level(int cpl) { return sizeof(double) * 4 * 64 * (size_t)cpl; }
inline size_t lds_bytes_per_wave(int L, int cpl, int max_depth, int levels) {
    return sizeof(double) * ((size_t)(6 + 4 * levels) * 64 * (size_t)cpl + (size_t)max_depth * 4 * (64 / L));
}

}  // namespace pk
}  // namespace dhmc
