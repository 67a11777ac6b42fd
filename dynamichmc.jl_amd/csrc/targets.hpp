// Device-side log densities: the stand-in for the user's
// LogDensityProblems.logdensity_and_gradient (reference src/hamiltonian.jl:204), evaluated
// inside the leapfrog with no host round trip.  One functor per DHMC_TARGET_* family
// (include/dhmc.h); the arithmetic order of each is part of its definition and is the same as
// the CPU definition the parity tests check against.
//
// Interface (q, g are the lane's NPL slots of the padded vectors; pads hold 0):
//   double eval(q, g, lane, D)   fills g = ∇ℓ(q) and returns either the lane's partial sum of
//                                the reduction that gives ℓ (kDeferred = true; the caller
//                                batches the wave reduction with the kinetic energy's and then
//                                calls finish()), or ℓ itself (kDeferred = false).
//   kElementwise                 element e of ∇ℓ and of the summand of ℓ depend on q_e alone (and eval takes any element base
//                                as its `lane` argument): such targets can be cut into 256-coordinate blocks, one wave
//                                each (dense_rounds_k3b.hpp; the multi-wave per-draw kernels of tools/experiments/mw and /w2).
//   kRecomputeGrad               ∇ℓ is cheap enough that a stored proposal keeps only q and the
//                                gradient is re-evaluated when the proposal becomes the chain's position.
//   kPointwiseGrad               element e of ∇ℓ is a function of q_e alone at no memory cost (grad1): the kernels then
//                                never keep ∇ℓ of the current point alive between leapfrogs — one D-vector less in
//                                registers — and recompute it in the first half step.  Same bits as eval's g.
//   kFiniteLqImpliesFiniteQ      a finite ℓq is only possible at a finite position, so the position scan
//                                of evaluate_ℓ (src/hamiltonian.jl:203) is needed only when ℓq is not finite.
//   kFiniteLqImpliesFiniteGrad   a finite ℓq implies a finite gradient, so the ∇ℓ scan of evaluate_ℓ
//                                (src/hamiltonian.jl:205) cannot change the outcome and is skipped.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/dhmc.h"
#include "../../include/dhmc_detmath.h"
#include "detmath_dev.hpp"
#include "run_params.hpp"
#include "wave.hpp"

namespace dhmc {


struct StdNormalT {
    static constexpr bool kDeferred = true;
    static constexpr bool kElementwise = true;
    static constexpr bool kFiniteLqImpliesFiniteGrad = true;
    static constexpr bool kPointwiseGrad = true;
    static constexpr bool kBigDims = false;
    static constexpr bool kRecomputeGrad = true;
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    __device__ explicit StdNormalT(const TargetParams&) {}
    template <int NPL>
    __device__ __forceinline__ double eval(const double (&q)[NPL], double (&g)[NPL], int, int) const {
        LaneAcc<1, NPL> acc;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            acc.add(0, k, q[k], q[k]);
            g[k] = -q[k];
        }
        return acc.fold(0);
    }
    __device__ __forceinline__ double grad1(double qk, int) const { return -qk; }   // element e of ∇ℓ from q_e alone
    __device__ __forceinline__ double finish(double s) const { return -0.5 * s; }
};

struct DiagNormalT {
    static constexpr bool kDeferred = true;
    static constexpr bool kElementwise = true;
    static constexpr bool kFiniteLqImpliesFiniteGrad = true;
    static constexpr bool kPointwiseGrad = false;
    static constexpr bool kBigDims = false;
    static constexpr bool kRecomputeGrad = true;
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    const double* mu;
    const double* prec;
    __device__ explicit DiagNormalT(const TargetParams& p) : mu(p.a), prec(p.b) {}
    template <int NPL>
    __device__ __forceinline__ double eval(const double (&q)[NPL], double (&g)[NPL], int lane, int) const {
        LaneAcc<1, NPL> acc;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            double d = q[k] - mu[lane + WAVE * k];
            double w = prec[lane + WAVE * k] * d;
            acc.add(0, k, d, w);
            g[k] = -w;
        }
        return acc.fold(0);
    }
    __device__ __forceinline__ double finish(double s) const { return -0.5 * s; }
};

// ℓ = -1/2 q'Pq, P symmetric tridiagonal: (Pq)_i = diag_i q_i + off_{i-1} q_{i-1} + off_i q_{i+1}
struct TridiagNormalT {
    static constexpr bool kDeferred = true;
    static constexpr bool kElementwise = false;
    static constexpr bool kFiniteLqImpliesFiniteGrad = true;
    static constexpr bool kPointwiseGrad = false;
    static constexpr bool kBigDims = false;
    static constexpr bool kRecomputeGrad = true;
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    const double* diag;
    const double* off;
    __device__ explicit TridiagNormalT(const TargetParams& p) : diag(p.a), off(p.b) {}
    template <int NPL>
    __device__ __forceinline__ double eval(const double (&q)[NPL], double (&g)[NPL], int lane, int D) const {
        LaneAcc<1, NPL> acc;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            int e = lane + WAVE * k;
            // neighbours: e-1 is lane-1 of this slot (lane 63 of slot k-1 for lane 0), e+1 likewise
            double left_same = __shfl_up(q[k], 1);
            double left_prev = (k > 0) ? __shfl(q[k > 0 ? k - 1 : 0], WAVE - 1) : 0.0;
            double qm = (lane == 0) ? left_prev : left_same;
            double right_same = __shfl_down(q[k], 1);
            double right_next = (k + 1 < NPL) ? __shfl(q[k + 1 < NPL ? k + 1 : k], 0) : 0.0;
            double qp = (lane == WAVE - 1) ? right_next : right_same;
            double t = diag[e] * q[k];
            if (e > 0 && e < D) t = t + off[e - 1] * qm;
            if (e < D - 1) t = t + off[e] * qp;
            acc.add(0, k, q[k], t);
            g[k] = -t;
        }
        return acc.fold(0);
    }
    // The same density over ONE 256-coordinate block of the row, for kernels that spread a chain over a workgroup with a
    // wave per block (dense_rounds_k3b.hpp): q, g are the block's slots of the lane, e0 the coordinate of the lane's slot 0,
    // `left` / `right` the coordinates just outside the block (anything where the row ends).  Returns the lane's partial sum
    // of the block — the fma chain LaneAcc runs over that block in eval() above: same bits.
    template <int NT>
    __device__ __forceinline__ double eval_block(const double (&q)[NT], double (&g)[NT], int e0, int lane, int D, double left, double right) const {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int e = e0 + WAVE * k;
            const double left_same = __shfl_up(q[k], 1);
            const double left_prev = (k > 0) ? __shfl(q[k > 0 ? k - 1 : 0], WAVE - 1) : left;
            const double qm = (lane == 0) ? left_prev : left_same;
            const double right_same = __shfl_down(q[k], 1);
            const double right_next = (k + 1 < NT) ? __shfl(q[k + 1 < NT ? k + 1 : k], 0) : right;
            const double qp = (lane == WAVE - 1) ? right_next : right_same;
            double t = diag[e] * q[k];
            if (e > 0 && e < D) t = t + off[e - 1] * qm;
            if (e < D - 1) t = t + off[e] * qp;
            acc = __builtin_fma(q[k], t, acc);
            g[k] = -t;
        }
        return acc;
    }
    __device__ __forceinline__ double finish(double s) const { return -0.5 * s; }
};

// ℓ = -1/2 (q-μ)'P(q-μ) with a full symmetric precision P (the reference tests' multivariate_normal(μ, L),
// test/utilities.jl:64, P = (LL')⁻¹): (Pd)_i is one fma chain over k ascending, streamed from L2 per chain —
// meant for the small correlated targets of the statistical tests, not for large D.
struct DenseNormalT {
    static constexpr bool kDeferred = true;
    static constexpr bool kElementwise = false;
    static constexpr bool kFiniteLqImpliesFiniteGrad = true;
    static constexpr bool kPointwiseGrad = false;
    static constexpr bool kBigDims = false;
    static constexpr bool kRecomputeGrad = true;
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    const double* mu;
    const double* P;
    int Dpad;
    __device__ explicit DenseNormalT(const TargetParams& p) : mu(p.a), P(p.b), Dpad(p.Dpad) {}
    template <int NPL>
    __device__ __forceinline__ double eval(const double (&q)[NPL], double (&g)[NPL], int lane, int D) const {
        double d[NPL], Pd[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) d[k] = q[k] - mu[lane + WAVE * k];
        sym_matvec<NPL>(P, Dpad, D, lane, d, Pd);
        LaneAcc<1, NPL> acc;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            acc.add(0, k, d[k], Pd[k]);
            g[k] = -Pd[k];
        }
        return acc.fold(0);
    }
    __device__ __forceinline__ double finish(double s) const { return -0.5 * s; }
};

// Neal's funnel: v = q_0 ~ N(0, 3²), q_i | v ~ N(0, e^v):
//   ℓ = -v²/18 - 1/2 e^{-v} Σ_{i>=1} q_i² - (D-1)/2 v
struct FunnelT {
    static constexpr bool kDeferred = false;
    static constexpr bool kElementwise = false;
    static constexpr bool kFiniteLqImpliesFiniteGrad = true;   // |e^{-v} q_i| <= max(e^{-v}, e^{-v} q_i^2)
    static constexpr bool kPointwiseGrad = false;
    static constexpr bool kBigDims = false;
    static constexpr bool kRecomputeGrad = true;
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    __device__ explicit FunnelT(const TargetParams&) {}
    template <int NPL>
    __device__ __forceinline__ double eval(const double (&q)[NPL], double (&g)[NPL], int lane, int D) const {
        double v = readlane_f64(q[0], 0);
        double ev = det_exp_u(-v);
        LaneAcc<1, NPL> acc;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            double x = (k == 0 && lane == 0) ? 0.0 : q[k];
            acc.add(0, k, x, x);
        }
        double S = wave_allreduce1(acc.fold(0), reduce_lanes(NPL, D));
        double hd = 0.5 * (double)(D - 1);
        double hes = (0.5 * ev) * S;
        double lq = (((v * v) * (-1.0 / 18.0)) - hes) - hd * v;
#pragma unroll
        for (int k = 0; k < NPL; ++k) g[k] = -(ev * q[k]);
        if (lane == 0) g[0] = ((v * (-1.0 / 9.0)) + hes) - hd;
        return lq;
    }
    __device__ __forceinline__ double finish(double s) const { return s; }
};

// Bernoulli-logit regression with a N(0, I) prior (BASELINE config 5's model), X resident in HBM and
// shared by all chains:  η_n = x_n·β,  ℓ = Σ_n [y_n η_n - log(1+e^{η_n})] - 1/2 β·β,  ∇ℓ = Xᵀ(y-σ(η)) - β.
// Round-1 form: one wave per chain walks the observations 64 at a time — lane = observation for the
// η pass (reads Xᵀ coalesced, β_d broadcast by readlane), lane = coordinate for the Xᵀr pass (reads
// X rows coalesced, r_n broadcast).  Every chain re-reads X, so this is cache-bandwidth bound; the
// production form shares X tiles across a workgroup's chains and contracts with fp64 MFMA.
struct LogisticT {
    static constexpr bool kDeferred = false;
    static constexpr bool kElementwise = false;
    static constexpr bool kFiniteLqImpliesFiniteGrad = true;   // |y - σ| <= 1, β finite when β·β is
    static constexpr bool kPointwiseGrad = false;
    static constexpr bool kBigDims = false;
    static constexpr bool kRecomputeGrad = false;              // the gradient is the expensive part: keep it with proposals
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    const double* X;
    const double* XT;
    const double* y;
    int64_t N, Npad;
    int Dpad;
    __device__ explicit LogisticT(const TargetParams& p) : X(p.a), XT(p.b), y(p.c), N(p.n), Npad(p.npad), Dpad(p.Dpad) {}
    template <int NPL>
    __device__ __forceinline__ double eval(const double (&q)[NPL], double (&g)[NPL], int lane, int D) const {
        double gb[NPL];                                  // the running block's chains; g: the blocks folded so far
#pragma unroll
        for (int k = 0; k < NPL; ++k) { g[k] = 0.0; gb[k] = 0.0; }
        double lpart = 0.0, S1 = 0.0;
        for (int64_t n0 = 0; n0 < Npad; n0 += WAVE) {
            if (n0 != 0 && n0 % DHMC_LOGISTIC_BLOCK == 0) {      // block complete (include/dhmc.h: blocks added in ascending order)
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    g[k] = (n0 == DHMC_LOGISTIC_BLOCK) ? gb[k] : g[k] + gb[k];
                    gb[k] = 0.0;
                }
                const double bs = wave_allreduce1(lpart);
                S1 = (n0 == DHMC_LOGISTIC_BLOCK) ? bs : S1 + bs;
                lpart = 0.0;
            }
            const int64_t n = n0 + lane;
            double eta = 0.0;
#pragma unroll
            for (int s2 = 0; s2 < NPL; ++s2) {
                const int kcount = (D - WAVE * s2) < WAVE ? (D - WAVE * s2) : WAVE;
                for (int l2 = 0; l2 < kcount; ++l2) {
                    const double bd = readlane_f64(q[s2], l2);
                    eta = __builtin_fma(XT[(size_t)(WAVE * s2 + l2) * Npad + n], bd, eta);
                }
            }
            const double t = det_exp_v(-__builtin_fabs(eta));
            const double sig = eta >= 0 ? 1.0 / (1.0 + t) : t / (1.0 + t);
            const double l1pe = (eta > 0 ? eta : 0.0) + det_log1p_nonneg_t<dm_v>(t);
            const bool valid = n < N;
            const double yn = y[n];
            const double r = valid ? yn - sig : 0.0;
            lpart = lpart + (valid ? yn * eta - l1pe : 0.0);
            const int rows = (N - n0) < WAVE ? (int)(N - n0) : WAVE;
            for (int l2 = 0; l2 < rows; ++l2) {
                const double rn = readlane_f64(r, l2);
                const double* __restrict__ xrow = X + (size_t)(n0 + l2) * Dpad;
#pragma unroll
                for (int k = 0; k < NPL; ++k) gb[k] = __builtin_fma(xrow[lane + WAVE * k], rn, gb[k]);
            }
        }
        const bool one_block = Npad <= DHMC_LOGISTIC_BLOCK;
        LaneAcc<1, NPL> qq;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            qq.add(0, k, q[k], q[k]);
            const double tot = one_block ? gb[k] : g[k] + gb[k];
            g[k] = tot - q[k];
        }
        double red[2] = {lpart, qq.fold(0)};
        wave_allreduce<2>(red);
        S1 = one_block ? red[0] : S1 + red[0];
        return S1 - 0.5 * red[1];
    }
    __device__ __forceinline__ double finish(double s) const { return s; }
};

// The reference's AlwaysDivergentTest (test/test_NUTS.jl:58-73)
struct AlwaysDivergentT {
    static constexpr bool kDeferred = false;
    static constexpr bool kElementwise = false;
    static constexpr bool kFiniteLqImpliesFiniteGrad = true;
    static constexpr bool kPointwiseGrad = false;
    static constexpr bool kBigDims = false;
    static constexpr bool kRecomputeGrad = true;
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    __device__ explicit AlwaysDivergentT(const TargetParams&) {}
    template <int NPL>
    __device__ __forceinline__ double eval(const double (&q)[NPL], double (&g)[NPL], int lane, int D) const {
        bool zero = true;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            g[k] = (lane + WAVE * k < D) ? 1.0 : 0.0;
            zero = zero && (q[k] == 0.0);
        }
        return wave_all(zero) ? 0.0 : -dm_inf();
    }
    __device__ __forceinline__ double finish(double s) const { return s; }
};

// DHMC_TARGET_EXTERNAL: the density lives in the host's callback (external_rounds.hpp), never in a kernel.  The
// functor exists so that the family dispatches like the others; the library only launches round-engine kernels
// for it (their use of the functor is limited to kRecomputeGrad == false: gradients are stored, never recomputed).
struct ExternalT {
    static constexpr bool kDeferred = false;
    static constexpr bool kElementwise = false;
    static constexpr bool kFiniteLqImpliesFiniteGrad = false;
    static constexpr bool kPointwiseGrad = false;
    static constexpr bool kBigDims = true;    // served by the streaming round-engine kernels up to 64 slots per lane (D <= 4096)
    static constexpr bool kRecomputeGrad = false;
    static constexpr bool kFiniteLqImpliesFiniteQ = false;
    __device__ explicit ExternalT(const TargetParams&) {}
    template <int NPL>
    __device__ __forceinline__ double eval(const double (&)[NPL], double (&g)[NPL], int, int) const {
#pragma unroll
        for (int k = 0; k < NPL; ++k) g[k] = dm_nan();
        return dm_nan();
    }
    __device__ __forceinline__ double finish(double s) const { return s; }
};

// Can the family be evaluated block by block (a wave per 256 coordinates)?  Coordinate-wise targets through eval() with the
// block's first coordinate as the element base; the tridiagonal normal through eval_block().
template <class T>
struct BlockEval {
    static constexpr bool value = T::kElementwise;
};
template <>
struct BlockEval<TridiagNormalT> {
    static constexpr bool value = true;
};

}  // namespace dhmc
