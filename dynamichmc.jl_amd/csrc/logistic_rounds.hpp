// Round engine for GEMM-shaped gradients: the logistic-regression family (BASELINE config 5's model) with a
// per-chain diagonal metric.  Same rounds as dense_rounds.hpp — every round is one leapfrog for every chain —
// but here the expensive part is the user's ∇log π (reference src/hamiltonian.jl:204,279), not M⁻¹:
//
//      K0<diag>   chains beginning a transition: p = W∘z, p♯ = M⁻¹∘p, π₀, τ₀, first half step
//      K1         q′ = q + ϵ·(M⁻¹∘pₘ)                                            (hamiltonian.jl:278)
//      K_act      the list of the chains that take a leapfrog this round (the rest have finished their transitions)
//      G_eta      H  = Q′ · Xᵀ     [A×Dpad]·[Dpad×Npad]   fp64 MFMA GEMM over the A listed rows  (η_n = x_n·β, one chain over d)
//      K_r        r_n = y_n − σ(η_n),  Σ_n [y_n η_n − log(1+e^{η_n})] in wave order;  H ← R
//      G_g        P_z = R · X over the observations of block z   [A×Npad]·[Npad×Dpad], split-K fp64 MFMA GEMM: (Xᵀr)_d is,
//                 per DHMC_LOGISTIC_BLOCK observations, one chain over n ascending (include/dhmc.h)
//      K2         G = ((P₀ + P₁) + P₂) + …,  ∇ℓ = G − q′,  ℓ = S₁ − ½ q′·q′,  p′ = pₘ + ϵ/2 ∇ℓ,  p♯ = M⁻¹∘p′  (:279-280)
//      K3         the leaf and what follows (dense_rounds.hpp: identical code; p♯ is simply M⁻¹∘p here)
//
// The rows of chains that are done are not multiplied: a call ends when its slowest chain does, and with trees of
// depth 5–7 side by side half of all row-rounds would be idle rows (BASELINE config 5: mean 964 leapfrogs per chain and
// call, maximum 1996).  The blocks of the Σ_n are what makes that pay: one chain of 10⁵ fma's per output takes 1.4 ms
// whether 1024 rows are multiplied or 10.
// X is read once per round for ALL chains instead of twice per chain per leapfrog.  Every number is the one
// the wave-per-chain functor (targets.hpp LogisticT) and the oracle compute: the GEMM accumulations are the
// same ascending fma chains, pads contribute fma(0, 0, acc) = acc.
#pragma once
#include "dense_rounds.hpp"

namespace dhmc {

struct LogisticRound {
    double* H;    // [C][Npad]  η, then r
    double* T;    // [C][Npad]  per-observation log-likelihood terms
    double* S1;   // [C]        Σ_n [y_n η_n − log1pexp(η_n)]
    double* P;    // [nz][C][Dpad]  R·X over the observations of block z (split-K partial products)
    int nz;       // ceil(Npad / DHMC_LOGISTIC_BLOCK)
    int* act;     // [C]  chains that take a leapfrog this round
    int* act_count;
};

// K_act: act <- the chains in PH_LEAF (any order: a row's result does not depend on its place in a tile)
__global__ __launch_bounds__(256) void rounds_active_list_kernel(RunParams P, RoundBuffers R, LogisticRound L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.C) return;
    const int chain = P.chain_base + i;
    if (R.ts[chain].phase == PH_LEAF) L.act[atomicAdd(L.act_count, 1)] = chain;
}

// K0 for a diagonal metric: z (in cp) -> p = W∘z, p♯ = M⁻¹∘p in the places K0 reads them from.
template <int NPL>
__global__ __launch_bounds__(64) void rounds_momentum_diag_kernel(RunParams P, RoundBuffers R) {
    if ((int)blockIdx.x >= *R.list_count) return;
    const int chain = R.list[blockIdx.x], lane = threadIdx.x;
    const size_t row = (size_t)chain * P.Dpad;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        const double p = P.st.W[row + e] * R.cp[row + e];       // rand_p, diagonal W (hamiltonian.jl:80,124)
        R.tbuf[row + e] = p;
        R.cps[row + e] = P.st.minv[row + e] * p;
    }
}

// K1: q′ = q + ϵ·(M⁻¹∘pₘ)
template <int NPL>
__global__ __launch_bounds__(64) void rounds_k1_diag_kernel(RunParams P, RoundBuffers R) {
    const int chain = P.chain_base + blockIdx.x, lane = threadIdx.x;
    const TreeState& S = R.ts[chain];
    if (S.phase != PH_LEAF) return;
    const size_t row = (size_t)chain * P.Dpad;
    const double eps_s = S.eps_s;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        const double t = P.st.minv[row + e] * R.cp[row + e];
        P.st.q[row + e] = P.st.q[row + e] + eps_s * t;
    }
}

// K_r, part 1: per-observation link, residual and log-likelihood term — elementwise over [C][Npad], any order:
// H <- r, T <- y η − log(1+e^η)  (0 for the padding observations).
__global__ __launch_bounds__(256) void logistic_link_kernel(RunParams P, RoundBuffers R, LogisticRound L) {
    if ((int)blockIdx.y >= *L.act_count) return;
    const int chain = L.act[blockIdx.y];
    const int64_t N = P.tp.n, Npad = P.tp.npad;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Npad) return;
    double* h = L.H + (size_t)chain * Npad;
    double* tt = L.T + (size_t)chain * Npad;
    const double eta = h[n];
    const double t = det_exp(-__builtin_fabs(eta));
    const double sig = eta >= 0 ? 1.0 / (1.0 + t) : t / (1.0 + t);
    const double l1pe = (eta > 0 ? eta : 0.0) + det_log1p_nonneg(t);
    const bool valid = n < N;
    const double yn = P.tp.c[n];
    h[n] = valid ? yn - sig : 0.0;
    tt[n] = valid ? yn * eta - l1pe : 0.0;
}

// K_r, part 2: S₁ = Σ_n T[n] in the ABI's wave order (lane l accumulates n = l, l+64, … ascending; butterfly).
__global__ __launch_bounds__(64) void logistic_sum_kernel(RunParams P, RoundBuffers R, LogisticRound L) {
    if ((int)blockIdx.x >= *L.act_count) return;
    const int chain = L.act[blockIdx.x], lane = threadIdx.x;
    const int64_t Npad = P.tp.npad;
    const double* tt = L.T + (size_t)chain * Npad;
    // the adds are one ordered chain per lane; the loads are not: 16 of them (8 KB per wave) in flight at a time
    double lpart = 0.0;
    constexpr int U = 16;
    int64_t n0 = 0;
    for (; n0 + U * WAVE <= Npad; n0 += U * WAVE) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = tt[n0 + u * WAVE + lane];
#pragma unroll
        for (int u = 0; u < U; ++u) lpart = lpart + v[u];
    }
    for (; n0 < Npad; n0 += WAVE) lpart = lpart + tt[n0 + lane];
    const double s1 = wave_allreduce1(lpart);
    if (lane == 0) L.S1[chain] = s1;
}

// K2: gradient, log density, second half step, p♯
template <int NPL>
__global__ __launch_bounds__(64) void rounds_k2_logistic_kernel(RunParams P, RoundBuffers R, LogisticRound L) {
    const int chain = P.chain_base + blockIdx.x, lane = threadIdx.x;
    TreeState& S = R.ts[chain];
    if (S.phase != PH_LEAF) return;
    const size_t row = (size_t)chain * P.Dpad;
    const double h = S.eps_s / 2;
    double q[NPL], g[NPL], p[NPL];
    ldv<NPL>(P.st.q + row, lane, q);
    ldv<NPL>(L.P + row, lane, g);        // (Xᵀ r): the blocks' partial products, added in ascending order
    for (int z = 1; z < L.nz; ++z) {
        const double* pz = L.P + (size_t)z * P.C * P.Dpad + row;
#pragma unroll
        for (int k = 0; k < NPL; ++k) g[k] = g[k] + pz[lane + WAVE * k];
    }
    ldv<NPL>(R.cp + row, lane, p);
    LaneAcc<1, NPL> qq;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        qq.add(0, k, q[k], q[k]);
        g[k] = g[k] - q[k];
    }
    double lq = uni_f64(L.S1[chain] - 0.5 * wave_allreduce1(qq.fold(0)));
    bool pos_finite = true;
    if (!dm_isfinite(lq)) pos_finite = all_finite<LogisticT, NPL>(q);
    lq = demote_lq(lq, pos_finite, true);
#pragma unroll
    for (int k = 0; k < NPL; ++k) p[k] = p[k] + h * g[k];
    stv<NPL>(P.st.g + row, lane, g);
    stv<NPL>(R.cp + row, lane, p);
#pragma unroll
    for (int k = 0; k < NPL; ++k) R.cps[row + lane + WAVE * k] = P.st.minv[row + lane + WAVE * k] * p[k];
    if (lane == 0) {
        S.lq_leaf = lq;
        if (!pos_finite) S.status |= DHMC_ST_NONFINITE_POSITION;
    }
}

}  // namespace dhmc
