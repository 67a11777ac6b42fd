// Round engine for GEMM-shaped gradients: the logistic-regression family (BASELINE config 5's model) with a
// per-chain diagonal metric.  Same rounds as dense_rounds.hpp — every round is one leapfrog for every chain —
// but here the expensive part is the user's ∇log π (reference src/hamiltonian.jl:204,279), not M⁻¹:
//
//      K0<diag>   chains beginning a transition: p = W∘z, p♯ = M⁻¹∘p, π₀, τ₀, first half step
//      K1         q′ = q + ϵ·(M⁻¹∘pₘ)                                            (hamiltonian.jl:278)
//      K_act      the list of the chains that take a leapfrog this round (the rest have finished their transitions)
//      G_eta      H  = Q′ · Xᵀ     [A×Dpad]·[Dpad×Npad]   fp64 MFMA GEMM over the A listed rows  (η_n = x_n·β, one chain over d)
//      K_r        r_n = y_n − σ(η_n),  Σ_n [y_n η_n − log(1+e^{η_n})] per block of observations in wave order;  H ← R
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
    double* S1P;  // [nz][C]    Σ [y_n η_n − log1pexp(η_n)] over the observations of block z
    double* S1;   // [C]        ℓ's data term, the blocks added in ascending order (external models: ℓ from the callback)
    double* P;    // [nz][C][Dpad]  R·X over the observations of block z (split-K partial products)
    int nz;       // ceil(Npad / DHMC_LOGISTIC_BLOCK)
    int* act;     // [C]  chains that take a leapfrog this round
    int* act_count;
};

// K_act: act <- the chains in PH_LEAF (any order: a row's result does not depend on its place in a tile)
static __global__ __launch_bounds__(256) void rounds_active_list_kernel(RunParams P, RoundBuffers R, LogisticRound L) {
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

// K_r: link, residual and the block's share of the log-likelihood — one wave per (block of observations, listed chain).
// H <- r (0 for the padding observations); S1P[z][chain] <- the block's Σ [y η − log(1+e^η)] in wave order (lane l adds
// its observations n = l mod 64 in ascending order, then the butterfly: include/dhmc.h).
static __global__ __launch_bounds__(64) void logistic_link_kernel(RunParams P, RoundBuffers R, LogisticRound L) {
    if ((int)blockIdx.y >= *L.act_count) return;
    const int chain = L.act[blockIdx.y], lane = threadIdx.x, z = blockIdx.x;
    const int64_t N = P.tp.n, Npad = P.tp.npad;
    double* h = L.H + (size_t)chain * Npad;
    const int64_t nb = (int64_t)z * DHMC_LOGISTIC_BLOCK;
    const int64_t ne = nb + DHMC_LOGISTIC_BLOCK < Npad ? nb + DHMC_LOGISTIC_BLOCK : Npad;
    double lpart = 0.0;
    constexpr int U = 8;                                  // loads in flight per lane
    for (int64_t n0 = nb; n0 < ne; n0 += U * WAVE) {
        double eta[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t n = n0 + u * WAVE + lane;
            eta[u] = n < ne ? h[n] : 0.0;
            y[u] = n < ne ? P.tp.c[n] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t n = n0 + u * WAVE + lane;
            const double t = det_exp_v(-__builtin_fabs(eta[u]));
            const double sig = eta[u] >= 0 ? 1.0 / (1.0 + t) : t / (1.0 + t);
            const double l1pe = (eta[u] > 0 ? eta[u] : 0.0) + det_log1p_nonneg_t<dm_v>(t);
            const bool valid = n < N;
            if (n < ne) h[n] = valid ? y[u] - sig : 0.0;
            if (valid) lpart = lpart + (y[u] * eta[u] - l1pe);
        }
    }
    const double bs = wave_allreduce1(lpart);
    if (lane == 0) L.S1P[(size_t)z * P.C + (chain - P.chain_base)] = bs;
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
    const size_t zs = (size_t)P.C * P.Dpad;
    constexpr int ZU = 8;                // blocks fetched together (the adds stay in order)
    int z = 1;
    for (; z + ZU <= L.nz; z += ZU) {
        double t[ZU][NPL];
#pragma unroll
        for (int u = 0; u < ZU; ++u) ldv<NPL>(L.P + (size_t)(z + u) * zs + row, lane, t[u]);
#pragma unroll
        for (int u = 0; u < ZU; ++u)
#pragma unroll
            for (int k = 0; k < NPL; ++k) g[k] = g[k] + t[u][k];
    }
    for (; z < L.nz; ++z) {
        const double* pz = L.P + (size_t)z * zs + row;
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
    double s1 = 0.0;                                      // the blocks' sums, added in ascending order: lane z fetches block z's
    for (int z0 = 0; z0 < L.nz; z0 += WAVE) {
        const int zl = z0 + lane;
        const double v = zl < L.nz ? L.S1P[(size_t)zl * P.C + (chain - P.chain_base)] : 0.0;
        const int cnt = (L.nz - z0) < WAVE ? (L.nz - z0) : WAVE;
        for (int i = 0; i < cnt; ++i) {
            const double b = readlane_f64(v, i);
            s1 = (z0 + i == 0) ? b : s1 + b;
        }
    }
    double lq = uni_f64(s1 - 0.5 * wave_allreduce1(qq.fold(0)));
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
