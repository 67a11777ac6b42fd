// The user's own log density on the device: DHMC_TARGET_EXTERNAL (include/dhmc.h).
//
// The reference takes ANY LogDensityProblems model (logdensity_and_gradient, src/hamiltonian.jl:204).  The built-in
// functor families of targets.hpp are evaluated inside the kernels; an external model is instead evaluated for ALL
// chains at once by a callback the host registers (dhmc_set_logdensity_callback: q [C][ld] in HBM -> ℓ [C], ∇ℓ
// [C][ld] in HBM, on the context's stream — e.g. a batched PyTorch function, dynamichmc.jl_amd/api.py
// TorchLogDensity).  That fits the round engine (dense_rounds.hpp / logistic_rounds.hpp) exactly: every round is
// one leapfrog for every chain,
//      K0<diag>  K1 (q′ = q + ϵ M⁻¹pₘ)   callback(Q′) -> (ℓ, ∇ℓ)   K2ext (evaluate_ℓ's rules, p′, p♯)   K3
// with a per-chain diagonal metric.  The kernels here are the two that touch the callback's outputs, and the
// batched initial step size search (stepsize.jl:46-85), whose trial leapfrogs need the callback as well.
#pragma once
#include "logistic_rounds.hpp"

namespace dhmc {

// All kernels here stream a chain's row slot by slot (no register arrays), so they serve up to 64 slots per lane
// (D <= 4096): external models are the one family not tied to the register-resident kernels' D <= 1024.

// evaluate_ℓ (hamiltonian.jl:202-217) on what the callback returned for one chain: the row's position must be finite
// (else the reference throws, :203), then (ℓ finite ∧ ∇ℓ finite) ∨ ℓ == -Inf keeps ℓ, anything else demotes to -Inf.
// `each(e, g_e)` is called for every slot with the gradient element (0 in the padding columns).
template <int NPL, class Each>
__device__ __forceinline__ double external_evaluate(const double* __restrict__ qrow, const double* __restrict__ grow, int lane, int D,
                                                    double lq_in, bool& pos_finite, bool& valid, Each each) {
    bool qfin = true, gfin = true;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        const double gv = e < D ? grow[e] : 0.0;      // padding columns of the callback's output are ignored
        qfin = qfin && dm_isfinite(qrow[e]);
        gfin = gfin && dm_isfinite(gv);
        each(k, e, gv);
    }
    pos_finite = wave_all(qfin);
    const bool grad_finite = wave_all(gfin);
    const double lq = uni_f64(lq_in);
    valid = pos_finite && ((dm_isfinite(lq) && grad_finite) || lq == -dm_inf());
    return valid ? lq : -dm_inf();
}

// The built-in normal families beyond the register-resident kernels' 1024 coordinates (the reference has no dimension
// limit, src/hamiltonian.jl:56-87): ℓ and ∇ℓ of ALL chains' positions q [C][ld] in one streaming kernel that stands where
// the host's callback stands for an external model — lq [C], grad [C][ld] — so the same round engine (init, step-size
// search, per-draw loops) serves them up to 4096 dimensions.  One wave per chain; the arithmetic (element order, the
// ABI's blocked dot product) is the functors' of targets.hpp / oracle/targets.hpp.
//   kind 0: standard normal; 1: diagonal normal (a = μ, b = precision); 2: tridiagonal precision (a = diag, b = off); 3: Neal's funnel; 4: AlwaysDivergentTest
template <int NPL>
__global__ __launch_bounds__(64) void builtin_normal_eval_kernel(int kind, int D, int ld, const double* __restrict__ q,
                                                                const double* __restrict__ a, const double* __restrict__ b,
                                                                double* __restrict__ lq, double* __restrict__ grad) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const double* qr = q + (size_t)chain * ld;
    double* gr = grad + (size_t)chain * ld;
    if (kind == 4) {                                   // the reference's AlwaysDivergentTest (test/test_NUTS.jl:58-73): ℓ = 0 at the origin, −Inf elsewhere
        bool zero = true;
#pragma unroll 4
        for (int k = 0; k < NPL; ++k) {
            const int e = lane + WAVE * k;
            gr[e] = e < D ? 1.0 : 0.0;
            zero = zero && (qr[e] == 0.0);
        }
        zero = wave_all(zero);
        if (lane == 0) lq[chain] = zero ? 0.0 : -dm_inf();
        return;
    }
    if (kind == 3) {                                   // Neal's funnel, the arithmetic of FunnelT::eval (targets.hpp) slot by slot
        const double v = uni_f64(qr[0]);
        const double ev = det_exp_u(-v);
        LaneAcc<1, NPL> fa;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int e = lane + WAVE * k;
            const double x = e == 0 ? 0.0 : qr[e];
            fa.add(0, k, x, x);
            gr[e] = -(ev * qr[e]);
        }
        const double S = wave_allreduce1(fa.fold(0));
        const double hd = 0.5 * (double)(D - 1);
        const double hes = (0.5 * ev) * S;
        if (lane == 0) {
            lq[chain] = (((v * v) * (-1.0 / 18.0)) - hes) - hd * v;
            gr[0] = ((v * (-1.0 / 9.0)) + hes) - hd;
        }
        return;
    }
    LaneAcc<1, NPL> acc;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        const double x = qr[e];
        double u, t;                                   // the dot is Σ u·t, the gradient −t
        if (kind == 0) {
            u = x; t = x;
        } else if (kind == 1) {
            u = x - a[e]; t = b[e] * u;
        } else {
            t = a[e] * x;
            if (e > 0 && e < D) t = t + b[e - 1] * qr[e - 1];
            if (e < D - 1) t = t + b[e] * qr[e + 1];
            u = x;
        }
        acc.add(0, k, u, t);
        gr[e] = -t;
    }
    const double s = wave_allreduce1(acc.fold(0));
    if (lane == 0) lq[chain] = -0.5 * s;
}

// DHMC_TARGET_DENSE_NORMAL beyond 1024 coordinates: d = q − μ, then P·d for all chains as ONE product d·P (k-ascending fma chains,
// what DenseNormalT's sym_matvec runs per chain), then ∇ℓ = −Pd and ℓ = −½ d·(Pd) in the ABI's dot order
template <int NPL>
__global__ __launch_bounds__(64) void builtin_dense_normal_pre_kernel(int ld, const double* __restrict__ q, const double* __restrict__ mu,
                                                                     double* __restrict__ d) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const size_t row = (size_t)chain * ld;
#pragma unroll 4
    for (int k = 0; k < NPL; ++k) d[row + lane + WAVE * k] = q[row + lane + WAVE * k] - mu[lane + WAVE * k];
}
template <int NPL>
__global__ __launch_bounds__(64) void builtin_dense_normal_post_kernel(int ld, const double* __restrict__ d, const double* __restrict__ Pd,
                                                                      double* __restrict__ lq, double* __restrict__ grad) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const size_t row = (size_t)chain * ld;
    LaneAcc<1, NPL> acc;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        const double t = Pd[row + e];
        acc.add(0, k, d[row + e], t);
        grad[row + e] = -t;
    }
    const double s = wave_allreduce1(acc.fold(0));
    if (lane == 0) lq[chain] = -0.5 * s;
}

// DHMC_TARGET_LOGISTIC beyond 1024 coefficients: the GEMM gradient of logistic_rounds.hpp for ALL chains as the batched evaluation —
// H = Q·Xᵀ, logistic_link_kernel (r and the blocks' log-likelihood sums), P_z = R·X block by block — and this kernel, the fold of
// rounds_k2_logistic_kernel: G = ((P₀ + P₁) + P₂) + …, ∇ℓ = G − q, ℓ = (the blocks' sums in ascending order) − ½ q·q
template <int NPL>
__global__ __launch_bounds__(64) void builtin_logistic_fold_kernel(int C, int ld, const double* __restrict__ q, LogisticRound L,
                                                                  double* __restrict__ lq_out, double* __restrict__ grad) {
    if ((int)blockIdx.x >= *L.act_count) return;
    const int chain = L.act[blockIdx.x], lane = threadIdx.x;       // the listed chains (all of them outside the rounds)
    const size_t row = (size_t)chain * ld, zs = (size_t)C * ld;
    LaneAcc<1, NPL> qq;
#pragma unroll 2
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        double g = L.P[row + e];
        for (int z = 1; z < L.nz; ++z) g = g + L.P[(size_t)z * zs + row + e];
        const double x = q[row + e];
        qq.add(0, k, x, x);
        grad[row + e] = g - x;
    }
    double s1 = 0.0;
    for (int z = 0; z < L.nz; ++z) {
        const double b = L.S1P[(size_t)z * C + chain];
        s1 = z == 0 ? b : s1 + b;
    }
    const double lq = s1 - 0.5 * wave_allreduce1(qq.fold(0));
    if (lane == 0) lq_out[chain] = lq;
}
static __global__ void builtin_all_rows_kernel(int C, int* __restrict__ act, int* __restrict__ act_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C) act[i] = i;
    if (i == 0) *act_count = C;
}

// initialize_warmup_state (mcmc.jl:129-132) without the density: positions (given, or random_position mcmc.jl:108 from
// the chain's stream exactly as init_kernel does), unit metric, ϵ unspecified, counters cleared
template <int NPL>
__global__ __launch_bounds__(64) void external_init_positions_kernel(InitParams P) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const int D = P.D;
    const size_t row = (size_t)chain * P.Dpad;
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    if (P.q0) {
#pragma unroll 4
        for (int k = 0; k < NPL; ++k) {
            const int e = lane + WAVE * k;
            P.st.q[row + e] = e < D ? P.q0[(size_t)chain * D + e] : 0.0;
        }
    } else {
#pragma unroll 2
        for (int kk = 0; kk < (NPL + 1) / 2; ++kk) {
            uint64_t r1, r2;
            stream_raw64(key, (uint32_t)(lane + WAVE * kk), PURPOSE_INIT_POSITION, 0u, r1, r2);
            const int e0 = lane + WAVE * (2 * kk), e1 = e0 + WAVE;
            P.st.q[row + e0] = e0 < D ? u01_closed_open(r1) * 4 - 2 : 0.0;
            if (2 * kk + 1 < NPL) P.st.q[row + e1] = e1 < D ? u01_closed_open(r2) * 4 - 2 : 0.0;
        }
    }
#pragma unroll 4
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        P.st.minv[row + e] = 1.0;                 // GaussianKineticEnergy(N) (hamiltonian.jl:87)
        P.st.W[row + e] = e < D ? 1.0 : 0.0;
    }
    if (lane == 0) {
        P.st.eps[chain] = dm_nan();               // ϵ = nothing (mcmc.jl:130)
        P.st.transition[chain] = 0;
        P.st.status[chain] = 0;
    }
}

// initialize_warmup_state's strict evaluation (mcmc.jl:131 -> hamiltonian.jl:212-216)
template <int NPL>
__global__ __launch_bounds__(64) void external_init_finish_kernel(int D, int Dpad, ChainArrays st, const double* __restrict__ lq_in,
                                                                 const double* __restrict__ grad_in) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const size_t row = (size_t)chain * Dpad;
    bool pos_finite, valid;
    double* gdst = st.g + row;
    const double lq = external_evaluate<NPL>(st.q + row, grad_in + row, lane, D, lq_in[chain], pos_finite, valid,
                                             [&](int k, int e, double gv) { gdst[e] = gv; });
    if (lane == 0) {
        st.lq[chain] = lq;
        uint32_t s = 0;
        if (!pos_finite) s |= DHMC_ST_NONFINITE_POSITION;
        else if (!valid) s |= DHMC_ST_INVALID_INITIAL;
        st.status[chain] = s;
    }
}

// K2 of the round engine for an external density: ∇ℓ(q′), ℓ(q′) from the callback; second half step; p♯
template <int NPL>
__global__ __launch_bounds__(64) void rounds_k2_external_kernel(RunParams P, RoundBuffers R, LogisticRound L) {
    const int chain = P.chain_base + blockIdx.x, lane = threadIdx.x;
    TreeState& S = R.ts[chain];
    if (S.phase != PH_LEAF) return;
    const size_t row = (size_t)chain * P.Dpad;
    const double h = S.eps_s / 2;
    bool pos_finite, valid;
    double* gdst = P.st.g + row;
    double* cp = R.cp + row;
    double* cps = R.cps + row;
    const double* minv = P.st.minv + row;
    const double lq = external_evaluate<NPL>(P.st.q + row, R.tbuf + row, lane, P.D, L.S1[chain], pos_finite, valid, [&](int k, int e, double gv) {
        gdst[e] = gv;
        const double p1 = cp[e] + h * gv;                                  // hamiltonian.jl:280
        cp[e] = p1;
        if (!P.one_product) cps[e] = minv[e] * p1;                         // (dense metric: overwritten by the product; one-product
    });                                                                    //  recurrence: cps holds M⁻¹pₘ for K3, dense_rounds.hpp)
    if (lane == 0) {
        S.lq_leaf = lq;
        if (!pos_finite) S.status |= DHMC_ST_NONFINITE_POSITION;
    }
}

// ---- warmup(::InitialStepsizeSearch) (mcmc.jl:134-148 -> stepsize.jl:46-85), all chains per callback ----------
struct ExtSearchState {
    double l0, eps;
    int32_t dbl, iter, active, pad_;
};
struct ExtSearchParams {
    int D, Dpad, C, chain_offset;
    uint64_t seed;
    double initial_eps, log_threshold;
    int maxiter;
    ChainArrays st;
    ExtSearchState* ss;   // [C]
    double* p0;           // [C][Dpad]  the search momentum
    double* trial;        // [C][Dpad]  trial positions handed to the callback
    const double* lq_in;  // [C]        callback outputs for the trial positions
    const double* grad_in;
    int* remaining;       // number of chains still searching after this step
};

template <int NPL>
__device__ __forceinline__ void ext_search_propose(const ExtSearchParams& P, size_t row, int lane, double eps) {
    const double h = eps / 2;
#pragma unroll 4
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        const double pm = P.p0[row + e] + h * P.st.g[row + e];              // hamiltonian.jl:277
        const double t = P.st.minv[row + e] * pm;
        P.trial[row + e] = P.st.q[row + e] + eps * t;                        // :278
    }
}

template <int NPL>
__global__ __launch_bounds__(64) void ext_search_begin_kernel(ExtSearchParams P) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const size_t row = (size_t)chain * P.Dpad;
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    // p = W∘z (sample_momentum's stream and order), K = ½ p·M⁻¹p in the ABI's order, slot by slot
    const uint32_t tr = P.st.transition[chain];
    LaneAcc<1, NPL> kacc;
#pragma unroll
    for (int kk = 0; kk < (NPL + 1) / 2; ++kk) {
        uint64_t r1, r2;
        stream_raw64(key, (uint32_t)(lane + WAVE * kk), PURPOSE_SEARCH_MOMENTUM, tr, r1, r2);
        double z0, z1;
        det_randn2_v(r1, r2, &z0, &z1);
        const int e0 = lane + WAVE * (2 * kk), e1 = e0 + WAVE;
        const double pa = P.st.W[row + e0] * z0;
        P.p0[row + e0] = pa;
        kacc.add(0, 2 * kk, pa, P.st.minv[row + e0] * pa);
        if (2 * kk + 1 < NPL) {
            const double pb = P.st.W[row + e1] * z1;
            P.p0[row + e1] = pb;
            kacc.add(0, 2 * kk + 1, pb, P.st.minv[row + e1] * pb);
        }
    }
    const double l0 = uni_f64(joint_logdensity(P.st.lq[chain], wave_allreduce1(kacc.fold(0)) / 2.0));
    ExtSearchState s{l0, P.initial_eps, 0, -1, 1, 0};
    if (!dm_isfinite(l0)) {   // stepsize.jl:77-79
        s.active = 0;
        if (lane == 0) P.st.status[chain] |= DHMC_ST_NONFINITE_START_DENSITY;
    } else {
        ext_search_propose<NPL>(P, row, lane, s.eps);
        if (lane == 0) atomicAdd(P.remaining, 1);
    }
    if (lane == 0) P.ss[chain] = s;
}

template <int NPL>
__global__ __launch_bounds__(64) void ext_search_step_kernel(ExtSearchParams P) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    ExtSearchState s = P.ss[chain];
    if (!s.active) return;
    const size_t row = (size_t)chain * P.Dpad;
    bool pos_finite, valid;
    const double h = s.eps / 2;
    LaneAcc<1, NPL> kacc;
    const double lq = external_evaluate<NPL>(P.trial + row, P.grad_in + row, lane, P.D, P.lq_in[chain], pos_finite, valid, [&](int k, int e, double gv) {
        const double pm = P.p0[row + e] + h * P.st.g[row + e];
        const double p1 = pm + h * gv;                                       // hamiltonian.jl:280
        kacc.add(0, k, p1, P.st.minv[row + e] * p1);
    });
    const double A = uni_f64(joint_logdensity(lq, wave_allreduce1(kacc.fold(0)) / 2.0)) - s.l0;   // stepsize.jl:81-83
    uint32_t st = 0;
    if (!pos_finite) st |= DHMC_ST_NONFINITE_POSITION;
    if (s.iter < 0) {                       // A(initial ϵ): which way to go (stepsize.jl:49-50)
        s.dbl = A > P.log_threshold;
        s.iter = 0;
        s.eps = s.dbl ? 2 * s.eps : s.eps / 2;
    } else if (s.dbl ? (A < P.log_threshold) : (A > P.log_threshold)) {
        s.active = 0;                       // crossed: this ϵ′ is the answer (:54)
    } else {
        s.iter += 1;
        if (s.iter >= P.maxiter) {          // :57-59
            s.active = 0;
            st |= DHMC_ST_STEPSIZE_SEARCH_FAILED;
        } else {
            s.eps = s.dbl ? 2 * s.eps : s.eps / 2;
        }
    }
    if (s.active) {
        ext_search_propose<NPL>(P, row, lane, s.eps);
        if (lane == 0) atomicAdd(P.remaining, 1);
    }
    if (lane == 0) {
        P.ss[chain] = s;
        if (!s.active) P.st.eps[chain] = s.eps;
        if (st) P.st.status[chain] |= st;
    }
}

// ---- the same with a dense (shared) metric: the M⁻¹ and W products are GEMMs over all chains between these kernels --

// K2a of the dense round engine for an external density: q′ = q + ϵ·(M⁻¹pₘ) (hamiltonian.jl:278), M⁻¹pₘ in tbuf
template <int NPL>
__global__ __launch_bounds__(64) void rounds_k2a_dense_external_kernel(RunParams P, RoundBuffers R) {
    const int chain = P.chain_base + blockIdx.x, lane = threadIdx.x;
    const TreeState& S = R.ts[chain];
    if (S.phase != PH_LEAF) return;
    const size_t row = (size_t)chain * P.Dpad;
    const double eps_s = S.eps_s, h = eps_s / 2;
#pragma unroll 4
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        double t;
        if (P.one_product) {                                               // M⁻¹pₘ = p♯ + (ϵ/2)·u, kept in p♯'s place (dense_rounds.hpp)
            t = R.cps[row + e] + h * R.cu[row + e];
            R.cps[row + e] = t;
        } else {
            t = R.tbuf[row + e];
        }
        P.st.q[row + e] = P.st.q[row + e] + eps_s * t;
    }
}

// search, step 0a: z ~ N(0, I) of the search momentum into `z` (then p₀ = z·Wᵀ and p₀♯ = p₀·M⁻¹ are GEMMs)
template <int NPL>
__global__ __launch_bounds__(64) void ext_search_dense_z_kernel(ExtSearchParams P, double* __restrict__ z) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const size_t row = (size_t)chain * P.Dpad;
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    const uint32_t tr = P.st.transition[chain];
#pragma unroll 2
    for (int kk = 0; kk < (NPL + 1) / 2; ++kk) {
        uint64_t r1, r2;
        stream_raw64(key, (uint32_t)(lane + WAVE * kk), PURPOSE_SEARCH_MOMENTUM, tr, r1, r2);
        double z0, z1;
        det_randn2_v(r1, r2, &z0, &z1);
        const int e0 = lane + WAVE * (2 * kk), e1 = e0 + WAVE;
        z[row + e0] = e0 < P.D ? z0 : 0.0;
        if (2 * kk + 1 < NPL) z[row + e1] = e1 < P.D ? z1 : 0.0;
    }
}
// search, step 0b: ℓ₀ from p₀ (P.p0) and p₀♯ (ps); first trial momentum pₘ = p₀ + ϵ/2 ∇ℓ into `pm`
template <int NPL>
__global__ __launch_bounds__(64) void ext_search_dense_begin_kernel(ExtSearchParams P, const double* __restrict__ ps, double* __restrict__ pm) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const size_t row = (size_t)chain * P.Dpad;
    LaneAcc<1, NPL> kacc;
#pragma unroll
    for (int k = 0; k < NPL; ++k) kacc.add(0, k, P.p0[row + lane + WAVE * k], ps[row + lane + WAVE * k]);
    const double l0 = uni_f64(joint_logdensity(P.st.lq[chain], wave_allreduce1(kacc.fold(0)) / 2.0));
    ExtSearchState s{l0, P.initial_eps, 0, -1, 1, 0};
    if (!dm_isfinite(l0)) {
        s.active = 0;
        if (lane == 0) P.st.status[chain] |= DHMC_ST_NONFINITE_START_DENSITY;
    } else {
        const double h = s.eps / 2;
#pragma unroll 4
        for (int k = 0; k < NPL; ++k) pm[row + lane + WAVE * k] = P.p0[row + lane + WAVE * k] + h * P.st.g[row + lane + WAVE * k];
        if (lane == 0) atomicAdd(P.remaining, 1);
    }
    if (lane == 0) P.ss[chain] = s;
}
// search, per trial: (a) q′ = q + ϵ·T into `trial` (T = pₘ·M⁻¹);  callback;  (b) evaluate_ℓ, p′ = pₘ + ϵ/2 ∇ℓ′ into `p1`;
// (c) with p′♯: A(ϵ), the decision, and the next pₘ
template <int NPL>
__global__ __launch_bounds__(64) void ext_search_dense_trial_kernel(ExtSearchParams P, const double* __restrict__ T) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const ExtSearchState s = P.ss[chain];
    if (!s.active) return;
    const size_t row = (size_t)chain * P.Dpad;
#pragma unroll 4
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        P.trial[row + e] = P.st.q[row + e] + s.eps * T[row + e];
    }
}
template <int NPL>
__global__ __launch_bounds__(64) void ext_search_dense_p1_kernel(ExtSearchParams P, uint32_t* __restrict__ flags, double* __restrict__ p1base,
                                                                size_t p1_stride) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const ExtSearchState s = P.ss[chain];
    if (!s.active) return;
    const size_t row = (size_t)chain * P.Dpad;
    bool pos_finite, valid;
    const double h = s.eps / 2;
    double* p1 = p1base + (size_t)chain * p1_stride;
    const double lq = external_evaluate<NPL>(P.trial + row, P.grad_in + row, lane, P.D, P.lq_in[chain], pos_finite, valid, [&](int k, int e, double gv) {
        const double pm = P.p0[row + e] + h * P.st.g[row + e];
        p1[e] = pm + h * gv;                                                 // hamiltonian.jl:280
    });
    if (lane == 0) {
        reinterpret_cast<double*>(flags)[2 * chain] = lq;                    // ℓ(q′) after evaluate_ℓ's rules, for step (c)
        flags[4 * chain + 2] = pos_finite ? 0u : 1u;
    }
}
template <int NPL>
__global__ __launch_bounds__(64) void ext_search_dense_decide_kernel(ExtSearchParams P, const double* __restrict__ p1base, size_t p1_stride,
                                                                    const double* __restrict__ p1s, const uint32_t* __restrict__ flags,
                                                                    double* __restrict__ pm) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    ExtSearchState s = P.ss[chain];
    if (!s.active) return;
    const size_t row = (size_t)chain * P.Dpad;
    LaneAcc<1, NPL> kacc;
    const double* p1 = p1base + (size_t)chain * p1_stride;
#pragma unroll
    for (int k = 0; k < NPL; ++k) kacc.add(0, k, p1[lane + WAVE * k], p1s[row + lane + WAVE * k]);
    const double lq = reinterpret_cast<const double*>(flags)[2 * chain];
    const double A = uni_f64(joint_logdensity(lq, wave_allreduce1(kacc.fold(0)) / 2.0)) - s.l0;
    uint32_t st = flags[4 * chain + 2] ? DHMC_ST_NONFINITE_POSITION : 0u;
    if (s.iter < 0) {
        s.dbl = A > P.log_threshold;
        s.iter = 0;
        s.eps = s.dbl ? 2 * s.eps : s.eps / 2;
    } else if (s.dbl ? (A < P.log_threshold) : (A > P.log_threshold)) {
        s.active = 0;
    } else {
        s.iter += 1;
        if (s.iter >= P.maxiter) {
            s.active = 0;
            st |= DHMC_ST_STEPSIZE_SEARCH_FAILED;
        } else {
            s.eps = s.dbl ? 2 * s.eps : s.eps / 2;
        }
    }
    if (s.active) {
        const double h = s.eps / 2;
#pragma unroll 4
        for (int k = 0; k < NPL; ++k) pm[row + lane + WAVE * k] = P.p0[row + lane + WAVE * k] + h * P.st.g[row + lane + WAVE * k];
        if (lane == 0) atomicAdd(P.remaining, 1);
    }
    if (lane == 0) {
        P.ss[chain] = s;
        if (!s.active) P.st.eps[chain] = s.eps;
        if (st) P.st.status[chain] |= st;
    }
}

// ---- Diagnostics.leapfrog_trajectory / explore_log_acceptance_ratios (src/diagnostics.jl:144-227) for a model whose density is
// the host's callback (or builtin_normal_eval_kernel): what probe_kernel (probe_kernels.hpp) does inside one kernel, here as
// lock-step leapfrogs of all chains with ONE batched evaluation per step —
//      momentum -> start (π₀)   then per step:  half (pₘ, q′)  [dense: pₘ·M⁻¹]  callback(Q′)  finish (evaluate_ℓ, p′)  [dense: p′·M⁻¹]  record
// with the arithmetic of leapfrog_leaf_m / leapfrog_leaf_dense slot by slot.  The chains' own state is never modified.
struct ExtProbeParams {
    int D, Dpad, C, chain_offset;
    uint64_t seed;
    ChainArrays st;
    double *q, *p, *g, *pm, *trial, *ps;   // [C][Dpad]: the travelling point, pₘ, the position handed to the callback, p♯ (dense: a GEMM's output)
    double *p0, *ps0;                      // [C][Dpad]: the momentum the trajectory starts from and its p♯
    double *pi0, *lq_cur;                  // [C]
    int32_t* alive;                        // [C]: the chain still steps (a trajectory stops after its first non-finite ℓ, diagnostics.jl:179)
    uint32_t* status;                      // [C]
    const double* lq_in;                   // the callback's outputs for `trial`
    const double* grad_in;
    int dense;
};

// p₀ (or z for the dense metric: p₀ = z·Wᵀ is a GEMM) from the caller's momenta p_in [C][n_mom][D] or from the chain's stream
template <int NPL>
__global__ __launch_bounds__(64) void ext_probe_momentum_kernel(ExtProbeParams P, const double* __restrict__ p_in, int n_mom, int m,
                                                               uint32_t momentum_index) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const size_t row = (size_t)chain * P.Dpad;
    if (p_in) {
        const double* src = p_in + ((size_t)chain * n_mom + m) * P.D;
#pragma unroll 4
        for (int k = 0; k < NPL; ++k) {
            const int e = lane + WAVE * k;
            P.p0[row + e] = e < P.D ? src[e] : 0.0;
        }
        return;
    }
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
#pragma unroll 2
    for (int kk = 0; kk < (NPL + 1) / 2; ++kk) {
        uint64_t r1, r2;
        stream_raw64(key, (uint32_t)(lane + WAVE * kk), PURPOSE_PROBE_MOMENTUM, momentum_index + (uint32_t)m, r1, r2);
        double z0, z1;
        det_randn2_v(r1, r2, &z0, &z1);
        const int e0 = lane + WAVE * (2 * kk), e1 = e0 + WAVE;
        if (P.dense) {
            P.p0[row + e0] = e0 < P.D ? z0 : 0.0;
            if (2 * kk + 1 < NPL) P.p0[row + e1] = e1 < P.D ? z1 : 0.0;
        } else {
            P.p0[row + e0] = P.st.W[row + e0] * z0;
            if (2 * kk + 1 < NPL) P.p0[row + e1] = P.st.W[row + e1] * z1;
        }
    }
}

// π₀ = logdensity(H, (Q, p₀)); the travelling point back at the start
template <int NPL>
__global__ __launch_bounds__(64) void ext_probe_start_kernel(ExtProbeParams P, int ratios) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const size_t row = (size_t)chain * P.Dpad;
    LaneAcc<1, NPL> kacc;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        const double pv = P.p0[row + e];
        double psv;
        if (P.dense) psv = P.ps0[row + e];
        else { psv = P.st.minv[row + e] * pv; P.ps0[row + e] = psv; }
        kacc.add(0, k, pv, psv);
    }
    const double pi0 = uni_f64(joint_logdensity(P.st.lq[chain], wave_allreduce1(kacc.fold(0)) / 2.0));
    if (lane == 0) {
        P.pi0[chain] = pi0;
        if (ratios && !dm_isfinite(pi0)) P.status[chain] |= DHMC_ST_NONFINITE_START_DENSITY;   // stepsize.jl:77-79 throws
    }
}
template <int NPL>
__global__ __launch_bounds__(64) void ext_probe_restart_kernel(ExtProbeParams P, int ratios) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const size_t row = (size_t)chain * P.Dpad;
#pragma unroll 4
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        const double qv = P.st.q[row + e];
        P.q[row + e] = qv; P.trial[row + e] = qv;
        P.p[row + e] = P.p0[row + e];
        P.g[row + e] = P.st.g[row + e];
    }
    if (lane == 0) {
        P.lq_cur[chain] = P.st.lq[chain];
        P.alive[chain] = ratios ? (dm_isfinite(P.pi0[chain]) ? 1 : 0) : 1;
    }
}

// first half step (hamiltonian.jl:277-278).  Diagonal metric: q′ here; dense: pₘ only, q′ after the GEMM (pos_kernel)
template <int NPL>
__global__ __launch_bounds__(64) void ext_probe_half_kernel(ExtProbeParams P, double eps) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const size_t row = (size_t)chain * P.Dpad;
    const bool go = P.alive[chain] && dm_isfinite(P.lq_cur[chain]);            // diagnostics.jl:179
    if (lane == 0 && !go) P.alive[chain] = 0;
    if (!go) return;
    const double h = eps / 2;
#pragma unroll 4
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        const double pm = P.p[row + e] + h * P.g[row + e];
        P.pm[row + e] = pm;
        if (!P.dense) P.trial[row + e] = P.q[row + e] + eps * (P.st.minv[row + e] * pm);
    }
}
template <int NPL>
__global__ __launch_bounds__(64) void ext_probe_pos_kernel(ExtProbeParams P, double eps) {   // dense: q′ = q + ϵ·(pₘ·M⁻¹), the product in P.ps
    const int chain = blockIdx.x, lane = threadIdx.x;
    if (!P.alive[chain]) return;
    const size_t row = (size_t)chain * P.Dpad;
#pragma unroll 4
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        P.trial[row + e] = P.q[row + e] + eps * P.ps[row + e];
    }
}

// evaluate_ℓ on the callback's outputs, second half step (hamiltonian.jl:279-280): the travelling point becomes (q′, p′, ∇ℓ′)
template <int NPL>
__global__ __launch_bounds__(64) void ext_probe_finish_kernel(ExtProbeParams P, double eps) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    if (!P.alive[chain]) return;
    const size_t row = (size_t)chain * P.Dpad;
    const double h = eps / 2;
    bool pos_finite, valid;
    const double lq = external_evaluate<NPL>(P.trial + row, P.grad_in + row, lane, P.D, P.lq_in[chain], pos_finite, valid, [&](int k, int e, double gv) {
        P.p[row + e] = P.pm[row + e] + h * gv;
        P.g[row + e] = gv;
        P.q[row + e] = P.trial[row + e];
    });
    if (lane == 0) {
        P.lq_cur[chain] = lq;
        if (!pos_finite) P.status[chain] |= DHMC_ST_NONFINITE_POSITION;          // hamiltonian.jl:203 throws
    }
}

// π of the travelling point and the outputs of this step.  Trajectory (out_lq != null): position index `idx` of npos, range[2 chain + hi_side]
// = pos; ratios: out_delta[(chain n_mom + m) n_eps + e] (passed as idx with npos = n_mom·n_eps)
template <int NPL>
__global__ __launch_bounds__(64) void ext_probe_record_kernel(ExtProbeParams P, int idx, int npos, int pos, int start, double* __restrict__ out_delta,
                                                             double* __restrict__ out_lq, double* __restrict__ out_q, double* __restrict__ out_p,
                                                             int32_t* __restrict__ out_range) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    if (!start && !P.alive[chain]) return;
    const size_t row = (size_t)chain * P.Dpad;
    double pi;
    const double lq = P.lq_cur[chain];
    if (start) {
        pi = P.pi0[chain];
    } else {
        LaneAcc<1, NPL> kacc;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int e = lane + WAVE * k;
            const double pv = P.p[row + e];
            kacc.add(0, k, pv, P.dense ? P.ps[row + e] : P.st.minv[row + e] * pv);
        }
        pi = uni_f64(joint_logdensity(lq, wave_allreduce1(kacc.fold(0)) / 2.0));
    }
    if (lane == 0) {
        out_delta[(size_t)chain * npos + idx] = pi - P.pi0[chain];               // diagnostics.jl:196, stepsize.jl:81-83
        if (out_lq) out_lq[(size_t)chain * npos + idx] = lq;
        if (out_range && !start) out_range[2 * chain + (pos > 0 ? 1 : 0)] = pos;
    }
    if (out_q || out_p) {
#pragma unroll 4
        for (int k = 0; k < NPL; ++k) {
            const int e = lane + WAVE * k;
            if (e < P.D) {
                if (out_q) out_q[((size_t)chain * npos + idx) * P.D + e] = P.q[row + e];
                if (out_p) out_p[((size_t)chain * npos + idx) * P.D + e] = P.p[row + e];
            }
        }
    }
}

}  // namespace dhmc
