// Dense (Symmetric) kinetic energy, production path: the per-draw loop as ROUNDS.
//
// With a full M⁻¹ the hot operation of a leapfrog (reference src/hamiltonian.jl:273-282) is the
// product M⁻¹·p (twice per step: ∇kinetic_energy(κ, pₘ) :278 and p♯ of the new point, used by
// kinetic_energy :103 and leaf_turn_statistic NUTS.jl:121).  One chain's matvec streams all of M⁻¹
// (8 MB at D = 1000) for 2 flops per byte; all chains together are a GEMM in which M⁻¹ is reused
// across chains.  So the dense path advances every chain by ONE leapfrog per round:
//
//     [G0,G0b,K0]   chains that begin a transition: p = z·Wᵀ, p♯ = p·M⁻¹ (gathered-row MFMA GEMMs),
//                   π₀, τ₀, first half step                                (rand_p :124, NUTS.jl:232-236)
//      G1           T  = Pₘ · M⁻¹   for all chains          (fp64 MFMA GEMM, gemm_f64_mfma.hpp)
//      K2           q′ = q + ϵT, ℓ(q′), ∇ℓ(q′), p′ = pₘ + ϵ/2 ∇ℓ           (:278-280)
//      G2           P♯ = P′ · M⁻¹   for all chains
//      K3           the leaf: K, Δ, divergence, merge cascade / U-turn tests, proposal selection, tree
//                   bookkeeping, end of transition (draw, statistics, dual averaging), and the half step
//                   that opens the chain's next leapfrog.
//
// That is the reference's recurrence: two products per leapfrog (RunParams::one_product = 0).  Both are products of
// vectors that differ by a multiple of ∇ℓ: M⁻¹pₘ = M⁻¹p + (ϵ/2)·M⁻¹∇ℓq and M⁻¹p′ = M⁻¹pₘ + (ϵ/2)·M⁻¹∇ℓq′.  With
// one_product = 1 (the default of a shared dense metric, include/dhmc.h dhmc_set_dense_products) a chain carries
// u = M⁻¹∇ℓq next to p♯ = M⁻¹p, and a round is
//
//     [G0,G0b,G0c,K0]  chains that begin a transition: p, p♯ as above and u = ∇ℓq·M⁻¹, all three fresh products
//      K2              T = p♯ + (ϵ/2)u (in p♯'s place);  q′ = q + ϵT, ℓ(q′), ∇ℓ(q′), p′ = pₘ + ϵ/2 ∇ℓ
//      G               U′ = ∇ℓ(Q′) · M⁻¹   for all chains — the round's ONE product (2·D² flops per leapfrog)
//      K3              p♯′ = T + (ϵ/2)u′, then as above
//
// i.e. the same map with a different rounding (the oracle restates it: oracle/hamiltonian.hpp leapfrog; the deviation
// from the two-product recurrence is bounded in tests/test_gpu_tolerance.py).
//
// A chain never waits for another: whatever transition and tree position it is in, its next leapfrog
// happens in the next round.  The tree logic is the same iterative adjacent_tree as in
// nuts_kernels.hpp, made resumable: every loop variable lives in a per-chain TreeState in HBM.
// Arithmetic (and therefore every output bit) is identical to the wave-per-chain dense kernel and to
// the oracle: the MFMA accumulation is the same k-ordered fma chain.
#pragma once
#include "gemm_f64_mfma.hpp"
#include "nuts_dense_kernel.hpp"

namespace dhmc {

enum : int32_t { PH_IDLE = 0, PH_NEED_MOMENTUM = 1, PH_LEAF = 2, PH_DONE = 3 };

struct TreeState {
    int32_t phase, n;
    uint32_t tr, dirs, directions0, j, nleaf, nrand, status;
    int32_t depth, dir, reg_edge, stored0, stored1, zeta_top, init_slot;
    int32_t k2_done, pad_;       // K3b has already done K2's work for the chain's next leapfrog (RunParams::fuse_k2)
    int64_t i, i_minus, i_plus, term_left, term_right, vtop_steps;
    uint64_t free_mask;
    unsigned long long total_steps;
    double eps, eps_s, pi0, omega_top, vtop_lsa, lq_cur, lq_leaf;
    DAState da;
    double lv_omega[LDS_LEVELS], lv_vlsa[LDS_LEVELS], lv_vsteps[LDS_LEVELS];
    double sl_lq[LDS_SLOTS], sl_pi[LDS_SLOTS];
    int32_t lv_zeta[LDS_LEVELS];
};

struct RoundBuffers {
    double* cp;     // [C][Dpad]  z, then p / pₘ / p′ of the point being integrated (GEMM input)
    double* cps;    // [C][Dpad]  p♯ (GEMM output)
    double* tbuf;   // [C][Dpad]  M⁻¹pₘ, or p₀ = W z (GEMM output)
    TreeState* ts;  // [C]
    int* list;      // [C] chains that begin a transition this round
    int* list_count;
    int* done_count;
    double* cu;     // [C][Dpad]  u = M⁻¹∇ℓq of the point being integrated (one_product: GEMM output)
};
// the point's q and ∇ℓ live in ChainArrays::q / ::g (the chain's position between transitions)

__device__ __forceinline__ void begin_transition_request(const RunParams& P, const RoundBuffers& R, TreeState& S, uint32_t tr,
                                                         int chain, int lane, int NPLr, double* cp_row) {
    // z ~ N(0, I) into the GEMM input row; K0 finishes the job once p = z·Wᵀ and p♯ are known
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    for (int kk = 0; kk < (NPLr + 1) / 2; ++kk) {
        uint64_t r1, r2;
        stream_raw64(key, (uint32_t)(lane + WAVE * kk), PURPOSE_MOMENTUM, tr, r1, r2);
        double z0, z1;
        det_randn2_v(r1, r2, &z0, &z1);
        const int e0 = lane + WAVE * (2 * kk), e1 = e0 + WAVE;
        cp_row[e0] = e0 < P.D ? z0 : 0.0;
        if (2 * kk + 1 < NPLr) cp_row[e1] = e1 < P.D ? z1 : 0.0;
    }
    if (lane == 0) {
        S.phase = PH_NEED_MOMENTUM;
        R.list[atomicAdd(R.list_count, 1)] = chain;
    }
}

// Start of dhmc_run: load the chain's scalars into its TreeState and ask for the first momentum.
template <int NPL>
__global__ __launch_bounds__(64) void rounds_start_kernel(RunParams P, RoundBuffers R) {
    const int chain = P.chain_base + blockIdx.x, lane = threadIdx.x;
    TreeState& S = R.ts[chain];
    const uint32_t tr0 = P.st.transition[chain];
    if (lane == 0) {
        S.n = 0;
        S.tr = tr0;
        S.status = P.st.status[chain];
        S.lq_cur = P.st.lq[chain];
        S.eps = P.st.eps[chain];
        S.da = P.st.da[chain];
        S.total_steps = 0;
        S.init_slot = 0;
        if (P.adapt && P.da_init) {   // initial_adaptation_state (stepsize.jl:134-138; mcmc.jl:266)
            double le = det_log_u(S.eps);
            S.da.mu = det_log_u(10.0) + le;
            S.da.m = 1;
            S.da.Hbar = 0.0;
            S.da.logeps = le;
            S.da.logeps_bar = 0.0;
        }
    }
    __syncthreads();
    // the chain's position occupies proposal slot 0 of its workspace
    double* ws = P.st.ws + (size_t)chain * P.nvec * P.Dpad;
    const double* q = P.st.q + (size_t)chain * P.Dpad;
    const double* g = P.st.g + (size_t)chain * P.Dpad;
    for (int k = 0; k < NPL; ++k) {
        ws[(size_t)wd_slot(P.max_depth, 0, 0) * P.Dpad + lane + WAVE * k] = q[lane + WAVE * k];
        ws[(size_t)wd_slot(P.max_depth, 0, 1) * P.Dpad + lane + WAVE * k] = g[lane + WAVE * k];
    }
    begin_transition_request(P, R, S, tr0, chain, lane, NPL, R.cp + (size_t)chain * P.Dpad);
}

// K0: chains in R.list have p₀ (tbuf) and p♯₀ (cps): π₀, τ₀, tree reset, first half step.
template <class T, int NPL>
__global__ __launch_bounds__(64) void rounds_k0_kernel(RunParams P, RoundBuffers R) {
    if ((int)blockIdx.x >= *R.list_count) return;
    const int chain = R.list[blockIdx.x], lane = threadIdx.x;
    const int Dpad = P.Dpad;
    TreeState& S = R.ts[chain];
    const size_t row = (size_t)chain * Dpad;
    double* ws = P.st.ws + (size_t)chain * P.nvec * Dpad;
    auto wsv = [&](int idx) -> double* { return ws + (size_t)idx * Dpad; };
    // streamed slot by slot (no register arrays: chains of up to 64 slots per lane come through here)
    const double* __restrict__ p0row = R.tbuf + row;
    const double* __restrict__ psrow = R.cps + row;
    LaneAcc<1, NPL> kacc;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        const double pk = p0row[e], psk = psrow[e];
        kacc.add(0, k, pk, psk);
        wsv(wd_top(0))[e] = pk; wsv(wd_top(1))[e] = psk;              // leaf τ of z₀ (NUTS.jl:120-123)
        wsv(wd_top(2))[e] = pk; wsv(wd_top(3))[e] = psk;
        wsv(wd_top(4))[e] = pk;
        if (P.one_product) wsv(wd_u0(P.max_depth))[e] = R.cu[row + e];   // u of the initial point: the anchor of either edge
    }
    const double lq_cur = S.lq_cur;
    const double pi0 = uni_f64(joint_logdensity(lq_cur, wave_allreduce1(kacc.fold(0)) / 2.0));
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    uint32_t w[4];
    philox4x32_10(0u, PURPOSE_DIRECTIONS, S.tr, key.seed_hi, key.k0, key.k1, w);
    const uint32_t dirs0 = uni_u32(w[0]);
    const double eps = uni_f64(P.adapt ? det_exp_u(S.da.logeps) : S.eps);   // current_ϵ (stepsize.jl:163)
    const bool fwd = (dirs0 & 1u) != 0;
    const double eps_s = fwd ? eps : -eps;
    const double h = eps_s / 2;
    const double* __restrict__ grow = P.st.g + row;
#pragma unroll 4
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        R.cp[row + e] = p0row[e] + h * grow[e];                          // pₘ of the first leapfrog (hamiltonian.jl:277)
    }
    if (lane == 0) {
        const int nslots = ws_nslots(P.max_depth);
        S.pi0 = pi0;
        S.sl_lq[S.init_slot] = lq_cur;
        S.sl_pi[S.init_slot] = pi0;
        S.directions0 = dirs0;
        S.dirs = dirs0 >> 1;
        S.dir = fwd ? 1 : 0;
        S.reg_edge = S.dir;
        S.stored0 = S.stored1 = 0;
        S.free_mask = ((nslots >= 64) ? ~0ull : ((1ull << nslots) - 1ull)) & ~(1ull << S.init_slot);
        S.zeta_top = S.init_slot;
        S.omega_top = 0.0;
        S.vtop_lsa = -dm_inf();
        S.vtop_steps = 0;
        S.depth = 0;
        S.i_minus = S.i_plus = 0;
        S.i = 0;
        S.term_left = 1; S.term_right = 0;
        S.nrand = 0;
        S.j = 0;
        S.nleaf = 1;
        S.eps_s = eps_s;
        S.k2_done = 0;
        S.phase = PH_LEAF;
    }
}

// K2: q′ = q + ϵ·(M⁻¹pₘ), (ℓq′, ∇ℓq′) = evaluate_ℓ(q′), p′ = pₘ + ϵ/2 ∇ℓq′   (hamiltonian.jl:278-280)
template <class T, int NPL>
__global__ __launch_bounds__(64) void rounds_k2_kernel(RunParams P, RoundBuffers R) {
    const int chain = P.chain_base + blockIdx.x, lane = threadIdx.x;
    TreeState& S = R.ts[chain];
    if (S.phase != PH_LEAF || S.k2_done) return;     // (k2_done: K3b did this step when it finished the previous leaf)
    const T tgt(P.tp);
    const size_t row = (size_t)chain * P.Dpad;
    const double eps_s = S.eps_s, h = eps_s / 2;
    double q[NPL], p[NPL], g[NPL], t[NPL];
    ldv<NPL>(P.st.q + row, lane, q);
    ldv<NPL>(R.cp + row, lane, p);
    if (P.one_product) {                             // M⁻¹pₘ = p♯ + (ϵ/2)·u, kept (in p♯'s place) for K3: p♯′ = M⁻¹pₘ + (ϵ/2)·u′
        ldv<NPL>(R.cps + row, lane, t);
#pragma unroll
        for (int k = 0; k < NPL; ++k) t[k] = t[k] + h * R.cu[row + lane + WAVE * k];
        stv<NPL>(R.cps + row, lane, t);
    } else {
        ldv<NPL>(R.tbuf + row, lane, t);
    }
#pragma unroll
    for (int k = 0; k < NPL; ++k) q[k] = q[k] + eps_s * t[k];
    const double lres = tgt.eval(q, g, lane, P.D);
    double lq = T::kDeferred ? tgt.finish(wave_allreduce1(lres)) : lres;
    lq = uni_f64(lq);
    bool pos_finite = true, gfin = true;
    if (!T::kFiniteLqImpliesFiniteQ || !dm_isfinite(lq)) pos_finite = all_finite<T, NPL>(q);
    if constexpr (!T::kFiniteLqImpliesFiniteGrad) gfin = all_finite<T, NPL>(g);
    lq = demote_lq(lq, pos_finite, gfin);
#pragma unroll
    for (int k = 0; k < NPL; ++k) p[k] = p[k] + h * g[k];
    stv<NPL>(P.st.q + row, lane, q);
    stv<NPL>(P.st.g + row, lane, g);
    stv<NPL>(R.cp + row, lane, p);
    if (lane == 0) {
        S.lq_leaf = lq;
        if (!pos_finite) S.status |= DHMC_ST_NONFINITE_POSITION;
    }
}

// K3: the leaf that the last leapfrog produced, and whatever follows it.
template <class T, int NPL>
__global__ __launch_bounds__(64) void rounds_k3_kernel(RunParams P, RoundBuffers R) {
    const int chain = P.chain_base + blockIdx.x, lane = threadIdx.x;
    if (R.ts[chain].phase != PH_LEAF) return;
    __shared__ TreeState S;   // the chain's state, worked on in LDS, written back at the end
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&R.ts[chain]);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&S);
        for (int w = lane; w < (int)(sizeof(TreeState) / 4); w += WAVE) dst[w] = src[w];
    }
    __syncthreads();
    const int D = P.D, Dpad = P.Dpad, max_depth = P.max_depth;
    const T tgt(P.tp);
    const size_t row = (size_t)chain * Dpad;
    double* const ws = P.st.ws + (size_t)chain * P.nvec * Dpad;
    auto wsv = [&](int idx) -> double* { return ws + (size_t)idx * Dpad; };
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    auto row_acc = [&](int idx) { const double* r = wsv(idx); return [r, lane](int k) { return r[lane + WAVE * k]; }; };

    // Only the leaf's (p, p♯) and the running summary (first, first♯, ρ) are held in registers: q and ∇ℓ of the
    // point are never computed on here, only copied (proposal slots, parked edges, the draw) — streamed row to row
    // so that the kernel fits two waves per SIMD without scratch.
    double p[NPL], ps[NPL], cf[NPL], cfs[NPL], cr[NPL];
    ldv<NPL>(R.cp + row, lane, p);
    if (P.one_product) {                             // p♯′ = M⁻¹pₘ + (ϵ/2)·u′ (K2 left M⁻¹pₘ in cps, the product left u′ in cu)
        const double h = S.eps_s / 2;
#pragma unroll
        for (int k = 0; k < NPL; ++k) ps[k] = R.cps[row + lane + WAVE * k] + h * R.cu[row + lane + WAVE * k];
        stv<NPL>(R.cps + row, lane, ps);
    } else {
        ldv<NPL>(R.cps + row, lane, ps);
    }
    auto copy_row = [&](const double* __restrict__ src, double* __restrict__ dst) {
#pragma unroll
        for (int k0 = 0; k0 < NPL; k0 += 4) {
            double t[4];
#pragma unroll
            for (int u = 0; u < 4 && k0 + u < NPL; ++u) t[u] = src[lane + WAVE * (k0 + u)];
#pragma unroll
            for (int u = 0; u < 4 && k0 + u < NPL; ++u) dst[lane + WAVE * (k0 + u)] = t[u];
        }
    };

    auto randexp = [&]() -> double {   // Random.randexp at NUTS.jl:44: the nrand-th draw of this transition
        uint64_t r1, r2;
        stream_raw64(key, S.nrand, PURPOSE_TREE, S.tr, r1, r2);
        S.nrand += 1;
        return uni_f64(det_randexp_t<dm_u>(r1));
    };
    auto save_leaf = [&](double lq_leaf, double pi_leaf) -> int {
        int s = __builtin_ctzll(S.free_mask);
        S.free_mask &= ~(1ull << s);
        copy_row(P.st.q + row, wsv(wd_slot(max_depth, s, 0)));
        if constexpr (!T::kRecomputeGrad) copy_row(P.st.g + row, wsv(wd_slot(max_depth, s, 1)));
        S.sl_lq[s] = lq_leaf;
        S.sl_pi[s] = pi_leaf;
        return s;
    };

    // ---- the leaf (NUTS.jl:148-159) -----------------------------------------------------------
    const bool fwd = S.dir == 1;
    const int64_t di = fwd ? 1 : -1;
    const uint32_t j = S.j, nleaf = S.nleaf;
    const int depth0 = S.depth;
    LaneAcc<1, NPL> kacc;
#pragma unroll
    for (int k = 0; k < NPL; ++k) kacc.add(0, k, p[k], ps[k]);
    const double lq_leaf = S.lq_leaf;
    const double pi_leaf = uni_f64(joint_logdensity(lq_leaf, wave_allreduce1(kacc.fold(0)) / 2.0));
    int64_t i = S.i + di;
    S.total_steps += 1;
    const double delta = pi_leaf - S.pi0;
    double v_lsa = delta < 0.0 ? delta : 0.0;
    int64_t v_steps = 1;
    bool invalid = false, finished = false, doubled = false;
    int level = 0;
    if (delta < P.min_delta) {
        S.term_left = S.term_right = i;
        invalid = true;
    } else {
#pragma unroll
        for (int k = 0; k < NPL; ++k) { cf[k] = p[k]; cfs[k] = ps[k]; cr[k] = p[k]; }
        double c_omega = delta;
        int c_zeta = -1;
        for (;;) {
            const bool sub = ((j >> level) & 1u) != 0;
            const bool top = !sub && (j == nleaf - 1) && (level == depth0);
            if (!sub && !top) break;
            auto a_cf = [&](int k) { return cf[k]; };
            auto a_cfs = [&](int k) { return cfs[k]; };
            auto a_p = [&](int k) { return p[k]; };
            auto a_ps = [&](int k) { return ps[k]; };
            auto a_cr = [&](int k) { return cr[k]; };
            bool turning;
            if (sub) {
                auto lf = row_acc(wd_stack(level, 0)), lfs = row_acc(wd_stack(level, 1));
                auto ll = row_acc(wd_stack(level, 2)), lls = row_acc(wd_stack(level, 3));
                auto lr = row_acc(wd_stack(level, 4));
                turning = fwd ? merge_core_dense<NPL>(lf, lfs, ll, lls, lr, a_cf, a_cfs, a_p, a_ps, a_cr, lf, lfs, cf, cfs, cr)
                              : merge_core_dense<NPL>(a_p, a_ps, a_cf, a_cfs, a_cr, ll, lls, lf, lfs, lr, lf, lfs, cf, cfs, cr);
                const double wl = S.lv_omega[level];
                double w;
                logaddexp_pair(S.lv_vlsa[level], v_lsa, wl, c_omega, lane, v_lsa, w);
                v_steps += (int64_t)S.lv_vsteps[level];
                if (turning) {
                    S.term_left = i - di * (((int64_t)2 << level) - 1);
                    S.term_right = i;
                    invalid = true;
                    level += 1;
                    break;
                }
                const double logprob2 = c_omega - w;
                const bool pick = logprob2 >= 0.0 || (randexp() > -logprob2);
                const int lz = S.lv_zeta[level];
                if (pick) {
                    S.free_mask |= (1ull << lz);
                } else {
                    if (c_zeta >= 0) S.free_mask |= (1ull << c_zeta);
                    c_zeta = lz;
                }
                c_omega = w;
                level += 1;
            } else {
                auto tm = row_acc(wd_top(0)), tms = row_acc(wd_top(1));
                auto tp = row_acc(wd_top(2)), tps = row_acc(wd_top(3));
                auto trr = row_acc(wd_top(4));
                turning = fwd ? merge_core_dense<NPL>(tm, tms, tp, tps, trr, a_cf, a_cfs, a_p, a_ps, a_cr, a_cf, a_cfs, cf, cfs, cr)
                              : merge_core_dense<NPL>(a_p, a_ps, a_cf, a_cfs, a_cr, tm, tms, tp, tps, trr, a_cf, a_cfs, cf, cfs, cr);
                double w, vt;
                logaddexp_pair(S.vtop_lsa, v_lsa, S.omega_top, c_omega, lane, vt, w);
                S.vtop_lsa = vt;
                S.vtop_steps += v_steps;
                const double logprob2 = c_omega - S.omega_top;
                const bool pick = logprob2 >= 0.0 || (randexp() > -logprob2);
                if (pick) {
                    if (c_zeta < 0) c_zeta = save_leaf(lq_leaf, pi_leaf);
                    if (S.zeta_top != S.init_slot) S.free_mask |= (1ull << S.zeta_top);
                    S.zeta_top = c_zeta;
                } else if (c_zeta >= 0) {
                    S.free_mask |= (1ull << c_zeta);
                }
                S.omega_top = w;
                S.depth += 1;
                if (fwd) S.i_plus = i; else S.i_minus = i;
                if (turning) {
                    S.term_left = S.i_minus;
                    S.term_right = S.i_plus;
                    finished = true;
                } else if (S.depth < max_depth) {
                    stv<NPL>(wsv(wd_top(fwd ? 2 : 0)), lane, p);
                    stv<NPL>(wsv(wd_top(fwd ? 3 : 1)), lane, ps);
                    stv<NPL>(wsv(wd_top(4)), lane, cr);
                    doubled = true;
                } else {
                    finished = true;   // depth == max_depth: REACHED_MAX_DEPTH stays in term
                }
                level = -1;
                break;
            }
        }
        if (level >= 0 && !invalid) {
            if (c_zeta < 0) c_zeta = save_leaf(lq_leaf, pi_leaf);
            stv<NPL>(wsv(wd_stack(level, 0)), lane, cf);
            stv<NPL>(wsv(wd_stack(level, 1)), lane, cfs);
            stv<NPL>(wsv(wd_stack(level, 2)), lane, p);
            stv<NPL>(wsv(wd_stack(level, 3)), lane, ps);
            stv<NPL>(wsv(wd_stack(level, 4)), lane, cr);
            S.lv_omega[level] = c_omega;
            S.lv_vlsa[level] = v_lsa;
            S.lv_vsteps[level] = (double)v_steps;
            S.lv_zeta[level] = c_zeta;
        }
    }
    if (invalid) {
        for (int l2 = level; l2 < depth0; ++l2) {
            if ((j >> l2) & 1u) {
                v_lsa = uni_f64(det_logaddexp_u(S.lv_vlsa[l2], v_lsa));
                v_steps += (int64_t)S.lv_vsteps[l2];
            }
        }
        S.vtop_lsa = uni_f64(det_logaddexp_u(S.vtop_lsa, v_lsa));
        S.vtop_steps += v_steps;
        finished = true;
    }
    S.i = i;

    if (finished) {
        // ---- end of the transition (NUTS.jl:238-240; mcmc.jl:272-278, 375-377) --------------------
        const double eps_used = fwd ? S.eps_s : -S.eps_s;
        double a = det_exp_u(S.vtop_lsa) / (double)S.vtop_steps;
        const double acc_rate = uni_f64(a < 1.0 ? a : 1.0);
        S.init_slot = S.zeta_top;
        S.lq_cur = S.sl_lq[S.init_slot];
        const double pi_stat = S.sl_pi[S.init_slot];
        const size_t o = (size_t)chain * P.N + S.n;
        {
            double q[NPL];
            ldv<NPL>(wsv(wd_slot(max_depth, S.init_slot, 0)), lane, q);
            stv<NPL>(P.st.q + row, lane, q);
            if (P.out.draws) {
                double* drow = P.out.draws + o * D;
#pragma unroll
                for (int k = 0; k < NPL; ++k)
                    if (lane + WAVE * k < D) drow[lane + WAVE * k] = q[k];
            }
            window_accumulate<NPL>(P, row, lane, q, S.n);
            if constexpr (T::kRecomputeGrad) {
                double g[NPL];
                (void)tgt.eval(q, g, lane, D);
                stv<NPL>(P.st.g + row, lane, g);
            }
        }
        if constexpr (!T::kRecomputeGrad) copy_row(wsv(wd_slot(max_depth, S.init_slot, 1)), P.st.g + row);
        if (lane == 0) {
            if (P.out.logdensities) P.out.logdensities[o] = S.lq_cur;
            if (P.out.eps) P.out.eps[o] = eps_used;
            if (P.out.pi) P.out.pi[o] = pi_stat;
            if (P.out.acceptance_rate) P.out.acceptance_rate[o] = acc_rate;
            if (P.out.steps) P.out.steps[o] = S.vtop_steps;
            if (P.out.term_left) P.out.term_left[o] = S.term_left;
            if (P.out.term_right) P.out.term_right[o] = S.term_right;
            if (P.out.depth) P.out.depth[o] = S.depth;
            if (P.out.directions) P.out.directions[o] = S.directions0;
        }
        if (P.adapt) {   // adapt_stepsize (stepsize.jl:147-156)
            S.da.m += 1;
            const double m = (double)S.da.m;
            S.da.Hbar += (P.delta - acc_rate - S.da.Hbar) / (m + (double)P.t0);
            S.da.logeps = S.da.mu - __builtin_sqrt(m) / P.gamma * S.da.Hbar;
            S.da.logeps_bar += det_pow_pos_u(m, -P.kappa) * (S.da.logeps - S.da.logeps_bar);
        }
        S.n += 1;
        S.tr += 1;
        if ((int64_t)S.n < P.N) {
            __syncthreads();
            begin_transition_request(P, R, S, S.tr, chain, lane, NPL, R.cp + row);
        } else {
            S.phase = PH_DONE;
            if (lane == 0) {
                P.st.lq[chain] = S.lq_cur;
                if (P.adapt) {
                    P.st.da[chain] = S.da;
                    if (P.da_finalize) P.st.eps[chain] = det_exp_u(S.da.logeps_bar);
                }
                P.st.transition[chain] = S.tr;
                P.st.status[chain] = S.status;
                if (P.leapfrog_counter) atomicAdd(P.leapfrog_counter, S.total_steps);
                atomicAdd(R.done_count, 1);
            }
        }
    } else {
        // ---- the chain's next leapfrog: same doubling, or the next one (trees.jl:290-293) ---------
        double eps_s = S.eps_s;
        if (doubled) {
            const bool nfwd = (S.dirs & 1u) != 0;
            S.dirs >>= 1;
            const int ndir = nfwd ? 1 : 0;
            if (S.reg_edge != ndir) {
                copy_row(P.st.q + row, wsv(wd_edge(S.reg_edge, 0)));
                if constexpr (!T::kRecomputeGrad) copy_row(P.st.g + row, wsv(wd_edge(S.reg_edge, 1)));
                if (S.reg_edge == 1) S.stored1 = 1; else S.stored0 = 1;
                const bool have = nfwd ? (S.stored1 != 0) : (S.stored0 != 0);
                const int qsrc = have ? wd_edge(ndir, 0) : wd_slot(max_depth, S.init_slot, 0);
                const int gsrc = have ? wd_edge(ndir, 1) : wd_slot(max_depth, S.init_slot, 1);
                if constexpr (T::kRecomputeGrad) {
                    double q[NPL], g[NPL];
                    ldv<NPL>(wsv(qsrc), lane, q);
                    (void)tgt.eval(q, g, lane, D);
                    stv<NPL>(P.st.q + row, lane, q);
                    stv<NPL>(P.st.g + row, lane, g);
                } else {
                    copy_row(wsv(qsrc), P.st.q + row);
                    copy_row(wsv(gsrc), P.st.g + row);
                }
                ldv<NPL>(wsv(wd_top(nfwd ? 2 : 0)), lane, p);
                if (P.one_product) {                 // the edge's p♯ and u travel with it
                    copy_row(R.cu + row, wsv(wd_edge_u(max_depth, S.reg_edge)));
                    copy_row(wsv(have ? wd_edge_u(max_depth, ndir) : wd_u0(max_depth)), R.cu + row);
                    copy_row(wsv(wd_top(nfwd ? 3 : 1)), R.cps + row);
                }
            }
            S.reg_edge = ndir;
            S.dir = ndir;
            S.i = nfwd ? S.i_plus : S.i_minus;
            S.j = 0;
            S.nleaf = 1u << S.depth;
            const double eps = eps_s < 0 ? -eps_s : eps_s;
            eps_s = nfwd ? eps : -eps;
            S.eps_s = eps_s;
        } else {
            S.j = j + 1;
        }
        const double h = eps_s / 2;
        const double* __restrict__ grow = P.st.g + row;
#pragma unroll
        for (int k = 0; k < NPL; ++k) p[k] = p[k] + h * grow[lane + WAVE * k];   // pₘ of the next leapfrog (hamiltonian.jl:277)
        stv<NPL>(R.cp + row, lane, p);
    }
    __syncthreads();
    {
        uint32_t* dst = reinterpret_cast<uint32_t*>(&R.ts[chain]);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&S);
        for (int w = lane; w < (int)(sizeof(TreeState) / 4); w += WAVE) dst[w] = src[w];
    }
}

}  // namespace dhmc
