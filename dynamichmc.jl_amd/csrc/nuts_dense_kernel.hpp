// Dense (Symmetric) kinetic energy variant of the per-draw loop kernel: GaussianKineticEnergy with a
// full M⁻¹ shared by all chains of the context (reference src/hamiltonian.jl:73,103,110,124).
//
// Round-1 status: functional and parity-exact, NOT yet the fast path.  One wavefront = one chain
// as in nuts_kernels.hpp, and each M⁻¹·v is a wave-local matvec that streams the symmetric
// matrix row by row from L2 / Infinity Cache (out_i = Σ_k M[k][i] v_k in ascending k, i.e. one
// k-ordered fma chain per output coordinate — exactly what a v_mfma_f64_16x16x4_f64 accumulation
// chain computes).  The production design for BASELINE config 3 groups 16 chains per workgroup so
// the M⁻¹ tile is shared through LDS and contracted with MFMA; see DESIGN.md §8.
//
// Differences from the diagonal kernel: p♯ = M⁻¹p is carried next to p everywhere (registers,
// suspended summaries, trajectory τ) instead of being recomputed, and every suspended summary and
// the trajectory τ live in the HBM workspace.
#pragma once
#include "nuts_kernels.hpp"

namespace dhmc {

struct DenseMetric {
    const double* Minv;  // [Dpad][Dpad] symmetric, pads 0
    const double* WT;    // [Dpad][Dpad] = Wᵀ (upper triangular), W Wᵀ = M, pads 0
    size_t stride;       // 0: one matrix shared by all chains; Dpad²: chain c's matrices start at c * stride (dense_per_chain)
    __device__ __forceinline__ DenseMetric of_chain(int chain) const { return DenseMetric{Minv + chain * stride, WT + chain * stride, 0}; }
};

// dense workspace vector indices
__host__ __device__ inline int wd_top(int which) { return which; }                     // 0 p₋, 1 p♯₋, 2 p₊, 3 p♯₊, 4 ρ
__host__ __device__ inline int wd_edge(int dir, int which) { return 5 + 2 * dir + which; }   // 0 q, 1 g
__host__ __device__ inline int wd_stack(int level, int which) { return 9 + 5 * level + which; }  // first, first♯, last, last♯, ρ
__host__ __device__ inline int wd_slot(int max_depth, int s, int which) { return 9 + 5 * max_depth + 2 * s + which; }
__host__ __device__ inline int wd_nvec_base(int max_depth) { return 9 + 5 * max_depth + 2 * ws_nslots(max_depth); }
// the one-product recurrence of the round engine (dense_rounds.hpp) carries u = M⁻¹∇ℓq next to q and ∇ℓq:
__host__ __device__ inline int wd_u0(int max_depth) { return wd_nvec_base(max_depth); }                     // u of the transition's initial point
__host__ __device__ inline int wd_edge_u(int max_depth, int dir) { return wd_nvec_base(max_depth) + 1 + dir; }   // u of a parked edge
__host__ __device__ inline int wd_nvec(int max_depth) { return wd_nvec_base(max_depth) + 3; }
__host__ __device__ inline size_t lds_bytes_dense() {
    return sizeof(double) * (3 * LDS_LEVELS + 2 * LDS_SLOTS) + sizeof(int) * LDS_LEVELS;
}

// rand_p (hamiltonian.jl:124): p = W z, and its p♯
template <int NPL>
__device__ __forceinline__ void sample_momentum_dense(const ChainKey& key, uint32_t purpose, uint32_t transition,
                                                      const DenseMetric& M, int Dpad, int D, int lane,
                                                      double (&p)[NPL], double (&ps)[NPL]) {
    double z[NPL];
#pragma unroll
    for (int kk = 0; kk < (NPL + 1) / 2; ++kk) {
        uint64_t r1, r2;
        stream_raw64(key, (uint32_t)(lane + WAVE * kk), purpose, transition, r1, r2);
        double z0, z1;
        det_randn2_v(r1, r2, &z0, &z1);
        z[2 * kk] = (lane + WAVE * (2 * kk) < D) ? z0 : 0.0;
        if (2 * kk + 1 < NPL) z[2 * kk + 1] = (lane + WAVE * (2 * kk + 1) < D) ? z1 : 0.0;
    }
    sym_matvec<NPL>(M.WT, Dpad, D, lane, z, p);      // Σ_k Wᵀ[k][i] z_k = (W z)_i, zeros above the diagonal
    sym_matvec<NPL>(M.Minv, Dpad, D, lane, p, ps);
}

// leapfrog (hamiltonian.jl:273-282) + the leaf's joint log density, dense metric
template <class T, int NPL>
__device__ __forceinline__ void leapfrog_leaf_dense(const T& tgt, const DenseMetric& M, int Dpad, int lane, int D,
                                                    double (&q)[NPL], double (&p)[NPL], double (&g)[NPL],
                                                    double (&ps)[NPL], double eps, double& lq_out, double& pi_out,
                                                    bool& pos_finite) {
    const double h = eps / 2;
#pragma unroll
    for (int k = 0; k < NPL; ++k) p[k] = p[k] + h * g[k];        // pₘ  (:277)
    sym_matvec<NPL>(M.Minv, Dpad, D, lane, p, ps);               // ∇kinetic_energy(κ, pₘ)
#pragma unroll
    for (int k = 0; k < NPL; ++k) q[k] = q[k] + eps * ps[k];     // :278
    const double lres = tgt.eval(q, g, lane, D);                 // :279
#pragma unroll
    for (int k = 0; k < NPL; ++k) p[k] = p[k] + h * g[k];        // :280
    sym_matvec<NPL>(M.Minv, Dpad, D, lane, p, ps);               // p♯ = M⁻¹ p′
    LaneAcc<1, NPL> kacc;
#pragma unroll
    for (int k = 0; k < NPL; ++k) kacc.add(0, k, p[k], ps[k]);
    double lq, K;
    if constexpr (T::kDeferred) {
        double r[2] = {lres, kacc.fold(0)};
        wave_allreduce<2>(r);
        lq = tgt.finish(r[0]);
        K = r[1] / 2.0;
    } else {
        lq = lres;
        K = wave_allreduce1(kacc.fold(0)) / 2.0;
    }
    lq = uni_f64(lq);
    pos_finite = true;
    bool gfin = true;
    if (!T::kFiniteLqImpliesFiniteQ || !dm_isfinite(lq)) pos_finite = all_finite<T, NPL>(q);
    if constexpr (!T::kFiniteLqImpliesFiniteGrad) gfin = all_finite<T, NPL>(g);
    lq = demote_lq(lq, pos_finite, gfin);
    lq_out = lq;
    pi_out = uni_f64(joint_logdensity(lq, K));
}

// The same step with ONE M⁻¹ product (dense_rounds.hpp, oracle/hamiltonian.hpp leapfrog with one_product): u = M⁻¹∇ℓq is carried next
// to ∇ℓq, M⁻¹pₘ = p♯ + (ϵ/2)·u and p♯′ = M⁻¹pₘ + (ϵ/2)·u′ follow by linearity, u′ = M⁻¹∇ℓq′ is the step's only product.
template <class T, int NPL>
__device__ __forceinline__ void leapfrog_leaf_dense1(const T& tgt, const DenseMetric& M, int Dpad, int lane, int D,
                                                     double (&q)[NPL], double (&p)[NPL], double (&g)[NPL],
                                                     double (&ps)[NPL], double (&u)[NPL], double eps, double& lq_out, double& pi_out,
                                                     bool& pos_finite) {
    const double h = eps / 2;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        p[k] = p[k] + h * g[k];                                  // pₘ  (:277)
        ps[k] = ps[k] + h * u[k];                                // M⁻¹pₘ
        q[k] = q[k] + eps * ps[k];                               // :278
    }
    const double lres = tgt.eval(q, g, lane, D);                 // :279
#pragma unroll
    for (int k = 0; k < NPL; ++k) p[k] = p[k] + h * g[k];        // :280
    sym_matvec<NPL>(M.Minv, Dpad, D, lane, g, u);                // u′ = M⁻¹ ∇ℓq′
    LaneAcc<1, NPL> kacc;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        ps[k] = ps[k] + h * u[k];                                // p♯′
        kacc.add(0, k, p[k], ps[k]);
    }
    double lq, K;
    if constexpr (T::kDeferred) {
        double r[2] = {lres, kacc.fold(0)};
        wave_allreduce<2>(r);
        lq = tgt.finish(r[0]);
        K = r[1] / 2.0;
    } else {
        lq = lres;
        K = wave_allreduce1(kacc.fold(0)) / 2.0;
    }
    lq = uni_f64(lq);
    pos_finite = true;
    bool gfin = true;
    if (!T::kFiniteLqImpliesFiniteQ || !dm_isfinite(lq)) pos_finite = all_finite<T, NPL>(q);
    if constexpr (!T::kFiniteLqImpliesFiniteGrad) gfin = all_finite<T, NPL>(g);
    lq = demote_lq(lq, pos_finite, gfin);
    lq_out = lq;
    pi_out = uni_f64(joint_logdensity(lq, K));
}

// combine_turn_statistics (NUTS.jl:132-139) with explicit p♯ vectors; x earlier in time, y later.
template <int NPL, class XM, class XMS, class XP, class XPS, class XR, class YM, class YMS, class YP, class YPS, class YR,
          class NF, class NFS>
__device__ __forceinline__ bool merge_core_dense(XM xm_, XMS xms_, XP xp_, XPS xps_, XR xr_, YM ym_, YMS yms_, YP yp_,
                                                 YPS yps_, YR yr_, NF nf_, NFS nfs_, double (&cf)[NPL],
                                                 double (&cfs)[NPL], double (&cr)[NPL]) {
    LaneAcc<6, NPL> A;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const double xms = xms_(k), xp = xp_(k), xps = xps_(k), xr = xr_(k);
        const double ym = ym_(k), yms = yms_(k), yps = yps_(k), yr = yr_(k);
        const double nf = nf_(k), nfs = nfs_(k);
        (void)xm_; (void)yp_;
        const double s1 = xr + ym;
        const double s2 = xp + yr;
        const double r = xr + yr;
        A.add(0, k, xms, s1);
        A.add(1, k, yms, s1);
        A.add(2, k, xps, s2);
        A.add(3, k, yps, s2);
        A.add(4, k, xms, r);
        A.add(5, k, yps, r);
        cf[k] = nf;
        cfs[k] = nfs;
        cr[k] = r;
    }
    double acc[6];
    A.fold_all(acc);
    wave_allreduce<6>(acc);
    return acc[0] < 0 || acc[1] < 0 || acc[2] < 0 || acc[3] < 0 || acc[4] < 0 || acc[5] < 0;
}

template <class T, int NPL>
__global__ __launch_bounds__(64, 1) void nuts_run_dense_kernel(RunParams P, DenseMetric Mall) {
    const int chain = P.launch_order ? P.launch_order[blockIdx.x] : (int)blockIdx.x;
    const DenseMetric M = Mall.of_chain(chain);
    const int lane = threadIdx.x;
    const int D = P.D, Dpad = P.Dpad;

    extern __shared__ double lds[];
    double* lv_omega = lds;
    double* lv_vlsa = lv_omega + LDS_LEVELS;
    double* lv_vsteps = lv_vlsa + LDS_LEVELS;
    double* sl_lq = lv_vsteps + LDS_LEVELS;
    double* sl_pi = sl_lq + LDS_SLOTS;
    int* lv_zeta = (int*)(sl_pi + LDS_SLOTS);

    const T tgt(P.tp);
    const size_t row = (size_t)chain * Dpad;
    double* const ws = P.st.ws + (size_t)chain * P.nvec * Dpad;
    auto wsv = [&](int idx) -> double* { return ws + (size_t)idx * Dpad; };
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    const int max_depth = P.max_depth;
    const int nslots = ws_nslots(max_depth);

    double q[NPL], p[NPL], g[NPL], ps[NPL], cf[NPL], cfs[NPL], cr[NPL];
    double u[NPL];                                   // one-product recurrence only: M⁻¹∇ℓq of the travelling point
    const bool one_product = P.one_product != 0;
    ldv<NPL>(P.st.q + row, lane, q);
    ldv<NPL>(P.st.g + row, lane, g);
    double lq_cur = P.st.lq[chain];
    double eps_fixed = P.st.eps[chain];
    DAState da = P.st.da[chain];
    uint32_t status = P.st.status[chain];
    const uint32_t tr0 = P.st.transition[chain];
    unsigned long long total_steps = 0;

    if (P.adapt && P.da_init) {
        double le = det_log_u(eps_fixed);
        da.mu = det_log_u(10.0) + le;
        da.m = 1;
        da.Hbar = 0.0;
        da.logeps = le;
        da.logeps_bar = 0.0;
    }

    int init_slot = 0;
    stv<NPL>(wsv(wd_slot(max_depth, init_slot, 0)), lane, q);
    if constexpr (!T::kRecomputeGrad) stv<NPL>(wsv(wd_slot(max_depth, init_slot, 1)), lane, g);

    uint64_t free_mask = 0;
    auto save_leaf = [&](double lq_leaf, double pi_leaf) -> int {
        int s = __builtin_ctzll(free_mask);
        free_mask &= ~(1ull << s);
        stv<NPL>(wsv(wd_slot(max_depth, s, 0)), lane, q);
        if constexpr (!T::kRecomputeGrad) stv<NPL>(wsv(wd_slot(max_depth, s, 1)), lane, g);
        sl_lq[s] = lq_leaf;
        sl_pi[s] = pi_leaf;
        return s;
    };
    auto row_acc = [&](int idx) { const double* r = wsv(idx); return [r, lane](int k) { return r[lane + WAVE * k]; }; };

    for (int64_t n = 0; n < P.N; ++n) {
        const uint32_t tr = tr0 + (uint32_t)n;
        const double eps = uni_f64(P.adapt ? det_exp_u(da.logeps) : eps_fixed);

        sample_momentum_dense<NPL>(key, PURPOSE_MOMENTUM, tr, M, Dpad, D, lane, p, ps);
        if (one_product) {                           // the transition's anchor: a fresh product (make_phasepoint in the oracle)
            sym_matvec<NPL>(M.Minv, Dpad, D, lane, g, u);
            stv<NPL>(wsv(wd_u0(max_depth)), lane, u);
        }
        uint32_t dirs;
        {
            uint32_t w[4];
            philox4x32_10(0u, PURPOSE_DIRECTIONS, tr, key.seed_hi, key.k0, key.k1, w);
            dirs = uni_u32(w[0]);
        }
        const uint32_t directions0 = dirs;
        double pi0;
        {
            LaneAcc<1, NPL> kacc;
#pragma unroll
            for (int k = 0; k < NPL; ++k) kacc.add(0, k, p[k], ps[k]);
            pi0 = uni_f64(joint_logdensity(lq_cur, wave_allreduce1(kacc.fold(0)) / 2.0));
        }
        // leaf τ of z₀ (NUTS.jl:120-123)
        stv<NPL>(wsv(wd_top(0)), lane, p); stv<NPL>(wsv(wd_top(1)), lane, ps);
        stv<NPL>(wsv(wd_top(2)), lane, p); stv<NPL>(wsv(wd_top(3)), lane, ps);
        stv<NPL>(wsv(wd_top(4)), lane, p);
        sl_lq[init_slot] = lq_cur;
        sl_pi[init_slot] = pi0;

        uint32_t nrand = 0, rexp_base = 0;
        double rexp_vals;
        auto rexp_fill = [&](uint32_t base) {
            uint64_t r1, r2;
            stream_raw64(key, base + (uint32_t)lane, PURPOSE_TREE, tr, r1, r2);
            rexp_vals = det_randexp_v(r1);
            rexp_base = base;
        };
        rexp_fill(0);
        auto randexp = [&]() -> double {
            if (nrand - rexp_base >= 64u) rexp_fill(nrand & ~63u);
            double v = readlane_f64(rexp_vals, (int)(nrand & 63u));
            nrand += 1;
            return v;
        };

        bool stored0 = false, stored1 = false;
        int reg_edge = 2;
        free_mask = ((nslots >= 64) ? ~0ull : ((1ull << nslots) - 1ull)) & ~(1ull << init_slot);
        int zeta_top = init_slot;
        double omega_top = 0.0;
        double vtop_lsa = -dm_inf();
        int64_t vtop_steps = 0;
        int depth = 0;
        int64_t i_minus = 0, i_plus = 0;
        int64_t term_left = 1, term_right = 0;

        bool finished = false;
        while (!finished && depth < max_depth) {
            const bool fwd = (dirs & 1u) != 0;
            dirs >>= 1;
            const int dir = fwd ? 1 : 0;
            if (reg_edge != 2 && reg_edge != dir) {
                stv<NPL>(wsv(wd_edge(reg_edge, 0)), lane, q);
                if constexpr (!T::kRecomputeGrad) stv<NPL>(wsv(wd_edge(reg_edge, 1)), lane, g);
                if (one_product) stv<NPL>(wsv(wd_edge_u(max_depth, reg_edge)), lane, u);
                if (reg_edge == 1) stored1 = true; else stored0 = true;
                const bool have = fwd ? stored1 : stored0;
                if (one_product) ldv<NPL>(wsv(have ? wd_edge_u(max_depth, dir) : wd_u0(max_depth)), lane, u);
                const int qsrc = have ? wd_edge(dir, 0) : wd_slot(max_depth, init_slot, 0);
                const int gsrc = have ? wd_edge(dir, 1) : wd_slot(max_depth, init_slot, 1);
                ldv<NPL>(wsv(qsrc), lane, q);
                if constexpr (T::kRecomputeGrad) (void)tgt.eval(q, g, lane, D);
                else ldv<NPL>(wsv(gsrc), lane, g);
                ldv<NPL>(wsv(wd_top(fwd ? 2 : 0)), lane, p);
                ldv<NPL>(wsv(wd_top(fwd ? 3 : 1)), lane, ps);
            }
            reg_edge = dir;
            int64_t i = fwd ? i_plus : i_minus;
            const int64_t di = fwd ? 1 : -1;
            const double eps_s = fwd ? eps : -eps;
            const uint32_t nleaf = 1u << depth;

            bool invalid = false;
            double v_lsa = 0.0;
            int64_t v_steps = 0;
            for (uint32_t j = 0; j < nleaf && !invalid && !finished; ++j) {
                double lq_leaf, pi_leaf;
                bool pos_finite;
                if (one_product) leapfrog_leaf_dense1<T, NPL>(tgt, M, Dpad, lane, D, q, p, g, ps, u, eps_s, lq_leaf, pi_leaf, pos_finite);
                else leapfrog_leaf_dense<T, NPL>(tgt, M, Dpad, lane, D, q, p, g, ps, eps_s, lq_leaf, pi_leaf, pos_finite);
                if (!pos_finite) status |= DHMC_ST_NONFINITE_POSITION;
                i += di;
                total_steps += 1;
                const double delta = pi_leaf - pi0;
                v_lsa = delta < 0.0 ? delta : 0.0;
                v_steps = 1;
                int level = 0;
                if (delta < P.min_delta) {
                    term_left = term_right = i;
                    invalid = true;
                } else {
#pragma unroll
                    for (int k = 0; k < NPL; ++k) { cf[k] = p[k]; cfs[k] = ps[k]; cr[k] = p[k]; }
                    double c_omega = delta;
                    int c_zeta = -1;
                    for (;;) {
                        const bool sub = ((j >> level) & 1u) != 0;
                        const bool top = !sub && (j == nleaf - 1) && (level == depth);
                        if (!sub && !top) break;
                        auto a_cf = [&](int k) { return cf[k]; };
                        auto a_cfs = [&](int k) { return cfs[k]; };
                        auto a_p = [&](int k) { return p[k]; };
                        auto a_ps = [&](int k) { return ps[k]; };
                        auto a_cr = [&](int k) { return cr[k]; };
                        bool turning;
                        if (sub) {
                            // suspended summary in build order: first, first♯, last, last♯, ρ
                            auto lf = row_acc(wd_stack(level, 0)), lfs = row_acc(wd_stack(level, 1));
                            auto ll = row_acc(wd_stack(level, 2)), lls = row_acc(wd_stack(level, 3));
                            auto lr = row_acc(wd_stack(level, 4));
                            turning = fwd ? merge_core_dense<NPL>(lf, lfs, ll, lls, lr, a_cf, a_cfs, a_p, a_ps, a_cr, lf, lfs, cf, cfs, cr)
                                          : merge_core_dense<NPL>(a_p, a_ps, a_cf, a_cfs, a_cr, ll, lls, lf, lfs, lr, lf, lfs, cf, cfs, cr);
                            const double wl = lv_omega[level];
                            double w;
                            logaddexp_pair(lv_vlsa[level], v_lsa, wl, c_omega, lane, v_lsa, w);
                            v_steps += (int64_t)lv_vsteps[level];
                            if (turning) {
                                term_left = i - di * (((int64_t)2 << level) - 1);
                                term_right = i;
                                invalid = true;
                                level += 1;
                                break;
                            }
                            const double logprob2 = c_omega - w;
                            const bool pick = logprob2 >= 0.0 || (randexp() > -logprob2);
                            const int lz = lv_zeta[level];
                            if (pick) {
                                free_mask |= (1ull << lz);
                            } else {
                                if (c_zeta >= 0) free_mask |= (1ull << c_zeta);
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
                            double w;
                            logaddexp_pair(vtop_lsa, v_lsa, omega_top, c_omega, lane, vtop_lsa, w);
                            vtop_steps += v_steps;
                            const double logprob2 = c_omega - omega_top;
                            const bool pick = logprob2 >= 0.0 || (randexp() > -logprob2);
                            if (pick) {
                                if (c_zeta < 0) c_zeta = save_leaf(lq_leaf, pi_leaf);
                                if (zeta_top != init_slot) free_mask |= (1ull << zeta_top);
                                zeta_top = c_zeta;
                            } else if (c_zeta >= 0) {
                                free_mask |= (1ull << c_zeta);
                            }
                            omega_top = w;
                            depth += 1;
                            if (fwd) i_plus = i; else i_minus = i;
                            if (turning) {
                                term_left = i_minus;
                                term_right = i_plus;
                                finished = true;
                            } else if (depth < max_depth) {
                                stv<NPL>(wsv(wd_top(fwd ? 2 : 0)), lane, p);
                                stv<NPL>(wsv(wd_top(fwd ? 3 : 1)), lane, ps);
                                stv<NPL>(wsv(wd_top(4)), lane, cr);
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
                        lv_omega[level] = c_omega;
                        lv_vlsa[level] = v_lsa;
                        lv_vsteps[level] = (double)v_steps;
                        lv_zeta[level] = c_zeta;
                    }
                }
                if (invalid) {
                    for (int l2 = level; l2 < depth; ++l2) {
                        if ((j >> l2) & 1u) {
                            v_lsa = uni_f64(det_logaddexp_u(lv_vlsa[l2], v_lsa));
                            v_steps += (int64_t)lv_vsteps[l2];
                        }
                    }
                    vtop_lsa = uni_f64(det_logaddexp_u(vtop_lsa, v_lsa));
                    vtop_steps += v_steps;
                    finished = true;
                }
            }
        }

        const double acc_rate = [&]() {
            double a = det_exp_u(vtop_lsa) / (double)vtop_steps;
            return uni_f64(a < 1.0 ? a : 1.0);
        }();
        init_slot = zeta_top;
        ldv<NPL>(wsv(wd_slot(max_depth, init_slot, 0)), lane, q);
        if constexpr (T::kRecomputeGrad) (void)tgt.eval(q, g, lane, D);
        else ldv<NPL>(wsv(wd_slot(max_depth, init_slot, 1)), lane, g);
        lq_cur = uni_f64(sl_lq[init_slot]);
        const double pi_stat = uni_f64(sl_pi[init_slot]);

        const size_t o = (size_t)chain * P.N + n;
        if (P.out.draws) {
            double* drow = P.out.draws + o * D;
#pragma unroll
            for (int k = 0; k < NPL; ++k)
                if (lane + WAVE * k < D) drow[lane + WAVE * k] = q[k];
        }
        if (lane == 0) {
            if (P.out.logdensities) P.out.logdensities[o] = lq_cur;
            if (P.out.eps) P.out.eps[o] = eps;
            if (P.out.pi) P.out.pi[o] = pi_stat;
            if (P.out.acceptance_rate) P.out.acceptance_rate[o] = acc_rate;
            if (P.out.steps) P.out.steps[o] = vtop_steps;
            if (P.out.term_left) P.out.term_left[o] = term_left;
            if (P.out.term_right) P.out.term_right[o] = term_right;
            if (P.out.depth) P.out.depth[o] = depth;
            if (P.out.directions) P.out.directions[o] = directions0;
        }

        if (P.adapt) {
            da.m += 1;
            const double m = (double)da.m;
            da.Hbar += (P.delta - acc_rate - da.Hbar) / (m + (double)P.t0);
            da.logeps = da.mu - __builtin_sqrt(m) / P.gamma * da.Hbar;
            da.logeps_bar += det_pow_pos_u(m, -P.kappa) * (da.logeps - da.logeps_bar);
        }
    }

    stv<NPL>(P.st.q + row, lane, q);
    stv<NPL>(P.st.g + row, lane, g);
    if (lane == 0) {
        P.st.lq[chain] = lq_cur;
        if (P.adapt) {
            P.st.da[chain] = da;
            if (P.da_finalize) P.st.eps[chain] = det_exp_u(da.logeps_bar);
        }
        P.st.transition[chain] = tr0 + (uint32_t)P.N;
        P.st.status[chain] = status;
        if (P.leapfrog_counter) atomicAdd(P.leapfrog_counter, total_steps);
        if (P.chain_work) P.chain_work[chain] = (unsigned)(total_steps > 0xffffffffull ? 0xffffffffull : total_steps);
    }
}

// warmup(::InitialStepsizeSearch) with a dense metric (mcmc.jl:134-148 -> stepsize.jl:46-85)
template <class T, int NPL>
__global__ __launch_bounds__(64, 1) void stepsize_search_dense_kernel(SearchParams P, DenseMetric Mall) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const DenseMetric M = Mall.of_chain(chain);
    const int D = P.D, Dpad = P.Dpad;
    const T tgt(P.tp);
    const size_t row = (size_t)chain * Dpad;
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    double q0[NPL], p0[NPL], g0[NPL], q[NPL], p[NPL], g[NPL], ps[NPL];
    ldv<NPL>(P.st.q + row, lane, q0);
    ldv<NPL>(P.st.g + row, lane, g0);
    const double lq0 = P.st.lq[chain];
    uint32_t status = P.st.status[chain];
    sample_momentum_dense<NPL>(key, PURPOSE_SEARCH_MOMENTUM, P.st.transition[chain], M, Dpad, D, lane, p0, ps);
    LaneAcc<1, NPL> kacc;
#pragma unroll
    for (int k = 0; k < NPL; ++k) kacc.add(0, k, p0[k], ps[k]);
    const double l0 = uni_f64(joint_logdensity(lq0, wave_allreduce1(kacc.fold(0)) / 2.0));
    if (!dm_isfinite(l0)) {
        if (lane == 0) P.st.status[chain] = status | DHMC_ST_NONFINITE_START_DENSITY;
        return;
    }
    auto A = [&](double eps) -> double {
#pragma unroll
        for (int k = 0; k < NPL; ++k) { q[k] = q0[k]; p[k] = p0[k]; g[k] = g0[k]; }
        double lq1, pi1;
        bool pfin;
        leapfrog_leaf_dense<T, NPL>(tgt, M, Dpad, lane, D, q, p, g, ps, eps, lq1, pi1, pfin);
        if (!pfin) status |= DHMC_ST_NONFINITE_POSITION;
        return pi1 - l0;
    };
    double eps = P.initial_eps;
    const double Ae = A(eps);
    const bool dbl = Ae > P.log_threshold;
    bool found = false;
    for (int it = 0; it < P.maxiter; ++it) {
        const double eps1 = dbl ? 2 * eps : eps / 2;
        const double Ae1 = A(eps1);
        if (dbl ? (Ae1 < P.log_threshold) : (Ae1 > P.log_threshold)) {
            eps = eps1;
            found = true;
            break;
        }
        eps = eps1;
    }
    if (!found) status |= DHMC_ST_STEPSIZE_SEARCH_FAILED;
    if (lane == 0) {
        P.st.eps[chain] = eps;
        P.st.status[chain] = status;
    }
}

}  // namespace dhmc
