// The per-draw loop kernel for chains of 512+ coordinates (BASELINE config 2: D = 1000): a WORKGROUP per chain,
// one VECTOR wave per 256-coordinate block of the chain plus one CONTROL wave.
//
// Why.  The one-wave-per-chain kernel (nuts_kernels.hpp) holds a 1000-dim chain as 16 slots per lane: eight live
// D-vectors fill all 512 registers of a SIMD lane, so a SIMD runs ONE wave and every dependent-latency chain of the
// tree logic (the 45-deep deterministic logaddexp, the DPP butterflies, LDS and L2 round trips) is exposed: the
// kernel sat at 49 % VALU-active / 10 % of the fp64 vector peak.  The chip's on-chip capacity still allows only four
// such chains per CU (8 vectors × 8 KB + LDS each), so the way to hide latency is not more chains per CU but the
// SAME four chains spread over all four SIMDs: each chain's vector work is cut into NW = Dpad/256 waves of four
// slots per lane (≈ 90 VGPRs), and every SIMD holds waves of several different chains that cover for each other.
//
//  * vector wave w owns coordinates 256 w + lane + 64 k, k = 0..3 — exactly one block of the ABI's dot product
//    (include/dhmc.h "Summation order", wave.hpp LaneAcc), so its per-lane partial sums are final: no wave-to-wave
//    fma chain.  It keeps (q, p, [∇ℓ]), the running subtree summary (first, Σp), the trajectory's (p₋, p₊, ρ) and
//    its part of M⁻¹ in registers, levels 0 and 1 of the suspended stack in LDS, deeper levels / parked edges /
//    proposal slots in the HBM workspace, and only ever touches its own coordinates of any of those;
//  * the control wave owns everything scalar of sample_tree (reference src/NUTS.jl:232-241, src/trees.jl:231-319):
//    it folds the blocks' partial sums and does the 64-lane butterfly, then the leaf's Δ / divergence test, the
//    acceptance statistic, both logaddexp's of a merge, the Exp(1) draws and proposal selection, slot bookkeeping,
//    termination, dual averaging and the per-transition outputs.  The two logaddexp's of a merge do not depend on
//    the dots, so they run WHILE the vector waves compute that merge's partial sums;
//  * they meet at SYNCHRONISATION POINTS (one per leaf, one per merge, one per transition start): the vector waves
//    leave partial sums in LDS, barrier, the control wave reduces and publishes a 32-bit VERDICT (divergent /
//    turning, a proposal slot to materialise the leaf in, the trajectory's current proposal), barrier.  Both sides
//    run the same replicated integer control (direction bits, depth, leaf counter, cascade level) from the verdicts,
//    so they execute the same barrier sequence by construction.
//
// Arithmetic, RNG consumption and every output bit equal nuts_run_kernel's and the oracle's (DHMC_MW=0 selects the
// one-wave kernel; tests/test_gpu_engines.py compares the two).
#pragma once
#include "nuts_kernels.hpp"

namespace dhmc {

constexpr int MW_NK = 4;                                     // slots per lane of a vector wave: one 256-coordinate block
enum : uint32_t { MWV_DIV = 1u, MWV_TURN = 2u, MWV_SCAN = 4u };
// verdict: bits 0-7 flags; 8-15: 1 + proposal slot the leaf held in registers must be written to now (0: none);
//          16-23: ζ of the whole trajectory (the slot that becomes the chain's position if the transition ends here)

// LDS of one chain: five Dpad-rows (suspended levels 0 and 1, the trajectory's two edge momenta), the vector waves'
// partial sums, and the control wave's per-level / per-slot scalars (sized by max_depth): 53.8 KB at Dpad = 1024,
// max_depth = 10 — three chains (15 waves) per CU.
__host__ __device__ inline int mw_nlev(int max_depth) { return max_depth; }
__host__ __device__ inline int mw_nslot(int max_depth) { return ws_nslots(max_depth); }
__host__ __device__ inline size_t mw_lds_bytes(int Dpad, int NW, int max_depth) {
    return sizeof(double) * ((size_t)5 * Dpad + 6 * NW * WAVE + 3 * mw_nlev(max_depth) + 2 * mw_nslot(max_depth) + 2) +
           sizeof(int) * (mw_nlev(max_depth) + 8);
}

struct MwLds {
    double *l0, *l1f, *l1l;          // [Dpad] each: suspended level 0 momentum, level 1 (first, last)
    double *tpm, *tpp;               // [Dpad] each: p₋ and p₊ of the whole trajectory (its ρ stays in registers)
    double* part;                    // [6][NW][64] partial sums of the vector waves
    double *lv_omega, *lv_vlsa, *lv_vsteps, *sl_lq, *sl_pi;   // control wave's per-level / per-slot scalars
    double* mb_f;                    // mailbox: [0] ϵ of the transition
    int* lv_zeta;
    uint32_t* mb_u;                  // mailbox: [0] verdict, [1] directions, [2 + w] position-scan flag of vector wave w
};

__device__ __forceinline__ MwLds mw_carve(double* lds, int Dpad, int NW, int max_depth) {
    const int nlev = mw_nlev(max_depth), nslot = mw_nslot(max_depth);
    MwLds L;
    L.l0 = lds;
    L.l1f = lds + Dpad;
    L.l1l = lds + 2 * Dpad;
    L.tpm = lds + 3 * Dpad;
    L.tpp = lds + 4 * Dpad;
    L.part = lds + 5 * Dpad;
    L.lv_omega = L.part + 6 * NW * WAVE;
    L.lv_vlsa = L.lv_omega + nlev;
    L.lv_vsteps = L.lv_vlsa + nlev;
    L.sl_lq = L.lv_vsteps + nlev;
    L.sl_pi = L.sl_lq + nslot;
    L.mb_f = L.sl_pi + nslot;
    L.lv_zeta = (int*)(L.mb_f + 2);
    L.mb_u = (uint32_t*)(L.lv_zeta + nlev);
    return L;
}

// Workgroup barrier that orders LDS traffic only (outstanding global stores — draws, proposal slots — keep flying).
__device__ __forceinline__ void mw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NK>
__device__ __forceinline__ void ldk(const double* __restrict__ at, double (&v)[NK]) {
#pragma unroll
    for (int k = 0; k < NK; ++k) v[k] = at[WAVE * k];
}
template <int NK>
__device__ __forceinline__ void stk(double* __restrict__ at, const double (&v)[NK]) {
#pragma unroll
    for (int k = 0; k < NK; ++k) at[WAVE * k] = v[k];
}

// combine_turn_statistics (NUTS.jl:132-139) of two adjacent subtrees on one block: x = earlier in time, y = later, each
// given as accessors k -> slot k of its (p₋, p₊, ρ); nf = the merged summary's build-order first momentum.  Leaves this
// block's six partial sums in LDS; cf <- nf, cr <- ρ of the merge.  (merge_core of nuts_kernels.hpp without the reduction.)
template <class XM, class XP, class XR, class YM, class YP, class YR, class NF>
__device__ __forceinline__ void mw_merge_core(XM xm_, XP xp_, XR xr_, YM ym_, YP yp_, YR yr_, NF nf_, const double (&m)[MW_NK],
                                              double (&cf)[MW_NK], double (&cr)[MW_NK], double* __restrict__ my_part, int PS) {
    double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < MW_NK; ++k) {
        const double xm = xm_(k), xp = xp_(k), xr = xr_(k);
        const double ym = ym_(k), yp = yp_(k), yr = yr_(k);
        const double nf = nf_(k);
        const double mk = m[k];
        const double s1 = xr + ym;      // x.ρ + y.p₋      (:134)
        const double s2 = xp + yr;      // x.p₊ + y.ρ      (:135)
        const double r = xr + yr;       // ρ               (:136)
        const double pa = mk * xm;      // x.p♯₋
        const double pb = mk * ym;      // y.p♯₋
        const double pc = mk * xp;      // x.p♯₊
        const double pd = mk * yp;      // y.p♯₊
        a[0] = __builtin_fma(pa, s1, a[0]);
        a[1] = __builtin_fma(pb, s1, a[1]);
        a[2] = __builtin_fma(pc, s2, a[2]);
        a[3] = __builtin_fma(pd, s2, a[3]);
        a[4] = __builtin_fma(pa, r, a[4]);
        a[5] = __builtin_fma(pd, r, a[5]);
        cf[k] = nf;
        cr[k] = r;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) my_part[i * PS] = a[i];
}

// ------------------------------------------------------------------------------------------------------------
// Vector wave `wv` of the chain: every D-vector operation of the transition on coordinates 256 wv + lane + 64 k.
// ------------------------------------------------------------------------------------------------------------
template <class T, int NW>
__device__ __forceinline__ void mw_vector_wave(const RunParams& P, const MwLds& L, const int chain, const int wv, const int lane) {
    constexpr int NK = MW_NK;
    const int D = P.D, Dpad = P.Dpad, max_depth = P.max_depth;
    const int eb = WAVE * NK * wv + lane;                        // this lane's first coordinate
    const T tgt(P.tp);
    const size_t row = (size_t)chain * Dpad + eb;
    double* const ws = P.st.ws + (size_t)chain * P.nvec * Dpad + eb;
    auto wsv = [&](int idx) -> double* { return ws + (size_t)idx * Dpad; };
    const double* const Wrow = P.st.W + row;
    double* const l0 = L.l0 + eb;
    double* const l1f = L.l1f + eb;
    double* const l1l = L.l1l + eb;
    double* const tpm = L.tpm + eb;                              // trajectory edges p₋, p₊: this lane's slots at [64 k]
    double* const tpp = L.tpp + eb;
    double* const my_part = L.part + wv * WAVE + lane;           // value n at my_part[n * PS]
    constexpr int PS = NW * WAVE;
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    const uint32_t idx_base = (uint32_t)(lane + 2 * WAVE * wv);  // momentum stream call index of slots (0,1); +64 for (2,3)
    const uint32_t tr0 = P.st.transition[chain];

    double m[NK], q[NK], p[NK], g[NK], cf[NK], cr[NK], trho[NK];
    ldk<NK>(P.st.minv + row, m);
    ldk<NK>(P.st.q + row, q);
    ldk<NK>(P.st.g + row, g);
#pragma unroll
    for (int k = 0; k < NK; ++k) { cf[k] = 0.0; cr[k] = 0.0; }

    auto sync = [&]() -> uint32_t {   // partial sums are in LDS: wait for the control wave's verdict
        mw_barrier();
        mw_barrier();
        return uni_u32(L.mb_u[0]);
    };

    int init_slot = 0;
    stk<NK>(wsv(ws_slot(max_depth, init_slot, 0)), q);

    for (int64_t n = 0; n < P.N; ++n) {
        const uint32_t tr = tr0 + (uint32_t)n;
        // ---- rand_p (hamiltonian.jl:124) and this block's part of K(p) for π₀ ---------------------------
#pragma unroll
        for (int kk = 0; kk < NK / 2; ++kk) {
            uint64_t r1, r2;
            stream_raw64(key, idx_base + (uint32_t)(WAVE * kk), PURPOSE_MOMENTUM, tr, r1, r2);
            double z0, z1;
            det_randn2(r1, r2, &z0, &z1);
            p[2 * kk] = Wrow[WAVE * (2 * kk)] * z0;
            p[2 * kk + 1] = Wrow[WAVE * (2 * kk + 1)] * z1;
        }
        {
            double kacc = 0.0;
#pragma unroll
            for (int k = 0; k < NK; ++k) kacc = __builtin_fma(p[k], m[k] * p[k], kacc);
            my_part[0] = kacc;
        }
        mw_barrier();
        mw_barrier();
        const double eps = uni_f64(L.mb_f[0]);
        uint32_t dirs = uni_u32(L.mb_u[1]);
        stk<NK>(tpm, p);                                                   // leaf τ of z₀ (NUTS.jl:120-123)
        stk<NK>(tpp, p);
#pragma unroll
        for (int k = 0; k < NK; ++k) trho[k] = p[k];

        // ---- sample_trajectory (trees.jl:283-319) ------------------------------------------------------
        bool stored0 = false, stored1 = false;
        int reg_edge = 2;
        int depth = 0;
        bool finished = false;
        uint32_t verdict = 0;
        while (!finished && depth < max_depth) {
            const bool fwd = (dirs & 1u) != 0;
            dirs >>= 1;
            const int dir = fwd ? 1 : 0;
            if (reg_edge != 2 && reg_edge != dir) {
                stk<NK>(wsv(ws_edge(reg_edge, 0)), q);                      // park the edge we leave ...
                if (reg_edge == 1) stored1 = true; else stored0 = true;
                const bool have = fwd ? stored1 : stored0;
                ldk<NK>(wsv(have ? ws_edge(dir, 0) : ws_slot(max_depth, init_slot, 0)), q);   // ... fetch the one we extend
                if constexpr (!T::kPointwiseGrad) (void)tgt.eval(q, g, eb, D);
                ldk<NK>(fwd ? tpp : tpm, p);
            }
            reg_edge = dir;
            const double eps_s = fwd ? eps : -eps;
            const double h = eps_s / 2;
            const uint32_t nleaf = 1u << depth;
            bool invalid = false;
            for (uint32_t j = 0; j < nleaf && !invalid && !finished; ++j) {
                // ---- leapfrog (hamiltonian.jl:273-282) on this block ---------------------------------
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    double gk;
                    if constexpr (T::kPointwiseGrad) gk = tgt.grad1(q[k], eb + WAVE * k);
                    else gk = g[k];
                    const double pm = p[k] + h * gk;                        // :277
                    const double t = m[k] * pm;
                    q[k] = q[k] + eps_s * t;                                // :278
                    p[k] = pm;
                }
                const double lres = tgt.eval(q, g, eb, D);                  // :279
                double kacc = 0.0;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    p[k] = p[k] + h * g[k];                                 // :280
                    kacc = __builtin_fma(p[k], m[k] * p[k], kacc);
                }
                my_part[0] = lres;
                my_part[PS] = kacc;
                verdict = sync();
                if (verdict & MWV_SCAN) {   // ℓq came out non-finite: evaluate_ℓ's position scan (hamiltonian.jl:203)
                    bool fin = true;
#pragma unroll
                    for (int k = 0; k < NK; ++k) fin = fin && dm_isfinite(q[k]);
                    const bool all = wave_all(fin);
                    if (lane == 0) L.mb_u[2 + wv] = all ? 1u : 0u;
                    verdict = sync();
                }
                int level = 0;
                if (verdict & MWV_DIV) {
                    invalid = true;
                } else {
                    for (;;) {
                        const bool sub = ((j >> level) & 1u) != 0;
                        const bool top = !sub && (j == nleaf - 1) && (level == depth);
                        if (!sub && !top) break;
                        const bool leafleaf = sub ? (level == 0) : (depth == 0);
                        if (leafleaf) {
                            // both subtrees are single leaves (merge_leaf_leaf of nuts_kernels.hpp): two distinct dots
                            double a0 = 0.0, a1 = 0.0;
#pragma unroll
                            for (int k = 0; k < NK; ++k) {
                                const double pa = sub ? l0[WAVE * k] : trho[k];
                                const double r = pa + p[k];
                                a0 = __builtin_fma(m[k] * pa, r, a0);
                                a1 = __builtin_fma(m[k] * p[k], r, a1);
                                cf[k] = pa;
                                cr[k] = r;
                            }
                            my_part[0] = a0;
                            my_part[PS] = a1;
                        } else {
                            // combine_turn_statistics (NUTS.jl:132-139): x earlier in time, y later (trees.jl:135-141)
                            auto a_cf = [&](int k) { return cf[k]; };
                            auto a_p = [&](int k) { return p[k]; };
                            auto a_cr = [&](int k) { return cr[k]; };
                            if (sub) {
                                if (level == 1) {
                                    auto a_lf = [&](int k) { return l1f[WAVE * k]; };
                                    auto a_ll = [&](int k) { return l1l[WAVE * k]; };
                                    auto a_lr = [&](int k) { return l1f[WAVE * k] + l1l[WAVE * k]; };
                                    if (fwd) mw_merge_core(a_lf, a_ll, a_lr, a_cf, a_p, a_cr, a_lf, m, cf, cr, my_part, PS);
                                    else mw_merge_core(a_p, a_cf, a_cr, a_ll, a_lf, a_lr, a_lf, m, cf, cr, my_part, PS);
                                } else {
                                    const double* Lf = wsv(ws_stack(level, 0));
                                    const double* Ll = wsv(ws_stack(level, 1));
                                    const double* Lr = wsv(ws_stack(level, 2));
                                    auto a_lf = [&](int k) { return Lf[WAVE * k]; };
                                    auto a_ll = [&](int k) { return Ll[WAVE * k]; };
                                    auto a_lr = [&](int k) { return Lr[WAVE * k]; };
                                    if (fwd) mw_merge_core(a_lf, a_ll, a_lr, a_cf, a_p, a_cr, a_lf, m, cf, cr, my_part, PS);
                                    else mw_merge_core(a_p, a_cf, a_cr, a_ll, a_lf, a_lr, a_lf, m, cf, cr, my_part, PS);
                                }
                            } else {
                                // top level (trees.jl:294-316): τ of the whole trajectory is time-ordered (tpm, tpp, trho)
                                auto a_tm = [&](int k) { return tpm[WAVE * k]; };
                                auto a_tp = [&](int k) { return tpp[WAVE * k]; };
                                auto a_tr = [&](int k) { return trho[k]; };
                                if (fwd) mw_merge_core(a_tm, a_tp, a_tr, a_cf, a_p, a_cr, a_cf, m, cf, cr, my_part, PS);
                                else mw_merge_core(a_p, a_cf, a_cr, a_tm, a_tp, a_tr, a_cf, m, cf, cr, my_part, PS);
                            }
                        }
                        verdict = sync();
                        const bool turning = (verdict & MWV_TURN) != 0;
                        if (sub) {
                            level += 1;
                            if (turning) { invalid = true; break; }      // trees.jl:255
                        } else {
                            const uint32_t slot = (verdict >> 8) & 0xffu;  // the new leaf won the doubling: materialise it
                            if (slot) stk<NK>(wsv(ws_slot(max_depth, (int)slot - 1, 0)), q);
                            depth += 1;
                            if (turning) {                                 // trees.jl:315-316
                                finished = true;
                            } else if (depth < max_depth) {
                                stk<NK>(fwd ? tpp : tpm, p);                // τ of the doubled trajectory: the new edge, Σp
#pragma unroll
                                for (int k = 0; k < NK; ++k) trho[k] = cr[k];
                            }
                            level = -1;
                            break;
                        }
                    }
                    if (level >= 0 && !invalid) {
                        // suspend the running subtree at `level` until its right sibling is built
                        const uint32_t slot = (verdict >> 8) & 0xffu;
                        if (slot) stk<NK>(wsv(ws_slot(max_depth, (int)slot - 1, 0)), q);
                        if (level == 0) {
                            stk<NK>(l0, p);
                        } else if (level == 1) {
                            stk<NK>(l1f, cf);
                            stk<NK>(l1l, p);
                        } else {
                            stk<NK>(wsv(ws_stack(level, 0)), cf);
                            stk<NK>(wsv(ws_stack(level, 1)), p);
                            stk<NK>(wsv(ws_stack(level, 2)), cr);
                        }
                    }
                }
                if (invalid) finished = true;                               // trees.jl:297
            }
        }

        // ---- the new position (NUTS.jl:238-240) and the draw (mcmc.jl:275,376) ---------------------------
        init_slot = (int)((verdict >> 16) & 0xffu);
        ldk<NK>(wsv(ws_slot(max_depth, init_slot, 0)), q);
        if constexpr (!T::kPointwiseGrad) (void)tgt.eval(q, g, eb, D);
        if (P.out.draws) {
            double* drow = P.out.draws + ((size_t)chain * P.N + n) * D + eb;
#pragma unroll
            for (int k = 0; k < NK; ++k)
                if (eb + WAVE * k < D) drow[WAVE * k] = q[k];
        }
    }
    stk<NK>(P.st.q + row, q);
    if constexpr (T::kPointwiseGrad) (void)tgt.eval(q, g, eb, D);
    stk<NK>(P.st.g + row, g);
}

// ------------------------------------------------------------------------------------------------------------
// Control wave: every scalar of the transition.
// ------------------------------------------------------------------------------------------------------------
template <int N, int NW>
__device__ __forceinline__ void mw_reduce(const double* __restrict__ part, int lane, double (&r)[N]) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
        double t[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) t[w] = part[(n * NW + w) * WAVE + lane];
#pragma unroll
        for (int s = 1; s < NW; s *= 2)
#pragma unroll
            for (int w = 0; w + s < NW; w += 2 * s) t[w] = t[w] + t[w + s];   // blocks folded per lane, adjacent pairs
        r[n] = t[0];
    }
    wave_allreduce<N>(r);
}

template <class T, int NW>
__device__ __forceinline__ void mw_control_wave(const RunParams& P, const MwLds& L, const int chain, const int lane) {
    const int max_depth = P.max_depth;
    const T tgt(P.tp);
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    const int nslots = ws_nslots(max_depth);
    double lq_cur = uni_f64(P.st.lq[chain]);
    const double eps_fixed = uni_f64(P.st.eps[chain]);
    DAState da = P.st.da[chain];
    uint32_t status = uni_u32(P.st.status[chain]);
    const uint32_t tr0 = uni_u32(P.st.transition[chain]);
    unsigned long long total_steps = 0;
    if (P.adapt && P.da_init) {  // initial_adaptation_state (stepsize.jl:134-138; mcmc.jl:266)
        const double le = det_log(eps_fixed);
        da.mu = det_log(10.0) + le;
        da.m = 1;
        da.Hbar = 0.0;
        da.logeps = le;
        da.logeps_bar = 0.0;
    }
    int init_slot = 0;
    uint64_t free_mask = 0;
    auto publish = [&](uint32_t verdict) {
        if (lane == 0) L.mb_u[0] = verdict;
        mw_barrier();
    };
    auto alloc_slot = [&](double lq_leaf, double pi_leaf) -> int {   // bookkeeping of save_leaf: the vector waves store q
        const int s = __builtin_ctzll(free_mask);
        free_mask &= ~(1ull << s);
        if (lane == 0) { L.sl_lq[s] = lq_leaf; L.sl_pi[s] = pi_leaf; }
        return s;
    };

    for (int64_t n = 0; n < P.N; ++n) {
        const uint32_t tr = tr0 + (uint32_t)n;
        const double eps = uni_f64(P.adapt ? det_exp(da.logeps) : eps_fixed);  // current_ϵ (stepsize.jl:163)
        uint32_t dirs;
        {
            uint32_t w[4];
            philox4x32_10(0u, PURPOSE_DIRECTIONS, tr, key.seed_hi, key.k0, key.k1, w);
            dirs = uni_u32(w[0]);
        }
        const uint32_t directions0 = dirs;
        if (lane == 0) {
            L.mb_f[0] = eps;
            L.mb_u[1] = dirs;
            L.sl_lq[init_slot] = lq_cur;
        }
        // Exp(1) draws of this transition, 64 at a time: lane l holds draw (rexp_base + l)
        uint32_t nrand = 0, rexp_base = 0;
        double rexp_vals;
        auto rexp_fill = [&](uint32_t base) {
            uint64_t r1, r2;
            stream_raw64(key, base + (uint32_t)lane, PURPOSE_TREE, tr, r1, r2);
            rexp_vals = det_randexp(r1);
            rexp_base = base;
        };
        rexp_fill(0);
        auto randexp = [&]() -> double {  // Random.randexp at NUTS.jl:44
            if (nrand - rexp_base >= 64u) rexp_fill(nrand & ~63u);
            const double v = readlane_f64(rexp_vals, (int)(nrand & 63u));
            nrand += 1;
            return v;
        };
        double pi0;
        mw_barrier();                                   // the blocks' parts of p·M⁻¹p are in LDS
        {
            double r[1];
            mw_reduce<1, NW>(L.part, lane, r);
            pi0 = uni_f64(joint_logdensity(lq_cur, r[0] / 2.0));
            if (lane == 0) L.sl_pi[init_slot] = pi0;
        }
        mw_barrier();                                   // ϵ and the directions are published

        free_mask = ((nslots >= 64) ? ~0ull : ((1ull << nslots) - 1ull)) & ~(1ull << init_slot);
        int zeta_top = init_slot;
        double omega_top = 0.0;
        double vtop_lsa = -dm_inf();
        int64_t vtop_steps = 0;
        int depth = 0;
        int64_t i_minus = 0, i_plus = 0;
        int64_t term_left = 1, term_right = 0;  // REACHED_MAX_DEPTH
        bool finished = false;
        while (!finished && depth < max_depth) {
            const bool fwd = (dirs & 1u) != 0;  // next_direction (trees.jl:31-34)
            dirs >>= 1;
            int64_t i = fwd ? i_plus : i_minus;
            const int64_t di = fwd ? 1 : -1;
            const uint32_t nleaf = 1u << depth;
            bool invalid = false;
            double v_lsa = 0.0;
            int64_t v_steps = 0;
            for (uint32_t j = 0; j < nleaf && !invalid && !finished; ++j) {
                // ---- the leaf (NUTS.jl:148-159) ---------------------------------------------------------
                mw_barrier();
                double lq_leaf, pi_leaf;
                {
                    double r[2];
                    mw_reduce<2, NW>(L.part, lane, r);
                    double lq = uni_f64(tgt.finish(r[0]));
                    const double K = r[1] / 2.0;
                    bool pos_finite = true;
                    if (!dm_isfinite(lq)) {      // evaluate_ℓ's position scan is the vector waves' (hamiltonian.jl:203)
                        publish(MWV_SCAN);
                        mw_barrier();
                        uint32_t all = 1u;
#pragma unroll
                        for (int w = 0; w < NW; ++w) all &= uni_u32(L.mb_u[2 + w]);
                        pos_finite = all != 0u;
                    }
                    lq = demote_lq(lq, pos_finite, true);
                    if (!pos_finite) status |= DHMC_ST_NONFINITE_POSITION;
                    lq_leaf = lq;
                    pi_leaf = uni_f64(joint_logdensity(lq, K));
                }
                i += di;
                total_steps += 1;
                const double delta = pi_leaf - pi0;             // NUTS.jl:150
                v_lsa = delta < 0.0 ? delta : 0.0;              // min(Δ, 0)   (NUTS.jl:79)
                v_steps = 1;
                int level = 0;
                const bool div = delta < P.min_delta;           // divergent leaf (NUTS.jl:151; trees.jl:236-237)
                publish((div ? MWV_DIV : 0u) | ((uint32_t)(__builtin_ctzll(free_mask) + 1) << 8) | ((uint32_t)zeta_top << 16));
                if (div) {
                    term_left = term_right = i;
                    invalid = true;
                } else {
                    double c_omega = delta;
                    int c_zeta = -1;  // -1: the proposal is the leaf the vector waves hold in registers
                    for (;;) {
                        const bool sub = ((j >> level) & 1u) != 0;
                        const bool top = !sub && (j == nleaf - 1) && (level == depth);
                        if (!sub && !top) break;
                        const bool leafleaf = sub ? (level == 0) : (depth == 0);
                        // v = v₋ ⊕ v₊ (trees.jl:249 / :294) and ω = logaddexp(ω₋, ω₊) (trees.jl:145): independent of the
                        // dots, so computed while the vector waves are still producing them
                        double v_new, w;
                        if (sub) logaddexp_pair(uni_f64(L.lv_vlsa[level]), v_lsa, uni_f64(L.lv_omega[level]), c_omega, lane, v_new, w);
                        else logaddexp_pair(vtop_lsa, v_lsa, omega_top, c_omega, lane, v_new, w);
                        mw_barrier();
                        bool turning;
                        if (leafleaf) {
                            double r[2];
                            mw_reduce<2, NW>(L.part, lane, r);
                            turning = r[0] < 0 || r[1] < 0;
                        } else {
                            double r[6];
                            mw_reduce<6, NW>(L.part, lane, r);
                            turning = r[0] < 0 || r[1] < 0 || r[2] < 0 || r[3] < 0 || r[4] < 0 || r[5] < 0;
                        }
                        if (sub) {
                            v_lsa = v_new;
                            v_steps += (int64_t)uni_f64(L.lv_vsteps[level]);
                            if (turning) {                       // trees.jl:255
                                term_left = i - di * (((int64_t)2 << level) - 1);
                                term_right = i;
                                publish(MWV_TURN | ((uint32_t)zeta_top << 16));
                                invalid = true;
                                level += 1;
                                break;
                            }
                            // combine_proposals_and_logweights(…, is_doubling = false) (trees.jl:258)
                            const double logprob2 = c_omega - w;
                            const bool pick = logprob2 >= 0.0 || (randexp() > -logprob2);
                            const int lz = uni_i32(L.lv_zeta[level]);
                            if (pick) {
                                free_mask |= (1ull << lz);
                            } else {
                                if (c_zeta >= 0) free_mask |= (1ull << c_zeta);
                                c_zeta = lz;
                            }
                            c_omega = w;
                            level += 1;
                            publish((c_zeta < 0 ? (uint32_t)(__builtin_ctzll(free_mask) + 1) << 8 : 0u) | ((uint32_t)zeta_top << 16));
                        } else {
                            // top level (trees.jl:294-316)
                            vtop_lsa = v_new;
                            vtop_steps += v_steps;
                            const double logprob2 = c_omega - omega_top;   // biased progressive (trees.jl:159-161)
                            const bool pick = logprob2 >= 0.0 || (randexp() > -logprob2);
                            uint32_t save = 0;
                            if (pick) {
                                if (c_zeta < 0) {
                                    c_zeta = alloc_slot(lq_leaf, pi_leaf);
                                    save = (uint32_t)(c_zeta + 1);
                                }
                                if (zeta_top != init_slot) free_mask |= (1ull << zeta_top);
                                zeta_top = c_zeta;
                            } else if (c_zeta >= 0) {
                                free_mask |= (1ull << c_zeta);
                            }
                            omega_top = w;
                            depth += 1;
                            if (fwd) i_plus = i; else i_minus = i;
                            if (turning) {                       // trees.jl:315-316
                                term_left = i_minus;
                                term_right = i_plus;
                                finished = true;
                            }
                            publish((turning ? MWV_TURN : 0u) | (save << 8) | ((uint32_t)zeta_top << 16));
                            level = -1;  // handled
                            break;
                        }
                    }
                    if (level >= 0 && !invalid) {
                        // the running subtree is suspended at `level`: the vector waves store its vectors (and the leaf, into
                        // the slot announced by the last verdict = the lowest free one)
                        if (c_zeta < 0) c_zeta = alloc_slot(lq_leaf, pi_leaf);
                        if (lane == 0) {
                            L.lv_omega[level] = c_omega;
                            L.lv_vlsa[level] = v_lsa;
                            L.lv_vsteps[level] = (double)v_steps;
                            L.lv_zeta[level] = c_zeta;
                        }
                    }
                }
                if (invalid) {
                    // unwind the recursion: every suspended left sibling contributes its visited statistic (trees.jl:244,249-250)
                    for (int l2 = level; l2 < depth; ++l2) {
                        if ((j >> l2) & 1u) {
                            v_lsa = uni_f64(det_logaddexp(uni_f64(L.lv_vlsa[l2]), v_lsa));
                            v_steps += (int64_t)uni_f64(L.lv_vsteps[l2]);
                        }
                    }
                    vtop_lsa = uni_f64(det_logaddexp(vtop_lsa, v_lsa));   // trees.jl:294
                    vtop_steps += v_steps;
                    finished = true;                                       // trees.jl:297
                }
            }
        }

        // ---- TreeStatisticsNUTS (NUTS.jl:238-240), the per-draw scalars (mcmc.jl:273-277, 376-377) -------------
        const double acc_rate = [&]() {
            const double a = det_exp(vtop_lsa) / (double)vtop_steps;       // NUTS.jl:87
            return uni_f64(a < 1.0 ? a : 1.0);
        }();
        init_slot = zeta_top;
        lq_cur = uni_f64(L.sl_lq[init_slot]);
        const double pi_stat = uni_f64(L.sl_pi[init_slot]);
        if (lane == 0) {
            const size_t o = (size_t)chain * P.N + n;
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
        if (P.adapt) {  // adapt_stepsize (stepsize.jl:147-156)
            da.m += 1;
            const double m = (double)da.m;
            da.Hbar += (P.delta - acc_rate - da.Hbar) / (m + (double)P.t0);
            da.logeps = da.mu - __builtin_sqrt(m) / P.gamma * da.Hbar;
            da.logeps_bar += det_pow_pos(m, -P.kappa) * (da.logeps - da.logeps_bar);
        }
    }

    if (lane == 0) {
        P.st.lq[chain] = lq_cur;
        if (P.adapt) {
            P.st.da[chain] = da;
            if (P.da_finalize) P.st.eps[chain] = det_exp(da.logeps_bar);   // final_ϵ (stepsize.jl:170; mcmc.jl:285)
        }
        P.st.transition[chain] = tr0 + (uint32_t)P.N;
        P.st.status[chain] = status;
        if (P.leapfrog_counter) atomicAdd(P.leapfrog_counter, total_steps);
    }
}

template <class T, int NW>
__global__ __launch_bounds__(WAVE * (NW + 1), 4) void nuts_run_mw_kernel(RunParams P) {
    static_assert(T::kElementwise && T::kDeferred && T::kRecomputeGrad && T::kFiniteLqImpliesFiniteQ && T::kFiniteLqImpliesFiniteGrad,
                  "the multi-wave kernel serves coordinate-wise targets whose position scan is needed only on a non-finite ℓq");
    extern __shared__ double lds[];
    const MwLds L = mw_carve(lds, P.Dpad, NW, P.max_depth);
    const int chain = blockIdx.x;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = uni_i32((int)(threadIdx.x >> 6));
    if (wv == NW) mw_control_wave<T, NW>(P, L, chain, lane);
    else mw_vector_wave<T, NW>(P, L, chain, wv, lane);
}

}  // namespace dhmc
