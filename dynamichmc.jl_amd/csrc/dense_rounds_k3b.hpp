// K3 of the round engines (dense_rounds.hpp) as a WORKGROUP per chain: one wave per 256-coordinate BLOCK of the chain
// (NPL 8: two waves, 16: four, 32: eight, 64: sixteen), wave w owning slots k = 4w … 4w+3, for chains of 512+
// coordinates.
//
// Why: the one-wave-per-chain K3 needs the whole 512-register budget of a SIMD lane at NPL = 16 (one wave per SIMD),
// so it can neither hide its HBM latency behind other waves nor share a CU with the other half-batch's GEMM.  This
// form holds 5 × 4 doubles per lane and co-resides with the MFMA waves.  (As compiled since K2 was fused in: 211 VGPRs, two waves per SIMD;
// forcing three or four by launch bounds spills and is 8 / 15 % slower on config 3 — profiles/r06_gemm_lds_rotation.txt §9.)
//
// The bits do not change: the ABI's dot product (wave.hpp LaneAcc) is, per lane, one fma chain per 256-coordinate
// block, the blocks' partial sums combined per lane by an adjacent-pairs tree, then the 64-lane butterfly — so every
// wave runs its own block's chain in parallel with the others, leaves 64 partial sums in LDS, and after one barrier
// every wave folds the blocks and does the butterfly itself (block_allreduce): no serial hand-over between waves.
// Every thread executes the scalar tree logic redundantly on private copies of the chain's mutable scalars (a
// read-modify-write of the shared TreeState by four unsynchronised waves would race); arrays of the TreeState are
// only ever overwritten with values every thread computes identically.  Gradients of suspended points are always
// kept in their workspace slots here (no recomputation at the end of a transition: same bits, the functor is a
// whole-vector operation of one wave).
#pragma once
#include "dense_rounds.hpp"

namespace dhmc {

// waves per chain: one per block of four slots per lane (NPL 8 -> 2 waves, 16 -> 4, 32 -> 8, 64 -> 16 = a 1024-thread workgroup)
__host__ __device__ constexpr int k3b_waves(int NPL) { return NPL / 4; }

template <int N, int WPC, class Step>
__device__ __forceinline__ void block_allreduce(int wave, int lane, double (*xch)[WPC][WAVE], Step step, double (&out)[N]) {
    double acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = 0.0;
    step(acc);                                   // this wave's block: one fma chain per dot
#pragma unroll
    for (int n = 0; n < N; ++n) xch[n][wave][lane] = acc[n];
    __syncthreads();
#pragma unroll
    for (int n = 0; n < N; ++n) {
        double t[WPC];
#pragma unroll
        for (int w = 0; w < WPC; ++w) t[w] = xch[n][w][lane];
#pragma unroll
        for (int s = 1; s < WPC; s *= 2)
#pragma unroll
            for (int w = 0; w + s < WPC; w += 2 * s) t[w] = t[w] + t[w + s];
        out[n] = t[0];
    }
    wave_allreduce<N>(out);
    __syncthreads();                             // xch is free again
}

template <class T, int NPL>
__global__ __launch_bounds__(WAVE * k3b_waves(NPL)) void rounds_k3b_kernel(RunParams P, RoundBuffers R) {
    constexpr int K3B_WPC = k3b_waves(NPL);
    constexpr int NT = NPL / K3B_WPC;
    static_assert(NT == 4, "one 256-coordinate block of the ABI's dot product per wave");
    const int chain = P.chain_base + blockIdx.x;
    if (R.ts[chain].phase != PH_LEAF) return;
    __shared__ TreeState S;
    __shared__ double xch[6][K3B_WPC][WAVE];
    __shared__ double qedge[K3B_WPC][2];         // fused K2: first / last coordinate of every wave's block of q′
    __shared__ int fin_flag;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&R.ts[chain]);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&S);
        for (int w = tid; w < (int)(sizeof(TreeState) / 4); w += WAVE * K3B_WPC) dst[w] = src[w];
    }
    __syncthreads();
    const int D = P.D, Dpad = P.Dpad, max_depth = P.max_depth;
    const int off = WAVE * NT * wave;                           // first element of this wave's slots
    const size_t row = (size_t)chain * Dpad;
    double* const ws = P.st.ws + (size_t)chain * P.nvec * Dpad;
    auto wsv = [&](int idx) -> double* { return ws + (size_t)idx * Dpad + off; };   // this wave's part of a workspace row
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    auto row_acc = [&](int idx) { const double* r = wsv(idx); return [r, lane](int k) { return r[lane + WAVE * k]; }; };
    auto copy_row = [&](const double* src, double* dst) {      // both already offset to this wave's part
        double t[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) t[k] = src[lane + WAVE * k];
#pragma unroll
        for (int k = 0; k < NT; ++k) dst[lane + WAVE * k] = t[k];
    };
    double* const q_row = P.st.q + row + off;
    double* const g_row = P.st.g + row + off;
    double* const cp_row = R.cp + row + off;

    // private copies of the scalars this kernel changes — made provably wave-uniform (SGPRs, scalar branches)
    auto uni_i64 = [](int64_t x) -> int64_t {
        const uint32_t lo = uni_u32((uint32_t)(uint64_t)x), hi = uni_u32((uint32_t)((uint64_t)x >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    uint32_t nrand = uni_u32(S.nrand), status = uni_u32(S.status), dirs = uni_u32(S.dirs), jleaf = uni_u32(S.j);
    uint32_t nleaf = uni_u32(S.nleaf), tr = uni_u32(S.tr);
    int32_t depth = uni_i32(S.depth), dir = uni_i32(S.dir), reg_edge = uni_i32(S.reg_edge), stored0 = uni_i32(S.stored0);
    int32_t stored1 = uni_i32(S.stored1), zeta_top = uni_i32(S.zeta_top), init_slot = uni_i32(S.init_slot);
    int32_t n_done = uni_i32(S.n), phase = uni_i32(S.phase);
    int64_t i_plus = uni_i64(S.i_plus), i_minus = uni_i64(S.i_minus), term_left = uni_i64(S.term_left);
    int64_t term_right = uni_i64(S.term_right), vtop_steps = uni_i64(S.vtop_steps);
    uint64_t free_mask = (uint64_t)uni_i64((int64_t)S.free_mask);
    unsigned long long total_steps = (unsigned long long)uni_i64((int64_t)S.total_steps);
    double eps_s = uni_f64(S.eps_s), omega_top = uni_f64(S.omega_top), vtop_lsa = uni_f64(S.vtop_lsa), lq_cur = uni_f64(S.lq_cur);
    DAState da = S.da;

    double p[NT], ps[NT], cf[NT], cfs[NT], cr[NT], u[NT];
    ldv<NT>(cp_row, lane, p);
    double* const cps_row = R.cps + row + off;
    double* const cu_row = R.cu + row + off;
    // … and whose finite ℓ implies a finite gradient: the fused step applies evaluate_ℓ's rules without the gradient scan
    // (a functor of the caller's with kElementwise but without kFiniteLqImpliesFiniteGrad keeps the separate K2 kernel)
    constexpr bool CAN_FUSE = BlockEval<T>::value && T::kFiniteLqImpliesFiniteGrad;
    const bool fuse = CAN_FUSE && P.fuse_k2 && P.one_product;
    // ∇ℓ of a stored point (proposal slot, parked edge) is not kept next to its q but re-evaluated from it, a block per wave
    // (same functor, same bits), for the families where that is cheap: two row copies less per suspended leaf
    const bool regrad = CAN_FUSE && T::kRecomputeGrad && P.fuse_k2;
    const T tgt(P.tp);
    auto eval_blockwise = [&](const double (&qv)[NT], double (&gv)[NT]) -> double {    // the lane's partial sum of ℓ's reduction
        if constexpr (!CAN_FUSE) {
            return 0.0;
        } else if constexpr (T::kElementwise) {
            return tgt.eval(qv, gv, off + lane, D);
        } else {                                                                       // neighbours across the blocks' borders
            __syncthreads();                                                           // (qedge free again)
            if (lane == 0) qedge[wave][0] = qv[0];
            if (lane == WAVE - 1) qedge[wave][1] = qv[NT - 1];
            __syncthreads();
            const double left = wave > 0 ? qedge[wave - 1][1] : 0.0;
            const double right = wave + 1 < K3B_WPC ? qedge[wave + 1][0] : 0.0;
            return tgt.eval_block(qv, gv, off + lane, lane, D, left, right);
        }
    };
    int32_t k2_done = 0;
    double lq_next = 0.0;
    if (P.one_product) {                             // p♯′ = M⁻¹pₘ + (ϵ/2)·u′ (dense_rounds.hpp)
        const double h = eps_s / 2;
#pragma unroll
        for (int k = 0; k < NT; ++k) { u[k] = cu_row[lane + WAVE * k]; ps[k] = cps_row[lane + WAVE * k] + h * u[k]; }
        if (!fuse) stv<NT>(cps_row, lane, ps);       // (fused: the row receives the next M⁻¹pₘ at the end instead)
    } else {
        ldv<NT>(cps_row, lane, ps);
    }

    auto randexp = [&]() -> double {   // Random.randexp at NUTS.jl:44: the nrand-th draw of this transition
        uint64_t r1, r2;
        stream_raw64(key, nrand, PURPOSE_TREE, tr, r1, r2);
        nrand += 1;
        return uni_f64(det_randexp_t<dm_u>(r1));
    };
    auto save_leaf = [&](double lq_leaf, double pi_leaf) -> int {
        int s = __builtin_ctzll(free_mask);
        free_mask &= ~(1ull << s);
        copy_row(q_row, wsv(wd_slot(max_depth, s, 0)));
        if (!regrad) copy_row(g_row, wsv(wd_slot(max_depth, s, 1)));
        S.sl_lq[s] = lq_leaf;
        S.sl_pi[s] = pi_leaf;
        return s;
    };
    // combine_turn_statistics (NUTS.jl:132-139) for x earlier, y later; writes the merged (first, first♯, ρ)
    auto merge = [&](auto xms_, auto xp_, auto xps_, auto xr_, auto ym_, auto yms_, auto yps_, auto yr_, auto nf_, auto nfs_) -> bool {
        double xms[NT], yms[NT], xps[NT], yps[NT], s1[NT], s2[NT], rr[NT], nf[NT], nfs[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            xms[k] = xms_(k); xps[k] = xps_(k); yms[k] = yms_(k); yps[k] = yps_(k);
            const double xp = xp_(k), xr = xr_(k), ym = ym_(k), yr = yr_(k);
            nf[k] = nf_(k); nfs[k] = nfs_(k);
            s1[k] = xr + ym;
            s2[k] = xp + yr;
            rr[k] = xr + yr;
        }
#pragma unroll
        for (int k = 0; k < NT; ++k) { cf[k] = nf[k]; cfs[k] = nfs[k]; cr[k] = rr[k]; }
        double acc[6];
        block_allreduce<6, K3B_WPC>(wave, lane, xch, [&](double (&a)[6]) {
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                a[0] = __builtin_fma(xms[k], s1[k], a[0]);
                a[1] = __builtin_fma(yms[k], s1[k], a[1]);
                a[2] = __builtin_fma(xps[k], s2[k], a[2]);
                a[3] = __builtin_fma(yps[k], s2[k], a[3]);
                a[4] = __builtin_fma(xms[k], rr[k], a[4]);
                a[5] = __builtin_fma(yps[k], rr[k], a[5]);
            }
        }, acc);
        return acc[0] < 0 || acc[1] < 0 || acc[2] < 0 || acc[3] < 0 || acc[4] < 0 || acc[5] < 0;
    };

    // The same merge when BOTH subtrees are single leaves (every level-0 merge: half of all merges): each side's
    // (p₋, p₊, ρ) is its p and both p♯'s are the leaf's, so the three sums of NUTS.jl:134-136 are all pa + pb (IEEE
    // addition commutes: the build direction does not matter) and the six dots are two values, each three times —
    // bit-identical to the general merge, with two rows read instead of five (a level-0 suspension stores only p, p♯).
    auto merge_leaf_leaf = [&](auto pa_, auto pas_) -> bool {
        double pas[NT], rr[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const double pa = pa_(k);
            pas[k] = pas_(k);
            rr[k] = pa + p[k];
            cf[k] = pa;
            cfs[k] = pas[k];
            cr[k] = rr[k];
        }
        double acc[2];
        block_allreduce<2, K3B_WPC>(wave, lane, xch, [&](double (&a)[2]) {
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                a[0] = __builtin_fma(pas[k], rr[k], a[0]);
                a[1] = __builtin_fma(ps[k], rr[k], a[1]);
            }
        }, acc);
        return acc[0] < 0 || acc[1] < 0;
    };

    // ---- the leaf (NUTS.jl:148-159) -----------------------------------------------------------
    const bool fwd = dir == 1;
    const int64_t di = fwd ? 1 : -1;
    const uint32_t j = jleaf;
    const int depth0 = depth;
    double kin[1];
    block_allreduce<1, K3B_WPC>(wave, lane, xch, [&](double (&a)[1]) {
#pragma unroll
        for (int k = 0; k < NT; ++k) a[0] = __builtin_fma(p[k], ps[k], a[0]);
    }, kin);
    const double lq_leaf = uni_f64(S.lq_leaf);
    const double pi_leaf = uni_f64(joint_logdensity(lq_leaf, kin[0] / 2.0));
    int64_t i = uni_i64(S.i) + di;
    total_steps += 1;
    const double delta = pi_leaf - uni_f64(S.pi0);
    double v_lsa = delta < 0.0 ? delta : 0.0;
    int64_t v_steps = 1;
    bool invalid = false, finished = false, doubled = false;
    int level = 0;
    if (delta < P.min_delta) {
        term_left = term_right = i;
        invalid = true;
    } else {
#pragma unroll
        for (int k = 0; k < NT; ++k) { cf[k] = p[k]; cfs[k] = ps[k]; cr[k] = p[k]; }
        double c_omega = delta;
        int c_zeta = -1;
        for (;;) {
            const bool sub = ((j >> level) & 1u) != 0;
            const bool top = !sub && (j == nleaf - 1) && (level == depth0);
            if (!sub && !top) break;
            auto a_cf = [&](int k) { return cf[k]; };
            auto a_cfs = [&](int k) { return cfs[k]; };
            auto a_ps = [&](int k) { return ps[k]; };
            auto a_cr = [&](int k) { return cr[k]; };
            bool turning;
            if (sub) {
                auto lf = row_acc(wd_stack(level, 0)), lfs = row_acc(wd_stack(level, 1));
                auto ll = row_acc(wd_stack(level, 2)), lls = row_acc(wd_stack(level, 3));
                auto lr = row_acc(wd_stack(level, 4));
                // x = (first, first♯, last, last♯, ρ); merge(x.p♯₋, x.p₊, x.p♯₊, x.ρ, y.p₋, y.p♯₋, y.p♯₊, y.ρ, new first, new first♯)
                if (level == 0) turning = merge_leaf_leaf(ll, lls);
                else turning = fwd ? merge(lfs, ll, lls, lr, a_cf, a_cfs, a_ps, a_cr, lf, lfs)
                                   : merge(a_ps, a_cf, a_cfs, a_cr, ll, lls, lfs, lr, lf, lfs);
                const double wl = S.lv_omega[level];
                double w;
                logaddexp_pair(S.lv_vlsa[level], v_lsa, wl, c_omega, lane, v_lsa, w);
                v_steps += (int64_t)S.lv_vsteps[level];
                if (turning) {
                    term_left = i - di * (((int64_t)2 << level) - 1);
                    term_right = i;
                    invalid = true;
                    level += 1;
                    break;
                }
                const double logprob2 = c_omega - w;
                const bool pick = logprob2 >= 0.0 || (randexp() > -logprob2);
                const int lz = S.lv_zeta[level];
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
                (void)tm;
                turning = fwd ? merge(tms, tp, tps, trr, a_cf, a_cfs, a_ps, a_cr, a_cf, a_cfs)
                              : merge(a_ps, a_cf, a_cfs, a_cr, tm, tms, tps, trr, a_cf, a_cfs);
                double w, vt;
                logaddexp_pair(vtop_lsa, v_lsa, omega_top, c_omega, lane, vt, w);
                vtop_lsa = vt;
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
                    stv<NT>(wsv(wd_top(fwd ? 2 : 0)), lane, p);
                    stv<NT>(wsv(wd_top(fwd ? 3 : 1)), lane, ps);
                    stv<NT>(wsv(wd_top(4)), lane, cr);
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
            if (level > 0) {                                 // a suspended leaf is its (p, p♯) alone
                stv<NT>(wsv(wd_stack(level, 0)), lane, cf);
                stv<NT>(wsv(wd_stack(level, 1)), lane, cfs);
                stv<NT>(wsv(wd_stack(level, 4)), lane, cr);
            }
            stv<NT>(wsv(wd_stack(level, 2)), lane, p);
            stv<NT>(wsv(wd_stack(level, 3)), lane, ps);
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
        vtop_lsa = uni_f64(det_logaddexp_u(vtop_lsa, v_lsa));
        vtop_steps += v_steps;
        finished = true;
    }

    if (finished) {
        // ---- end of the transition (NUTS.jl:238-240; mcmc.jl:272-278, 375-377) --------------------
        const double eps_used = fwd ? eps_s : -eps_s;
        double a = det_exp_u(vtop_lsa) / (double)vtop_steps;
        const double acc_rate = uni_f64(a < 1.0 ? a : 1.0);
        init_slot = zeta_top;
        lq_cur = S.sl_lq[init_slot];
        const double pi_stat = S.sl_pi[init_slot];
        const size_t o = (size_t)chain * P.N + n_done;
        {
            double qv[NT];
            ldv<NT>(wsv(wd_slot(max_depth, init_slot, 0)), lane, qv);
            stv<NT>(q_row, lane, qv);
            if (P.out.draws) {
                double* drow = P.out.draws + o * D + off;
#pragma unroll
                for (int k = 0; k < NT; ++k)
                    if (off + lane + WAVE * k < D) drow[lane + WAVE * k] = qv[k];
            }
            window_accumulate<NT>(P, row + off, lane, qv, n_done);
            if (regrad) {
                double gv[NT];
                (void)eval_blockwise(qv, gv);
                stv<NT>(g_row, lane, gv);
            }
        }
        if (!regrad) copy_row(wsv(wd_slot(max_depth, init_slot, 1)), g_row);
        if (tid == 0) {
            if (P.out.logdensities) P.out.logdensities[o] = lq_cur;
            if (P.out.eps) P.out.eps[o] = eps_used;
            if (P.out.pi) P.out.pi[o] = pi_stat;
            if (P.out.acceptance_rate) P.out.acceptance_rate[o] = acc_rate;
            if (P.out.steps) P.out.steps[o] = vtop_steps;
            if (P.out.term_left) P.out.term_left[o] = term_left;
            if (P.out.term_right) P.out.term_right[o] = term_right;
            if (P.out.depth) P.out.depth[o] = depth;
            if (P.out.directions) P.out.directions[o] = S.directions0;
        }
        if (P.adapt) {   // adapt_stepsize (stepsize.jl:147-156)
            da.m += 1;
            const double m = (double)da.m;
            da.Hbar += (P.delta - acc_rate - da.Hbar) / (m + (double)P.t0);
            da.logeps = da.mu - __builtin_sqrt(m) / P.gamma * da.Hbar;
            da.logeps_bar += det_pow_pos_u(m, -P.kappa) * (da.logeps - da.logeps_bar);
        }
        n_done += 1;
        tr += 1;
        if ((int64_t)n_done < P.N) {
            // z ~ N(0, I) of the next transition into the GEMM input row; K0 finishes the job (begin_transition_request)
            __syncthreads();
#pragma unroll
            for (int k2 = 0; k2 < NT / 2; ++k2) {
                const int kk = wave * (NT / 2) + k2;
                uint64_t r1, r2;
                stream_raw64(key, (uint32_t)(lane + WAVE * kk), PURPOSE_MOMENTUM, tr, r1, r2);
                double z0, z1;
                det_randn2_v(r1, r2, &z0, &z1);
                const int e0 = lane + WAVE * (2 * kk), e1 = e0 + WAVE;
                R.cp[row + e0] = e0 < D ? z0 : 0.0;
                R.cp[row + e1] = e1 < D ? z1 : 0.0;
            }
            phase = PH_NEED_MOMENTUM;
            if (tid == 0) R.list[atomicAdd(R.list_count, 1)] = chain;
        } else {
            phase = PH_DONE;
            if (tid == 0) {
                P.st.lq[chain] = lq_cur;
                if (P.adapt) {
                    P.st.da[chain] = da;
                    if (P.da_finalize) P.st.eps[chain] = det_exp_u(da.logeps_bar);
                }
                P.st.transition[chain] = tr;
                P.st.status[chain] = status;
                if (P.leapfrog_counter) atomicAdd(P.leapfrog_counter, total_steps);
                atomicAdd(R.done_count, 1);
            }
        }
    } else {
        // ---- the chain's next leapfrog: same doubling, or the next one (trees.jl:290-293) ---------
        if (doubled) {
            const bool nfwd = (dirs & 1u) != 0;
            dirs >>= 1;
            const int ndir = nfwd ? 1 : 0;
            if (reg_edge != ndir) {
                copy_row(q_row, wsv(wd_edge(reg_edge, 0)));
                if (!regrad) copy_row(g_row, wsv(wd_edge(reg_edge, 1)));
                if (reg_edge == 1) stored1 = 1; else stored0 = 1;
                const bool have = nfwd ? (stored1 != 0) : (stored0 != 0);
                const int qsrc = have ? wd_edge(ndir, 0) : wd_slot(max_depth, init_slot, 0);
                const int gsrc = have ? wd_edge(ndir, 1) : wd_slot(max_depth, init_slot, 1);
                if (regrad) {
                    double qe[NT], ge[NT];
                    ldv<NT>(wsv(qsrc), lane, qe);
                    stv<NT>(q_row, lane, qe);
                    (void)eval_blockwise(qe, ge);
                    stv<NT>(g_row, lane, ge);
                } else {
                    copy_row(wsv(qsrc), q_row);
                    copy_row(wsv(gsrc), g_row);
                }
                ldv<NT>(wsv(wd_top(nfwd ? 2 : 0)), lane, p);
                if (fuse) {                          // the edge's p♯ and u travel with it: in registers here
                    stv<NT>(wsv(wd_edge_u(max_depth, reg_edge)), lane, u);
                    ldv<NT>(wsv(have ? wd_edge_u(max_depth, ndir) : wd_u0(max_depth)), lane, u);
                    ldv<NT>(wsv(wd_top(nfwd ? 3 : 1)), lane, ps);
                } else if (P.one_product) {
                    copy_row(cu_row, wsv(wd_edge_u(max_depth, reg_edge)));
                    copy_row(wsv(have ? wd_edge_u(max_depth, ndir) : wd_u0(max_depth)), cu_row);
                    copy_row(wsv(wd_top(nfwd ? 3 : 1)), cps_row);
                }
            }
            reg_edge = ndir;
            dir = ndir;
            i = nfwd ? i_plus : i_minus;
            jleaf = 0;
            nleaf = 1u << depth;
            const double eps = eps_s < 0 ? -eps_s : eps_s;
            eps_s = nfwd ? eps : -eps;
        } else {
            jleaf = j + 1;
        }
        const double h = eps_s / 2;
#pragma unroll
        for (int k = 0; k < NT; ++k) p[k] = p[k] + h * g_row[lane + WAVE * k];   // pₘ of the next leapfrog (hamiltonian.jl:277)
        if constexpr (CAN_FUSE) {
            if (fuse) {
                // K2 of the chain's next leapfrog, here (dense_rounds.hpp rounds_k2_kernel, one-product form): the same
                // operations on the same operands, a wave per 256-coordinate block — the rows q, ∇ℓ, p and M⁻¹pₘ are written once
                // instead of written by K3, read and written again by K2.
                double t[NT], qv[NT], gv[NT];
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    t[k] = ps[k] + h * u[k];                                       // M⁻¹pₘ = p♯ + (ϵ/2)·u
                    qv[k] = q_row[lane + WAVE * k] + eps_s * t[k];                 // hamiltonian.jl:278
                }
                stv<NT>(cps_row, lane, t);
                const double part = eval_blockwise(qv, gv);
                double lsum[1];
                block_allreduce<1, K3B_WPC>(wave, lane, xch, [&](double (&a)[1]) { a[0] = part; }, lsum);
                double lq = uni_f64(tgt.finish(lsum[0]));
                bool pos_finite = true;
                if (!T::kFiniteLqImpliesFiniteQ || !dm_isfinite(lq)) {            // evaluate_ℓ's position scan (hamiltonian.jl:203)
                    bool fin = true;
#pragma unroll
                    for (int k = 0; k < NT; ++k) fin = fin && dm_isfinite(qv[k]);
                    if (tid == 0) fin_flag = 1;
                    __syncthreads();
                    if (!wave_all(fin) && lane == 0) fin_flag = 0;
                    __syncthreads();
                    pos_finite = fin_flag != 0;
                }
                static_assert(T::kFiniteLqImpliesFiniteGrad, "block evaluation: families whose gradient needs no scan");
                lq = demote_lq(lq, pos_finite, true);
                if (!pos_finite) status |= DHMC_ST_NONFINITE_POSITION;
                lq_next = lq;
#pragma unroll
                for (int k = 0; k < NT; ++k) p[k] = p[k] + h * gv[k];             // p′ (hamiltonian.jl:280)
                stv<NT>(q_row, lane, qv);
                stv<NT>(g_row, lane, gv);
                k2_done = 1;
            }
        }
        stv<NT>(cp_row, lane, p);
    }
    __syncthreads();
    if (tid == 0) {
        S.k2_done = k2_done;
        if (k2_done) S.lq_leaf = lq_next;
        S.nrand = nrand; S.status = status; S.dirs = dirs; S.j = jleaf; S.nleaf = nleaf; S.tr = tr;
        S.depth = depth; S.dir = dir; S.reg_edge = reg_edge; S.stored0 = stored0; S.stored1 = stored1;
        S.zeta_top = zeta_top; S.init_slot = init_slot; S.n = n_done; S.phase = phase;
        S.i = i; S.i_plus = i_plus; S.i_minus = i_minus; S.term_left = term_left; S.term_right = term_right; S.vtop_steps = vtop_steps;
        S.free_mask = free_mask; S.total_steps = total_steps;
        S.eps_s = eps_s; S.omega_top = omega_top; S.vtop_lsa = vtop_lsa; S.lq_cur = lq_cur; S.da = da;
    }
    __syncthreads();
    {
        uint32_t* dst = reinterpret_cast<uint32_t*>(&R.ts[chain]);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&S);
        for (int w = tid; w < (int)(sizeof(TreeState) / 4); w += WAVE * K3B_WPC) dst[w] = src[w];
    }
}

}  // namespace dhmc
