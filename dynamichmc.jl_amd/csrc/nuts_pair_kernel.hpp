// The per-draw loop of a SHORT chain (at most 64 coordinates, diagonal metric) as a PAIR of wavefronts: an integrator and a tree
// builder, in one workgroup, connected by a ring of leaf records in LDS.
//
// Why.  A launch ends with its slowest chain, and a chain's leapfrogs are sequential: Neal's funnel at 1000 transitions has a chain
// with 13.5 × the mean's work (836 716 leapfrogs), and the call lasts exactly as long as that chain × the kernel's latency per
// leapfrog (profiles/r05_packed_engine.txt) — 3 700 clocks in nuts_run_kernel<T,1>, of which the leapfrog itself (integrator,
// density, kinetic energy) is ≈ 40 % and the tree logic (turn checks, logaddexp, multinomial picks, suspensions) the rest.  Nothing
// in the tree logic feeds the next leapfrog except the decision to stop.  So the two halves run as two instruction streams on two
// SIMDs of a CU:
//
//   wave A (integrator)   samples the momentum, walks the doublings in the order of the direction bits (both trajectory edges in
//                         its registers), and for every leaf writes a record (p, q, ℓq, π, flags) into the ring; for the odd leaf
//                         of a pair it also takes the pair's leaf·leaf turn check (it holds both momenta).  It runs ahead of the
//                         tree by at most the ring's depth and stops when the builder says the tree has ended.
//   wave B (tree builder) consumes the records in order: leaf scalars, the merge cascade of the iterative adjacent_tree, proposals,
//                         the end of the transition (outputs, dual averaging) — nuts_run_kernel's loop with "take a record" where
//                         the leapfrog stood — and hands the next position and step size back through a mailbox.
//
// Leaves integrated beyond the tree's end are discarded (the integrator cannot know a turn before the builder has found it); they
// cost nothing on the critical path.  Arithmetic, random streams and merge order are nuts_run_kernel's: the same bits
// (tests/test_gpu_pair.py compares the two kernels and the oracle).  Every wait is bounded: a wave that waits longer than
// PAIR_SPIN_LIMIT polls raises DHMC_ST_KERNEL_PROTOCOL in the chain's status and both waves leave (a logic error must not hang the
// device).  Used by dhmc_run for launches that the previous launch showed to be held open by a few chains (families whose
// gradient is recomputed from a stored position: every built-in functor family but the logistic regression).
#pragma once
#include "nuts_kernels.hpp"

namespace dhmc {

constexpr int PAIR_RING = 4;                     // leaf records in flight
constexpr unsigned PAIR_SPIN_LIMIT = 1u << 24;   // polls of one wait (≈ 64 clocks each plus the poll itself: ≈ 1 s)
constexpr uint32_t DHMC_ST_KERNEL_PROTOCOL = 0x40000000u;   // internal: the pair kernel's handshake timed out (a bug, never a model's fault)

// LDS of one pair (doubles): B's rows exactly as nuts_run_kernel<T,1,true> lays them out (M⁻¹, level 0, level 1 first / last, six
// more levels × 3), then the ring (p, q rows and four scalars per record), the mailbox (q row + four scalars) and 8 control words.
__host__ __device__ constexpr int pair_rows_b() { return 4 + 3 * lds_extra_levels(1); }
__host__ __device__ constexpr size_t pair_lds_bytes() {
    return sizeof(double) * ((size_t)WAVE * pair_rows_b() + (size_t)PAIR_RING * (2 * WAVE + 4) + WAVE + 4 + 8);
}

__device__ __forceinline__ void pair_publish(volatile unsigned* flag, unsigned v, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) *flag = v;
}
__device__ __forceinline__ unsigned pair_peek(volatile unsigned* flag) {
    const unsigned v = *flag;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return uni_u32(v);
}

template <class T>
__global__ __launch_bounds__(128) void nuts_run_pair_kernel(RunParams P) {
    static_assert(T::kRecomputeGrad, "the integrator re-evaluates ∇ℓ at the position the builder hands back");
    constexpr int NPL = 1;
    const int chain = P.launch_order ? P.launch_order[blockIdx.x] : (int)blockIdx.x;
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int D = P.D, Dpad = P.Dpad;
    extern __shared__ double lds[];
    double* const m_lds = lds;
    double* const l0_lds = lds + WAVE;
    double* const l1f_lds = lds + 2 * WAVE;
    double* const l1l_lds = lds + 3 * WAVE;
    constexpr int NXL = lds_extra_levels(1);
    double* const xl_lds = lds + 4 * WAVE;
    double* const ring_p = lds + (size_t)WAVE * pair_rows_b();           // [RING][64]
    double* const ring_q = ring_p + (size_t)PAIR_RING * WAVE;             // [RING][64]
    double* const ring_s = ring_q + (size_t)PAIR_RING * WAVE;             // [RING][4]: ℓq, π, flags (as an integer in a double's bits), -
    double* const mb_q = ring_s + (size_t)PAIR_RING * 4;                  // [64] the position the next transition starts from
    double* const mb_s = mb_q + WAVE;                                     // [4]: ℓq, ϵ, -, -
    volatile unsigned* const ctl = reinterpret_cast<volatile unsigned*>(mb_s + 4);
    volatile unsigned* const c_head = ctl + 0;     // A: records of transition seq_a published so far (record 0 = the start point)
    volatile unsigned* const c_tail = ctl + 1;     // B: records consumed
    volatile unsigned* const c_seq_a = ctl + 2;    // A: the transition its records belong to (+1; 0 = none yet)
    volatile unsigned* const c_seq_b = ctl + 3;    // B: the transition whose start is in the mailbox (+1; 0 = none yet)
    volatile unsigned* const c_quit = ctl + 4;     // B: all transitions done (or either: protocol error)
    if (threadIdx.x < 8) ctl[threadIdx.x] = 0u;
    const size_t row = (size_t)chain * Dpad;
    if (wave == 1) m_lds[lane] = P.st.minv[row + lane];
    __syncthreads();

    const T tgt(P.tp);
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    const int max_depth = P.max_depth;
    const int nl = uni_i32(reduce_lanes(NPL, D));
    const uint32_t tr0 = P.st.transition[chain];
    bool broken = false;
    auto wait_for = [&](auto cond) -> bool {       // bounded spin; false: give up (and tell the partner)
        unsigned spins = 0;
        while (!cond()) {
            if (pair_peek(c_quit) == 2u) { broken = true; return false; }
            __builtin_amdgcn_s_sleep(1);
            if (++spins > PAIR_SPIN_LIMIT) { broken = true; pair_publish(c_quit, 2u, lane); return false; }
        }
        return true;
    };

    if (wave == 0) {
        // ================================================= A: the integrator =================================================
        const double mreg = P.st.minv[row + lane];
        auto mk = [&](int) -> double { return mreg; };
        const double* const Wrow = P.st.W + row;
        double q[1], p[1], g[1], cf[1], cr[1];
        for (int64_t n = 0; n < P.N; ++n) {
            const unsigned want = (unsigned)n + 1u;
            if (!wait_for([&] { return pair_peek(c_seq_b) == want || pair_peek(c_quit) != 0u; })) return;
            if (pair_peek(c_quit) != 0u) return;
            const uint32_t tr = tr0 + (uint32_t)n;
            q[0] = mb_q[lane];
            const double lq_cur = uni_f64(mb_s[0]);
            const double eps = uni_f64(mb_s[1]);
            (void)tgt.eval(q, g, lane, D);                               // ∇ℓ of the position (proposals keep q only)
            sample_momentum<NPL>(key, PURPOSE_MOMENTUM, tr, Wrow, lane, p);
            uint32_t dirs;
            {
                uint32_t w[4];
                philox4x32_10(0u, PURPOSE_DIRECTIONS, tr, key.seed_hi, key.k0, key.k1, w);
                dirs = uni_u32(w[0]);
            }
            double pi0;
            {
                LaneAcc<1, NPL> kacc;
                kacc.add(0, 0, p[0], mk(0) * p[0]);
                pi0 = uni_f64(joint_logdensity(lq_cur, wave_allreduce1(kacc.fold(0), nl) / 2.0));
            }
            // record 0: the start point (p₀, q₀, ℓq₀, π₀)
            ring_p[lane] = p[0];
            ring_q[lane] = q[0];
            if (lane == 0) { ring_s[0] = lq_cur; ring_s[1] = pi0; ring_s[2] = __longlong_as_double(1ll); }
            unsigned head = 1u;
            pair_publish(c_head, head, lane);
            pair_publish(c_seq_a, want, lane);
            // both edges of the trajectory (z₋, z₊) in registers; the registers (q, p, g) hold the one being extended
            double qe[2] = {q[0], q[0]}, pe[2] = {p[0], p[0]}, ge[2] = {g[0], g[0]};
            bool ended = false;                                          // the builder has moved on, or a divergent leaf was produced
            for (int depth = 0; depth < max_depth && !ended; ++depth) {
                const bool fwd = (dirs & 1u) != 0;
                dirs >>= 1;
                const int dir = fwd ? 1 : 0;
                q[0] = qe[dir]; p[0] = pe[dir]; g[0] = ge[dir];
                const double eps_s = fwd ? eps : -eps;
                const uint32_t nleaf = 1u << depth;
                double p_prev = 0.0;
                for (uint32_t j = 0; j < nleaf; ++j) {
                    // room in the ring, and the builder still on this transition
                    if (!wait_for([&] { return head - pair_peek(c_tail) < (unsigned)PAIR_RING || pair_peek(c_seq_b) != want || pair_peek(c_quit) != 0u; })) return;
                    if (pair_peek(c_seq_b) != want || pair_peek(c_quit) != 0u) { ended = true; break; }
                    double lq_leaf, pi_leaf;
                    bool pos_finite;
                    leapfrog_leaf_m<T, NPL>(tgt, mk, lane, D, q, p, g, eps_s, lq_leaf, pi_leaf, pos_finite, nl);
                    unsigned flags = pos_finite ? 1u : 0u;
                    if (j & 1u) {                                        // the pair's leaf·leaf turn check (merge_leaf_leaf: its two dots)
                        const double pa = p_prev;
                        if (merge_leaf_leaf<NPL>([&](int) { return pa; }, mk, cf, cr, p, nl)) flags |= 2u;
                    }
                    p_prev = p[0];
                    const unsigned slot = head % (unsigned)PAIR_RING;
                    ring_p[slot * WAVE + lane] = p[0];
                    ring_q[slot * WAVE + lane] = q[0];
                    if (lane == 0) {
                        ring_s[slot * 4 + 0] = lq_leaf;
                        ring_s[slot * 4 + 1] = pi_leaf;
                        ring_s[slot * 4 + 2] = __longlong_as_double((long long)flags);
                    }
                    head += 1u;
                    pair_publish(c_head, head, lane);
                    if (pi_leaf - pi0 < P.min_delta) { ended = true; break; }   // divergent (NUTS.jl:151): the tree ends at this leaf
                }
                qe[dir] = q[0]; pe[dir] = p[0]; ge[dir] = g[0];
            }
        }
        return;
    }

    // ===================================================== B: the tree builder =====================================================
    double* const ws = P.st.ws + (size_t)chain * P.nvec * Dpad;
    auto wsv = [&](int idx) -> double* { return ws + (size_t)idx * Dpad; };
    auto mk = [&](int) -> double { return m_lds[lane]; };
    const int nslots = ws_nslots(max_depth);
    LaneArrF64 lv_omega, lv_vlsa;
    LaneArrI64 lv_vsteps;
    LaneArrI32 lv_zeta;
    LaneArrF64 sl_lq, sl_pi;
    double q[1], p[1], g[1], cf[1], cr[1];
    double tpm[1], tpp[1], trho[1];
    ldv<NPL>(P.st.q + row, lane, q);
    double lq_cur = P.st.lq[chain];
    double eps_fixed = P.st.eps[chain];
    DAState da = P.st.da[chain];
    uint32_t status = P.st.status[chain];
    unsigned long long total_steps = 0;
    if (P.adapt && P.da_init) {  // initial_adaptation_state (stepsize.jl:134-138; mcmc.jl:266)
        double le = det_log_u(eps_fixed);
        da.mu = det_log_u(10.0) + le;
        da.m = 1;
        da.Hbar = 0.0;
        da.logeps = le;
        da.logeps_bar = 0.0;
    }
    int init_slot = 0;
    stv<NPL>(wsv(ws_slot(max_depth, init_slot, 0)), lane, q);
    uint64_t free_mask = 0;
    auto save_leaf = [&](double lq_leaf, double pi_leaf) -> int {
        int s = __builtin_ctzll(free_mask);
        free_mask &= ~(1ull << s);
        stv<NPL>(wsv(ws_slot(max_depth, s, 0)), lane, q);
        sl_lq.set(s, lq_leaf, lane);
        sl_pi.set(s, pi_leaf, lane);
        return s;
    };
    int64_t n = 0;
    for (; n < P.N; ++n) {
        const uint32_t tr = tr0 + (uint32_t)n;
        const unsigned want = (unsigned)n + 1u;
        const double eps = uni_f64(P.adapt ? det_exp_u(da.logeps) : eps_fixed);  // current_ϵ (stepsize.jl:163)
        // the mailbox: where the integrator starts this transition
        mb_q[lane] = q[0];
        if (lane == 0) { mb_s[0] = lq_cur; mb_s[1] = eps; }
        unsigned tail = 0u;
        pair_publish(c_tail, tail, lane);
        pair_publish(c_seq_b, want, lane);
        uint32_t dirs;
        {
            uint32_t w[4];
            philox4x32_10(0u, PURPOSE_DIRECTIONS, tr, key.seed_hi, key.k0, key.k1, w);
            dirs = uni_u32(w[0]);
        }
        const uint32_t directions0 = dirs;
        // record 0: p₀ and π₀
        if (!wait_for([&] { return pair_peek(c_seq_a) == want && pair_peek(c_head) >= 1u; })) break;
        p[0] = ring_p[lane];
        const double pi0 = uni_f64(ring_s[1]);
        tail = 1u;
        pair_publish(c_tail, tail, lane);
        tpm[0] = p[0]; tpp[0] = p[0]; trho[0] = p[0];
        sl_lq.set(init_slot, lq_cur, lane);
        sl_pi.set(init_slot, pi0, lane);

        uint32_t nrand = 0, rexp_base = 0;
        double rexp_vals;
        auto rexp_fill = [&](uint32_t base) {
            uint64_t r1, r2;
            stream_raw64(key, base + (uint32_t)lane, PURPOSE_TREE, tr, r1, r2);
            rexp_vals = det_randexp_v(r1);
            rexp_base = base;
        };
        rexp_fill(0);
        auto randexp = [&]() -> double {  // Random.randexp at NUTS.jl:44
            if (nrand - rexp_base >= 64u) rexp_fill(nrand & ~63u);
            double v = readlane_f64(rexp_vals, (int)(nrand & 63u));
            nrand += 1;
            return v;
        };

        free_mask = ((nslots >= 64) ? ~0ull : ((1ull << nslots) - 1ull)) & ~(1ull << init_slot);
        int zeta_top = init_slot;
        double omega_top = 0.0;
        double vtop_lsa = -dm_inf();
        int64_t vtop_steps = 0;
        int depth = 0;
        int64_t i_minus = 0, i_plus = 0;
        int64_t term_left = 1, term_right = 0;  // REACHED_MAX_DEPTH
        bool finished = false;
        while (!finished && depth < max_depth && !broken) {
            const bool fwd = (dirs & 1u) != 0;  // next_direction (trees.jl:31-34)
            dirs >>= 1;
            int64_t i = fwd ? i_plus : i_minus;
            const int64_t di = fwd ? 1 : -1;
            const uint32_t nleaf = 1u << depth;
            bool invalid = false;
            double v_lsa = 0.0;
            int64_t v_steps = 0;
            for (uint32_t j = 0; j < nleaf && !invalid && !finished; ++j) {
                // ---- the leaf: the integrator's next record ---------------------------------------------------------------
                if (!wait_for([&] { return pair_peek(c_head) > tail; })) { finished = true; break; }
                const unsigned slot = tail % (unsigned)PAIR_RING;
                p[0] = ring_p[slot * WAVE + lane];
                q[0] = ring_q[slot * WAVE + lane];
                const double lq_leaf = uni_f64(ring_s[slot * 4 + 0]);
                const double pi_leaf = uni_f64(ring_s[slot * 4 + 1]);
                const unsigned flags = uni_u32((unsigned)__double_as_longlong(ring_s[slot * 4 + 2]));
                tail += 1u;
                pair_publish(c_tail, tail, lane);
                if (!(flags & 1u)) status |= DHMC_ST_NONFINITE_POSITION;
                i += di;
                const double delta = pi_leaf - pi0;             // NUTS.jl:150
                v_lsa = delta < 0.0 ? delta : 0.0;              // min(Δ, 0)   (NUTS.jl:79)
                v_steps = 1;
                int level = 0;
                if (delta < P.min_delta) {                      // divergent leaf (NUTS.jl:151; trees.jl:236-237)
                    term_left = term_right = i;
                    invalid = true;
                } else {
                    double c_omega = delta;
                    int c_zeta = -1;  // -1: the proposal is the leaf held in registers
                    for (;;) {
                        const bool sub = ((j >> level) & 1u) != 0;
                        const bool top = !sub && (j == nleaf - 1) && (level == depth);
                        if (!sub && !top) break;
                        auto a_cf = [&](int k) { return cf[k]; };
                        auto a_p = [&](int k) { return p[k]; };
                        auto a_cr = [&](int k) { return cr[k]; };
                        bool turning;
                        if (sub) {
                            if (level == 0) {                    // the integrator took this pair's two dots (flags bit 1)
                                const double pa = l0_lds[lane];
                                cf[0] = pa;
                                cr[0] = pa + p[0];
                                turning = (flags & 2u) != 0;
                            } else if (level == 1) {
                                auto a_lf = [&](int) { return l1f_lds[lane]; };
                                auto a_ll = [&](int) { return l1l_lds[lane]; };
                                auto a_lr = [&](int) { return l1f_lds[lane] + l1l_lds[lane]; };
                                turning = fwd ? merge_core<NPL>(a_lf, a_ll, a_lr, a_cf, a_p, a_cr, a_lf, mk, cf, cr, nl)
                                              : merge_core<NPL>(a_p, a_cf, a_cr, a_ll, a_lf, a_lr, a_lf, mk, cf, cr, nl);
                            } else if (level < 2 + NXL) {
                                const double* Lf = xl_lds + (size_t)(3 * (level - 2)) * WAVE;
                                const double* Ll = Lf + WAVE;
                                const double* Lr = Ll + WAVE;
                                auto a_lf = [&](int) { return Lf[lane]; };
                                auto a_ll = [&](int) { return Ll[lane]; };
                                auto a_lr = [&](int) { return Lr[lane]; };
                                turning = fwd ? merge_core<NPL>(a_lf, a_ll, a_lr, a_cf, a_p, a_cr, a_lf, mk, cf, cr, nl)
                                              : merge_core<NPL>(a_p, a_cf, a_cr, a_ll, a_lf, a_lr, a_lf, mk, cf, cr, nl);
                            } else {
                                const double* Lf = wsv(ws_stack(level, 0));
                                const double* Ll = wsv(ws_stack(level, 1));
                                const double* Lr = wsv(ws_stack(level, 2));
                                auto a_lf = [&](int) { return Lf[lane]; };
                                auto a_ll = [&](int) { return Ll[lane]; };
                                auto a_lr = [&](int) { return Lr[lane]; };
                                turning = fwd ? merge_core<NPL>(a_lf, a_ll, a_lr, a_cf, a_p, a_cr, a_lf, mk, cf, cr, nl)
                                              : merge_core<NPL>(a_p, a_cf, a_cr, a_ll, a_lf, a_lr, a_lf, mk, cf, cr, nl);
                            }
                            const double wl = lv_omega.get(level);
                            double w;
                            logaddexp_pair(lv_vlsa.get(level), v_lsa, wl, c_omega, lane, v_lsa, w);
                            v_steps += lv_vsteps.get(level);
                            if (turning) {                       // trees.jl:255
                                term_left = i - di * (((int64_t)2 << level) - 1);
                                term_right = i;
                                invalid = true;
                                level += 1;
                                break;
                            }
                            const double logprob2 = c_omega - w;
                            const bool pick = logprob2 >= 0.0 || (randexp() > -logprob2);
                            const int lz = lv_zeta.get(level);
                            if (pick) {
                                free_mask |= (1ull << lz);
                            } else {
                                if (c_zeta >= 0) free_mask |= (1ull << c_zeta);
                                c_zeta = lz;
                            }
                            c_omega = w;
                            level += 1;
                        } else {
                            auto a_tm = [&](int) { return tpm[0]; };
                            auto a_tp = [&](int) { return tpp[0]; };
                            auto a_tr = [&](int) { return trho[0]; };
                            if (depth == 0) {
                                turning = merge_leaf_leaf<NPL>(a_tr, mk, cf, cr, p, nl);
                            } else {
                                turning = fwd ? merge_core<NPL>(a_tm, a_tp, a_tr, a_cf, a_p, a_cr, a_cf, mk, cf, cr, nl)
                                              : merge_core<NPL>(a_p, a_cf, a_cr, a_tm, a_tp, a_tr, a_cf, mk, cf, cr, nl);
                            }
                            double w;
                            logaddexp_pair(vtop_lsa, v_lsa, omega_top, c_omega, lane, vtop_lsa, w);
                            vtop_steps += v_steps;
                            const double logprob2 = c_omega - omega_top;   // biased progressive (trees.jl:159-161)
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
                            if (turning) {                       // trees.jl:315-316
                                term_left = i_minus;
                                term_right = i_plus;
                                finished = true;
                            } else if (depth < max_depth) {
                                if (fwd) tpp[0] = p[0]; else tpm[0] = p[0];
                                trho[0] = cr[0];
                            }
                            level = -1;  // handled
                            break;
                        }
                    }
                    if (level >= 0 && !invalid) {
                        if (c_zeta < 0) c_zeta = save_leaf(lq_leaf, pi_leaf);
                        if (level == 0) {
                            l0_lds[lane] = p[0];
                        } else if (level == 1) {
                            l1f_lds[lane] = cf[0];
                            l1l_lds[lane] = p[0];
                        } else if (level < 2 + NXL) {
                            double* Lf = xl_lds + (size_t)(3 * (level - 2)) * WAVE;
                            Lf[lane] = cf[0];
                            Lf[WAVE + lane] = p[0];
                            Lf[2 * WAVE + lane] = cr[0];
                        } else {
                            stv<NPL>(wsv(ws_stack(level, 0)), lane, cf);
                            stv<NPL>(wsv(ws_stack(level, 1)), lane, p);
                            stv<NPL>(wsv(ws_stack(level, 2)), lane, cr);
                        }
                        lv_omega.set(level, c_omega, lane);
                        lv_vlsa.set(level, v_lsa, lane);
                        lv_vsteps.set(level, v_steps, lane);
                        lv_zeta.set(level, c_zeta, lane);
                    }
                }
                if (invalid) {
                    for (int l2 = level; l2 < depth; ++l2) {
                        if ((j >> l2) & 1u) {
                            v_lsa = uni_f64(det_logaddexp_u(lv_vlsa.get(l2), v_lsa));
                            v_steps += lv_vsteps.get(l2);
                        }
                    }
                    vtop_lsa = uni_f64(det_logaddexp_u(vtop_lsa, v_lsa)); // trees.jl:294
                    vtop_steps += v_steps;
                    finished = true;                                       // trees.jl:297
                }
            }
        }
        if (broken) break;

        // ---- TreeStatisticsNUTS and the new position (NUTS.jl:238-240) -----------------
        const double acc_rate = [&]() {
            double a = det_exp_u(vtop_lsa) / (double)vtop_steps;           // NUTS.jl:87
            return uni_f64(a < 1.0 ? a : 1.0);
        }();
        total_steps += (unsigned long long)vtop_steps;
        init_slot = zeta_top;
        ldv<NPL>(wsv(ws_slot(max_depth, init_slot, 0)), lane, q);
        lq_cur = sl_lq.get(init_slot);
        const double pi_stat = sl_pi.get(init_slot);
        const size_t o = (size_t)chain * (P.out_stride ? P.out_stride : P.N) + n;
        if (P.out.draws) {
            double* drow = P.out.draws + o * D;
            if (lane < D) drow[lane] = q[0];
        }
        window_accumulate<NPL>(P, (size_t)chain * P.Dpad, lane, q, n);
        if (lane == 0) {
            if (P.out.logdensities) P.out.logdensities[o] = lq_cur;        // mcmc.jl:276,377
            if (P.out.eps) P.out.eps[o] = eps;                             // mcmc.jl:273
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
            da.logeps_bar += det_pow_pos_u(m, -P.kappa) * (da.logeps - da.logeps_bar);
        }
    }
    if (broken) status |= DHMC_ST_KERNEL_PROTOCOL;
    pair_publish(c_quit, broken ? 2u : 1u, lane);        // the integrator may leave
    // ---- write the chain back (WarmupState + adaptation state) --------------------------
    stv<NPL>(P.st.q + row, lane, q);
    (void)tgt.eval(q, g, lane, D);
    stv<NPL>(P.st.g + row, lane, g);
    if (lane == 0) {
        P.st.lq[chain] = lq_cur;
        if (P.adapt) {
            P.st.da[chain] = da;
            if (P.da_finalize) P.st.eps[chain] = det_exp_u(da.logeps_bar); // final_ϵ (stepsize.jl:170; mcmc.jl:285)
        }
        P.st.transition[chain] = tr0 + (uint32_t)n;
        P.st.status[chain] = status;
        if (P.leapfrog_counter) atomicAdd(P.leapfrog_counter, total_steps);
        if (P.chain_work) P.chain_work[chain] = (unsigned)(total_steps > 0xffffffffull ? 0xffffffffull : total_steps);
    }
}

template <class T>
int launch_run_pair(const RunParams& P, hipStream_t s) {
    if constexpr (!T::kRecomputeGrad || T::kBigDims) {
        return DHMC_ERR_UNSUPPORTED;
    } else {
        if (P.Dpad != WAVE) return DHMC_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((nuts_run_pair_kernel<T>), dim3(P.C), dim3(2 * WAVE), pair_lds_bytes(), s, P);
        return DHMC_OK;
    }
}

}  // namespace dhmc
