// The per-draw loop of a SHORT chain (at most 64 coordinates, diagonal metric) as a PIPELINE of four wavefronts in one workgroup — a
// CU's four SIMDs: an integrator, a turn-statistic builder, a visited-statistic builder and a proposal builder, joined by a ring of
// leaf records in LDS.
//
// Why.  A launch ends with its slowest chain, and a chain's leapfrogs are sequential: Neal's funnel at 1000 transitions has a chain
// with 13.5 × the mean's work (836 716 leapfrogs), and the call lasts exactly as long as that chain × the kernel's latency per
// leapfrog (profiles/r05_packed_engine.txt) — 3 700 clocks in nuts_run_kernel<T,1>, of which the leapfrog itself (integrator,
// density, kinetic energy) is ≈ 40 %, the merges' vector work ≈ 25 % and their scalar work (logaddexp, picks, bookkeeping) the
// rest.  Nothing downstream feeds the next leapfrog except the decision to stop; the turn statistics (p₋, p₊, ρ and the six dots of
// a merge) never depend on the proposals' weights or picks, and the weights never on the statistics — only on WHETHER a merge was
// turning; and the visited statistic (the acceptance rate's log Σ min(1, e^Δ)) depends on the leaves' Δ and on which merges happen,
// while only the end of the transition reads it.  So the parts run as four instruction streams on the four SIMDs of a CU:
//
//   wave A   the integrator: samples the momentum, walks the doublings in the order of the direction bits (both trajectory edges in
//            its registers), and for every leaf writes a record (p, q, ℓq, π, flags) into the ring; for the odd leaf of a pair it
//            also takes the pair's leaf·leaf turn check (it holds both momenta).  It runs ahead of the tree by at most the ring's
//            depth and stops when the tree has ended.
//   wave B1  the turn-statistic builder: the running summary (first, Σp), the suspended stack and the trajectory's τ, every merge's
//            dots; per leaf it publishes ONE code — which merge of the leaf's cascade was turning, if any.
//   wave B3  the visited-statistic builder: v = v₋ ⊕ v₊ where the tree merges, the unwinding where a subtree is invalid; one result
//            per transition (log Σ α, steps).
//   wave B2  the proposal builder: Δ, ω = logaddexp(ω₋, ω₊), multinomial / biased-progressive picks with the Exp(1) stream, proposal
//            slots (it reads q from the ring), termination record, outputs, dual averaging — nuts_run_kernel's loop with "take a
//            record and a code" where the leapfrog and the dots stood — and hands the next position and step size back through a
//            mailbox.
//
// All four find a tree's end by themselves (divergent leaf: π of the record; turning: B1's code; depth limit), so nobody waits for
// a message that is not coming; leaves integrated beyond the tree's end are discarded.  Counters that a faster wave reads across a
// transition boundary carry the transition they belong to (tail1_seq, tail3_seq, mseq, aseq): a stale count is read as zero.  Arithmetic, random
// streams and merge order are nuts_run_kernel's: the same bits (tests/test_gpu_pipeline.py compares the kernels and the oracle).
// Every wait is bounded: a wave that polls longer than PAIR_SPIN_LIMIT times raises DHMC_ST_KERNEL_PROTOCOL in the chain's status
// and all waves leave (a logic error must not hang the device).  Used by dhmc_run for launches that the previous launch showed to
// be held open by a few chains (families whose gradient is recomputed from a stored position: every built-in functor family but
// the logistic regression).  (A two-wave form — integrator ‖ whole tree builder — was measured first: 1.22 × the wave-per-chain
// kernel on config 4, three waves (B2 and B3 as one) 1.57 ×; profiles/r05_pipeline_kernel.txt.)
#pragma once
#include "nuts_kernels.hpp"

namespace dhmc {

// (PAIR_RING, PIPE_NXL, pipeline_lds_bytes: run_params.hpp)
constexpr unsigned PAIR_SPIN_LIMIT = 1u << 24;   // polls of one wait (≈ 64 clocks each plus the poll itself: ≈ 1 s)
// (DHMC_ST_KERNEL_PROTOCOL, include/dhmc.h: the handshake timed out — a bug, never a model's fault)

// The handshakes live in LDS only (ring, mailbox, counters), and the LDS executes one wavefront's operations in issue order: a record
// written before its counter is visible before it, a counter read before a record is read before it.  So the "fences" are compiler
// barriers — a workgroup-scope release fence would also drain the wave's GLOBAL stores (proposal slots, deep stack rows, draws),
// ≈ 1 µs of waiting per leaf for stores nobody else reads.
// That ordering is how every GCN / CDNA LDS works, not an architected guarantee (include/dhmc.h says so).  -DDHMC_PIPE_FENCED builds the
// variant that does not rely on it: the publishing wave waits for its outstanding LDS operations (s_waitcnt lgkmcnt(0): the record
// is IN the LDS) before it writes the counter — an LDS-only fence, the wave's global stores stay in flight.
__device__ __forceinline__ void pair_publish(volatile unsigned* flag, unsigned v, int lane) {
#ifdef DHMC_PIPE_FENCED
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
    if (lane == 0) *flag = v;
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned pair_peek(volatile unsigned* flag) {
    const unsigned v = *flag;
    asm volatile("" ::: "memory");
    return uni_u32(v);
}
// all control words in one round trip (four ds_read_b128)
struct PairCtl { unsigned w[16]; };
__device__ __forceinline__ PairCtl pair_load_ctl(volatile unsigned* ctl) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const volatile v4u* p4 = reinterpret_cast<const volatile v4u*>(ctl);
    const v4u a = p4[0], b = p4[1], c = p4[2], d = p4[3];
    asm volatile("" ::: "memory");
    PairCtl r;
    r.w[0] = uni_u32(a.x); r.w[1] = uni_u32(a.y); r.w[2] = uni_u32(a.z); r.w[3] = uni_u32(a.w);
    r.w[4] = uni_u32(b.x); r.w[5] = uni_u32(b.y); r.w[6] = uni_u32(b.z); r.w[7] = uni_u32(b.w);
    r.w[8] = uni_u32(c.x); r.w[9] = uni_u32(c.y); r.w[10] = uni_u32(c.z); r.w[11] = uni_u32(c.w);
    r.w[12] = uni_u32(d.x); r.w[13] = uni_u32(d.y); r.w[14] = uni_u32(d.z); r.w[15] = uni_u32(d.w);
    return r;
}

constexpr int PIPE_TOP = 64;      // B1's code for "the top-level merge was turning" (0 … 31: the sub-merge at that level; -1: none)

template <class T, int NPL>
__global__ __launch_bounds__(256) void nuts_run_pipeline_kernel(RunParams P) {
    // the integrator re-evaluates ∇ℓ at the position the builder hands back: families that store gradients with their proposals
    // (T::kRecomputeGrad false) are not launched here (launch_run_pipeline; a caller's functor: dhmc_create reads its traits)
    if (!T::kRecomputeGrad) return;
    constexpr int DP = WAVE * NPL;
    const int chain = P.launch_order ? P.launch_order[blockIdx.x] : (int)blockIdx.x;
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int D = P.D, Dpad = P.Dpad;
    extern __shared__ double lds[];
    constexpr int NXL = PIPE_NXL;
    double* const l0_lds = lds;
    double* const l1f_lds = lds + DP;
    double* const l1l_lds = lds + 2 * DP;
    double* const xl_lds = lds + 3 * DP;
    double* const ring_p = lds + (size_t)DP * (3 + 3 * NXL);              // [RING][DP]
    double* const ring_q = ring_p + (size_t)PAIR_RING * DP;               // [RING][DP]
    double* const ring_s = ring_q + (size_t)PAIR_RING * DP;               // [RING][4]: ℓq, π, flags, -
    double* const mres = ring_s + (size_t)PAIR_RING * 4;                  // [RING]: B1's code of the leaf (an integer in a double's bits)
    double* const mb_q = mres + PAIR_RING;                                // [DP]
    double* const mb_s = mb_q + DP;                                       // [4]: ℓq, ϵ
    // a row of a chain in LDS: lane l holds elements l, l + 64, … (the wave kernel's slot layout)
    auto ld_row = [&](const double* rowp, double (&v)[NPL]) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) v[k] = rowp[lane + WAVE * k];
    };
    auto st_row = [&](double* rowp, const double (&v)[NPL]) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) rowp[lane + WAVE * k] = v[k];
    };
    volatile unsigned* const ctl = reinterpret_cast<volatile unsigned*>(mb_s + 4);
    volatile unsigned* const c_head = ctl + 0;       // A   (the order of these words is the enum W_* below)
    volatile unsigned* const c_seq_a = ctl + 1;      // A
    volatile unsigned* const c_tail1 = ctl + 2;      // B1: records consumed …
    volatile unsigned* const c_tail1_seq = ctl + 3;  //     … of this transition
    volatile unsigned* const c_mhead = ctl + 4;      // B1: codes published (same index as the records) …
    volatile unsigned* const c_mseq = ctl + 5;       //     … of this transition
    volatile unsigned* const c_tail2 = ctl + 6;      // B2: records consumed (reset before seq_b is published)
    volatile unsigned* const c_seq_b = ctl + 7;      // B2: the transition whose start is in the mailbox
    volatile unsigned* const c_quit = ctl + 8;       // B2: done (1) / anyone: protocol error (2)
    volatile unsigned* const c_tail3 = ctl + 9;      // B3: records (and codes) consumed …
    volatile unsigned* const c_tail3_seq = ctl + 10; //     … of this transition
    volatile unsigned* const c_aseq = ctl + 11;      // B3: the transition whose visited statistic is in mb_s[2..3]
    // (a call in rounds runs these blocks beside the packed kernel's waves, one of which may share the SIMD: the chain here is the one
    // the round waits for, so its instructions go first.  Only then: priorities between the four roles change nothing — each has its
    // SIMD — and an unconditional s_setprio here cost 2.5 % of the kernel's latency, whatever the level)
    if (P.prog) __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x < 16) ctl[threadIdx.x] = 0u;
    __syncthreads();

    const size_t row = (size_t)chain * Dpad;
    const T tgt(P.tp);
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    const int max_depth = P.max_depth;
    const int nl = uni_i32(reduce_lanes(NPL, D));
    const uint32_t tr0 = P.st.transition[chain];
    // a call in rounds (RunParams::prog): the chain has n_done transitions of the call behind it and NN to go to the round's target;
    // the loops below count the launch's transitions (so do the waves' sequence numbers), records and window counts the call's
    const int64_t n_done = P.prog ? (int64_t)P.prog[chain] : 0;
    const int64_t NN = P.N > n_done ? P.N - n_done : 0;
    bool broken = false;
    enum { W_HEAD = 0, W_SEQ_A = 1, W_TAIL1 = 2, W_TAIL1_SEQ = 3, W_MHEAD = 4, W_MSEQ = 5, W_TAIL2 = 6, W_SEQ_B = 7, W_QUIT = 8,
           W_TAIL3 = 9, W_TAIL3_SEQ = 10, W_ASEQ = 11 };
    PairCtl cw;                                    // the control words as of the last poll
#ifdef DHMC_PHASE_TIMING     // tools/experiments/pipeline_stage_timing.py: the clocks each wave spends waiting, and in all
    unsigned long long pt_wait = 0;
    const unsigned long long pt_start = __builtin_readcyclecounter();
#define PIPE_PT_WAIT_BEGIN const unsigned long long pt_t0 = __builtin_readcyclecounter();
#define PIPE_PT_WAIT_END pt_wait += __builtin_readcyclecounter() - pt_t0;
    unsigned long long pt_b2[4] = {0ull, 0ull, 0ull, 0ull}, pt_mark = 0ull;   // B2's leaf loop: record read / cascade / suspension / rest
#define PIPE_B2_BEGIN pt_mark = __builtin_readcyclecounter();
#define PIPE_B2_MARK(i) { const unsigned long long t_ = __builtin_readcyclecounter(); pt_b2[i] += t_ - pt_mark; pt_mark = t_; }
#define PIPE_B2_FLUSH if (lane == 0) { atomicAdd(&g_phase[8], pt_b2[0]); atomicAdd(&g_phase[9], pt_b2[1]); atomicAdd(&g_phase[10], pt_b2[2]); atomicAdd(&g_phase[11], pt_b2[3]); }
#define PIPE_PT_FLUSH(role, leaves) if (lane == 0) { atomicAdd(&g_phase[role], pt_wait); atomicAdd(&g_phase[4 + role], __builtin_readcyclecounter() - pt_start); if (role == 2) { atomicAdd(&g_phase[14], (unsigned long long)(leaves)); atomicAdd(&g_phase[15], 1ull); } }
#else
#define PIPE_PT_WAIT_BEGIN
#define PIPE_PT_WAIT_END
#define PIPE_PT_FLUSH(role, leaves)
#define PIPE_B2_BEGIN
#define PIPE_B2_MARK(i)
#define PIPE_B2_FLUSH
#endif
    auto wait_for = [&](auto cond) -> bool {       // bounded spin over one load of all control words per poll; false: give up
        unsigned spins = 0;
        PIPE_PT_WAIT_BEGIN
        for (;;) {
            cw = pair_load_ctl(ctl);
            if (cw.w[W_QUIT] == 2u) { broken = true; return false; }
            if (cond(cw)) { PIPE_PT_WAIT_END return true; }
            __builtin_amdgcn_s_sleep(1);
            if (++spins > PAIR_SPIN_LIMIT) { broken = true; pair_publish(c_quit, 2u, lane); return false; }
        }
    };
    auto uni_i64v = [](long long x) -> long long {
        const uint32_t lo = uni_u32((uint32_t)(unsigned long long)x), hi = uni_u32((uint32_t)((unsigned long long)x >> 32));
        return (long long)(((unsigned long long)hi << 32) | lo);
    };
    auto directions_of = [&](uint32_t tr) -> uint32_t {
        uint32_t w[4];
        philox4x32_10(0u, PURPOSE_DIRECTIONS, tr, key.seed_hi, key.k0, key.k1, w);
        return uni_u32(w[0]);
    };

    if (wave == 0) {
        // ================================================= A: the integrator =================================================
        double mreg[NPL];
        ldv<NPL>(P.st.minv + row, lane, mreg);
        auto mk = [&](int k) -> double { return mreg[k]; };
        const double* const Wrow = P.st.W + row;
        double q[NPL], p[NPL], g[NPL], cf[NPL], cr[NPL];
        for (int64_t n = 0; n < NN; ++n) {
            const unsigned want = (unsigned)n + 1u;
            if (!wait_for([&](const PairCtl& c) { return c.w[W_SEQ_B] == want || c.w[W_QUIT] != 0u; })) return;
            if (cw.w[W_QUIT] != 0u) return;
            const uint32_t tr = tr0 + (uint32_t)n;
            ld_row(mb_q, q);
            const double lq_cur = uni_f64(mb_s[0]);
            const double eps = uni_f64(mb_s[1]);
            (void)tgt.eval(q, g, lane, D);
            sample_momentum<NPL>(key, PURPOSE_MOMENTUM, tr, Wrow, lane, p);
            uint32_t dirs = directions_of(tr);
            double pi0;
            {
                LaneAcc<1, NPL> kacc;
#pragma unroll
                for (int k = 0; k < NPL; ++k) kacc.add(0, k, p[k], mk(k) * p[k]);
                pi0 = uni_f64(joint_logdensity(lq_cur, wave_allreduce1(kacc.fold(0), nl) / 2.0));
            }
            st_row(ring_p, p);
            st_row(ring_q, q);
            if (lane == 0) { ring_s[0] = lq_cur; ring_s[1] = pi0; ring_s[2] = __longlong_as_double(1ll); }
            unsigned head = 1u;
            pair_publish(c_head, head, lane);
            pair_publish(c_seq_a, want, lane);
            auto consumed = [&](const PairCtl& c) -> unsigned {          // what ALL readers have taken of this transition's records
                const unsigned t2 = c.w[W_TAIL2];
                const unsigned t1 = c.w[W_TAIL1_SEQ] == want ? c.w[W_TAIL1] : 0u;
                const unsigned t3 = c.w[W_TAIL3_SEQ] == want ? c.w[W_TAIL3] : 0u;
                const unsigned t = t1 < t2 ? t1 : t2;
                return t < t3 ? t : t3;
            };
            double qm[NPL], pm[NPL], gm[NPL], qp[NPL], pp[NPL], gp[NPL];     // the trajectory's two edges (backward, forward)
#pragma unroll
            for (int k = 0; k < NPL; ++k) { qm[k] = q[k]; pm[k] = p[k]; gm[k] = g[k]; qp[k] = q[k]; pp[k] = p[k]; gp[k] = g[k]; }
            bool ended = false;
            unsigned seen = 0u;                                          // records both readers had consumed at the last poll
            for (int depth = 0; depth < max_depth && !ended; ++depth) {
                const bool fwd = (dirs & 1u) != 0;
                dirs >>= 1;
#pragma unroll
                for (int k = 0; k < NPL; ++k) { q[k] = fwd ? qp[k] : qm[k]; p[k] = fwd ? pp[k] : pm[k]; g[k] = fwd ? gp[k] : gm[k]; }
                const double eps_s = fwd ? eps : -eps;
                const uint32_t nleaf = 1u << depth;
                double p_prev[NPL];
#pragma unroll
                for (int k = 0; k < NPL; ++k) p_prev[k] = 0.0;
                for (uint32_t j = 0; j < nleaf; ++j) {
                    // room in the ring by the LAST poll's counts is room now (the readers only advance): poll when that runs out —
                    // which is also when the integrator learns that the tree has ended (at most a ring's worth of leaves late)
                    if (head - seen >= (unsigned)PAIR_RING) {
                        if (!wait_for([&](const PairCtl& c) { return head - consumed(c) < (unsigned)PAIR_RING || c.w[W_SEQ_B] != want || c.w[W_QUIT] != 0u; })) return;
                        if (cw.w[W_SEQ_B] != want || cw.w[W_QUIT] != 0u) { ended = true; break; }
                        seen = consumed(cw);
                    }
                    double lq_leaf, pi_leaf;
                    bool pos_finite;
                    leapfrog_leaf_m<T, NPL>(tgt, mk, lane, D, q, p, g, eps_s, lq_leaf, pi_leaf, pos_finite, nl);
                    unsigned flags = pos_finite ? 1u : 0u;
                    if (j & 1u) {
                        if (merge_leaf_leaf<NPL>([&](int k) { return p_prev[k]; }, mk, cf, cr, p, nl)) flags |= 2u;
                    }
#pragma unroll
                    for (int k = 0; k < NPL; ++k) p_prev[k] = p[k];
                    const unsigned slot = head % (unsigned)PAIR_RING;
                    st_row(ring_p + (size_t)slot * DP, p);
                    st_row(ring_q + (size_t)slot * DP, q);
                    if (lane == 0) {
                        ring_s[slot * 4 + 0] = lq_leaf;
                        ring_s[slot * 4 + 1] = pi_leaf;
                        ring_s[slot * 4 + 2] = __longlong_as_double((long long)flags);
                    }
                    head += 1u;
                    pair_publish(c_head, head, lane);
                    if (pi_leaf - pi0 < P.min_delta) { ended = true; break; }
                }
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    if (fwd) { qp[k] = q[k]; pp[k] = p[k]; gp[k] = g[k]; }
                    else { qm[k] = q[k]; pm[k] = p[k]; gm[k] = g[k]; }
                }
            }
        }
        PIPE_PT_FLUSH(0, 0)
        return;
    }

    if (wave == 1) {
        // ========================================== B1: turn statistics, merge by merge ==========================================
        double mreg[NPL];
        ldv<NPL>(P.st.minv + row, lane, mreg);
        auto mk = [&](int k) -> double { return mreg[k]; };
        double* const ws = P.st.ws + (size_t)chain * P.nvec * Dpad;
        auto wsv = [&](int idx) -> double* { return ws + (size_t)idx * Dpad; };
        double p[NPL], cf[NPL], cr[NPL], tpm[NPL], tpp[NPL], trho[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) { cf[k] = 0.0; cr[k] = 0.0; }
        for (int64_t n = 0; n < NN; ++n) {
            const unsigned want = (unsigned)n + 1u;
            if (!wait_for([&](const PairCtl& c) { return (c.w[W_SEQ_A] == want && c.w[W_HEAD] >= 1u) || c.w[W_QUIT] != 0u; })) return;
            if (cw.w[W_QUIT] != 0u) return;
            ld_row(ring_p, p);
            const double pi0 = uni_f64(ring_s[1]);
            unsigned tail = 1u;
            pair_publish(c_tail1, tail, lane);
            pair_publish(c_tail1_seq, want, lane);
            pair_publish(c_mhead, tail, lane);
            pair_publish(c_mseq, want, lane);
#pragma unroll
            for (int k = 0; k < NPL; ++k) { tpm[k] = p[k]; tpp[k] = p[k]; trho[k] = p[k]; }
            uint32_t dirs = directions_of(tr0 + (uint32_t)n);
            int depth = 0;
            bool finished = false;
            unsigned avail = 1u, seen2 = 0u;                             // A's head and B2's tail at the last poll
            while (!finished && depth < max_depth) {
                const bool fwd = (dirs & 1u) != 0;
                dirs >>= 1;
                const uint32_t nleaf = 1u << depth;
                for (uint32_t j = 0; j < nleaf && !finished; ++j) {
                    if (!(avail > tail && tail - seen2 < (unsigned)PAIR_RING)) {     // (by the last poll's counts: they only advance)
                        auto readers = [&](const PairCtl& c) -> unsigned {             // what both readers of the codes have taken
                            const unsigned t3 = c.w[W_TAIL3_SEQ] == want ? c.w[W_TAIL3] : 0u;
                            return c.w[W_TAIL2] < t3 ? c.w[W_TAIL2] : t3;
                        };
                        if (!wait_for([&](const PairCtl& c) { return c.w[W_HEAD] > tail && tail - readers(c) < (unsigned)PAIR_RING; })) return;
                        avail = cw.w[W_HEAD];
                        seen2 = readers(cw);
                    }
                    const unsigned slot = tail % (unsigned)PAIR_RING;
                    ld_row(ring_p + (size_t)slot * DP, p);
                    const double pi_leaf = uni_f64(ring_s[slot * 4 + 1]);
                    const unsigned flags = uni_u32((unsigned)__double_as_longlong(ring_s[slot * 4 + 2]));
                    int code = -1, level = 0;
                    bool invalid = false;
                    if (pi_leaf - pi0 < P.min_delta) {
                        invalid = true;                           // divergent leaf: the tree ends here (no merge)
                    } else {
                        for (;;) {
                            const bool sub = ((j >> level) & 1u) != 0;
                            const bool top = !sub && (j == nleaf - 1) && (level == depth);
                            if (!sub && !top) break;
                            auto a_cf = [&](int k) { return cf[k]; };
                            auto a_p = [&](int k) { return p[k]; };
                            auto a_cr = [&](int k) { return cr[k]; };
                            bool turning;
                            if (sub) {
                                if (level == 0) {
#pragma unroll
                                    for (int k = 0; k < NPL; ++k) {
                                        const double pa = l0_lds[lane + WAVE * k];
                                        cf[k] = pa;
                                        cr[k] = pa + p[k];
                                    }
                                    turning = (flags & 2u) != 0;
                                } else if (level == 1) {
                                    auto a_lf = [&](int k) { return l1f_lds[lane + WAVE * k]; };
                                    auto a_ll = [&](int k) { return l1l_lds[lane + WAVE * k]; };
                                    auto a_lr = [&](int k) { return l1f_lds[lane + WAVE * k] + l1l_lds[lane + WAVE * k]; };
                                    turning = fwd ? merge_core<NPL>(a_lf, a_ll, a_lr, a_cf, a_p, a_cr, a_lf, mk, cf, cr, nl)
                                                  : merge_core<NPL>(a_p, a_cf, a_cr, a_ll, a_lf, a_lr, a_lf, mk, cf, cr, nl);
                                } else if (level < 2 + NXL) {
                                    const double* Lf = xl_lds + (size_t)(3 * (level - 2)) * DP;
                                    const double* Ll = Lf + DP;
                                    const double* Lr = Ll + DP;
                                    auto a_lf = [&](int k) { return Lf[lane + WAVE * k]; };
                                    auto a_ll = [&](int k) { return Ll[lane + WAVE * k]; };
                                    auto a_lr = [&](int k) { return Lr[lane + WAVE * k]; };
                                    turning = fwd ? merge_core<NPL>(a_lf, a_ll, a_lr, a_cf, a_p, a_cr, a_lf, mk, cf, cr, nl)
                                                  : merge_core<NPL>(a_p, a_cf, a_cr, a_ll, a_lf, a_lr, a_lf, mk, cf, cr, nl);
                                } else {
                                    const double* Lf = wsv(ws_stack(level, 0));
                                    const double* Ll = wsv(ws_stack(level, 1));
                                    const double* Lr = wsv(ws_stack(level, 2));
                                    auto a_lf = [&](int k) { return Lf[lane + WAVE * k]; };
                                    auto a_ll = [&](int k) { return Ll[lane + WAVE * k]; };
                                    auto a_lr = [&](int k) { return Lr[lane + WAVE * k]; };
                                    turning = fwd ? merge_core<NPL>(a_lf, a_ll, a_lr, a_cf, a_p, a_cr, a_lf, mk, cf, cr, nl)
                                                  : merge_core<NPL>(a_p, a_cf, a_cr, a_ll, a_lf, a_lr, a_lf, mk, cf, cr, nl);
                                }
                                if (turning) { code = level; invalid = true; break; }
                                level += 1;
                            } else {
                                auto a_tm = [&](int k) { return tpm[k]; };
                                auto a_tp = [&](int k) { return tpp[k]; };
                                auto a_tr = [&](int k) { return trho[k]; };
                                if (depth == 0) turning = merge_leaf_leaf<NPL>(a_tr, mk, cf, cr, p, nl);
                                else turning = fwd ? merge_core<NPL>(a_tm, a_tp, a_tr, a_cf, a_p, a_cr, a_cf, mk, cf, cr, nl)
                                                   : merge_core<NPL>(a_p, a_cf, a_cr, a_tm, a_tp, a_tr, a_cf, mk, cf, cr, nl);
                                depth += 1;
                                if (turning) {
                                    code = PIPE_TOP;
                                    finished = true;
                                } else if (depth < max_depth) {
#pragma unroll
                                    for (int k = 0; k < NPL; ++k) {
                                        if (fwd) tpp[k] = p[k]; else tpm[k] = p[k];
                                        trho[k] = cr[k];
                                    }
                                }
                                level = -1;
                                break;
                            }
                        }
                        if (level >= 0 && !invalid) {             // suspend the running summary
                            if (level == 0) {
                                st_row(l0_lds, p);
                            } else if (level == 1) {
                                st_row(l1f_lds, cf);
                                st_row(l1l_lds, p);
                            } else if (level < 2 + NXL) {
                                double* Lf = xl_lds + (size_t)(3 * (level - 2)) * DP;
                                st_row(Lf, cf);
                                st_row(Lf + DP, p);
                                st_row(Lf + 2 * DP, cr);
                            } else {
                                stv<NPL>(wsv(ws_stack(level, 0)), lane, cf);
                                stv<NPL>(wsv(ws_stack(level, 1)), lane, p);
                                stv<NPL>(wsv(ws_stack(level, 2)), lane, cr);
                            }
                        }
                    }
                    if (invalid) finished = true;
                    if (lane == 0) mres[slot] = __longlong_as_double((long long)code);
                    tail += 1u;
                    pair_publish(c_mhead, tail, lane);
                    pair_publish(c_tail1, tail, lane);
                }
            }
        }
        PIPE_PT_FLUSH(1, 0)
        return;
    }

    if (wave == 2) {
        // ===================================== B3: the visited statistic (NUTS.jl:59-89) =====================================
        // v = (log Σ min(1, e^Δ), steps) of every subtree, merged where the tree merges (trees.jl:249,294) and unwound where a subtree
        // is invalid (trees.jl:244,249-250): it depends on the leaves' Δ and on WHICH merges happen — the records' π and B1's codes —
        // and on nothing else; the acceptance rate of the transition is its only consumer.
        LaneArrF64 lv_vlsa;
        LaneArrI64 lv_vsteps;
        for (int64_t n = 0; n < NN; ++n) {
            const unsigned want = (unsigned)n + 1u;
            if (!wait_for([&](const PairCtl& c) { return (c.w[W_SEQ_A] == want && c.w[W_HEAD] >= 1u) || c.w[W_QUIT] != 0u; })) return;
            if (cw.w[W_QUIT] != 0u) return;
            const double pi0 = uni_f64(ring_s[1]);
            unsigned tail = 1u;
            pair_publish(c_tail3, tail, lane);
            pair_publish(c_tail3_seq, want, lane);
            double vtop_lsa = -dm_inf();
            int64_t vtop_steps = 0;
            int depth = 0;
            bool finished = false;
            unsigned avail = 1u;
            while (!finished && depth < max_depth) {
                const uint32_t nleaf = 1u << depth;
                bool invalid = false;
                for (uint32_t j = 0; j < nleaf && !invalid && !finished; ++j) {
                    if (!(avail > tail)) {
                        if (!wait_for([&](const PairCtl& c) { return c.w[W_HEAD] > tail && c.w[W_MSEQ] == want && c.w[W_MHEAD] > tail; })) return;
                        avail = cw.w[W_HEAD] < cw.w[W_MHEAD] ? cw.w[W_HEAD] : cw.w[W_MHEAD];
                    }
                    const unsigned slot = tail % (unsigned)PAIR_RING;
                    const double pi_leaf = uni_f64(ring_s[slot * 4 + 1]);
                    const int code = uni_i32((int)__double_as_longlong(mres[slot]));
                    tail += 1u;
                    if ((tail & 3u) == 0u) pair_publish(c_tail3, tail, lane);
                    const double delta = pi_leaf - pi0;             // NUTS.jl:150
                    double v_lsa = delta < 0.0 ? delta : 0.0;       // min(Δ, 0)   (NUTS.jl:79)
                    int64_t v_steps = 1;
                    int level = 0;
                    if (delta < P.min_delta) {
                        invalid = true;
                    } else {
                        for (;;) {
                            const bool sub = ((j >> level) & 1u) != 0;
                            const bool top = !sub && (j == nleaf - 1) && (level == depth);
                            if (!sub && !top) break;
                            if (sub) {
                                v_lsa = uni_f64(det_logaddexp_u(lv_vlsa.get(level), v_lsa));      // v = v₋ ⊕ v₊ (trees.jl:249)
                                v_steps += lv_vsteps.get(level);
                                if (code == level) { invalid = true; level += 1; break; }          // turning (trees.jl:255)
                                level += 1;
                            } else {
                                vtop_lsa = uni_f64(det_logaddexp_u(vtop_lsa, v_lsa));             // trees.jl:294
                                vtop_steps += v_steps;
                                depth += 1;
                                if (code == PIPE_TOP) finished = true;
                                level = -1;
                                break;
                            }
                        }
                        if (level >= 0 && !invalid) {
                            lv_vlsa.set(level, v_lsa, lane);
                            lv_vsteps.set(level, v_steps, lane);
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
            if (lane == 0) { mb_s[2] = vtop_lsa; mb_s[3] = __longlong_as_double((long long)vtop_steps); }
            pair_publish(c_aseq, want, lane);
        }
        PIPE_PT_FLUSH(3, 0)
        return;
    }

    // ================================================== B2: the scalar builder ==================================================
    double* const ws = P.st.ws + (size_t)chain * P.nvec * Dpad;
    auto wsv = [&](int idx) -> double* { return ws + (size_t)idx * Dpad; };
    const int nslots = ws_nslots(max_depth);
    LaneArrF64 lv_omega;
    LaneArrI32 lv_zeta;
    LaneArrF64 sl_lq, sl_pi;
    double q[NPL], g[NPL];
    ldv<NPL>(P.st.q + row, lane, q);
    double lq_cur = P.st.lq[chain];
    double eps_fixed = P.st.eps[chain];
    DAState da = P.st.da[chain];
    uint32_t status = P.st.status[chain];
    unsigned long long total_steps = 0;
    if (P.adapt && P.da_init && n_done == 0) {  // initial_adaptation_state (stepsize.jl:134-138; mcmc.jl:266)
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
    for (; n < NN; ++n) {
        const uint32_t tr = tr0 + (uint32_t)n;
        const unsigned want = (unsigned)n + 1u;
        const double eps = uni_f64(P.adapt ? det_exp_u(da.logeps) : eps_fixed);  // current_ϵ (stepsize.jl:163)
        st_row(mb_q, q);
        if (lane == 0) { mb_s[0] = lq_cur; mb_s[1] = eps; }
        unsigned tail = 0u;
        pair_publish(c_tail2, tail, lane);
        pair_publish(c_seq_b, want, lane);
        uint32_t dirs = directions_of(tr);
        const uint32_t directions0 = dirs;
        if (!wait_for([&](const PairCtl& c) { return c.w[W_SEQ_A] == want && c.w[W_HEAD] >= 1u; })) break;
        const double pi0 = uni_f64(ring_s[1]);
        tail = 1u;
        pair_publish(c_tail2, tail, lane);
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
        int depth = 0;
        int64_t i_minus = 0, i_plus = 0;
        int64_t term_left = 1, term_right = 0;  // REACHED_MAX_DEPTH
        bool finished = false;
        unsigned avail = 1u;
        while (!finished && depth < max_depth && !broken) {
            const bool fwd = (dirs & 1u) != 0;  // next_direction (trees.jl:31-34)
            dirs >>= 1;
            int64_t i = fwd ? i_plus : i_minus;
            const int64_t di = fwd ? 1 : -1;
            const uint32_t nleaf = 1u << depth;
            bool invalid = false;
            for (uint32_t j = 0; j < nleaf && !invalid && !finished; ++j) {
                // the leaf's record (A) and its merge code (B1)
                PIPE_B2_BEGIN
                if (!(avail > tail)) {                          // records WITH their codes available at the last poll
                    if (!wait_for([&](const PairCtl& c) { return c.w[W_HEAD] > tail && c.w[W_MSEQ] == want && c.w[W_MHEAD] > tail; })) { finished = true; break; }
                    avail = cw.w[W_HEAD] < cw.w[W_MHEAD] ? cw.w[W_HEAD] : cw.w[W_MHEAD];
                }
                const unsigned slot = tail % (unsigned)PAIR_RING;
                ld_row(ring_q + (size_t)slot * DP, q);
                const double lq_leaf = uni_f64(ring_s[slot * 4 + 0]);
                const double pi_leaf = uni_f64(ring_s[slot * 4 + 1]);
                const unsigned flags = uni_u32((unsigned)__double_as_longlong(ring_s[slot * 4 + 2]));
                const int code = uni_i32((int)__double_as_longlong(mres[slot]));
                tail += 1u;
                if ((tail & 3u) == 0u) pair_publish(c_tail2, tail, lane);      // (the ring is 16 deep: the writers need this count to a few leaves only)
                if (!(flags & 1u)) status |= DHMC_ST_NONFINITE_POSITION;
                i += di;
                const double delta = pi_leaf - pi0;             // NUTS.jl:150
                int level = 0;
                PIPE_B2_MARK(0)
                if (delta < P.min_delta) {                      // divergent leaf (NUTS.jl:151; trees.jl:236-237)
                    term_left = term_right = i;
                    invalid = true;
                } else {
                    double c_omega = delta;
                    int c_zeta = -1;
                    for (;;) {
                        const bool sub = ((j >> level) & 1u) != 0;
                        const bool top = !sub && (j == nleaf - 1) && (level == depth);
                        if (!sub && !top) break;
                        if (sub) {
                            const bool turning = code == level;
                            if (turning) {                       // trees.jl:255 (before any proposal mixing)
                                term_left = i - di * (((int64_t)2 << level) - 1);
                                term_right = i;
                                invalid = true;
                                level += 1;
                                break;
                            }
                            const double w = uni_f64(det_logaddexp_u(lv_omega.get(level), c_omega));   // ω = logaddexp(ω₋, ω₊) (trees.jl:145)
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
                            const bool turning = code == PIPE_TOP;
                            const double w = uni_f64(det_logaddexp_u(omega_top, c_omega));
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
                            }
                            level = -1;
                            break;
                        }
                    }
                    PIPE_B2_MARK(1)
                    if (level >= 0 && !invalid) {
                        if (c_zeta < 0) c_zeta = save_leaf(lq_leaf, pi_leaf);
                        lv_omega.set(level, c_omega, lane);
                        lv_zeta.set(level, c_zeta, lane);
                    }
                    PIPE_B2_MARK(2)
                }
                if (invalid) finished = true;                              // trees.jl:297 (the visited statistic's unwinding: B3)
            }
        }
        if (broken) break;
        // the visited statistic of the transition (B3)
        if (!wait_for([&](const PairCtl& c) { return c.w[W_ASEQ] == want; })) break;
        const double vtop_lsa = uni_f64(mb_s[2]);
        const int64_t vtop_steps = (int64_t)uni_i64v(__double_as_longlong(mb_s[3]));

        const double acc_rate = [&]() {
            double a = det_exp_u(vtop_lsa) / (double)vtop_steps;           // NUTS.jl:87
            return uni_f64(a < 1.0 ? a : 1.0);
        }();
        total_steps += (unsigned long long)vtop_steps;
        init_slot = zeta_top;
        ldv<NPL>(wsv(ws_slot(max_depth, init_slot, 0)), lane, q);
        lq_cur = sl_lq.get(init_slot);
        const double pi_stat = sl_pi.get(init_slot);
        const size_t o = (size_t)chain * (P.out_stride ? P.out_stride : P.N) + (size_t)(n_done + n);
        if (P.out.draws) {
            double* drow = P.out.draws + o * D;
#pragma unroll
            for (int k = 0; k < NPL; ++k)
                if (lane + WAVE * k < D) drow[lane + WAVE * k] = q[k];
        }
        window_accumulate<NPL>(P, (size_t)chain * P.Dpad, lane, q, n_done + n);
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
    pair_publish(c_quit, broken ? 2u : 1u, lane);
    PIPE_PT_FLUSH(2, total_steps)
    PIPE_B2_FLUSH
    stv<NPL>(P.st.q + row, lane, q);
    (void)tgt.eval(q, g, lane, D);
    stv<NPL>(P.st.g + row, lane, g);
    if (lane == 0) {
        P.st.lq[chain] = lq_cur;
        if (P.adapt) {
            P.st.da[chain] = da;
            if (P.da_finalize && !broken) P.st.eps[chain] = det_exp_u(da.logeps_bar); // final_ϵ (stepsize.jl:170; mcmc.jl:285); not from a broken call
        }
        P.st.transition[chain] = tr0 + (uint32_t)n;
        P.st.status[chain] = status;
        if (P.prog) P.prog[chain] = (int)(n_done + n);
        if (P.leapfrog_counter) atomicAdd(P.leapfrog_counter, total_steps);
        if (P.chain_work) {
            const unsigned long long w = total_steps + (P.prog ? (unsigned long long)P.chain_work[chain] : 0ull);
            P.chain_work[chain] = (unsigned)(w > 0xffffffffull ? 0xffffffffull : w);
        }
    }
}

#ifndef __HIPCC_RTC__      // (the host side; a caller's functor is compiled with hiprtc: kernels only)
template <class T>
int launch_run_pipeline(const RunParams& P, hipStream_t s) {
    if constexpr (!T::kRecomputeGrad || T::kBigDims) {
        return DHMC_ERR_UNSUPPORTED;
    } else {
        constexpr size_t pad = 0;
#define DHMC_PIPE_LAUNCH(NPL_)                                                                                                  \
    if (P.Dpad == WAVE * NPL_) {                                                                                                \
        static bool once = [] {                                                                                                 \
            (void)hipFuncSetAttribute((const void*)nuts_run_pipeline_kernel<T, NPL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(pipeline_lds_bytes(NPL_) + pad)); \
            return true;                                                                                                        \
        }();                                                                                                                    \
        (void)once;                                                                                                             \
        hipLaunchKernelGGL((nuts_run_pipeline_kernel<T, NPL_>), dim3(P.C), dim3(4 * WAVE), pipeline_lds_bytes(NPL_) + pad, s, P); \
        return DHMC_OK;                                                                                                         \
    }
        DHMC_PIPE_LAUNCH(1) DHMC_PIPE_LAUNCH(2) DHMC_PIPE_LAUNCH(4)
#undef DHMC_PIPE_LAUNCH
        return DHMC_ERR_UNSUPPORTED;
    }
}

#endif
}  // namespace dhmc
