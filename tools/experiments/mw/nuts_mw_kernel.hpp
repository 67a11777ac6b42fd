// The per-draw loop kernel for chains of 512+ coordinates (BASELINE config 2: D = 1000): a WORKGROUP per chain,
// one VECTOR wave per 256-coordinate block of the chain plus one CONTROL wave.
//
// Why.  The one-wave-per-chain kernel (nuts_kernels.hpp) holds a 1000-dim chain as 16 slots per lane: eight live
// D-vectors fill all 512 registers of a SIMD lane, so a SIMD runs ONE wave and every dependent-latency chain of the
// tree logic (the ≈75-deep deterministic logaddexp with its two fp64 divisions, the DPP butterflies, LDS and L2 round
// trips) is exposed: the kernel sat at 49 % VALU-active / 10 % of the fp64 vector peak.  Here each chain's vector work
// is cut into NW = Dpad/256 waves of four slots per lane (128 VGPRs), three chains (15 waves) share a CU, every SIMD
// holds waves of different chains that cover for each other — and the scalar chain is taken off the vector waves'
// critical path altogether:
//
//  * vector wave w owns coordinates 256 w + lane + 64 k, k = 0..3 — exactly one block of the ABI's dot product
//    (include/dhmc.h "Summation order", wave.hpp LaneAcc), so its per-lane partial sums are final: no wave-to-wave
//    fma chain.  It keeps (q, p, [∇ℓ]), the running subtree summary (first, Σp), the trajectory's ρ, its part of M⁻¹
//    and the previous leaf's position in registers; levels 0 and 1 of the suspended stack and the trajectory's edge
//    momenta in LDS; deeper levels / parked edges / proposal slots in the HBM workspace — and only ever touches its own
//    coordinates of any of those;
//  * the control wave owns everything scalar of sample_tree (reference src/NUTS.jl:232-241, src/trees.jl:231-319).
//    Its per-level / per-slot arrays live in the LANES of a few VGPRs (v_readlane / v_writelane), not in memory;
//  * they meet at SYNCHRONISATION POINTS: the vector waves leave per-lane partial sums in LDS, barrier, each vector
//    wave folds the blocks and does the 64-lane butterfly for ITS SHARE of the dots (value n belongs to wave n mod NW:
//    the reductions of one merge run in parallel on four SIMDs instead of one after the other on one), leaves the
//    scalars in LDS, barrier.  Every wave then reads the scalars and takes the same decisions from them — divergent?
//    turning? — with a few uniform instructions: a single wave issues one instruction every four cycles, so a control
//    wave in the decision loop was the bottleneck, not a help.  The control wave reads the same scalars and does
//    everything slow (both logaddexp's of the merge, the Exp(1) draw, proposal selection, slot bookkeeping,
//    termination, acceptance statistic) while the vector waves are already integrating the next leapfrog.  The one
//    thing the vector waves need from it — whether the leaf they just left must be kept as a proposal, and in which
//    slot — reaches them one leaf late through a mailbox word: they hold a copy of that leaf's position (qkeep) until
//    the next leaf's synchronisation says where to store it, if anywhere;
//  * one synchronisation per leaf (which also carries the level-0 / depth-0 leaf+leaf merge: half of all merges, two
//    dots, computed speculatively with the leapfrog), one per higher merge, one per transition (final proposal, next
//    ϵ and directions; the next momentum is sampled before it, while the control wave does dual averaging).
//    Both sides run the same replicated integer control (direction bits, depth, leaf counter, cascade level) from
//    the same scalars, so they execute the same barrier sequence by construction.
//
// Arithmetic, RNG consumption and every output bit equal nuts_run_kernel's and the oracle's (DHMC_MW=0 selects the
// one-wave kernel; tests/test_gpu_engines.py compares the two).
#pragma once
#include "nuts_kernels.hpp"

namespace dhmc {

constexpr int MW_NK = 4;                                     // slots per lane of a vector wave: one 256-coordinate block
// mailbox words (control wave -> vector waves): pend = 1 + proposal slot the PREVIOUS leaf's position (qkeep) must be
// written to (0: none), read at every leaf and at the end of a transition; zeta = ζ of the trajectory = the slot that
// becomes the chain's position, read at the end of a transition.

// LDS of one chain: five Dpad-rows (suspended levels 0 and 1, the trajectory's two edge momenta), the vector waves'
// per-lane partial sums, the reduced scalars and a small mailbox: 53.3 KB at Dpad = 1024.
__host__ __device__ inline size_t mw_lds_bytes(int Dpad, int NW) {
    return sizeof(double) * ((size_t)5 * Dpad + 6 * NW * WAVE + 6 + 2) + sizeof(uint32_t) * (4 + NW);
}

struct MwLds {
    double *l0, *l1f, *l1l;          // [Dpad] each: suspended level 0 momentum, level 1 (first, last)
    double *tpm, *tpp;               // [Dpad] each: p₋ and p₊ of the whole trajectory (its ρ stays in registers)
    double* part;                    // [6][NW][64] per-lane partial sums of the vector waves
    double* red;                     // [6] the reduced dots of the current synchronisation point
    double* mb_f;                    // mailbox: [0] ϵ of the transition, [1] ℓq of the chain's position
    uint32_t* mb_u;                  // mailbox: [0], [1] pend (by parity of the synchronisation point), [2] zeta, [3] directions,
                                     //          [4 + w] position-scan flag of vector wave w
};

__device__ __forceinline__ MwLds mw_carve(double* lds, int Dpad, int NW) {
    MwLds L;
    L.l0 = lds;
    L.l1f = lds + Dpad;
    L.l1l = lds + 2 * Dpad;
    L.tpm = lds + 3 * Dpad;
    L.tpp = lds + 4 * Dpad;
    L.part = lds + 5 * Dpad;
    L.red = L.part + 6 * NW * WAVE;
    L.mb_f = L.red + 6;
    L.mb_u = (uint32_t*)(L.mb_f + 2);
    return L;
}

// Timeline instrumentation for tools/experiments (compiled in with -DMW_TRACE only): chain 0's waves append
// (event << 56 | clock) words to a device array that dhmc_debug_mw_trace copies out.
#ifdef MW_TRACE
__device__ unsigned long long g_mw_trace[8][4096];
__device__ unsigned int g_mw_trace_n[8];
#define MW_T(wave_, ev_)                                                                                   \
    do {                                                                                                   \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                                                  \
            const unsigned int i_ = g_mw_trace_n[wave_]++;                                                 \
            if (i_ < 4096u) g_mw_trace[wave_][i_] = ((unsigned long long)(ev_) << 56) | (__builtin_readcyclecounter() & 0x00ffffffffffffffull); \
        }                                                                                                  \
    } while (0)
#else
#define MW_T(wave_, ev_) do {} while (0)
#endif

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

// One vector wave's share of a synchronisation point's reductions: value n (n < N) belongs to wave n mod NW.  Blocks
// folded per lane by adjacent pairs, then the 64-lane butterfly (the ABI's order); the scalar goes to red[n].
template <int NW>
__device__ __forceinline__ double mw_fold_blocks(const double* __restrict__ part, int n, int lane) {
    double t[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) t[w] = part[(n * NW + w) * WAVE + lane];
#pragma unroll
    for (int s = 1; s < NW; s *= 2)
#pragma unroll
        for (int w = 0; w + s < NW; w += 2 * s) t[w] = t[w] + t[w + s];
    return t[0];
}
template <int NW>
__device__ __forceinline__ void mw_reduce_share(const MwLds& L, int N, int wv, int lane) {
    for (int n0 = wv; n0 < N; n0 += 2 * NW) {
        const int n1 = n0 + NW;
        if (n1 < N) {                 // two values on this wave: their butterflies interleave
            double r[2] = {mw_fold_blocks<NW>(L.part, n0, lane), mw_fold_blocks<NW>(L.part, n1, lane)};
            wave_allreduce<2>(r);
            if (lane == 0) { L.red[n0] = r[0]; L.red[n1] = r[1]; }
        } else {
            const double r = wave_allreduce1(mw_fold_blocks<NW>(L.part, n0, lane));
            if (lane == 0) L.red[n0] = r;
        }
    }
}

// What every wave derives from the scalars of a leaf synchronisation (same data, same code: same decisions everywhere).
struct MwLeaf {
    double lq, pi;
    bool div, turning, pos_finite;
};
// leaf (NUTS.jl:148-159) from ℓ's reduced sum and p·M⁻¹p; evaluate_ℓ's demotion rules (hamiltonian.jl:202-217)
template <class T>
__device__ __forceinline__ MwLeaf mw_leaf_decide(const T& tgt, const double* red, bool fused, bool pos_finite, double pi0, double min_delta) {
    MwLeaf R;
    double lq = uni_f64(tgt.finish(uni_f64(red[0])));
    const double K = uni_f64(red[1]) / 2.0;
    lq = demote_lq(lq, pos_finite, true);
    R.lq = lq;
    R.pi = uni_f64(joint_logdensity(lq, K));
    R.div = (R.pi - pi0) < min_delta;                               // NUTS.jl:150-151
    R.turning = fused && (uni_f64(red[2]) < 0 || uni_f64(red[3]) < 0);
    R.pos_finite = pos_finite;
    return R;
}

// ------------------------------------------------------------------------------------------------------------
// Vector wave `wv` of the chain: every D-vector operation of the transition on coordinates 256 wv + lane + 64 k.
// Partial sums of a leaf synchronisation: [0] Σ of ℓ's summand, [1] p·M⁻¹p, [2], [3] the two dots of the leaf+leaf
// merge that follows (odd leaves, and the first leaf of a transition), [4] p₀·M⁻¹p₀ (first leaf of a transition).
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

    double m[NK], q[NK], p[NK], g[NK], cf[NK], cr[NK], trho[NK], qkeep[NK];
    ldk<NK>(P.st.minv + row, m);
    ldk<NK>(P.st.q + row, q);
    ldk<NK>(P.st.g + row, g);
#pragma unroll
    for (int k = 0; k < NK; ++k) { cf[k] = 0.0; cr[k] = 0.0; qkeep[k] = q[k]; }

    uint32_t par = 0;                 // parity of the synchronisation point: which pend word the control wave wrote for it
    auto sync = [&](int N) {          // N per-lane partial sums are in LDS: reduce my share, leave the scalars in L.red
        par ^= 1u;
        MW_T(wv, 1);
        mw_barrier();
        MW_T(wv, 2);
        mw_reduce_share<NW>(L, N, wv, lane);
        MW_T(wv, 3);
        mw_barrier();
        MW_T(wv, 4);
    };
    auto sample_momentum_block = [&](uint32_t tr) {             // rand_p (hamiltonian.jl:124) on this block
#pragma unroll
        for (int kk = 0; kk < NK / 2; ++kk) {
            uint64_t r1, r2;
            stream_raw64(key, idx_base + (uint32_t)(WAVE * kk), PURPOSE_MOMENTUM, tr, r1, r2);
            double z0, z1;
            det_randn2(r1, r2, &z0, &z1);
            p[2 * kk] = Wrow[WAVE * (2 * kk)] * z0;
            p[2 * kk + 1] = Wrow[WAVE * (2 * kk + 1)] * z1;
        }
    };
    auto flush_keep = [&]() -> uint32_t {                       // the previous leaf became a proposal: materialise it
        const uint32_t slot = uni_u32(L.mb_u[par]);
        if (slot) stk<NK>(wsv(ws_slot(max_depth, (int)slot - 1, 0)), qkeep);
        return slot;
    };

    int init_slot = 0;
    stk<NK>(wsv(ws_slot(max_depth, init_slot, 0)), q);
    sample_momentum_block(tr0);
    sync(0);                                                     // ϵ, directions and ℓq of the first transition

    for (int64_t n = 0; n < P.N; ++n) {
        const uint32_t tr = tr0 + (uint32_t)n;
        const double eps = uni_f64(L.mb_f[0]);
        const double lq_cur = uni_f64(L.mb_f[1]);
        uint32_t dirs = uni_u32(L.mb_u[3]);
        double pi0 = 0.0;
        double kin0 = 0.0;                                       // this block's part of K(p₀) for π₀
#pragma unroll
        for (int k = 0; k < NK; ++k) kin0 = __builtin_fma(p[k], m[k] * p[k], kin0);
        stk<NK>(tpm, p);                                         // leaf τ of z₀ (NUTS.jl:120-123)
        stk<NK>(tpp, p);
#pragma unroll
        for (int k = 0; k < NK; ++k) trho[k] = p[k];

        // ---- sample_trajectory (trees.jl:283-319) ------------------------------------------------------
        bool stored0 = false, stored1 = false;
        int reg_edge = 2;
        int depth = 0;
        bool finished = false;
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
                // an odd leaf is followed by its level-0 merge, the first leaf of a transition by the depth-0 top merge:
                // both subtrees are single leaves (merge_leaf_leaf of nuts_kernels.hpp: two distinct dots) — computed
                // now, speculatively, so that the leaf and that merge cost one synchronisation
                const bool fused = (j & 1u) != 0 || depth == 0;
                if (fused) {
                    double a0 = 0.0, a1 = 0.0;
#pragma unroll
                    for (int k = 0; k < NK; ++k) {
                        const double pa = depth == 0 ? trho[k] : l0[WAVE * k];
                        const double r = pa + p[k];
                        a0 = __builtin_fma(m[k] * pa, r, a0);
                        a1 = __builtin_fma(m[k] * p[k], r, a1);
                        cf[k] = pa;
                        cr[k] = r;
                    }
                    my_part[2 * PS] = a0;
                    my_part[3 * PS] = a1;
                    if (depth == 0) my_part[4 * PS] = kin0;
                }
                sync(depth == 0 ? 5 : (fused ? 4 : 2));
                if (depth == 0) pi0 = uni_f64(joint_logdensity(lq_cur, uni_f64(L.red[4]) / 2.0));   // logdensity(H, z₀) (NUTS.jl:234)
                bool pos_finite = true;
                if (!dm_isfinite(uni_f64(tgt.finish(uni_f64(L.red[0]))))) {
                    // ℓq came out non-finite: evaluate_ℓ's position scan (hamiltonian.jl:203), every wave for its block
                    bool fin = true;
#pragma unroll
                    for (int k = 0; k < NK; ++k) fin = fin && dm_isfinite(q[k]);
                    const bool all = wave_all(fin);
                    if (lane == 0) L.mb_u[4 + wv] = all ? 1u : 0u;
                    mw_barrier();
                    mw_barrier();
                    uint32_t ok = 1u;
#pragma unroll
                    for (int w = 0; w < NW; ++w) ok &= uni_u32(L.mb_u[4 + w]);
                    pos_finite = ok != 0u;
                }
                const MwLeaf lf = mw_leaf_decide(tgt, L.red, fused, pos_finite, pi0, P.min_delta);
                MW_T(wv, 5);
                (void)flush_keep();                                         // where the previous leaf goes, if anywhere
#pragma unroll
                for (int k = 0; k < NK; ++k) qkeep[k] = q[k];               // this leaf's position, until its fate is known
                int level = 0;
                if (lf.div) {
                    invalid = true;
                } else {
                    bool first = fused;
                    bool turning = lf.turning;
                    for (;;) {
                        const bool sub = ((j >> level) & 1u) != 0;
                        const bool top = !sub && (j == nleaf - 1) && (level == depth);
                        if (!sub && !top) break;
                        if (!first) {
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
                            MW_T(wv, 6);
                            sync(6);
                            turning = uni_f64(L.red[0]) < 0 || uni_f64(L.red[1]) < 0 || uni_f64(L.red[2]) < 0 ||
                                      uni_f64(L.red[3]) < 0 || uni_f64(L.red[4]) < 0 || uni_f64(L.red[5]) < 0;
                        }
                        first = false;
                        if (sub) {
                            level += 1;
                            if (turning) { invalid = true; break; }      // trees.jl:255
                        } else {
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

        // ---- the next momentum (while the control wave finishes the transition), then the new position
        //      (NUTS.jl:238-240) and the draw (mcmc.jl:275,376) ----------------------------------------------
        MW_T(wv, 7);
        if (n + 1 < P.N) sample_momentum_block(tr + 1u);
        MW_T(wv, 8);
        sync(0);
        const uint32_t last_slot = flush_keep();
        init_slot = (int)uni_u32(L.mb_u[2]);
        if (last_slot == (uint32_t)(init_slot + 1)) {
#pragma unroll
            for (int k = 0; k < NK; ++k) q[k] = qkeep[k];                   // the last leaf is the draw: no round trip
        } else {
            ldk<NK>(wsv(ws_slot(max_depth, init_slot, 0)), q);
        }
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
    LaneArrF64 lv_omega, lv_vlsa, lv_vsteps;      // per suspended level (lane = level): ω, visited statistic
    LaneArrI32 lv_zeta;                           //   … and the proposal slot
    LaneArrF64 sl_lq, sl_pi;                      // per proposal slot (lane = slot): ℓq and π of the point stored there
    int init_slot = 0;
    uint64_t free_mask = 0;
    uint32_t pend = 0;                            // 1 + slot the vector waves must write the leaf they last left to
    uint32_t par = 0;
    // a synchronisation point seen from here: say where the leaf the vector waves last left goes (if this is a point
    // where they look), then wait for them to reduce; the scalars are in L.red afterwards
    auto csync = [&](bool leaf_or_final) {
        par ^= 1u;
        if (lane == 0) L.mb_u[par] = leaf_or_final ? pend : 0u;
        if (leaf_or_final) pend = 0;
        MW_T(NW, 11);
        mw_barrier();
        MW_T(NW, 12);
        mw_barrier();
        MW_T(NW, 13);
    };
    auto alloc_slot = [&](double lq_leaf, double pi_leaf) -> int {   // bookkeeping of save_leaf: the vector waves store q
        const int s = __builtin_ctzll(free_mask);
        free_mask &= ~(1ull << s);
        sl_lq.set(s, lq_leaf, lane);
        sl_pi.set(s, pi_leaf, lane);
        pend = (uint32_t)(s + 1);
        return s;
    };

    // ϵ and the directions of a transition (stepsize.jl:163; trees.jl:23), left in the mailbox for the vector waves
    double eps = 0.0;
    uint32_t dirs = 0, directions0 = 0;
    uint32_t nrand = 0, rexp_base = 0;
    double rexp_vals = 0.0;
    uint32_t tr = tr0;
    auto rexp_fill = [&](uint32_t base) {         // Exp(1) draws of this transition, 64 at a time: lane l holds draw (base + l)
        uint64_t r1, r2;
        stream_raw64(key, base + (uint32_t)lane, PURPOSE_TREE, tr, r1, r2);
        rexp_vals = det_randexp(r1);
        rexp_base = base;
    };
    auto randexp = [&]() -> double {              // Random.randexp at NUTS.jl:44
        if (nrand - rexp_base >= 64u) rexp_fill(nrand & ~63u);
        const double v = readlane_f64(rexp_vals, (int)(nrand & 63u));
        nrand += 1;
        return v;
    };
    auto begin_transition = [&]() {
        eps = uni_f64(P.adapt ? det_exp(da.logeps) : eps_fixed);
        uint32_t w[4];
        philox4x32_10(0u, PURPOSE_DIRECTIONS, tr, key.seed_hi, key.k0, key.k1, w);
        dirs = uni_u32(w[0]);
        directions0 = dirs;
        if (lane == 0) {
            L.mb_f[0] = eps;
            L.mb_u[3] = dirs;
        }
        nrand = 0;
        rexp_fill(0);
    };

    begin_transition();
    if (lane == 0) { L.mb_f[1] = lq_cur; L.mb_u[2] = 0u; }
    csync(true);

    for (int64_t n = 0; n < P.N; ++n) {
        sl_lq.set(init_slot, lq_cur, lane);
        double pi0 = 0.0;
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
                // ---- the leaf (NUTS.jl:148-159), and the leaf+leaf merge's turn test when it follows at once -----
                const bool fused = (j & 1u) != 0 || depth == 0;
                csync(true);
                if (depth == 0) {
                    pi0 = uni_f64(joint_logdensity(lq_cur, uni_f64(L.red[4]) / 2.0));   // logdensity(H, z₀) (NUTS.jl:234)
                    sl_pi.set(init_slot, pi0, lane);
                }
                bool pos_finite = true;
                if (!dm_isfinite(uni_f64(tgt.finish(uni_f64(L.red[0]))))) {   // evaluate_ℓ's position scan is the vector waves' (hamiltonian.jl:203)
                    mw_barrier();
                    mw_barrier();
                    uint32_t ok = 1u;
#pragma unroll
                    for (int w = 0; w < NW; ++w) ok &= uni_u32(L.mb_u[4 + w]);
                    pos_finite = ok != 0u;
                }
                const MwLeaf lf = mw_leaf_decide(tgt, L.red, fused, pos_finite, pi0, P.min_delta);
                if (!pos_finite) status |= DHMC_ST_NONFINITE_POSITION;
                const double lq_leaf = lf.lq, pi_leaf = lf.pi;
                bool turning = lf.turning;
                const double delta = pi_leaf - pi0;             // NUTS.jl:150
                const bool div = lf.div;                        // divergent leaf (NUTS.jl:151; trees.jl:236-237)
                MW_T(NW, 14);
                // ---- from here on the vector waves are already integrating the next leapfrog ---------------
                i += di;
                total_steps += 1;
                v_lsa = delta < 0.0 ? delta : 0.0;              // min(Δ, 0)   (NUTS.jl:79)
                v_steps = 1;
                int level = 0;
                if (div) {
                    term_left = term_right = i;
                    invalid = true;
                } else {
                    double c_omega = delta;
                    int c_zeta = -1;  // -1: the proposal is the leaf the vector waves keep in qkeep
                    bool first = fused;
                    for (;;) {
                        const bool sub = ((j >> level) & 1u) != 0;
                        const bool top = !sub && (j == nleaf - 1) && (level == depth);
                        if (!sub && !top) break;
                        if (!first) {
                            csync(false);
                            turning = uni_f64(L.red[0]) < 0 || uni_f64(L.red[1]) < 0 || uni_f64(L.red[2]) < 0 ||
                                      uni_f64(L.red[3]) < 0 || uni_f64(L.red[4]) < 0 || uni_f64(L.red[5]) < 0;
                        }
                        first = false;
                        // v = v₋ ⊕ v₊ (trees.jl:249 / :294) and ω = logaddexp(ω₋, ω₊) (trees.jl:145), one pass on even / odd lanes
                        double v_new, w;
                        if (sub) logaddexp_pair(lv_vlsa.get(level), v_lsa, lv_omega.get(level), c_omega, lane, v_new, w);
                        else logaddexp_pair(vtop_lsa, v_lsa, omega_top, c_omega, lane, v_new, w);
                        if (sub) {
                            v_lsa = v_new;
                            v_steps += (int64_t)lv_vsteps.get(level);
                            if (turning) {                       // trees.jl:255
                                term_left = i - di * (((int64_t)2 << level) - 1);
                                term_right = i;
                                invalid = true;
                                level += 1;
                                break;
                            }
                            // combine_proposals_and_logweights(…, is_doubling = false) (trees.jl:258)
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
                            // top level (trees.jl:294-316)
                            vtop_lsa = v_new;
                            vtop_steps += v_steps;
                            const double logprob2 = c_omega - omega_top;   // biased progressive (trees.jl:159-161)
                            const bool pick = logprob2 >= 0.0 || (randexp() > -logprob2);
                            if (pick) {
                                if (c_zeta < 0) c_zeta = alloc_slot(lq_leaf, pi_leaf);
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
                            level = -1;  // handled
                            break;
                        }
                    }
                    if (level >= 0 && !invalid) {
                        // the running subtree is suspended at `level` (the vector waves store its vectors)
                        if (c_zeta < 0) c_zeta = alloc_slot(lq_leaf, pi_leaf);
                        lv_omega.set(level, c_omega, lane);
                        lv_vlsa.set(level, v_lsa, lane);
                        lv_vsteps.set(level, (double)v_steps, lane);
                        lv_zeta.set(level, c_zeta, lane);
                    }
                }
                if (invalid) {
                    // unwind the recursion: every suspended left sibling contributes its visited statistic (trees.jl:244,249-250)
                    for (int l2 = level; l2 < depth; ++l2) {
                        if ((j >> l2) & 1u) {
                            v_lsa = uni_f64(det_logaddexp(lv_vlsa.get(l2), v_lsa));
                            v_steps += (int64_t)lv_vsteps.get(l2);
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
        lq_cur = sl_lq.get(init_slot);
        const double pi_stat = sl_pi.get(init_slot);
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
        // ---- the end of the transition for the vector waves: where the last leaf goes, the new position's slot and
        //      ℓq; the next ϵ and directions
        tr += 1u;
        if (n + 1 < P.N) begin_transition();
        if (lane == 0) { L.mb_f[1] = lq_cur; L.mb_u[2] = (uint32_t)zeta_top; }
        csync(true);
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
__global__ __launch_bounds__(WAVE * (NW + 1), 5) void nuts_run_mw_kernel(RunParams P) {
    static_assert(T::kElementwise && T::kDeferred && T::kRecomputeGrad && T::kFiniteLqImpliesFiniteQ && T::kFiniteLqImpliesFiniteGrad,
                  "the multi-wave kernel serves coordinate-wise targets whose position scan is needed only on a non-finite ℓq");
    extern __shared__ double lds[];
    const MwLds L = mw_carve(lds, P.Dpad, NW);
    const int chain = blockIdx.x;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = uni_i32((int)(threadIdx.x >> 6));
    if (wv == NW) mw_control_wave<T, NW>(P, L, chain, lane);
    else mw_vector_wave<T, NW>(P, L, chain, wv, lane);
}

}  // namespace dhmc
