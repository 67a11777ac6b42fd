// HIP kernels of the many-chain NUTS hot path for MI355X (gfx950).  One wavefront = one chain.
//
// What runs here, per chain and per transition, is the body of the reference's per-draw loops
// (src/mcmc.jl:271-280 and :374-379): sample_tree (src/NUTS.jl:232-241) — momentum refresh
// (hamiltonian.jl:124), tree doubling (trees.jl:283-319) with the recursive adjacent_tree
// (trees.jl:231-262) unrolled into an iterative binary-counter merge, leapfrog
// (hamiltonian.jl:273-282), joint log density (hamiltonian.jl:251-256), the generalised
// U-turn test (NUTS.jl:130-139), multinomial / biased progressive proposal selection
// (trees.jl:143-161, NUTS.jl:43-53), the acceptance statistic (NUTS.jl:59-89) — followed by
// the dual-averaging update (stepsize.jl:147-156).
//
// Mapping to the machine
//  * lane l of the chain's wave owns coordinates l, l+64, ... of every D-vector (wave.hpp);
//    the phase point being integrated (q, p, ∇ℓ) and the running subtree summary of the
//    merge cascade (first momentum, Σp) stay in VGPRs across leapfrog steps; control flow
//    (direction, depth, validity, accept) is wave-uniform and runs on the scalar unit.
//  * per-chain diagonal M⁻¹ is staged once per launch in LDS;
//  * suspended subtree summaries (one per set bit of the leaf counter), the two trajectory
//    edges and the candidate proposals live in a per-chain HBM workspace that is touched
//    LIFO, so it is served by L2 / Infinity Cache.  Proposals are never copied: a merge
//    picks a slot index (the reference's pointer selection, NUTS.jl:52), and a leaf's
//    position is written to a slot only if it survives its whole merge cascade.
//  * dot products: NPL lane-local fma's + one batched 64-lane butterfly per group of dots.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dhmc.h"
#include "../../include/dhmc_detmath.h"
#include "detmath_dev.hpp"
#include "philox_dev.hpp"
#include "run_params.hpp"
#include "targets.hpp"
#include "wave.hpp"

namespace dhmc {

// Scheduling fence between 256-coordinate blocks of a per-slot loop (experiment switch -DDHMC_SCHED_BLOCKS): keeps the compiler
// from hoisting the loads of all 16 slots to the top of a loop, which is what decides the kernel's register budget.
#ifdef DHMC_SCHED_BLOCKS
#define DHMC_BLOCK_FENCE(k) do { if (((k) & 3) == 3) __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DHMC_BLOCK_FENCE(k) do { } while (0)
#endif


// One chain's position after its transition number `n_in_call` of this launch joins the window's moments: rows at `row`
// (chain * Dpad [+ the wave's offset]), lane l's slot k is coordinate l + 64k of that row.  Pads hold zeros throughout.
template <int N>
__device__ __forceinline__ void window_accumulate(const RunParams& P, size_t row, int lane, const double (&q)[N], int64_t n_in_call) {
#ifdef DHMC_NO_WINDOW      // A/B builds only (tools/experiments/build_variant_fast.sh): what the window's code costs a run without one
    return;
#endif
    if (!P.win_mean) return;
    const double rn = 1.0 / (double)(P.win_n0 + n_in_call + 1);
    double* __restrict__ mrow = P.win_mean + row;
    double* __restrict__ srow = P.win_m2 + row;
    // four slots at a time (eight loads in flight): the whole row at once would cost the per-draw kernel 32 more live registers
    // at the end of a transition, and the compiler pays for those with scratch in the tree loop
    constexpr int B = N < 4 ? N : 4;
#pragma unroll
    for (int k0 = 0; k0 < N; k0 += B) {
        double mean[B], m2[B];
#pragma unroll
        for (int k = 0; k < B; ++k) { mean[k] = mrow[lane + WAVE * (k0 + k)]; m2[k] = srow[lane + WAVE * (k0 + k)]; }
#pragma unroll
        for (int k = 0; k < B; ++k) {
            dm_window_update(q[k0 + k], rn, &mean[k], &m2[k]);
            mrow[lane + WAVE * (k0 + k)] = mean[k];
            srow[lane + WAVE * (k0 + k)] = m2[k];
        }
        asm volatile("" ::: "memory");
    }
}


// LDS carve (one wave per block): m[Dpad], the level-0 suspended momentum [Dpad], (optionally)
// the level-1 suspended summary first/last [2][Dpad] — its ρ is first+last, recomputed — then
// per-level and per-slot scalars.
constexpr int LDS_LEVELS = 32;
constexpr int LDS_SLOTS = 36;
// Short chains keep MORE of the suspended stack in LDS: levels 2 .. 1+lds_extra_levels(NPL) (first, last, ρ: 3 rows
// each), ≈11-14 KB per wave in all.  A level that lives in the HBM workspace costs an exposed L2/HBM round trip
// (≈2 µs, more than a whole small-D leapfrog) every time a subtree of that size is merged — 1/8 of all leaves for
// levels ≥ 3 — and a wave-per-chain kernel on a short chain has nothing else to hide it behind.
__host__ __device__ constexpr int lds_extra_levels(int NPL) { return NPL == 1 ? 6 : NPL == 2 ? 3 : NPL == 4 ? 1 : 0; }
// The per-level / per-slot scalars of the tree logic live in the lanes of a few registers (wave.hpp LaneArr), not in LDS.
// Wide chains (16 slots per lane, kTrajInLds): M⁻¹ stays in registers and the rows are (level 0, level 1 first / last,
// trajectory p₋ / p₊) — 40 KB per wave, four waves fill the CU's 160 KB exactly.
// (Every target family since round 3: round 2 restricted it to coordinate-wise targets after a fault in a fuzz sweep that
// no longer reproduces — docs/DESIGN_history_rounds1-4.md §10 — and a tridiagonal-precision normal at D = 1000 runs 2.9× faster with it.)
__host__ __device__ constexpr bool traj_in_lds(int NPL, bool l1_in_lds) { return NPL == 16 && l1_in_lds; }
__host__ __device__ inline size_t lds_bytes(int Dpad, bool l1_in_lds, int extra_levels) {
    if (traj_in_lds(Dpad / 64, l1_in_lds)) return sizeof(double) * (size_t)Dpad * 5;
    return sizeof(double) * ((size_t)Dpad * ((l1_in_lds ? 4 : 2) + 3 * extra_levels));
}

__device__ __forceinline__ double joint_logdensity(double lq, double K) {  // hamiltonian.jl:251-256
    if (!dm_isfinite(lq)) return -dm_inf();
    return lq - (dm_isfinite(K) ? K : dm_inf());
}

// evaluate_ℓ(ℓ, q) (hamiltonian.jl:202-217) + the kinetic part of logdensity for the point
// (q, p): returns ℓq after the -Inf demotion rules and K = p·M⁻¹p / 2.  `p` must already be
// the full-step momentum computed from g.  Sets *pos_bad when the position has a non-finite
// coordinate (the reference throws there, :203).
template <class T, int NPL>
__device__ __forceinline__ bool all_finite(const double (&v)[NPL]) {
    bool fin = true;
#pragma unroll
    for (int k = 0; k < NPL; ++k) fin = fin && dm_isfinite(v[k]);
    return wave_all(fin);
}

__device__ __forceinline__ double demote_lq(double lq, bool pos_finite, bool grad_finite) {
    if (!pos_finite) return -dm_inf();
    bool ok = (dm_isfinite(lq) && grad_finite) || lq == -dm_inf();
    return ok ? lq : -dm_inf();
}

// One leapfrog step in registers (hamiltonian.jl:273-282) followed by the leaf's joint log
// density.  eps is signed (backward motion = negative ϵ, NUTS.jl:30).
// `merge0` (wave-uniform; `pa0_` = slot k of the suspended level-0 leaf's momentum): the leaf that completes a pair also takes
// the pair's leaf·leaf turn check — merge_leaf_leaf's two dots — and their partial sums ride in the SAME butterfly as ℓ and K
// (four values instead of two, one reduction latency instead of two; the suspended row's LDS round trip runs under the
// leapfrog's arithmetic).  Same operations on the same operands, value by value, so the same bits as the separate merge; if the
// leaf turns out divergent the merge's result is simply not looked at (cf / cr are dead then).
template <class T, int NPL, bool XL = false, class MK, class PA0>
__device__ __forceinline__ void leapfrog_leaf_m(const T& tgt, MK mk_, int lane, int D,
                                                double (&q)[NPL], double (&p)[NPL], double (&g)[NPL],
                                                double eps, double& lq_out, double& pi_out, bool& pos_finite, int nl,
                                                bool merge0, PA0 pa0_, double (&cf)[NPL], double (&cr)[NPL], bool& turning0) {
    const double h = eps / 2;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        double gk;
        if constexpr (T::kPointwiseGrad) gk = tgt.grad1(q[k], lane + WAVE * k);   // ∇ℓq recomputed, not carried
        else gk = g[k];
        double pm = p[k] + h * gk;                   // :277
        double t = mk_(k) * pm;                      // ∇kinetic_energy(κ, pₘ) = M⁻¹ pₘ
        q[k] = q[k] + eps * t;                       // :278
        p[k] = pm;
        DHMC_BLOCK_FENCE(k);
    }
    const double lres = tgt.eval(q, g, lane, D);     // :279 -> hamiltonian.jl:204
    LaneAcc<1, NPL> kacc;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        p[k] = p[k] + h * g[k];                      // :280
        double ps = mk_(k) * p[k];                   // p♯ = M⁻¹ p'
        kacc.add(0, k, p[k], ps);
        DHMC_BLOCK_FENCE(k);
    }
    double lq, K;
    turning0 = false;
    if (merge0) {
        LaneAcc<2, NPL> A;                           // merge_leaf_leaf, word for word
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const double pa = pa0_(k);
            const double mk = mk_(k);
            const double r = pa + p[k];
            A.add(0, k, mk * pa, r);
            A.add(1, k, mk * p[k], r);
            cf[k] = pa;
            cr[k] = r;
        }
        if constexpr (T::kDeferred) {
            double r[4] = {lres, kacc.fold(0), A.fold(0), A.fold(1)};
            wave_totals<4, XL>(r, nl);
            lq = tgt.finish(r[0]);
            K = r[1] / 2.0;
            turning0 = r[2] < 0 || r[3] < 0;
        } else {
            double r[3] = {kacc.fold(0), A.fold(0), A.fold(1)};
            wave_totals<3, XL>(r, nl);
            lq = lres;
            K = r[0] / 2.0;
            turning0 = r[1] < 0 || r[2] < 0;
        }
    } else if constexpr (T::kDeferred) {
        double r[2] = {lres, kacc.fold(0)};
        wave_totals<2, XL>(r, nl);
        lq = tgt.finish(r[0]);
        K = r[1] / 2.0;
    } else {
        lq = lres;
        K = wave_total1<XL>(kacc.fold(0), nl) / 2.0;
    }
    // evaluate_ℓ's checks (hamiltonian.jl:203-211).  Every shipped family has "ℓq finite =>
    // all q finite" (kFiniteLqImpliesFiniteQ), so the coordinate scan runs only on the rare
    // non-finite ℓq; the gradient scan only for families that need it.
    lq = uni_f64(lq);
    pos_finite = true;
    bool gfin = true;
    if (!T::kFiniteLqImpliesFiniteQ || !dm_isfinite(lq)) {
        double t[NPL];                               // opaque copies: the scan cannot be hoisted into the hot path
#pragma unroll
        for (int k = 0; k < NPL; ++k) asm volatile("" : "=v"(t[k]) : "0"(q[k]));
        pos_finite = all_finite<T, NPL>(t);
    }
    if constexpr (!T::kFiniteLqImpliesFiniteGrad) gfin = all_finite<T, NPL>(g);
    lq = demote_lq(lq, pos_finite, gfin);
    lq_out = lq;
    pi_out = uni_f64(joint_logdensity(lq, K));
}
// Measured (round 4, D = 1000 × 4096 chains, same box): 3.19e8 leapfrog-steps/s without the fusion, 3.01e8 with it — the merge's
// rows live through the density evaluation and the allocator pays in AGPR moves (the leaf region +650 clocks per leapfrog, the
// merges −315): bit-exact, slower, off.  -DDHMC_FUSE_LEAF_MERGE0 builds it.
// The permuted wave of wave.hpp ("the same tree on a permuted wave") for coordinate-wise targets of 128+ coordinates; -DDHMC_XL=0
// builds the plain lane order (A/B timing).
#ifndef DHMC_XL
#define DHMC_XL 1
#endif
constexpr bool kXlLanes = DHMC_XL != 0;
#ifdef DHMC_FUSE_LEAF_MERGE0
constexpr bool kFuseLeafMerge0 = true;
#else
constexpr bool kFuseLeafMerge0 = false;
#endif
// the leaf alone
template <class T, int NPL, class MK>
__device__ __forceinline__ void leapfrog_leaf_m(const T& tgt, MK mk_, int lane, int D,
                                                double (&q)[NPL], double (&p)[NPL], double (&g)[NPL],
                                                double eps, double& lq_out, double& pi_out, bool& pos_finite, int nl = 64) {
    double cf_[NPL], cr_[NPL];
    bool t0;
    leapfrog_leaf_m<T, NPL, false>(tgt, mk_, lane, D, q, p, g, eps, lq_out, pi_out, pos_finite, nl, false, [](int) { return 0.0; }, cf_, cr_, t0);
}

// M⁻¹ staged in LDS (or any lane-strided row)
template <class T, int NPL>
__device__ __forceinline__ void leapfrog_leaf(const T& tgt, const double* __restrict__ m_lds, int lane, int D,
                                              double (&q)[NPL], double (&p)[NPL], double (&g)[NPL],
                                              double eps, double& lq_out, double& pi_out, bool& pos_finite) {
    leapfrog_leaf_m<T, NPL>(tgt, LdsRow{m_lds, lane}, lane, D, q, p, g, eps, lq_out, pi_out, pos_finite);
}

// combine_turn_statistics (NUTS.jl:132-139) of two adjacent subtrees, time-ordered
// (trees.jl:135-141): x = earlier in time, y = later, each given as accessors k -> slot k of
// its (p₋, p₊, ρ).  nf(k) is what becomes the merged summary's build-order first momentum.
// On return cf = nf, cr = ρ of the merge.  Returns turning.
template <int NPL, bool XL = false, class XM, class XP, class XR, class YM, class YP, class YR, class NF, class MK>
__device__ __forceinline__ bool merge_core(XM xm_, XP xp_, XR xr_, YM ym_, YP yp_, YR yr_, NF nf_, MK mk_,
                                           double (&cf)[NPL], double (&cr)[NPL], int nl = 64) {
    LaneAcc<6, NPL> A;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const double xm = xm_(k), xp = xp_(k), xr = xr_(k);
        const double ym = ym_(k), yp = yp_(k), yr = yr_(k);
        const double nf = nf_(k);
        const double mk = mk_(k);
        const double s1 = xr + ym;      // x.ρ + y.p₋      (:134)
        const double s2 = xp + yr;      // x.p₊ + y.ρ      (:135)
        const double r = xr + yr;       // ρ               (:136)
        const double a = mk * xm;       // x.p♯₋
        const double b = mk * ym;       // y.p♯₋
        const double c = mk * xp;       // x.p♯₊
        const double d = mk * yp;       // y.p♯₊
        A.add(0, k, a, s1);
        A.add(1, k, b, s1);
        A.add(2, k, c, s2);
        A.add(3, k, d, s2);
        A.add(4, k, a, r);
        A.add(5, k, d, r);
        cf[k] = nf;
        cr[k] = r;
        DHMC_BLOCK_FENCE(k);
    }
    double acc[6];
    A.fold_all(acc);
    if constexpr (XL) {                  // only the signs are asked for: no total leaves the vector unit (wave.hpp)
        double u[2];
        xl_reduce<6>(acc, u);
        return xl_any_negative<6>(u);
    } else {
        wave_allreduce<6>(acc, nl);
        return acc[0] < 0 || acc[1] < 0 || acc[2] < 0 || acc[3] < 0 || acc[4] < 0 || acc[5] < 0;
    }
}

// The same merge when BOTH subtrees are single leaves with momenta pa (suspended) and pb
// (running): every (p₋, p₊, ρ) collapses to the leaf's p, the three sums of NUTS.jl:134-136 are
// all pa + pb (IEEE addition commutes, so the build direction does not matter) and the six
// dots are two distinct values, (M⁻¹pa)·ρ and (M⁻¹pb)·ρ, each appearing three times: the
// result is bit-identical to merge_core.  Half of all merges of a tree are of this kind.
template <int NPL, bool XL = false, class PA, class MK>
__device__ __forceinline__ bool merge_leaf_leaf(PA pa_, MK mk_,
                                                double (&cf)[NPL], double (&cr)[NPL], const double (&pb)[NPL], int nl = 64) {
    LaneAcc<2, NPL> A;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const double pa = pa_(k);
        const double mk = mk_(k);
        const double r = pa + pb[k];
        A.add(0, k, mk * pa, r);
        A.add(1, k, mk * pb[k], r);
        cf[k] = pa;
        cr[k] = r;
        DHMC_BLOCK_FENCE(k);
    }
    double acc[2];
    A.fold_all(acc);
    if constexpr (XL) {
        double u[1];
        xl_reduce<2>(acc, u);
        return xl_any_negative<2>(u);
    } else {
        wave_allreduce<2>(acc, nl);
        return acc[0] < 0 || acc[1] < 0;
    }
}

// The two logaddexp's of a merge (visited statistic and ω).  Wave-uniform arguments: each is ≈ 20 vector instructions with its
// softplus cell fetched by one scalar load (detmath_dev.hpp); ABI v1's pair — exp, log and two divisions evaluated in even / odd
// lanes — was 177.
__device__ __forceinline__ void logaddexp_pair(double a1, double b1, double a2, double b2, int,
                                               double& r1, double& r2) {
    double v1, v2;
    det_logaddexp_pair_u(a1, b1, a2, b2, v1, v2);
    r1 = uni_f64(v1);
    r2 = uni_f64(v2);
}

// p = W .* randn (hamiltonian.jl:124) from the chain's stream.  The Box–Muller pairs are taken in batches of four: Philox and the
// argument reductions of the whole batch first, then ALL its table rows (log cell, sin/cos cell: per-lane gathers) and W
// slots requested together, then the arithmetic.  A lone wave waits out every memory round trip it takes one at a time, and
// these rows come from L2 as often as not (the chain's workspace traffic streams through the CU's 32 KB L1): pair by pair the
// refresh of a 1024-coordinate row cost 16 exposed round trips (round 4: +500 clocks per leapfrog against ABI v1's table-free
// polynomials, more than the shorter arithmetic saved), in batches it costs four.
template <int NPL>
__device__ __forceinline__ void sample_momentum(const ChainKey& key, uint32_t purpose, uint32_t transition,
                                                const double* __restrict__ Wrow, int lane, double (&p)[NPL]) {
    constexpr int NP = (NPL + 1) / 2;
    constexpr int B = NP < 4 ? NP : 4;
#pragma unroll
    for (int b0 = 0; b0 < NP; b0 += B) {
        dm_randn2_reduced R[B];
        double lrow[B][3], srow[B][2], w0[B], w1[B];
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const int kk = b0 + i;
            uint64_t r1, r2;
            stream_raw64(key, (uint32_t)(lane + WAVE * kk), purpose, transition, r1, r2);
            dm_randn2_reduce(r1, r2, &R[i]);
        }
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const int kk = b0 + i;
            const double* __restrict__ lp = DM_LOG_TBL[R[i].jl];
            const double* __restrict__ sp = DM_SINCOS_TBL[R[i].js];
            lrow[i][0] = lp[0]; lrow[i][1] = lp[1]; lrow[i][2] = lp[2];
            srow[i][0] = sp[0]; srow[i][1] = sp[1];
            w0[i] = Wrow[lane + WAVE * (2 * kk)];
            w1[i] = (2 * kk + 1 < NPL) ? Wrow[lane + WAVE * (2 * kk + 1)] : 0.0;
        }
        asm volatile("" ::: "memory");           // every request of the batch is in flight before the first answer is used
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const int kk = b0 + i;
            double z0, z1;
            dm_randn2_finish<dm_v>(R[i], lrow[i][0], lrow[i][1], lrow[i][2], srow[i][0], srow[i][1], &z0, &z1);
            p[2 * kk] = w0[i] * z0;
            if (2 * kk + 1 < NPL) p[2 * kk + 1] = w1[i] * z1;
        }
    }
}

// Phase timing for tools/experiments/phase_timing.py (compiled in with -DDHMC_PHASE_TIMING only): every wave
// accumulates the s_memtime clocks it spends in each region of the per-draw loop; lane 0 adds them to g_phase.
// (Indicative only: the clock reads serialise the wave's outstanding LDS/scalar-memory operations.)
#ifdef DHMC_PHASE_TIMING
__device__ unsigned long long g_phase[16];
#define PH_DECL unsigned long long ph_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ph_t0 = __builtin_readcyclecounter(); int ph_cur = 0;
#define PH(i) { unsigned long long t_ = __builtin_readcyclecounter(); ph_acc[ph_cur] += t_ - ph_t0; ph_t0 = t_; ph_cur = (i); }
#define PH_FLUSH { PH(0); if (lane == 0) { for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&g_phase[i_], ph_acc[i_]); atomicAdd(&g_phase[15], 1ull); } }
#elif defined(DHMC_PHASE_MARK)   // tools/isa_regions.py: the region boundaries as comments in the assembly (static instruction counts per region)
#define PH_DECL
#define PH(i) asm volatile("; DHMC_PH " #i);
#define PH_FLUSH asm volatile("; DHMC_PH 9");
#else
#define PH_DECL
#define PH(i)
#define PH_FLUSH
#endif

// ------------------------------------------------------------------------------------------
// The per-draw loop kernel.  L1LDS: keep the level-1 suspended summary in LDS as well (needs
// 4 Dpad-rows of LDS per wave, i.e. one wave per SIMD at Dpad = 1024).
// ------------------------------------------------------------------------------------------
template <class T, int NPL, bool L1LDS>
__global__ __launch_bounds__(64, (NPL >= 8 && L1LDS) ? 1 : 2) void nuts_run_kernel(RunParams P) {
    const int chain = P.launch_order ? P.launch_order[blockIdx.x] : (int)blockIdx.x;
    // XL (wave.hpp, "the same tree on a permuted wave"): hardware lane `plane` holds the coordinates of LOGICAL lane `lane`.
    // Rule: whatever the ABI lays out — the chain's state, outputs, window moments, the random stream's coordinate counters, the
    // target's parameters — is indexed by `lane`; the kernel's own rows (LDS, workspace), its lane arrays and its Exp(1) buffer
    // by `plane`.
    constexpr bool XL = kXlLanes && T::kElementwise && NPL >= 2;
    const int plane = threadIdx.x;
    const int lane = XL ? xl_logical_lane(plane) : plane;
    const int lane_ = lane, plane_ = plane;
    const int D = P.D, Dpad = P.Dpad;

    extern __shared__ double lds[];
    constexpr bool TPL = traj_in_lds(NPL, L1LDS);                           // M⁻¹ in registers; trajectory edges p₋, p₊ in LDS
    double* m_lds = lds;                                   // [Dpad]   (!TPL)
    double* l0_lds = lds + (TPL ? 0 : 1) * Dpad;           // [Dpad]   level-0 suspended momentum
    double* l1f_lds = lds + (TPL ? 1 : 2) * Dpad;          // [Dpad]   level-1 first   (L1LDS only)
    double* l1l_lds = lds + (TPL ? 2 : 3) * Dpad;          // [Dpad]   level-1 last    (L1LDS only)
    double* tpm_lds = lds + 3 * Dpad;                      // [Dpad]   trajectory p₋   (TPL only)
    double* tpp_lds = lds + 4 * Dpad;                      // [Dpad]   trajectory p₊   (TPL only)
    constexpr int NXL = L1LDS ? lds_extra_levels(NPL) : 0; // levels 2 .. 1+NXL in LDS too (short chains)
    double* xl_lds = lds + 4 * Dpad;                       // [NXL][3][Dpad]  first, last, ρ
    LaneArrF64 lv_omega, lv_vlsa;                          // per suspended level (lane = level): ω, visited statistic
    LaneArrI64 lv_vsteps;
    LaneArrI32 lv_zeta;                                    //   … and the proposal slot
    LaneArrF64 sl_lq, sl_pi;                               // per proposal slot (lane = slot): ℓq and π of the point stored there

    const T tgt(P.tp);
    const size_t row = (size_t)chain * Dpad;
    double* const ws = P.st.ws + (size_t)chain * P.nvec * Dpad;
    auto wsv = [&](int idx) -> double* { return ws + (size_t)idx * Dpad; };

    double mreg[TPL ? NPL : 1];
    if constexpr (TPL) {
        ldv<NPL>(P.st.minv + row, lane, mreg);
    } else {
        const double* mrow = P.st.minv + row;
#pragma unroll
        for (int k = 0; k < NPL; ++k) m_lds[plane + WAVE * k] = mrow[lane + WAVE * k];
    }
    auto mk = [&](int k) -> double {                       // slot k of M⁻¹
        if constexpr (TPL) return mreg[k];
        else return m_lds[plane + WAVE * k];
    };
    const double* Wrow = P.st.W + row;
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    const int max_depth = P.max_depth;
    const int nslots = ws_nslots(max_depth);
    const int nl = uni_i32(reduce_lanes(NPL, D));          // lanes that can hold nonzero partial sums (wave.hpp wave_allreduce)

    double q[NPL], p[NPL], g[NPL], cf[NPL], cr[NPL];
    // TWS (the two-waves-per-SIMD layout of a wide chain, L1LDS = false at 16 slots per lane): the trajectory's turn statistic
    // (p₋, p₊, ρ — touched once per doubling) lives in workspace rows like the level >= 1 summaries, M⁻¹ and the level-0 leaf in
    // 16 KB of LDS, and only the phase point and the running summary (q, p, cf, cr: 128 VGPRs) in registers — half the
    // registers and less than half the LDS of the one-wave-per-SIMD layout, so that a second chain shares the SIMD and each
    // chain's memory and dependency latencies run under the other's instructions.
    constexpr bool TWS = NPL == 16 && !L1LDS;
    double* const tpm_ws = ws + (size_t)ws_edge(0, 1) * Dpad;
    double* const tpp_ws = ws + (size_t)ws_edge(1, 1) * Dpad;
    double* const trho_ws = ws + (size_t)ws_rho_top() * Dpad;
    double tpm[(TPL || TWS) ? 1 : NPL], tpp[(TPL || TWS) ? 1 : NPL], trho[TWS ? 1 : NPL];     // turn statistic of the whole trajectory: p₋, p₊, ρ
    auto a_tm = [&](int k) -> double {
        if constexpr (TPL) return tpm_lds[plane + WAVE * k];
        else if constexpr (TWS) return tpm_ws[plane + WAVE * k];
        else return tpm[k];
    };
    auto a_tp = [&](int k) -> double {
        if constexpr (TPL) return tpp_lds[plane + WAVE * k];
        else if constexpr (TWS) return tpp_ws[plane + WAVE * k];
        else return tpp[k];
    };
    ldv<NPL>(P.st.q + row, lane, q);
    ldv<NPL>(P.st.g + row, lane, g);
    double lq_cur = P.st.lq[chain];
    double eps_fixed = P.st.eps[chain];
    DAState da = P.st.da[chain];
    uint32_t status = P.st.status[chain];
    const uint32_t tr0 = P.st.transition[chain];
    unsigned long long total_steps = 0;
    constexpr int64_t n_done = 0;
    const int64_t NN = P.N;

    if (P.adapt && P.da_init && n_done == 0) {  // initial_adaptation_state (stepsize.jl:134-138; mcmc.jl:266)
        double le = det_log_u(eps_fixed);
        da.mu = det_log_u(10.0) + le;
        da.m = 1;
        da.Hbar = 0.0;
        da.logeps = le;
        da.logeps_bar = 0.0;
    }

    // the chain's current position occupies proposal slot `init_slot` of the workspace
    int init_slot = 0;
    stv<NPL>(wsv(ws_slot(max_depth, init_slot, 0)), plane, q);
    if constexpr (!T::kRecomputeGrad) stv<NPL>(wsv(ws_slot(max_depth, init_slot, 1)), plane, g);

    // write a leaf held in registers into a fresh proposal slot
    uint64_t free_mask = 0;
    auto save_leaf = [&](double lq_leaf, double pi_leaf) -> int {
        int s = __builtin_ctzll(free_mask);
        free_mask &= ~(1ull << s);
        stv<NPL>(wsv(ws_slot(max_depth, s, 0)), plane, q);
        if constexpr (!T::kRecomputeGrad) stv<NPL>(wsv(ws_slot(max_depth, s, 1)), plane, g);
        sl_lq.set(s, lq_leaf, plane);
        sl_pi.set(s, pi_leaf, plane);
        return s;
    };

    // draw n of this call := the chain's position after transition n (mcmc.jl:275,376)
    auto store_draw = [&](int64_t n_) {
        int lane = lane_;                                   // (once per transition: not worth loop-invariant address registers)
        asm volatile("" : "+v"(lane));
        if (P.out.draws) {
            double* drow = P.out.draws + ((size_t)chain * (P.out_stride ? P.out_stride : P.N) + n_) * D;
#pragma unroll
            for (int k = 0; k < NPL; ++k)
                if (lane + WAVE * k < D) drow[lane + WAVE * k] = q[k];
        }
        window_accumulate<NPL>(P, (size_t)chain * P.Dpad, lane, q, n_);
    };

    PH_DECL
    for (int64_t n = 0; n < NN; ++n) {
        PH(1)   // momentum refresh + transition setup
        const uint32_t tr = tr0 + (uint32_t)n;
        const double eps = uni_f64(P.adapt ? det_exp_u(da.logeps) : eps_fixed);  // current_ϵ (stepsize.jl:163)

        // ---- sample_tree (NUTS.jl:232-241): p, directions, π₀ --------------------------
        // (once per transition: an opaque copy of the lane index keeps the refresh's loop invariants — W's per-slot addresses, the
        // first Philox products of every counter — from being hoisted out of the transition loop into registers that the tree loop
        // then pays for with spills: round 6, 60 accumulation registers and every scratch reload of the merges)
        int lane_cold = lane;
        asm volatile("" : "+v"(lane_cold));
        sample_momentum<NPL>(key, PURPOSE_MOMENTUM, tr, Wrow, lane_cold, p);
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
            for (int k = 0; k < NPL; ++k) kacc.add(0, k, p[k], mk(k) * p[k]);
            pi0 = uni_f64(joint_logdensity(lq_cur, wave_total1<XL>(kacc.fold(0), nl) / 2.0));
        }
        // leaf τ of z₀ (NUTS.jl:120-123)
        if constexpr (TPL) {
            stv<NPL>(tpm_lds, plane, p);
            stv<NPL>(tpp_lds, plane, p);
        } else if constexpr (TWS) {
            stv<NPL>(tpm_ws, plane, p);
            stv<NPL>(tpp_ws, plane, p);
            stv<NPL>(trho_ws, plane, p);
        } else {
#pragma unroll
            for (int k = 0; k < NPL; ++k) { tpm[k] = p[k]; tpp[k] = p[k]; }
        }
        if constexpr (!TWS) {
#pragma unroll
            for (int k = 0; k < NPL; ++k) trho[k] = p[k];
        }
        sl_lq.set(init_slot, lq_cur, plane);
        sl_pi.set(init_slot, pi0, plane);

        // Exp(1) draws of this transition, 64 at a time: lane l holds draw (rexp_base + l)
        uint32_t nrand = 0, rexp_base = 0;
        double rexp_vals;
        auto rexp_fill = [&](uint32_t base) {
            uint64_t r1, r2;
            stream_raw64(key, base + (uint32_t)plane, PURPOSE_TREE, tr, r1, r2);
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

        // ---- sample_trajectory (trees.jl:283-319) ---------------------------------------
        // Edges z₋ (0), z₊ (1).  The registers (q,p,g) hold the edge being extended; the other
        // edge's q (and ∇ℓ unless it is recomputed) is parked in the workspace only when the
        // build direction switches; an edge that was never extended is still the initial point.
        bool stored0 = false, stored1 = false;
        int reg_edge = 2;  // which edge the registers hold: 0, 1 or 2 = both
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
            const int dir = fwd ? 1 : 0;
            PH(2)   // edge switch
            if (reg_edge != 2 && reg_edge != dir) {
                int plane = plane_, lane = lane_;           // (once per doubling at most)
                asm volatile("" : "+v"(plane));
                asm volatile("" : "+v"(lane));
                // park the edge we leave ...
                stv<NPL>(wsv(ws_edge(reg_edge, 0)), plane, q);
                if constexpr (!T::kRecomputeGrad) stv<NPL>(wsv(ws_edge(reg_edge, 2)), plane, g);
                if (reg_edge == 1) stored1 = true; else stored0 = true;
                // ... and fetch the one we extend now
                const bool have = fwd ? stored1 : stored0;
                const int qsrc = have ? ws_edge(dir, 0) : ws_slot(max_depth, init_slot, 0);
                const int gsrc = have ? ws_edge(dir, 2) : ws_slot(max_depth, init_slot, 1);
                ldv<NPL>(wsv(qsrc), plane, q);
                if constexpr (T::kPointwiseGrad) {}
                else if constexpr (T::kRecomputeGrad) (void)tgt.eval(q, g, lane, D);
                else ldv<NPL>(wsv(gsrc), plane, g);
                if (fwd) {
#pragma unroll
                    for (int k = 0; k < NPL; ++k) p[k] = a_tp(k);
                } else {
#pragma unroll
                    for (int k = 0; k < NPL; ++k) p[k] = a_tm(k);
                }
                // The edge's rows are waited for HERE, not at their first use: the first use is the top of the leaf loop, whose other
                // predecessor (its own back edge) has nothing outstanding — the compiler's counted waits (one per slot: 24 s_waitcnt
                // at 16 slots per lane) then sat in every leapfrog for the sake of one edge switch per doubling.
                __builtin_amdgcn_s_waitcnt(0x0070);         // vmcnt(0) lgkmcnt(0)
            }
            reg_edge = dir;
            int64_t i = fwd ? i_plus : i_minus;
            const int64_t di = fwd ? 1 : -1;
            const double eps_s = fwd ? eps : -eps;
            const uint32_t nleaf = 1u << depth;

            // ---- adjacent_tree(rng, trajectory, z, i, depth, is_forward), iteratively ----
            bool invalid = false;
            double v_lsa = 0.0;       // visited statistic of the running subtree
            int64_t v_steps = 0;
            for (uint32_t j = 0; j < nleaf && !invalid && !finished; ++j) {
                double lq_leaf, pi_leaf;
                bool pos_finite;
                PH(3)   // leaf
                bool turning0;
                leapfrog_leaf_m<T, NPL, XL>(tgt, mk, lane, D, q, p, g, eps_s, lq_leaf, pi_leaf, pos_finite, nl, kFuseLeafMerge0 && (j & 1u) != 0,
                                            [&](int k) { return l0_lds[plane + WAVE * k]; }, cf, cr, turning0);
                PH(4)   // leaf scalars
                if (!pos_finite) status |= DHMC_ST_NONFINITE_POSITION;
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
                        // element accessors of the running summary (build order: cf, p, cr)
                        auto a_cf = [&](int k) { return cf[k]; };
                        auto a_p = [&](int k) { return p[k]; };
                        auto a_cr = [&](int k) { return cr[k]; };
                        bool turning;
                        PH(5)   // merge, vector part
                        if (sub) {
                            if (level == 0) {
                                if constexpr (kFuseLeafMerge0) turning = turning0;      // taken with the leaf's own reduction (leapfrog_leaf_m)
                                else turning = merge_leaf_leaf<NPL, XL>([&](int k) { return l0_lds[plane + WAVE * k]; }, mk, cf, cr, p, nl);
                            } else if (L1LDS && level == 1) {
                                auto a_lf = [&](int k) { return l1f_lds[plane + WAVE * k]; };
                                auto a_ll = [&](int k) { return l1l_lds[plane + WAVE * k]; };
                                auto a_lr = [&](int k) { return l1f_lds[plane + WAVE * k] + l1l_lds[plane + WAVE * k]; };
                                turning = fwd ? merge_core<NPL, XL>(a_lf, a_ll, a_lr, a_cf, a_p, a_cr, a_lf, mk, cf, cr, nl)
                                              : merge_core<NPL, XL>(a_p, a_cf, a_cr, a_ll, a_lf, a_lr, a_lf, mk, cf, cr, nl);
                            } else if (NXL > 0 && level < 2 + NXL) {
                                const double* Lf = xl_lds + (size_t)(3 * (level - 2)) * Dpad;
                                const double* Ll = Lf + Dpad;
                                const double* Lr = Ll + Dpad;
                                auto a_lf = [&](int k) { return Lf[plane + WAVE * k]; };
                                auto a_ll = [&](int k) { return Ll[plane + WAVE * k]; };
                                auto a_lr = [&](int k) { return Lr[plane + WAVE * k]; };
                                turning = fwd ? merge_core<NPL, XL>(a_lf, a_ll, a_lr, a_cf, a_p, a_cr, a_lf, mk, cf, cr, nl)
                                              : merge_core<NPL, XL>(a_p, a_cf, a_cr, a_ll, a_lf, a_lr, a_lf, mk, cf, cr, nl);
                            } else {
                                const double* Lf = wsv(ws_stack(level, 0));
                                const double* Ll = wsv(ws_stack(level, 1));
                                const double* Lr = wsv(ws_stack(level, 2));
                                auto a_lf = [&](int k) { return Lf[plane + WAVE * k]; };
                                auto a_ll = [&](int k) { return Ll[plane + WAVE * k]; };
                                auto a_lr = [&](int k) { return Lr[plane + WAVE * k]; };
                                turning = fwd ? merge_core<NPL, XL>(a_lf, a_ll, a_lr, a_cf, a_p, a_cr, a_lf, mk, cf, cr, nl)
                                              : merge_core<NPL, XL>(a_p, a_cf, a_cr, a_ll, a_lf, a_lr, a_lf, mk, cf, cr, nl);
                            }
                            // v = v₋ ⊕ v₊ (trees.jl:249) and ω = logaddexp(ω₋, ω₊) (trees.jl:145), one pass
                            PH(6)   // merge, scalar part
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
                            // top level (trees.jl:294-316): merge with τ of the whole trajectory, which is
                            // time-ordered (tpm, tpp, trho) whatever the direction
                            auto a_tr = [&](int k) -> double {
                                if constexpr (TWS) return trho_ws[plane + WAVE * k];
                                else return trho[k];
                            };
                            if (depth == 0) {
                                turning = merge_leaf_leaf<NPL, XL>(a_tr, mk, cf, cr, p, nl);
                            } else {
                                turning = fwd ? merge_core<NPL, XL>(a_tm, a_tp, a_tr, a_cf, a_p, a_cr, a_cf, mk, cf, cr, nl)
                                              : merge_core<NPL, XL>(a_p, a_cf, a_cr, a_tm, a_tp, a_tr, a_cf, mk, cf, cr, nl);
                            }
                            double w;
                            PH(6)
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
                                // τ of the doubled trajectory: the new edge momentum and Σp
                                if constexpr (TPL) {
                                    stv<NPL>(fwd ? tpp_lds : tpm_lds, plane, p);
                                } else if constexpr (TWS) {
                                    stv<NPL>(fwd ? tpp_ws : tpm_ws, plane, p);
                                    stv<NPL>(trho_ws, plane, cr);
                                } else if (fwd) {
#pragma unroll
                                    for (int k = 0; k < NPL; ++k) tpp[k] = p[k];
                                } else {
#pragma unroll
                                    for (int k = 0; k < NPL; ++k) tpm[k] = p[k];
                                }
                                if constexpr (!TWS) {
#pragma unroll
                                    for (int k = 0; k < NPL; ++k) trho[k] = cr[k];
                                }
                            }
                            level = -1;  // handled
                            break;
                        }
                    }
                    PH(7)   // suspend
                    if (level >= 0 && !invalid) {
                        // suspend the running subtree at `level` until its right sibling is built
                        if (c_zeta < 0) c_zeta = save_leaf(lq_leaf, pi_leaf);
                        if (level == 0) {
                            stv<NPL>(l0_lds, plane, p);
                        } else if (L1LDS && level == 1) {
                            stv<NPL>(l1f_lds, plane, cf);
                            stv<NPL>(l1l_lds, plane, p);
                        } else if (NXL > 0 && level < 2 + NXL) {
                            double* Lf = xl_lds + (size_t)(3 * (level - 2)) * Dpad;
                            stv<NPL>(Lf, plane, cf);
                            stv<NPL>(Lf + Dpad, plane, p);
                            stv<NPL>(Lf + 2 * Dpad, plane, cr);
                        } else {
                            stv<NPL>(wsv(ws_stack(level, 0)), plane, cf);
                            stv<NPL>(wsv(ws_stack(level, 1)), plane, p);
                            stv<NPL>(wsv(ws_stack(level, 2)), plane, cr);
                        }
                        lv_omega.set(level, c_omega, plane);
                        lv_vlsa.set(level, v_lsa, plane);
                        lv_vsteps.set(level, v_steps, plane);
                        lv_zeta.set(level, c_zeta, plane);
                    }
                }
                if (invalid) {
                    // unwind the recursion: every suspended left sibling contributes its visited
                    // statistic (trees.jl:244,249-250)
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

        // ---- TreeStatisticsNUTS and the new position (NUTS.jl:238-240) -----------------
        PH(8)   // end of transition
        const double acc_rate = [&]() {
            double a = det_exp_u(vtop_lsa) / (double)vtop_steps;           // NUTS.jl:87
            return uni_f64(a < 1.0 ? a : 1.0);
        }();
        total_steps += (unsigned long long)vtop_steps;
        init_slot = zeta_top;
        {
            int plane = plane_, lane = lane_;               // (once per transition)
            asm volatile("" : "+v"(plane));
            asm volatile("" : "+v"(lane));
            ldv<NPL>(wsv(ws_slot(max_depth, init_slot, 0)), plane, q);
            if constexpr (T::kPointwiseGrad) {}
            else if constexpr (T::kRecomputeGrad) (void)tgt.eval(q, g, lane, D);
            else ldv<NPL>(wsv(ws_slot(max_depth, init_slot, 1)), plane, g);
        }
        lq_cur = sl_lq.get(init_slot);
        const double pi_stat = sl_pi.get(init_slot);

        const size_t o = (size_t)chain * (P.out_stride ? P.out_stride : P.N) + (size_t)(n_done + n);
        // (tried in round 4: the draw stored after the NEXT transition's momentum refresh, so that the proposal slot's round trip
        // runs under it — 3.10e8 against 3.16e8 leapfrog-steps/s on one box, the loads kept in flight across the loop's back edge
        // cost more in waits at the loop head than the round trip they hide)
        store_draw(n_done + n);
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

    // ---- write the chain back (WarmupState + adaptation state) --------------------------
    PH_FLUSH
    stv<NPL>(P.st.q + row, lane, q);
    if constexpr (T::kPointwiseGrad) (void)tgt.eval(q, g, lane, D);
    stv<NPL>(P.st.g + row, lane, g);
    if (lane == 0) {
        P.st.lq[chain] = lq_cur;
        if (P.adapt) {
            P.st.da[chain] = da;
            if (P.da_finalize) P.st.eps[chain] = det_exp_u(da.logeps_bar); // final_ϵ (stepsize.jl:170; mcmc.jl:285)
        }
        P.st.transition[chain] = tr0 + (uint32_t)NN;
        P.st.status[chain] = status;
        if (P.leapfrog_counter) atomicAdd(P.leapfrog_counter, total_steps);
        if (P.chain_work) P.chain_work[chain] = (unsigned)(total_steps > 0xffffffffull ? 0xffffffffull : total_steps);
    }
}

// ------------------------------------------------------------------------------------------
// initialize_warmup_state (mcmc.jl:129-132): position (given or random_position, mcmc.jl:108),
// strict evaluate_ℓ (hamiltonian.jl:202-217).
// ------------------------------------------------------------------------------------------
struct InitParams {
    int D, Dpad, C, chain_offset;
    uint64_t seed;
    const double* q0;  // [C][D] unpadded device pointer, or null for random
    ChainArrays st;
    TargetParams tp;
};

template <class T, int NPL>
__global__ __launch_bounds__(64) void init_kernel(InitParams P) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const int D = P.D, Dpad = P.Dpad;
    const T tgt(P.tp);
    const size_t row = (size_t)chain * Dpad;
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    double q[NPL], g[NPL];
    if (P.q0) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            int e = lane + WAVE * k;
            q[k] = e < D ? P.q0[(size_t)chain * D + e] : 0.0;
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < (NPL + 1) / 2; ++kk) {
            uint64_t r1, r2;
            stream_raw64(key, (uint32_t)(lane + WAVE * kk), PURPOSE_INIT_POSITION, 0u, r1, r2);
            int e0 = lane + WAVE * (2 * kk), e1 = e0 + WAVE;
            q[2 * kk] = e0 < D ? u01_closed_open(r1) * 4 - 2 : 0.0;
            if (2 * kk + 1 < NPL) q[2 * kk + 1] = e1 < D ? u01_closed_open(r2) * 4 - 2 : 0.0;
        }
    }
    const bool pfin = all_finite<T, NPL>(q);
    const double lres = tgt.eval(q, g, lane, D);
    const bool gfin = T::kFiniteLqImpliesFiniteGrad ? true : all_finite<T, NPL>(g);
    double lq = T::kDeferred ? tgt.finish(wave_allreduce1(lres)) : lres;
    uint32_t status = 0;
    if (!pfin) {
        status |= DHMC_ST_NONFINITE_POSITION;
        lq = -dm_inf();
#pragma unroll
        for (int k = 0; k < NPL; ++k) g[k] = 0.0;
    } else {
        bool ok = (dm_isfinite(lq) && gfin) || lq == -dm_inf();
        if (!ok) {  // strict: the reference throws (hamiltonian.jl:212-216)
            status |= DHMC_ST_INVALID_INITIAL;
            lq = -dm_inf();
        }
    }
    stv<NPL>(P.st.q + row, lane, q);
    stv<NPL>(P.st.g + row, lane, g);
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        int e = lane + WAVE * k;
        P.st.minv[row + e] = 1.0;                 // GaussianKineticEnergy(N) (hamiltonian.jl:87)
        P.st.W[row + e] = e < D ? 1.0 : 0.0;
    }
    if (lane == 0) {
        P.st.lq[chain] = lq;
        P.st.eps[chain] = dm_nan();               // ϵ = nothing (mcmc.jl:130)
        P.st.transition[chain] = 0;
        P.st.status[chain] = status;
    }
}

// ------------------------------------------------------------------------------------------
// warmup(::InitialStepsizeSearch) (mcmc.jl:134-148): z = (Q, rand_p), then
// find_initial_stepsize (stepsize.jl:46-60) on A(ϵ) = logdensity(leapfrog(z, ϵ)) - logdensity(z)
// (stepsize.jl:75-85).
// ------------------------------------------------------------------------------------------
struct SearchParams {
    int D, Dpad, C, chain_offset;
    uint64_t seed;
    double initial_eps, log_threshold;
    int maxiter;
    ChainArrays st;
    TargetParams tp;
};

template <class T, int NPL>
__global__ __launch_bounds__(64) void stepsize_search_kernel(SearchParams P) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const int D = P.D, Dpad = P.Dpad;
    extern __shared__ double lds[];
    double* m_lds = lds;
    const T tgt(P.tp);
    const size_t row = (size_t)chain * Dpad;
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
#pragma unroll
    for (int k = 0; k < NPL; ++k) m_lds[lane + WAVE * k] = P.st.minv[row + lane + WAVE * k];
    double q0[NPL], p0[NPL], g0[NPL], q[NPL], p[NPL], g[NPL];
    ldv<NPL>(P.st.q + row, lane, q0);
    ldv<NPL>(P.st.g + row, lane, g0);
    const double lq0 = P.st.lq[chain];
    uint32_t status = P.st.status[chain];
    sample_momentum<NPL>(key, PURPOSE_SEARCH_MOMENTUM, P.st.transition[chain], P.st.W + row, lane, p0);
    LaneAcc<1, NPL> kacc;
#pragma unroll
    for (int k = 0; k < NPL; ++k) kacc.add(0, k, p0[k], m_lds[lane + WAVE * k] * p0[k]);
    const double l0 = uni_f64(joint_logdensity(lq0, wave_allreduce1(kacc.fold(0)) / 2.0));
    if (!dm_isfinite(l0)) {  // stepsize.jl:77-79
        if (lane == 0) P.st.status[chain] = status | DHMC_ST_NONFINITE_START_DENSITY;
        return;
    }
    auto A = [&](double eps) -> double {
#pragma unroll
        for (int k = 0; k < NPL; ++k) { q[k] = q0[k]; p[k] = p0[k]; g[k] = g0[k]; }
        double lq1, pi1;
        bool pfin;
        leapfrog_leaf<T, NPL>(tgt, m_lds, lane, D, q, p, g, eps, lq1, pi1, pfin);
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
    if (!found) status |= DHMC_ST_STEPSIZE_SEARCH_FAILED;  // stepsize.jl:57-59
    if (lane == 0) {
        P.st.eps[chain] = eps;
        P.st.status[chain] = status;
    }
}

// ------------------------------------------------------------------------------------------
// ℓ and ∇ℓ of every chain's row through a functor whose chains are too wide for the register-resident kernels
// (1024 < D <= 4096): the evaluation the streaming round engine asks for between its kernels where an external model's callback
// stands (external_rounds.hpp; dhmc_capi.hip external_eval) — a caller's functor beyond 1024 coordinates (hamiltonian.jl:146-147,204
// put no limit on the dimension).  One wave per chain, the functor's own arithmetic; evaluate_ℓ's rules are applied by the engine.
// ------------------------------------------------------------------------------------------
template <class T, int NPL>
__global__ __launch_bounds__(64) void functor_eval_kernel(TargetParams tp, int D, int ld, const double* __restrict__ q,
                                                          double* __restrict__ lq, double* __restrict__ grad) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const T tgt(tp);
    double qv[NPL], gv[NPL];
    ldv<NPL>(q + (size_t)chain * ld, lane, qv);
    const double lres = tgt.eval(qv, gv, lane, D);
    double l = lres;
    if constexpr (T::kDeferred) l = tgt.finish(wave_allreduce1(lres));
    stv<NPL>(grad + (size_t)chain * ld, lane, gv);
    if (lane == 0) lq[chain] = l;
}

// ------------------------------------------------------------------------------------------
// End-of-stage metric: κ = GaussianKineticEnergy(Diagonal(var(posterior_matrix; dims=2)))
// (mcmc.jl:209,281-284; hamiltonian.jl:80) per chain from its own draws [C][N][D].
// Statistics.var with dims reduces sequentially over draws: mean = (Σx)/n, Σ(x-mean)²/(n-1).
// ------------------------------------------------------------------------------------------
template <int NPL>
__global__ __launch_bounds__(64) void metric_diag_kernel(int D, int Dpad, int64_t N, const double* __restrict__ draws,
                                                         double* __restrict__ minv, double* __restrict__ W) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const double* base = draws + (size_t)chain * N * D;
    constexpr int KC = NPL < 16 ? NPL : 16;          // slots per pass (wide chains: several passes over the draws)
    for (int k0 = 0; k0 < NPL; k0 += KC) {
        double s[KC], ss[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) { s[k] = 0.0; ss[k] = 0.0; }
        for (int64_t i = 0; i < N; ++i)
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                int e = lane + WAVE * (k0 + k);
                if (e < D) s[k] = s[k] + base[(size_t)i * D + e];
            }
#pragma unroll
        for (int k = 0; k < KC; ++k) s[k] = s[k] / (double)N;
        for (int64_t i = 0; i < N; ++i)
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                int e = lane + WAVE * (k0 + k);
                if (e < D) {
                    double d = base[(size_t)i * D + e] - s[k];
                    ss[k] = ss[k] + d * d;
                }
            }
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            int e = lane + WAVE * (k0 + k);
            size_t o = (size_t)chain * Dpad + e;
            if (e < D) {
                double var = ss[k] / (double)(N - 1);
                minv[o] = var;
                W[o] = __builtin_sqrt(1.0 / var);
            } else {
                minv[o] = 1.0;
                W[o] = 0.0;
            }
        }
    }
}

}  // namespace dhmc
