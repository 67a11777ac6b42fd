// Round 6 experiment, NOT part of the library: config 5's fused eta + link kernel with the ROLES split over the workgroup's waves
// (profiles/r06_config5_fused_link.txt, "roles"), here with the clock counters that showed where it waits.  Bit-equal to the shipped kernel
// (tests/test_gpu_configs.py / test_gpu_engines.py logistic cases with the kernel wired in: 9 passed) and slower: 1.56-1.63 ms against
// 1.30 ms.  Per workgroup of 620 000 clocks (32 groups x 4 stages): the product waves stand at barriers for 45 % of them, the
// staging + link waves for 6 %: the link of a group's 512 elements per wave costs ~11 000 clocks on ONE wave per SIMD (phase 3 alone 6 400:
// two divisions per element) plus 3 500 waiting for X^T, against 8 200 for the group's 128 matrix instructions — one link wave per SIMD
// cannot keep up with one product wave, and the product waves' 235 registers leave no room for a second one.
// To try it: paste both pieces into csrc/logistic_rounds.hpp (the phases after logistic_link_batch, the kernel after
// logistic_eta_link_kernel) and launch logistic_eta_link_roles_kernel<Dpad> with 512 threads on the same grid.
#pragma once

// logistic_link_batch's phases as three calls that may stand far apart in the instruction stream (the role-split kernel below runs them
// in successive LDS stages, so that a phase's table gathers return under the next stage's hand-over): the same operations on the same
// operands.  phase1: the reductions of e^{-|η|} and its table requests (or, for the functions' rare branches, everything at once);
// phase2: e^{-|η|}, the reduction of log(1 + t) and its table requests; phase3: σ(η) and log(1 + e^η).
template <int NE>
struct LogisticLinkPhases {
    // carried from phase to phase: η and the exp table's cells (1 -> 2), η, e^{-|η|} and the log table's rows (2 -> 3); the cheap
    // integer reductions are done again where they are needed rather than kept in registers across an LDS stage
    double eta[NE], T[NE], t[NE], row[NE][3], sig[NE], l1pe[NE];
    bool whole;                                   // the wave went through the functions themselves in phase1
    __device__ __forceinline__ void phase1() {
        bool rare = false;
        int j[NE], e[NE];
        double x[NE], r[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            x[i] = -__builtin_fabs(eta[i]);
            dm_exp_reduce(x[i], &j[i], &e[i], &r[i]);
            rare = rare || (e[i] < -1021 && x[i] >= -745.2);
        }
        whole = __ballot(rare) != 0ull;
        if (__builtin_expect(whole, 0)) {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const double tt = det_exp_v(x[i]);
                sig[i] = eta[i] >= 0 ? 1.0 / (1.0 + tt) : tt / (1.0 + tt);
                l1pe[i] = (eta[i] > 0 ? eta[i] : 0.0) + det_log1p_nonneg_t<dm_v>(tt);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) T[i] = DM_EXP2_TBL[j[i]];
    }
    __device__ __forceinline__ void phase2() {
        if (whole) return;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            int j, e, jl, el;
            double r, m;
            const double x = -__builtin_fabs(eta[i]);
            dm_exp_reduce(x, &j, &e, &r);
            const double y = dm_exp_poly<dm_v>(r, T[i]);
            double v = dm_from_bits(dm_bits(y) + ((uint64_t)(int64_t)e << 52));
            v = x < -745.2 ? 0.0 : v;
            t[i] = dm_isnan(x) ? x : v;
            dm_log_reduce(dm_bits(1.0 + t[i]), 0, &jl, &el, &m);
            const double* __restrict__ lp = DM_LOG_TBL[jl];
            row[i][0] = lp[0]; row[i][1] = lp[1]; row[i][2] = lp[2];
        }
    }
    __device__ __forceinline__ void phase3() {
        if (whole) return;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            int jl, el;
            double m;
            const double w = 1.0 + t[i];
            dm_log_reduce(dm_bits(w), 0, &jl, &el, &m);
            const double lg = dm_log_finish<dm_v>(m, el, row[i][0], row[i][1], row[i][2]) + (t[i] - (w - 1.0)) / w;
            const double l1p = (w == 1.0) ? t[i] : (!dm_isfinite(w) ? w : lg);
            sig[i] = (eta[i] >= 0 ? 1.0 : t[i]) / w;
            l1pe[i] = (eta[i] > 0 ? eta[i] : 0.0) + l1p;
        }
    }
};



// The same work with the ROLES split over the workgroup's waves (round 6, second form): waves 0-3 only multiply — a wave per SIMD, the
// chain rows of Q′ in its registers, nothing but LDS reads and matrix instructions in its loop; waves 4-7 — the other wave of each
// SIMD — bring Xᵀ's columns into the double-buffered LDS tile and run the link of the PREVIOUS group's 64 × 32 elements, which the product
// waves leave in LDS (Es, double-buffered): the link's vector instructions issue under the other wave's matrix instructions instead of
// between a wave's own.  One barrier per LDS stage for all eight waves; the elements, their order and every operation are the one-role
// kernel's.  One workgroup per CU (the product waves' 128 fragment registers leave room for two waves per SIMD).
static __device__ int lk_dbg_count = 0;
template <int I, int N, class F>
__device__ __forceinline__ void lk_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        lk_static_for<I + 1, N>(f);
    }
}

template <int DP>
__global__ __launch_bounds__(512, 1) void logistic_eta_link_roles_kernel(RunParams P, LogisticRound L, const double* __restrict__ Q, int ntile_rows) {
    constexpr int KS = DP / 4, NST = DP / LK_TK;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int rt = sq % ntile_rows, grp = (sq / ntile_rows) * 8 + xcd;           // grp = 2 z + lh
    const int z = grp >> 1, lh = grp & 1;
    if (z >= L.nz) return;
    const int count = *L.act_count;
    const int row0 = rt * 64;
    if (row0 >= count) return;
    const int64_t N = P.tp.n, Npad = P.tp.npad;
    const int64_t nb = (int64_t)z * DHMC_LOGISTIC_BLOCK;
    const int64_t ne = nb + DHMC_LOGISTIC_BLOCK < Npad ? nb + DHMC_LOGISTIC_BLOCK : Npad;
    const int nm = (int)((ne - nb) / WAVE);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, w4 = wv & 3;
    const int l0 = LK_TL * lh;

    __shared__ double Bs[2][LK_TK * LK_LS];
    __shared__ double Es[2][4 * 8 * WAVE];                                         // [group parity][wave][j][r][lane]: the accumulators as they are

    unsigned long long t_wait = 0, t_all = __builtin_readcyclecounter();
#define LK_BARRIER() do { const unsigned long long tb_ = __builtin_readcyclecounter(); __syncthreads(); t_wait += __builtin_readcyclecounter() - tb_; } while (0)
    if (wv < 4) {
        // ---- product waves -------------------------------------------------------------------------------------------------------
        int arow = row0 + 16 * w4 + (lane & 15);
        arow = arow < count ? arow : count - 1;
        const double* __restrict__ ap = Q + (size_t)L.act[arow] * P.Dpad + (lane >> 4);
        double a[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = ap[4 * ks];
        const int b_rd = (lane >> 4) * LK_LS + (lane & 15);
        int buf = 0;
#pragma nounroll
        for (int m = 0; m < nm; ++m) {
            mfma_d4 acc[2];
            acc[0] = mfma_d4{0.0, 0.0, 0.0, 0.0};
            acc[1] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                LK_BARRIER();                                                   // stage (m, st) is in Bs[buf]
                const double* bs = Bs[buf];
                // the stage's B-fragments in batches of 8 k-steps, the next batch's LDS reads in flight under this one's products (a lone
                // wave per SIMD: nothing else covers the read latency)
                constexpr int NB = LK_TK / 4 / 8;
                double bb[2][8][2];
#pragma unroll
                for (int i = 0; i < 8; ++i) { bb[0][i][0] = bs[4 * i * LK_LS + b_rd]; bb[0][i][1] = bs[4 * i * LK_LS + b_rd + 16]; }
#pragma unroll
                for (int h = 0; h < NB; ++h) {
                    if (h + 1 < NB) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            bb[(h + 1) & 1][i][0] = bs[4 * (8 * (h + 1) + i) * LK_LS + b_rd];
                            bb[(h + 1) & 1][i][1] = bs[4 * (8 * (h + 1) + i) * LK_LS + b_rd + 16];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(LK_TK / 4) * st + 8 * h + i], bb[h & 1][i][0], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(LK_TK / 4) * st + 8 * h + i], bb[h & 1][i][1], acc[1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                buf ^= 1;
            }
            double* es = Es[m & 1] + (size_t)w4 * 8 * WAVE + lane;                 // read by the link waves after the next barrier
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) es[(4 * j + r) * WAVE] = acc[j][r];
        }
        LK_BARRIER();                                                           // the last group's η is in Es
        if (blockIdx.x == 1000 && t == 0 && atomicAdd(&lk_dbg_count, 1) < 4) printf("[roles] product wave: %llu clocks, %llu at barriers, %d groups x %d stages\n", __builtin_readcyclecounter() - t_all, t_wait, nm, NST);
        return;
    }
    // ---- staging + link waves ------------------------------------------------------------------------------------------------------
    const int tl = t - 256;
    const int b_k = tl >> 2, b_c = 8 * (tl & 3);
    const double* __restrict__ bsrc = P.tp.b + (size_t)b_k * Npad + nb + l0 + b_c;
    // PF stages of Xᵀ are in flight ahead of the one being handed over (one workgroup per CU: nothing else hides the loads)
    constexpr int PF = NST >= 2 ? 2 : 1;
    double bv[PF][8];
    auto bload = [&](int m, auto stc) __attribute__((always_inline)) {             // stage (m, st) into ring slot st % PF
        constexpr int st = decltype(stc)::value;
        const gemm_d2* s2 = reinterpret_cast<const gemm_d2*>(bsrc + (size_t)(LK_TK * st) * Npad + WAVE * m);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const gemm_d2 v = s2[i]; bv[st % PF][2 * i] = v[0]; bv[st % PF][2 * i + 1] = v[1]; }
    };
    mfma_d4 lp[2];
    lp[0] = mfma_d4{0.0, 0.0, 0.0, 0.0};
    lp[1] = mfma_d4{0.0, 0.0, 0.0, 0.0};
    double* hrow[4];
    int srow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int lr = row0 + 16 * w4 + (lane >> 4) + 4 * r;
        const int g = L.act[lr < count ? lr : count - 1];
        hrow[r] = L.H + (size_t)g * Npad + nb + l0 + (lane & 15);
        srow[r] = g - P.chain_base;
    }
    // the link of a group's 8 elements per lane runs one group behind the products, a phase per LDS stage (LogisticLinkPhases)
    LogisticLinkPhases<8> K8;                                                      // the lane's elements 4 j + r
    double yv[2];
    int mt = 0;                                                                    // the group the held η belong to
    auto take = [&](int m1) __attribute__((always_inline)) {                       // group m1's η (and y) into registers
        const double* es = Es[m1 & 1] + (size_t)w4 * 8 * WAVE + lane;
#pragma unroll
        for (int e = 0; e < 8; ++e) K8.eta[e] = es[e * WAVE];
        const int64_t n_lo = nb + (int64_t)WAVE * m1 + l0 + (lane & 15);
        yv[0] = n_lo < N ? P.tp.c[n_lo] : 0.0;
        yv[1] = n_lo + 16 < N ? P.tp.c[n_lo + 16] : 0.0;
        mt = m1;
    };
    auto finish = [&](auto jc) __attribute__((always_inline)) {                    // r stored, the terms added (elements (j, r) of group mt)
        constexpr int j = decltype(jc)::value;
        const int64_t n_lo = nb + (int64_t)WAVE * mt + l0 + (lane & 15);
        const double y = yv[j];
        const bool valid = n_lo + 16 * j < N;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hrow[r][WAVE * mt + 16 * j] = valid ? y - K8.sig[4 * j + r] : 0.0;
            if (valid) lp[j][r] = lp[j][r] + (y * K8.eta[4 * j + r] - K8.l1pe[4 * j + r]);
        }
    };
    auto phase = [&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k == 0) { K8.phase1(); }
        else if constexpr (k == 1) { K8.phase2(); }
        else { K8.phase3(); finish(std::integral_constant<int, 0>{}); finish(std::integral_constant<int, 1>{}); }
    };
    unsigned long long t_ph[3] = {0, 0, 0}, t_stw = 0;
    lk_static_for<0, PF>([&](auto stc) __attribute__((always_inline)) { bload(0, stc); });
    int buf = 0;
#pragma nounroll
    for (int m = 0; m < nm; ++m) {
        lk_static_for<0, NST>([&](auto stc) __attribute__((always_inline)) {
            constexpr int st = decltype(stc)::value;
            const unsigned long long tw_ = __builtin_readcyclecounter();
            gemm_d2* bw = reinterpret_cast<gemm_d2*>(Bs[buf] + b_k * LK_LS + b_c);
#pragma unroll
            for (int i = 0; i < 4; ++i) bw[i] = gemm_d2{bv[st % PF][2 * i], bv[st % PF][2 * i + 1]};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t_stw += __builtin_readcyclecounter() - tw_;
            LK_BARRIER();                                                       // stage (m, st) handed over; group m-1's η is in Es
            if constexpr (st + PF < NST) bload(m, std::integral_constant<int, (st + PF) % NST>{});          // PF stages on, same ring slot
            else if (m + 1 < nm) bload(m + 1, std::integral_constant<int, (st + PF) % NST>{});
            if (m > 0) {                                                           // phase k of group m-1 in stage min(k, NST-1)
                if (st == 0) take(m - 1);
                lk_static_for<0, 3>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr ((k < NST - 1 ? k : NST - 1) == st) { const unsigned long long tp_ = __builtin_readcyclecounter(); phase(kc); t_ph[k] += __builtin_readcyclecounter() - tp_; }
                });
            }
            buf ^= 1;
        });
    }
    LK_BARRIER();
    if (blockIdx.x == 1000 && t == 256 && lk_dbg_count < 8) printf("[roles] staging+link wave: %llu clocks, %llu at barriers, phases %llu %llu %llu, waiting for X^T + LDS write %llu\n", __builtin_readcyclecounter() - t_all, t_wait, t_ph[0], t_ph[1], t_ph[2], t_stw);
    take(nm - 1);
    lk_static_for<0, 3>([&](auto kc) __attribute__((always_inline)) { phase(kc); });
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) L.S1L[((size_t)z * P.C + srow[r]) * WAVE + l0 + 16 * j + (lane & 15)] = lp[j][r];
}

