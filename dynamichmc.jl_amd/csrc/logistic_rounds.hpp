// Round engine for GEMM-shaped gradients: the logistic-regression family (BASELINE config 5's model) with a
// per-chain diagonal metric.  Same rounds as dense_rounds.hpp — every round is one leapfrog for every chain —
// but here the expensive part is the user's ∇log π (reference src/hamiltonian.jl:204,279), not M⁻¹:
//
//      K0<diag>   chains beginning a transition: p = W∘z, p♯ = M⁻¹∘p, π₀, τ₀, first half step
//      K1         q′ = q + ϵ·(M⁻¹∘pₘ)                                            (hamiltonian.jl:278)
//      K_act      the list of the chains that take a leapfrog this round (the rest have finished their transitions)
//      G_eta      H  = Q′ · Xᵀ     [A×Dpad]·[Dpad×Npad]   fp64 MFMA GEMM over the A listed rows  (η_n = x_n·β, one chain over d)
//      K_r        r_n = y_n − σ(η_n),  Σ_n [y_n η_n − log(1+e^{η_n})] per block of observations in wave order;  H ← R
//      G_g        P_z = R · X over the observations of block z   [A×Npad]·[Npad×Dpad], split-K fp64 MFMA GEMM: (Xᵀr)_d is,
//                 per DHMC_LOGISTIC_BLOCK observations, one chain over n ascending (include/dhmc.h)
//      K2         G = ((P₀ + P₁) + P₂) + …,  ∇ℓ = G − q′,  ℓ = S₁ − ½ q′·q′,  p′ = pₘ + ϵ/2 ∇ℓ,  p♯ = M⁻¹∘p′  (:279-280)
//      K3         the leaf and what follows (dense_rounds.hpp: identical code; p♯ is simply M⁻¹∘p here)
//
// The rows of chains that are done are not multiplied: a call ends when its slowest chain does, and with trees of
// depth 5–7 side by side half of all row-rounds would be idle rows (BASELINE config 5: mean 964 leapfrogs per chain and
// call, maximum 1996).  The blocks of the Σ_n are what makes that pay: one chain of 10⁵ fma's per output takes 1.4 ms
// whether 1024 rows are multiplied or 10.
// X is read once per round for ALL chains instead of twice per chain per leapfrog.  Every number is the one
// the wave-per-chain functor (targets.hpp LogisticT) and the oracle compute: the GEMM accumulations are the
// same ascending fma chains, pads contribute fma(0, 0, acc) = acc.
#pragma once
#include <type_traits>
#include "dense_rounds.hpp"

namespace dhmc {

struct LogisticRound {
    double* H;    // [C][Npad]  η, then r
    double* S1P;  // [nz][C]    Σ [y_n η_n − log1pexp(η_n)] over the observations of block z
    double* S1L;  // [nz][C][64]  the same sum's 64 per-lane partial sums before the butterfly (the fused η + link kernel)
    double* S1;   // [C]        ℓ's data term, the blocks added in ascending order (external models: ℓ from the callback)
    double* P;    // [nz][C][Dpad]  R·X over the observations of block z (split-K partial products)
    int nz;       // ceil(Npad / DHMC_LOGISTIC_BLOCK)
    int* act;     // [C]  chains that take a leapfrog this round
    int* act_count;
};

// K_act: act <- the chains in PH_LEAF (any order: a row's result does not depend on its place in a tile)
static __global__ __launch_bounds__(256) void rounds_active_list_kernel(RunParams P, RoundBuffers R, LogisticRound L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.C) return;
    const int chain = P.chain_base + i;
    if (R.ts[chain].phase == PH_LEAF) L.act[atomicAdd(L.act_count, 1)] = chain;
}

// K0 for a diagonal metric: z (in cp) -> p = W∘z, p♯ = M⁻¹∘p in the places K0 reads them from.
template <int NPL>
__global__ __launch_bounds__(64) void rounds_momentum_diag_kernel(RunParams P, RoundBuffers R) {
    if ((int)blockIdx.x >= *R.list_count) return;
    const int chain = R.list[blockIdx.x], lane = threadIdx.x;
    const size_t row = (size_t)chain * P.Dpad;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        const double p = P.st.W[row + e] * R.cp[row + e];       // rand_p, diagonal W (hamiltonian.jl:80,124)
        R.tbuf[row + e] = p;
        R.cps[row + e] = P.st.minv[row + e] * p;
    }
}

// K1: q′ = q + ϵ·(M⁻¹∘pₘ)
template <int NPL>
__global__ __launch_bounds__(64) void rounds_k1_diag_kernel(RunParams P, RoundBuffers R) {
    const int chain = P.chain_base + blockIdx.x, lane = threadIdx.x;
    const TreeState& S = R.ts[chain];
    if (S.phase != PH_LEAF) return;
    const size_t row = (size_t)chain * P.Dpad;
    const double eps_s = S.eps_s;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        const double t = P.st.minv[row + e] * R.cp[row + e];
        P.st.q[row + e] = P.st.q[row + e] + eps_s * t;
    }
}

// (the link of NE observations at once — logistic_link_batch — lives with the scalar math's device policies: detmath_dev.hpp)

// K_r: link, residual and the block's share of the log-likelihood — one wave per (block of observations, listed chain).
// H <- r (0 for the padding observations); S1P[z][chain] <- the block's Σ [y η − log(1+e^η)] in wave order (lane l adds
// its observations n = l mod 64 in ascending order, then the butterfly: include/dhmc.h).
static __global__ __launch_bounds__(64) void logistic_link_kernel(RunParams P, RoundBuffers R, LogisticRound L) {
    if ((int)blockIdx.y >= *L.act_count) return;
    const int chain = L.act[blockIdx.y], lane = threadIdx.x, z = blockIdx.x;
    const int64_t N = P.tp.n, Npad = P.tp.npad;
    double* h = L.H + (size_t)chain * Npad;
    const int64_t nb = (int64_t)z * DHMC_LOGISTIC_BLOCK;
    const int64_t ne = nb + DHMC_LOGISTIC_BLOCK < Npad ? nb + DHMC_LOGISTIC_BLOCK : Npad;
    double lpart = 0.0;
    constexpr int U = 8;                                  // loads in flight per lane
    for (int64_t n0 = nb; n0 < ne; n0 += U * WAVE) {
        double eta[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t n = n0 + u * WAVE + lane;
            eta[u] = n < ne ? h[n] : 0.0;
            y[u] = n < ne ? P.tp.c[n] : 0.0;
        }
        double sig[U], l1pe[U];
        logistic_link_batch<U>(eta, sig, l1pe);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t n = n0 + u * WAVE + lane;
            const bool valid = n < N;
            if (n < ne) h[n] = valid ? y[u] - sig[u] : 0.0;
            if (valid) lpart = lpart + (y[u] * eta[u] - l1pe[u]);
        }
    }
    const double bs = wave_allreduce1(lpart);
    if (lane == 0) L.S1P[(size_t)z * P.C + (chain - P.chain_base)] = bs;
}

// ------------------------------------------------------------------------------------------------------------------------------
// G_eta and K_r in ONE kernel (round 6): η is never written.  A workgroup owns 64 listed chains × the 32 observations
// n ≡ l (mod 64), l in [32 lh, 32 lh + 32), of EVERY group of 64 observations of block z — i.e. 32 of the 64 per-lane partial sums
// the ABI's wave order defines for that block (include/dhmc.h: the terms of the observations n = l mod 64 added in ascending order
// from +0, then the xor butterfly) — and walks the block's groups m = 0, 1, … in ascending order: per group a 64 × 32 × Dpad
// product on the matrix cores (the chain's row of Q′ stays in registers as MFMA A-fragments for the whole block; Xᵀ's 32
// columns stream through a double-buffered LDS tile), then per element the link — σ(η), r = y − σ, the term y η − log(1 + e^η),
// operation for operation logistic_link_kernel's — with r stored where η would have been and the term added to the element's
// running partial sum.  After the last group the 64 × 32 partial sums go to S1L[z][chain][l]; logistic_block_sums_kernel runs
// the butterfly over l.  η = x_n·β is the same ascending k chain of fma's as gemm_rows_f64_kernel's (v_mfma_f64_16x16x4_f64).
// Traffic per round: r written once (8 B per observation and chain) instead of η written, η read, r written.
// Workgroup id -> (z, lh, row tile) keeps the row tiles of one (z, lh) on one XCD back to back: its 2 MB of Xᵀ come from that L2.
// ------------------------------------------------------------------------------------------------------------------------------

#ifndef DHMC_LK_NE
#define DHMC_LK_NE 2
#endif
constexpr int LK_NE = DHMC_LK_NE;                              // arguments of the link in flight together (of a lane's 4 per accumulator)
// columns per workgroup; k per HALF stage (what one batch of loads / LDS stores covers: 64 k × 32 columns = 8 doubles per thread).  An LDS stage
// is two halves — 128 k under ONE barrier (Dpad = 64: one half) — and the row of k holds its 32 columns with bit 4 of the column flipped for odd
// k (k and k + 1 of a fragment read land on opposite bank halves: no padding, 64 KB for the two buffers, two workgroups per CU)
constexpr int LK_TL = 32, LK_TK = 64;

template <int DP>                                              // DP = Dpad (64, 128 or 256): the A-fragments are DP / 4 registers
__global__ __launch_bounds__(256, 2) void logistic_eta_link_kernel(RunParams P, LogisticRound L, const double* __restrict__ Q, int ntile_rows) {
    constexpr int KS = DP / 4;
    constexpr int NH = DP / LK_TK;                             // half stages per group of observations: 1, 2, 4
    constexpr int HPS = NH >= 2 ? 2 : 1;                       // halves per LDS stage
    constexpr int NST = NH / HPS;                              // stages (barriers) per group: 1, 1, 2
    constexpr int SROWS = HPS * LK_TK;                         // k rows per stage
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
    const int nm = (int)((ne - nb) / WAVE);                                        // groups of 64 observations in this block (Npad % 64 == 0)
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int l0 = LK_TL * lh;

    __shared__ __attribute__((aligned(16))) double Bs[2][SROWS * LK_TL];

    // A-fragments: lane (r16, kk) holds Q′[row r16 of the wave's 16][4 ks + kk], ks = 0 .. KS-1
    int arow = row0 + 16 * wv + (lane & 15);
    arow = arow < count ? arow : count - 1;
    const double* __restrict__ ap = Q + (size_t)L.act[arow] * P.Dpad + (lane >> 4);
    double a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = ap[4 * ks];

    // a half stage (64 k × 32 columns): thread t -> k row t / 4 of the half, the 8 columns from 8 (t % 4) — 64 contiguous bytes of Xᵀ's row
    const int b_k = t >> 2, b_c = 8 * (t & 3);
    const double* __restrict__ bsrc = P.tp.b + (size_t)b_k * Npad + nb + l0 + b_c;   // + 64 m + (size_t)64 hs Npad
    const int b_wr = b_k * LK_TL + (b_c ^ (16 * (b_k & 1)));                        // (64 is even: the half's first row has the parity of 0)
    double bv[8];
    auto bload = [&](int m, int hs) {                                               // half stage hs = 0 .. NH-1 of group m
        const gemm_d2* s2 = reinterpret_cast<const gemm_d2*>(bsrc + (size_t)(LK_TK * hs) * Npad + WAVE * m);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const gemm_d2 v = s2[i]; bv[2 * i] = v[0]; bv[2 * i + 1] = v[1]; }
    };
    auto bstore = [&](double* bs, int half) {
        gemm_d2* bw = reinterpret_cast<gemm_d2*>(bs + half * (LK_TK * LK_TL) + b_wr);
#pragma unroll
        for (int i = 0; i < 4; ++i) bw[i] = gemm_d2{bv[2 * i], bv[2 * i + 1]};
    };
    auto next_half = [&](int m, int hs, int& m1, int& hs1) { hs1 = hs + 1; m1 = m; if (hs1 == NH) { hs1 = 0; m1 = m + 1; } };
    // fragment reads: lane (c16, g) reads row 4 kk + g, columns c16 (b0) and 16 + c16 (b1), bit 4 flipped for odd rows (g odd)
    const int g = lane >> 4;
    const int b_rd0 = g * LK_TL + ((lane & 15) ^ (16 * (g & 1)));
    const int b_rd1 = g * LK_TL + ((16 + (lane & 15)) ^ (16 * (g & 1)));

    mfma_d4 lp[2];                                   // the running partial sums of the lane's 8 (chain, l) pairs
    lp[0] = mfma_d4{0.0, 0.0, 0.0, 0.0};
    lp[1] = mfma_d4{0.0, 0.0, 0.0, 0.0};
    // rows of the lane's accumulator registers: (lane >> 4) + 4 r of the wave's 16.  Rows past the list's end stand for its last row:
    // they compute and store that row's own values once more (no branch on a loop invariant inside the group loop)
    double* hrow[4];
    int srow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int lr = row0 + 16 * wv + (lane >> 4) + 4 * r;
        const int gg = L.act[lr < count ? lr : count - 1];
        hrow[r] = L.H + (size_t)gg * Npad + nb + l0 + (lane & 15);
        srow[r] = gg - P.chain_base;
    }

    // Pipeline of half stages.  Entering stage (m, st), bv holds the LAST half of that stage; its earlier half (HPS = 2) was stored into the
    // stage's buffer in the middle of the stage before.  One barrier per stage: it publishes both halves and says that every wave is done
    // with the other buffer, which then receives the next stage's first half in the middle of this one.
    int buf = 0;
    bload(0, 0);
    if constexpr (HPS == 2) {
        bstore(Bs[0], 0);
        bload(0, 1);
    }
#ifdef DHMC_LK_CLOCKS      // timing build (tools/gpu_scripts/r6b/l_c5_clocks.sh): where a wave's clocks go — wait for the staged loads, barrier, products, link
    unsigned long long ck_load = 0, ck_bar = 0, ck_mfma = 0, ck_link = 0;
    const unsigned long long ck_begin = __builtin_amdgcn_s_memtime();
#define LK_CK(acc_, t_) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_ += now_ - t_; t_ = now_; }
#else
#define LK_CK(acc_, t_)
#endif
#pragma nounroll
    for (int m = 0; m < nm; ++m) {
        mfma_d4 acc[2];
        acc[0] = mfma_d4{0.0, 0.0, 0.0, 0.0};
        acc[1] = mfma_d4{0.0, 0.0, 0.0, 0.0};
        const int64_t n_lo = nb + (int64_t)WAVE * m + l0 + (lane & 15);          // the lane's observations: n_lo, n_lo + 16
        const double y0 = n_lo < N ? P.tp.c[n_lo] : 0.0;
        const double y1 = n_lo + 16 < N ? P.tp.c[n_lo + 16] : 0.0;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            double* bs = Bs[buf];
            const int hs_last = HPS * st + HPS - 1;                                 // the half bv holds
#ifdef DHMC_LK_CLOCKS
            unsigned long long ck_t = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            LK_CK(ck_load, ck_t)
#endif
            bstore(bs, HPS - 1);
            int m1, hs1;
            next_half(m, hs_last, m1, hs1);
            if (m1 < nm) bload(m1, hs1);             // the next stage's first half: in flight under the barrier's wait and this stage's first products
            __syncthreads();                         // stage visible; every wave is done with the other buffer's previous contents
            LK_CK(ck_bar, ck_t)
#pragma unroll
            for (int hh = 0; hh < HPS; ++hh) {
#pragma unroll
                for (int kk = 0; kk < LK_TK / 4; ++kk) {
                    const int kr = (hh * LK_TK + 4 * kk) * LK_TL;
                    const double b0 = bs[kr + b_rd0];
                    const double b1 = bs[kr + b_rd1];
                    const int ka = (SROWS / 4) * st + (LK_TK / 4) * hh + kk;
                    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ka], b0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ka], b1, acc[1], 0, 0, 0);
                }
                if (HPS == 2 && hh == 0) {           // middle of the stage: the next stage's first half into the other buffer, its last half requested
                    if (m1 < nm) {
                        bstore(Bs[buf ^ 1], 0);
                        int m2, hs2;
                        next_half(m1, hs1, m2, hs2);
                        bload(m2, hs2);              // (m2 == m1: a stage's two halves belong to one group)
                    }
                }
            }
            buf ^= 1;
#ifdef DHMC_LK_CLOCKS
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            LK_CK(ck_mfma, ck_t)
#endif
        }
#ifdef DHMC_LK_CLOCKS
        unsigned long long ck_t2 = __builtin_amdgcn_s_memtime();
#endif
        // the link of the lane's 8 elements (logistic_link_kernel's operations, in phases; four at a time: the A-fragments hold half
        // the wave's registers)
        const bool whole = nb + (int64_t)WAVE * (m + 1) <= N;                     // no padding observation in this group (uniform)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t n = n_lo + 16 * j;
            const double y = j == 0 ? y0 : y1;
            const bool valid = n < N;
#pragma unroll
            for (int h = 0; h < 4 / LK_NE; ++h) {
                double eta[LK_NE], sig[LK_NE], l1pe[LK_NE];
#pragma unroll
                for (int r = 0; r < LK_NE; ++r) eta[r] = acc[j][LK_NE * h + r];
                logistic_link_batch<LK_NE>(eta, sig, l1pe);
#if defined(DHMC_LK_ABLATE) && DHMC_LK_ABLATE == 1          // timing build: the link twice, same results (tools/gpu_scripts/r6b/k_c5_ablate.sh)
                {
                    double eta2[LK_NE], sig2[LK_NE], l2[LK_NE];
#pragma unroll
                    for (int r = 0; r < LK_NE; ++r) { eta2[r] = eta[r]; asm volatile("" : "+v"(eta2[r])); }
                    logistic_link_batch<LK_NE>(eta2, sig2, l2);
#pragma unroll
                    for (int r = 0; r < LK_NE; ++r) asm volatile("" :: "v"(sig2[r]), "v"(l2[r]));
                }
#endif
                if (whole) {                                                       // every group but the data's last: no per-lane tests
#pragma unroll
                    for (int r = 0; r < LK_NE; ++r) {
                        hrow[LK_NE * h + r][WAVE * m + 16 * j] = y - sig[r];
                        lp[j][LK_NE * h + r] = lp[j][LK_NE * h + r] + (y * eta[r] - l1pe[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < LK_NE; ++r) {
                        hrow[LK_NE * h + r][WAVE * m + 16 * j] = valid ? y - sig[r] : 0.0;
                        if (valid) lp[j][LK_NE * h + r] = lp[j][LK_NE * h + r] + (y * eta[r] - l1pe[r]);
                    }
                }
            }
        }
#ifdef DHMC_LK_CLOCKS
        LK_CK(ck_link, ck_t2)
#endif
    }
#ifdef DHMC_LK_CLOCKS
    {
        __shared__ int dummy_;
        static __device__ int printed_;
        const unsigned long long total = __builtin_amdgcn_s_memtime() - ck_begin;
        if (count >= 1024 && lane == 0 && (blockIdx.x % 397) == 5 && atomicAdd(&printed_, 1) < 24)
            printf("LKCLK block %d wave %d groups %d total %llu load-wait %llu barrier %llu products %llu link %llu\n", (int)blockIdx.x, wv, nm, total, ck_load, ck_bar, ck_mfma, ck_link);
        (void)dummy_;
    }
#endif
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) L.S1L[((size_t)z * P.C + srow[r]) * WAVE + l0 + 16 * j + (lane & 15)] = lp[j][r];
}

// the butterfly over the 64 per-lane partial sums of (block z, listed chain): S1P[z][chain]
static __global__ __launch_bounds__(64) void logistic_block_sums_kernel(RunParams P, LogisticRound L) {
    if ((int)blockIdx.y >= *L.act_count) return;
    const int chain = L.act[blockIdx.y], lane = threadIdx.x, z = blockIdx.x;
    const size_t o = (size_t)z * P.C + (chain - P.chain_base);
    const double bs = wave_allreduce1(L.S1L[o * WAVE + lane]);
    if (lane == 0) L.S1P[o] = bs;
}

// does the fused η + link kernel serve this context?  (then the blocks' sums exist as 64 lane partials each: S1L)
__host__ __device__ inline bool logistic_link_is_fused(const LogisticRound& L, int ld, int64_t npad) {
    return L.S1L && (ld == 64 || ld == 128 || ld == 256) && npad % WAVE == 0;
}

// η, the link and the blocks' sums of the rows listed in L.act: fused where the chain's row fits the A-fragments, else G_eta + K_r.
// sums_kernel = false: the caller's next kernel folds the lane partials S1L itself (rounds_k2_logistic_kernel); S1P is then not written.
inline void launch_logistic_eta_link(const RunParams& P, const RoundBuffers& R, const LogisticRound& L, const double* Q, int C, hipStream_t s,
                                     bool sums_kernel = true) {
    const int ld = P.Dpad, npad = (int)P.tp.npad;
    if (logistic_link_is_fused(L, ld, P.tp.npad)) {
        const int ntr = (C + 63) / 64;
        const int ngrp = 2 * L.nz;
        const dim3 grid((unsigned)(8 * ntr * ((ngrp + 7) / 8)));
        if (ld == 256) hipLaunchKernelGGL((logistic_eta_link_kernel<256>), grid, dim3(256), 0, s, P, L, Q, ntr);
        else if (ld == 128) hipLaunchKernelGGL((logistic_eta_link_kernel<128>), grid, dim3(256), 0, s, P, L, Q, ntr);
        else hipLaunchKernelGGL((logistic_eta_link_kernel<64>), grid, dim3(256), 0, s, P, L, Q, ntr);
        if (sums_kernel) hipLaunchKernelGGL(logistic_block_sums_kernel, dim3((unsigned)L.nz, C), dim3(WAVE), 0, s, P, L);
    } else {
        launch_gemm_list(Q, ld, P.tp.b, npad, L.H, npad, C, ld, npad, L.act, L.act_count, s);          // η = Q′·Xᵀ
        hipLaunchKernelGGL(logistic_link_kernel, dim3((unsigned)L.nz, C), dim3(WAVE), 0, s, P, R, L);   // r, the blocks' sums
    }
}

// K2: gradient, log density, second half step, p♯
template <int NPL>
__global__ __launch_bounds__(64) void rounds_k2_logistic_kernel(RunParams P, RoundBuffers R, LogisticRound L) {
    const int chain = P.chain_base + blockIdx.x, lane = threadIdx.x;
    TreeState& S = R.ts[chain];
    if (S.phase != PH_LEAF) return;
    const size_t row = (size_t)chain * P.Dpad;
    const double h = S.eps_s / 2;
    double q[NPL], g[NPL], p[NPL];
    ldv<NPL>(P.st.q + row, lane, q);
    ldv<NPL>(L.P + row, lane, g);        // (Xᵀ r): the blocks' partial products, added in ascending order
    const size_t zs = (size_t)P.C * P.Dpad;
    constexpr int ZU = 8;                // blocks fetched together (the adds stay in order)
    int z = 1;
    for (; z + ZU <= L.nz; z += ZU) {
        double t[ZU][NPL];
#pragma unroll
        for (int u = 0; u < ZU; ++u) ldv<NPL>(L.P + (size_t)(z + u) * zs + row, lane, t[u]);
#pragma unroll
        for (int u = 0; u < ZU; ++u)
#pragma unroll
            for (int k = 0; k < NPL; ++k) g[k] = g[k] + t[u][k];
    }
    for (; z < L.nz; ++z) {
        const double* pz = L.P + (size_t)z * zs + row;
#pragma unroll
        for (int k = 0; k < NPL; ++k) g[k] = g[k] + pz[lane + WAVE * k];
    }
    ldv<NPL>(R.cp + row, lane, p);
    LaneAcc<1, NPL> qq;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        qq.add(0, k, q[k], q[k]);
        g[k] = g[k] - q[k];
    }
    double s1 = 0.0;                                      // the blocks' sums, added in ascending order
    if (logistic_link_is_fused(L, P.Dpad, P.tp.npad)) {   // … each the butterfly of its 64 lane partials (what logistic_block_sums_kernel computes),
        constexpr int ZB = 8;                             //     eight blocks' butterflies side by side
        const double* __restrict__ sl = L.S1L + (size_t)(chain - P.chain_base) * WAVE + lane;
        for (int z0 = 0; z0 < L.nz; z0 += ZB) {
            double v[ZB];
#pragma unroll
            for (int u = 0; u < ZB; ++u) v[u] = z0 + u < L.nz ? sl[(size_t)(z0 + u) * P.C * WAVE] : 0.0;
            wave_allreduce<ZB>(v);
#pragma unroll
            for (int u = 0; u < ZB; ++u)
                if (z0 + u < L.nz) s1 = (z0 + u == 0) ? v[u] : s1 + v[u];
        }
    } else
    for (int z0 = 0; z0 < L.nz; z0 += WAVE) {             //     … lane z fetches block z's
        const int zl = z0 + lane;
        const double v = zl < L.nz ? L.S1P[(size_t)zl * P.C + (chain - P.chain_base)] : 0.0;
        const int cnt = (L.nz - z0) < WAVE ? (L.nz - z0) : WAVE;
        for (int i = 0; i < cnt; ++i) {
            const double b = readlane_f64(v, i);
            s1 = (z0 + i == 0) ? b : s1 + b;
        }
    }
    double lq = uni_f64(s1 - 0.5 * wave_allreduce1(qq.fold(0)));
    bool pos_finite = true;
    if (!dm_isfinite(lq)) pos_finite = all_finite<LogisticT, NPL>(q);
    lq = demote_lq(lq, pos_finite, true);
#pragma unroll
    for (int k = 0; k < NPL; ++k) p[k] = p[k] + h * g[k];
    stv<NPL>(P.st.g + row, lane, g);
    stv<NPL>(R.cp + row, lane, p);
#pragma unroll
    for (int k = 0; k < NPL; ++k) R.cps[row + lane + WAVE * k] = P.st.minv[row + lane + WAVE * k] * p[k];
    if (lane == 0) {
        S.lq_leaf = lq;
        if (!pos_finite) S.status |= DHMC_ST_NONFINITE_POSITION;
    }
}

}  // namespace dhmc
