// Leapfrog probes: the two Diagnostics functions of the reference that call the hot path directly,
// batched over chains (one wavefront = one chain, state in registers as in nuts_kernels.hpp):
//   * leapfrog_trajectory          src/diagnostics.jl:214-227 (iterator :176-186, Δ :194-197)
//   * explore_log_acceptance_ratios src/diagnostics.jl:144-152 (local_log_acceptance_ratio, stepsize.jl:75-85)
// Both start from the chain's current (q, ℓq, ∇ℓq) and never modify the chain: failures are reported in a
// separate status array, not in the context's.
#pragma once
#include "nuts_dense_kernel.hpp"

namespace dhmc {

struct ProbeParams {
    int D, Dpad, C, chain_offset;
    uint64_t seed;
    ChainArrays st;
    TargetParams tp;
    uint32_t momentum_index;   // stream index of the first sampled momentum (purpose PURPOSE_PROBE_MOMENTUM)
    const double* p_in;        // caller's momenta [C][n_mom][D], or null: p = rand_p
    int n_mom;                 // 1 for a trajectory
    // trajectory
    double eps;
    int first, last;           // positions first..last, first <= 0 <= last
    double* out_lq;            // [C][npos]
    double* out_q;             // [C][npos][D] or null
    double* out_p;             // [C][npos][D] or null
    int32_t* out_range;        // [C][2]: positions actually visited (lo, hi)
    // acceptance ratios
    const double* eps_list;    // [n_eps]
    int n_eps;
    double* out_delta;         // trajectory: [C][npos]; ratios: [C][n_mom][n_eps]
    uint32_t* out_status;      // [C]
};

template <class T, int NPL, bool DENSE>
struct ProbeState {
    const T& tgt;
    const double* m_lds;
    DenseMetric M;
    int lane, D, Dpad;
    double q[NPL], p[NPL], g[NPL], ps[NPL];
    __device__ __forceinline__ void step(double eps, double& lq, double& pi, bool& pfin) {
        if constexpr (DENSE) leapfrog_leaf_dense<T, NPL>(tgt, M, Dpad, lane, D, q, p, g, ps, eps, lq, pi, pfin);
        else leapfrog_leaf<T, NPL>(tgt, m_lds, lane, D, q, p, g, eps, lq, pi, pfin);
    }
};

template <int NPL>
__device__ __forceinline__ void store_unpadded(double* dst, int lane, int D, const double (&v)[NPL]) {
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int e = lane + WAVE * k;
        if (e < D) dst[e] = v[k];
    }
}

// MODE 0: trajectory, MODE 1: acceptance ratios
template <class T, int NPL, bool DENSE, int MODE>
__global__ __launch_bounds__(64, 1) void probe_kernel(ProbeParams P, DenseMetric Mall) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const DenseMetric M = Mall.of_chain(chain);
    const int D = P.D, Dpad = P.Dpad;
    extern __shared__ double lds[];
    const T tgt(P.tp);
    const size_t row = (size_t)chain * Dpad;
    const ChainKey key{(uint32_t)P.seed, (uint32_t)(P.chain_offset + chain), (uint32_t)(P.seed >> 32)};
    if constexpr (!DENSE) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) lds[lane + WAVE * k] = P.st.minv[row + lane + WAVE * k];
    }
    ProbeState<T, NPL, DENSE> S{tgt, lds, M, lane, D, Dpad, {}, {}, {}, {}};
    double q0[NPL], g0[NPL], p0[NPL], ps0[NPL];
    ldv<NPL>(P.st.q + row, lane, q0);
    ldv<NPL>(P.st.g + row, lane, g0);
    const double lq0 = P.st.lq[chain];
    uint32_t status = 0;
    const int npos = P.last - P.first + 1;

    for (int m = 0; m < P.n_mom; ++m) {
        // p = rand_p(rng, κ) (hamiltonian.jl:124) or the caller's
        if (P.p_in) {
            const double* src = P.p_in + ((size_t)chain * P.n_mom + m) * D;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const int e = lane + WAVE * k;
                p0[k] = e < D ? src[e] : 0.0;
            }
            if constexpr (DENSE) sym_matvec<NPL>(M.Minv, Dpad, D, lane, p0, ps0);
        } else {
            if constexpr (DENSE) sample_momentum_dense<NPL>(key, PURPOSE_PROBE_MOMENTUM, P.momentum_index + (uint32_t)m, M, Dpad, D, lane, p0, ps0);
            else sample_momentum<NPL>(key, PURPOSE_PROBE_MOMENTUM, P.momentum_index + (uint32_t)m, P.st.W + row, lane, p0);
        }
        LaneAcc<1, NPL> kacc;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            if constexpr (!DENSE) ps0[k] = lds[lane + WAVE * k] * p0[k];
            kacc.add(0, k, p0[k], ps0[k]);
        }
        const double pi0 = uni_f64(joint_logdensity(lq0, wave_allreduce1(kacc.fold(0)) / 2.0));
        auto restart = [&]() {
#pragma unroll
            for (int k = 0; k < NPL; ++k) { S.q[k] = q0[k]; S.p[k] = p0[k]; S.g[k] = g0[k]; S.ps[k] = ps0[k]; }
        };

        if constexpr (MODE == 1) {
            double* out = P.out_delta + ((size_t)chain * P.n_mom + m) * P.n_eps;
            if (!dm_isfinite(pi0)) {   // stepsize.jl:77-79 throws
                status |= DHMC_ST_NONFINITE_START_DENSITY;
                continue;
            }
            for (int e = 0; e < P.n_eps; ++e) {
                restart();
                double lq1, pi1;
                bool pfin;
                S.step(P.eps_list[e], lq1, pi1, pfin);
                if (!pfin) status |= DHMC_ST_NONFINITE_POSITION;
                if (lane == 0) out[e] = pi1 - pi0;   // stepsize.jl:81-83
            }
        } else {
            double* od = P.out_delta + (size_t)chain * npos;
            double* ol = P.out_lq + (size_t)chain * npos;
            auto record = [&](int pos, double lq, double pi) {
                const int idx = pos - P.first;
                if (lane == 0) { od[idx] = pi - pi0; ol[idx] = lq; }   // diagnostics.jl:196
                if (P.out_q) store_unpadded<NPL>(P.out_q + ((size_t)chain * npos + idx) * D, lane, D, S.q);
                if (P.out_p) store_unpadded<NPL>(P.out_p + ((size_t)chain * npos + idx) * D, lane, D, S.p);
            };
            restart();
            record(0, lq0, pi0);
            int lo = 0, hi = 0;
            for (int dir = 0; dir < 2; ++dir) {
                const double eps = dir == 0 ? P.eps : -P.eps;        // diagnostics.jl:223,225
                const int count = dir == 0 ? P.last : -P.first;
                restart();
                double lq = lq0;
                for (int i = 1; i <= count; ++i) {
                    if (!dm_isfinite(lq)) break;                     // diagnostics.jl:179
                    double pi;
                    bool pfin;
                    S.step(eps, lq, pi, pfin);
                    if (!pfin) status |= DHMC_ST_NONFINITE_POSITION; // hamiltonian.jl:203 throws
                    const int pos = dir == 0 ? i : -i;
                    record(pos, lq, pi);
                    if (dir == 0) hi = pos; else lo = pos;
                }
            }
            if (lane == 0) { P.out_range[2 * chain] = lo; P.out_range[2 * chain + 1] = hi; }
        }
    }
    if (lane == 0) P.out_status[chain] = status;
}

}  // namespace dhmc
