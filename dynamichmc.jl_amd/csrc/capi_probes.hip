// The Diagnostics probes (include/dhmc.h; reference src/diagnostics.jl leapfrog_trajectory / explore_log_acceptance_ratios): the
// wave-per-chain probe kernels (probe_kernels.hpp), and for models evaluated between kernels the lock-step ExtProbe.
#include "capi_internal.hpp"

using namespace capi;

extern "C" {

// ---- Diagnostics probes (probe_kernels.hpp) ---------------------------------------------------
namespace {
int probe_finish(dhmc_ctx* c, const DevBuf& dst, uint32_t* status) {
    const int C = c->cfg.chains;
    std::vector<uint32_t> st(C);
    HIP_TRY(c, hipMemcpyAsync(st.data(), dst.p, sizeof(uint32_t) * C, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int rc = DHMC_OK;
    for (int i = 0; i < C; ++i) {
        if (status) status[i] = st[i];
        if (st[i]) rc = DHMC_ERR_CHAIN_FAILURE;
    }
    return rc;
}

// the same two probes for a model evaluated by the host's callback (external_rounds.hpp): lock-step leapfrogs of all chains
struct ExtProbe {
    DevBuf rows[8], pi0, lq, alive;
    ExtProbeParams E{};
    dhmc_ctx* c = nullptr;
    int init(dhmc_ctx* ctx, uint32_t* d_status) {
        c = ctx;
        const size_t C = c->cfg.chains, n = C * c->Dpad;
        for (DevBuf& b : rows) {
            HIP_TRY(c, hipMalloc(&b.p, sizeof(double) * n));
            HIP_TRY(c, hipMemsetAsync(b.p, 0, sizeof(double) * n, c->stream));
        }
        HIP_TRY(c, hipMalloc(&pi0.p, sizeof(double) * C));
        HIP_TRY(c, hipMalloc(&lq.p, sizeof(double) * C));
        HIP_TRY(c, hipMalloc(&alive.p, sizeof(int32_t) * C));
        HIP_TRY(c, hipMemsetAsync(d_status, 0, sizeof(uint32_t) * C, c->stream));
        E.D = c->cfg.dim; E.Dpad = c->Dpad; E.C = (int)C; E.chain_offset = c->cfg.chain_offset; E.seed = c->cfg.seed; E.st = c->st;
        E.q = (double*)rows[0].p; E.p = (double*)rows[1].p; E.g = (double*)rows[2].p; E.pm = (double*)rows[3].p;
        E.trial = (double*)rows[4].p; E.ps = (double*)rows[5].p; E.p0 = (double*)rows[6].p; E.ps0 = (double*)rows[7].p;
        E.pi0 = (double*)pi0.p; E.lq_cur = (double*)lq.p; E.alive = (int32_t*)alive.p; E.status = d_status;
        E.lq_in = c->lr.S1; E.grad_in = c->rb.tbuf; E.dense = c->cfg.metric == DHMC_METRIC_DENSE;
        return DHMC_OK;
    }
    // p₀ (the caller's m-th momentum, or rand_p from the chains' streams), p₀♯, π₀
    int momentum(const double* d_p_in, int n_mom, int m, uint32_t momentum_index, bool ratios) {
        const int C = E.C, ld = E.Dpad;
        if (E.dense && !d_p_in) {                      // z into pₘ's row (free here), p₀ = z·Wᵀ
            ExtProbeParams Z = E;
            Z.p0 = E.pm;
            DHMC_EXT_NPL(ext_probe_momentum_kernel, dim3(C), Z, d_p_in, n_mom, m, momentum_index)
            launch_gemm_rows(E.pm, c->d_WT, E.p0, ld, C, nullptr, nullptr, c->stream);
        } else {
            DHMC_EXT_NPL(ext_probe_momentum_kernel, dim3(C), E, d_p_in, n_mom, m, momentum_index)
        }
        if (E.dense) launch_gemm_rows(E.p0, c->d_Minv, E.ps0, ld, C, nullptr, nullptr, c->stream);
        DHMC_EXT_NPL(ext_probe_start_kernel, dim3(C), E, (int)ratios)
        return DHMC_OK;
    }
    void restart(bool ratios) { DHMC_EXT_NPL(ext_probe_restart_kernel, dim3(E.C), E, (int)ratios) }
    int step(double eps) {                             // one leapfrog of every chain that still steps
        const int C = E.C, ld = E.Dpad;
        DHMC_EXT_NPL(ext_probe_half_kernel, dim3(C), E, eps)
        if (E.dense) {
            launch_gemm_rows(E.pm, c->d_Minv, E.ps, ld, C, nullptr, nullptr, c->stream);
            DHMC_EXT_NPL(ext_probe_pos_kernel, dim3(C), E, eps)
        }
        if (int rc = external_eval(c, E.trial)) return rc;
        DHMC_EXT_NPL(ext_probe_finish_kernel, dim3(C), E, eps)
        if (E.dense) launch_gemm_rows(E.p, c->d_Minv, E.ps, ld, C, nullptr, nullptr, c->stream);
        return DHMC_OK;
    }
    void record(int idx, int npos, int pos, bool start, double* od, double* ol, double* oq, double* op, int32_t* orange) {
        DHMC_EXT_NPL(ext_probe_record_kernel, dim3(E.C), E, idx, npos, pos, (int)start, od, ol, oq, op, orange)
    }
};
}  // namespace

int dhmc_leapfrog_trajectory(dhmc_ctx* c, double eps, int32_t first, int32_t last, uint32_t momentum_index,
                             const double* p, double* delta, double* logdensity, double* q_out, double* p_out,
                             int32_t* range, uint32_t* status) {
    if (!c || !delta || !logdensity) return DHMC_ERR_INVALID_ARGUMENT;
    if (!(first <= 0 && 0 <= last)) return DHMC_ERR_INVALID_ARGUMENT;   // diagnostics.jl:218
    if (c->external) DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int C = c->cfg.chains, D = c->cfg.dim;
    const size_t npos = (size_t)last - first + 1;
    DevBuf dp, dd, dl, dq, dpo, dr, dst;
    HIP_TRY(c, hipMalloc(&dd.p, sizeof(double) * C * npos));
    HIP_TRY(c, hipMalloc(&dl.p, sizeof(double) * C * npos));
    HIP_TRY(c, hipMalloc(&dr.p, sizeof(int32_t) * 2 * C));
    HIP_TRY(c, hipMalloc(&dst.p, sizeof(uint32_t) * C));
    // positions that are not visited stay NaN (all-ones bit pattern)
    HIP_TRY(c, hipMemsetAsync(dd.p, 0xFF, sizeof(double) * C * npos, c->stream));
    HIP_TRY(c, hipMemsetAsync(dl.p, 0xFF, sizeof(double) * C * npos, c->stream));
    if (q_out) {
        HIP_TRY(c, hipMalloc(&dq.p, sizeof(double) * C * npos * D));
        HIP_TRY(c, hipMemsetAsync(dq.p, 0xFF, sizeof(double) * C * npos * D, c->stream));
    }
    if (p_out) {
        HIP_TRY(c, hipMalloc(&dpo.p, sizeof(double) * C * npos * D));
        HIP_TRY(c, hipMemsetAsync(dpo.p, 0xFF, sizeof(double) * C * npos * D, c->stream));
    }
    if (p) {
        HIP_TRY(c, hipMalloc(&dp.p, sizeof(double) * C * D));
        HIP_TRY(c, hipMemcpyAsync(dp.p, p, sizeof(double) * C * D, hipMemcpyHostToDevice, c->stream));
    }
    if (c->external || c->logistic_batched) {   // the density is evaluated for all chains between kernels: one batched evaluation per step
        ExtProbe X;
        int rc;
        if ((rc = X.init(c, (uint32_t*)dst.p))) return rc;
        HIP_TRY(c, hipMemsetAsync(dr.p, 0, sizeof(int32_t) * 2 * C, c->stream));
        if ((rc = X.momentum((const double*)dp.p, 1, 0, momentum_index, false))) return rc;
        X.restart(false);
        X.record(-first, (int)npos, 0, true, (double*)dd.p, (double*)dl.p, (double*)dq.p, (double*)dpo.p, (int32_t*)dr.p);
        for (int dir = 0; dir < 2; ++dir) {
            const double e = dir == 0 ? eps : -eps;                          // diagnostics.jl:223,225
            const int count = dir == 0 ? last : -first;
            X.restart(false);
            for (int i = 1; i <= count; ++i) {
                if ((rc = X.step(e))) return rc;
                const int pos = dir == 0 ? i : -i;
                X.record(pos - first, (int)npos, pos, false, (double*)dd.p, (double*)dl.p, (double*)dq.p, (double*)dpo.p, (int32_t*)dr.p);
            }
        }
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(delta, dd.p, sizeof(double) * C * npos, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(logdensity, dl.p, sizeof(double) * C * npos, hipMemcpyDeviceToHost, c->stream));
        if (q_out) HIP_TRY(c, hipMemcpyAsync(q_out, dq.p, sizeof(double) * C * npos * D, hipMemcpyDeviceToHost, c->stream));
        if (p_out) HIP_TRY(c, hipMemcpyAsync(p_out, dpo.p, sizeof(double) * C * npos * D, hipMemcpyDeviceToHost, c->stream));
        if (range) HIP_TRY(c, hipMemcpyAsync(range, dr.p, sizeof(int32_t) * 2 * C, hipMemcpyDeviceToHost, c->stream));
        return probe_finish(c, dst, status);
    }
    ProbeParams P{};
    P.D = D; P.Dpad = c->Dpad; P.C = C; P.chain_offset = c->cfg.chain_offset; P.seed = c->cfg.seed;
    P.st = c->st; P.tp = c->tp; P.momentum_index = momentum_index; P.p_in = (const double*)dp.p; P.n_mom = 1;
    P.eps = eps; P.first = first; P.last = last;
    P.out_delta = (double*)dd.p; P.out_lq = (double*)dl.p; P.out_q = (double*)dq.p; P.out_p = (double*)dpo.p;
    P.out_range = (int32_t*)dr.p; P.out_status = (uint32_t*)dst.p;
    int rc = dispatch(c, Op::ProbeTrajectory, &P);
    if (rc) return rc;
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(delta, dd.p, sizeof(double) * C * npos, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(logdensity, dl.p, sizeof(double) * C * npos, hipMemcpyDeviceToHost, c->stream));
    if (q_out) HIP_TRY(c, hipMemcpyAsync(q_out, dq.p, sizeof(double) * C * npos * D, hipMemcpyDeviceToHost, c->stream));
    if (p_out) HIP_TRY(c, hipMemcpyAsync(p_out, dpo.p, sizeof(double) * C * npos * D, hipMemcpyDeviceToHost, c->stream));
    if (range) HIP_TRY(c, hipMemcpyAsync(range, dr.p, sizeof(int32_t) * 2 * C, hipMemcpyDeviceToHost, c->stream));
    return probe_finish(c, dst, status);
}

int dhmc_explore_log_acceptance_ratios(dhmc_ctx* c, const double* eps, int32_t n_eps, int32_t n_momenta,
                                       uint32_t momentum_index, const double* ps, double* out, uint32_t* status) {
    if (!c || !eps || !out || n_eps <= 0 || n_momenta <= 0) return DHMC_ERR_INVALID_ARGUMENT;
    if (c->external) DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int C = c->cfg.chains, D = c->cfg.dim;
    const size_t nout = (size_t)C * n_momenta * n_eps;
    DevBuf de, dp, dout, dst;
    HIP_TRY(c, hipMalloc(&de.p, sizeof(double) * n_eps));
    HIP_TRY(c, hipMalloc(&dout.p, sizeof(double) * nout));
    HIP_TRY(c, hipMalloc(&dst.p, sizeof(uint32_t) * C));
    HIP_TRY(c, hipMemcpyAsync(de.p, eps, sizeof(double) * n_eps, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(dout.p, 0xFF, sizeof(double) * nout, c->stream));
    if (ps) {
        HIP_TRY(c, hipMalloc(&dp.p, sizeof(double) * C * n_momenta * D));
        HIP_TRY(c, hipMemcpyAsync(dp.p, ps, sizeof(double) * C * n_momenta * D, hipMemcpyHostToDevice, c->stream));
    }
    if (c->external || c->logistic_batched) {
        ExtProbe X;
        int rc;
        if ((rc = X.init(c, (uint32_t*)dst.p))) return rc;
        for (int m = 0; m < n_momenta; ++m) {
            if ((rc = X.momentum((const double*)dp.p, n_momenta, m, momentum_index, true))) return rc;
            for (int e = 0; e < n_eps; ++e) {
                X.restart(true);
                if ((rc = X.step(eps[e]))) return rc;
                X.record(m * n_eps + e, n_momenta * n_eps, 0, false, (double*)dout.p, nullptr, nullptr, nullptr, nullptr);
            }
        }
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(out, dout.p, sizeof(double) * nout, hipMemcpyDeviceToHost, c->stream));
        return probe_finish(c, dst, status);
    }
    ProbeParams P{};
    P.D = D; P.Dpad = c->Dpad; P.C = C; P.chain_offset = c->cfg.chain_offset; P.seed = c->cfg.seed;
    P.st = c->st; P.tp = c->tp; P.momentum_index = momentum_index; P.p_in = (const double*)dp.p; P.n_mom = n_momenta;
    P.eps_list = (const double*)de.p; P.n_eps = n_eps; P.out_delta = (double*)dout.p; P.out_status = (uint32_t*)dst.p;
    int rc = dispatch(c, Op::ProbeRatios, &P);
    if (rc) return rc;
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, dout.p, sizeof(double) * nout, hipMemcpyDeviceToHost, c->stream));
    return probe_finish(c, dst, status);
}

}  // extern "C"
