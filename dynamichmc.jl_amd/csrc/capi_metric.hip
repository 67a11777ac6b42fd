// The kinetic energy's metric (include/dhmc.h): setters / getters of the diagonal and the dense M⁻¹, the warmup windows' updates
// from draws (mcmc.jl:368-375 sample_M⁻¹ + regularize_M⁻¹; pooled over chains, and over ranks through dhmc_set_metric_allreduce)
// and the device factorisation behind GaussianKineticEnergy(Symmetric(M⁻¹)) (hamiltonian.jl:73; dense_factor.hpp).
#include "capi_internal.hpp"
#include "dense_factor.hpp"
#include "metric_dense_adapt.hpp"

using namespace capi;

namespace capi {
// upload S (symmetric M⁻¹) and Wᵀ padded to [Dpad][Dpad]
int upload_dense_metric(dhmc_ctx* c, const std::vector<double>& S, const std::vector<double>& W) {
    const int D = c->cfg.dim;
    const size_t Dp = c->Dpad;
    std::vector<double> a(Dp * Dp, 0.0), b(Dp * Dp, 0.0);
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
            a[(size_t)i * Dp + j] = S[(size_t)i * D + j];
            b[(size_t)j * Dp + i] = W[(size_t)i * D + j];   // transpose: WT[k][i] = W[i][k]
        }
    const size_t nmat = c->per_chain_dense ? (size_t)c->cfg.chains : 1;
    for (size_t m = 0; m < nmat; ++m) {
        HIP_TRY(c, hipMemcpyAsync(c->d_Minv + m * Dp * Dp, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_WT + m * Dp * Dp, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DHMC_OK;
}
// κ := GaussianKineticEnergy(Symmetric(src)) (hamiltonian.jl:73) entirely on the device (dense_factor.hpp): src is a
// device matrix with row stride lsrc whose upper triangle is read.  The context's metric is replaced only if src is
// finite and positive definite; otherwise DHMC_ERR_INVALID_ARGUMENT (the reference's cholesky throws).
int device_dense_metric(dhmc_ctx* c, const double* src, int lsrc, int slot) {   // slot: a chain of a per-chain dense context, -1: all
    const int D = c->cfg.dim, ld = c->Dpad;
    const size_t n = (size_t)ld * ld;
    double* Stmp = c->d_fwork + 3 * n;
    double* WTtmp = c->d_fwork + n;              // the X buffer: free again once M = XᵀX exists
    int flags[2] = {0, 0};
    HIP_TRY(c, hipMemsetAsync(c->d_fflags, 0, 2 * sizeof(int), c->stream));
    hipLaunchKernelGGL(df_symmetrize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, src, lsrc, D, Stmp, ld, c->d_fflags);
    df_dense_metric(Stmp, Stmp, WTtmp, D, ld, c->d_fwork, c->d_fflags, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(flags, c->d_fflags, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (flags[0] || flags[1]) return DHMC_ERR_INVALID_ARGUMENT;
    const size_t nmat = c->per_chain_dense ? (size_t)c->cfg.chains : 1;
    for (size_t m = 0; m < nmat; ++m) {
        if (slot >= 0 && (size_t)slot != m) continue;
        HIP_TRY(c, hipMemcpyAsync(c->d_Minv + m * n, Stmp, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_WT + m * n, WTtmp, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DHMC_OK;
}
void launch_metric(const dhmc_ctx* c, const double* draws, int64_t N) {
    int D = c->cfg.dim, Dp = c->Dpad, C = c->cfg.chains;
    switch (c->NPL) {
    case 1: hipLaunchKernelGGL((metric_diag_kernel<1>), dim3(C), dim3(WAVE), 0, c->stream, D, Dp, N, draws, c->st.minv, c->st.W); break;
    case 2: hipLaunchKernelGGL((metric_diag_kernel<2>), dim3(C), dim3(WAVE), 0, c->stream, D, Dp, N, draws, c->st.minv, c->st.W); break;
    case 4: hipLaunchKernelGGL((metric_diag_kernel<4>), dim3(C), dim3(WAVE), 0, c->stream, D, Dp, N, draws, c->st.minv, c->st.W); break;
    case 8: hipLaunchKernelGGL((metric_diag_kernel<8>), dim3(C), dim3(WAVE), 0, c->stream, D, Dp, N, draws, c->st.minv, c->st.W); break;
    case 16: hipLaunchKernelGGL((metric_diag_kernel<16>), dim3(C), dim3(WAVE), 0, c->stream, D, Dp, N, draws, c->st.minv, c->st.W); break;
    case 32: hipLaunchKernelGGL((metric_diag_kernel<32>), dim3(C), dim3(WAVE), 0, c->stream, D, Dp, N, draws, c->st.minv, c->st.W); break;
    default: hipLaunchKernelGGL((metric_diag_kernel<64>), dim3(C), dim3(WAVE), 0, c->stream, D, Dp, N, draws, c->st.minv, c->st.W); break;
    }
}
}  // namespace capi

extern "C" {

int dhmc_set_metric_diag(dhmc_ctx* c, const double* minv, int per_chain, int on_device) {
    if (!c || !minv || c->cfg.metric != DHMC_METRIC_DIAG) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int D = c->cfg.dim, C = c->cfg.chains;
    const size_t n = per_chain ? (size_t)C * D : (size_t)D;
    if (!on_device) {
        for (size_t i = 0; i < n; ++i)
            if (!(minv[i] > 0) || !std::isfinite(minv[i])) return DHMC_ERR_INVALID_ARGUMENT;
    } else {                                   // the same @argcheck (hamiltonian.jl:63) for a device array
        DevBuf flag;
        int bad = 0;
        HIP_TRY(c, hipMalloc(&flag.p, sizeof(int)));
        HIP_TRY(c, hipMemsetAsync(flag.p, 0, sizeof(int), c->stream));
        hipLaunchKernelGGL(check_positive_finite_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, minv, n, (int*)flag.p);
        HIP_TRY(c, hipMemcpyAsync(&bad, flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (bad) return DHMC_ERR_INVALID_ARGUMENT;
    }
    Staged s;
    int rc = stage_in(c, minv, n * sizeof(double), on_device, &s);
    if (rc) return rc;
    size_t tot = (size_t)C * c->Dpad;
    hipLaunchKernelGGL(set_metric_diag_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, D, c->Dpad, C,
                       (const double*)s.dev, per_chain, c->st.minv, c->st.W);
    HIP_TRY(c, hipGetLastError());
    stage_free(c, &s);
    return DHMC_OK;
}

int dhmc_get_metric_diag(dhmc_ctx* c, double* minv, int on_device) {
    if (!c || !minv) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return copy_out_padded(c, c->st.minv, minv, on_device);
}

int dhmc_set_metric_dense(dhmc_ctx* c, const double* minv, int on_device) {
    if (!c || !minv || c->cfg.metric != DHMC_METRIC_DENSE) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int D = c->cfg.dim;
    Staged s;
    int rc = stage_in(c, minv, sizeof(double) * (size_t)D * D, on_device, &s);
    if (rc) return rc;
    rc = device_dense_metric(c, (const double*)s.dev, D);      // symmetrise, check, factorise: all on the device
    stage_free(c, &s);
    return rc;
}

int dhmc_set_dense_products(dhmc_ctx* c, int32_t products) {
    if (!c || c->cfg.metric != DHMC_METRIC_DENSE || (products != 1 && products != 2)) return DHMC_ERR_INVALID_ARGUMENT;
    c->dense_products = products;
    return DHMC_OK;
}
int dhmc_get_dense_products(const dhmc_ctx* c) { return (c && c->cfg.metric == DHMC_METRIC_DENSE) ? c->dense_products : 0; }

int dhmc_get_metric_dense_chain(dhmc_ctx* c, int32_t chain, double* minv, double* W) {
    if (!c || c->cfg.metric != DHMC_METRIC_DENSE || chain < 0 || chain >= c->cfg.chains) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int D = c->cfg.dim;
    const size_t Dp = c->Dpad;
    const size_t off = c->per_chain_dense ? (size_t)chain * Dp * Dp : 0;
    std::vector<double> a(Dp * Dp), b(Dp * Dp);
    HIP_TRY(c, hipMemcpyAsync(a.data(), c->d_Minv + off, a.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(b.data(), c->d_WT + off, b.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
            if (minv) minv[(size_t)i * D + j] = a[(size_t)i * Dp + j];
            if (W) W[(size_t)i * D + j] = b[(size_t)j * Dp + i];
        }
    return DHMC_OK;
}

int dhmc_get_metric_dense(dhmc_ctx* c, double* minv, double* W) { return dhmc_get_metric_dense_chain(c, 0, minv, W); }

int dhmc_update_metric_diag(dhmc_ctx* c, const double* draws, int64_t n, double lambda, int on_device) {
    if (!c || !draws || c->cfg.metric != DHMC_METRIC_DIAG) return DHMC_ERR_INVALID_ARGUMENT;
    if (n < 2 || !(lambda >= 0)) return DHMC_ERR_INVALID_ARGUMENT;  // mcmc.jl:191-192 (N >= 20 is the host wrapper's check)
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    Staged s;
    int rc = stage_in(c, draws, sizeof(double) * (size_t)c->cfg.chains * n * c->cfg.dim, on_device, &s);
    if (rc) return rc;
    launch_metric(c, (const double*)s.dev, n);
    HIP_TRY(c, hipGetLastError());
    stage_free(c, &s);
    return DHMC_OK;
}

// ---- the same estimate without the draws: running moments updated by the kernels at the end of every transition --------------
int dhmc_metric_window_begin(dhmc_ctx* c) {
    if (!c || c->cfg.metric != DHMC_METRIC_DIAG) return DHMC_ERR_INVALID_ARGUMENT;
    DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const size_t n = 2 * (size_t)c->cfg.chains * c->Dpad;
    if (!c->d_win) {
        int rc = dev_alloc(c, &c->d_win, n);
        if (rc) return rc;
    }
    HIP_TRY(c, hipMemsetAsync(c->d_win, 0, n * sizeof(double), c->stream));
    c->win_n = 0;
    return DHMC_OK;
}

int64_t dhmc_metric_window_count(const dhmc_ctx* c) { return c ? c->win_n : -1; }

int dhmc_metric_window_end(dhmc_ctx* c) {
    if (!c) return DHMC_ERR_INVALID_ARGUMENT;
    c->win_n = -1;
    return DHMC_OK;
}

int dhmc_update_metric_diag_window(dhmc_ctx* c, double lambda) {
    if (!c || c->cfg.metric != DHMC_METRIC_DIAG || c->win_n < 0) return DHMC_ERR_INVALID_ARGUMENT;
    if (c->win_n < 2 || !(lambda >= 0)) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const size_t n = (size_t)c->cfg.chains * c->Dpad;
    hipLaunchKernelGGL(window_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->cfg.dim, c->Dpad, c->cfg.chains,
                       (const double*)(c->d_win + n), c->win_n, c->st.minv, c->st.W);
    HIP_TRY(c, hipGetLastError());
    c->win_n = -1;
    return DHMC_OK;
}

int dhmc_update_metric_dense(dhmc_ctx* c, const double* draws, int64_t n, double lambda, int on_device) {
    if (!c || !draws || c->cfg.metric != DHMC_METRIC_DENSE) return DHMC_ERR_INVALID_ARGUMENT;
    if (n < 2 || !(lambda >= 0)) return DHMC_ERR_INVALID_ARGUMENT;  // mcmc.jl:191-192
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int D = c->cfg.dim, ld = c->Dpad;
    // shared M⁻¹: one estimate from the pooled draws of all chains; per-chain: every chain from its own n draws (mcmc.jl:281-285)
    const int nest = c->per_chain_dense ? c->cfg.chains : 1;
    const int64_t J = c->per_chain_dense ? n : (int64_t)c->cfg.chains * n;
    Staged s;
    int rc = DHMC_OK;
    int refused = 0, first_refused = -1;
    if (!c->per_chain_dense && c->metric_allreduce) {
        // Job-wide estimate (include/dhmc.h dhmc_set_metric_allreduce): column sums + row count over the ranks, then the scatter about
        // the job's mean over the ranks.  EVERY rank makes the same sequence of collective calls whatever happens to it locally: a
        // rank that cannot stage its draws or allocate its buffers takes part in the first all-reduce with zeros and a raised error
        // slot, all ranks read the reduced slot, and all of them return the error together before the second collective — a rank
        // that left early would leave the others blocked in it.  The first buffer is the factorisation's work space (free until
        // device_dense_metric below), so that taking part needs no allocation.
        DevBuf bmean, bS;
        double local_err = 0.0;
        if (stage_in(c, draws, sizeof(double) * (size_t)c->cfg.chains * n * D, on_device, &s) != DHMC_OK) local_err = 1.0;
        if (hipMalloc(&bmean.p, sizeof(double) * ld) != hipSuccess) local_err = 1.0;
        if (hipMalloc(&bS.p, sizeof(double) * (size_t)ld * ld) != hipSuccess) local_err = 1.0;
        double* const mean = (double*)bmean.p;
        double* const S = (double*)bS.p;
        double* const sums = c->d_fwork;                      // D + 2 slots: column sums, row count, error flag
        const double* x = (const double*)s.dev;
        // (no early return before the first collective: a HIP error here is this rank's local_err like any other — the memset that
        // failed leaves garbage column sums, which no rank uses once the reduced error slot is raised)
        if (hipMemsetAsync(sums, 0, sizeof(double) * (size_t)(D + 2), c->stream) != hipSuccess) local_err = 1.0;
        if (local_err == 0.0) {
            hipLaunchKernelGGL(pooled_mean_kernel, dim3((D + 255) / 256), dim3(256), 0, c->stream, D, J, x, sums, (size_t)0, (size_t)0, 1);
            if (hipGetLastError() != hipSuccess) local_err = 1.0;
        }
        double tail[2] = {local_err != 0.0 ? 0.0 : (double)J, local_err};
        if (hipMemcpyAsync(sums + D, tail, 2 * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            // the slot could not be written: write it by a kernel-free path of last resort (a blocking copy); if that fails too the
            // device is gone and the collective cannot be entered by this rank in any case
            tail[1] = local_err = 1.0; tail[0] = 0.0;
            (void)hipMemcpy(sums + D, tail, 2 * sizeof(double), hipMemcpyHostToDevice);
        }
        if (c->metric_allreduce(c->metric_allreduce_user, sums, (int64_t)D + 2, (void*)c->stream) != 0) {
            c->err = "dhmc_update_metric_dense: the all-reduce callback failed (column sums)"; stage_free(c, &s); return DHMC_ERR_CALLBACK;
        }
        // (after the collective every rank is on its own again until the second one, which all reach or none: a rank whose copy back
        // fails cannot know the job's verdict and reports its own error; the others' second collective then times out in the
        // communicator, as for any rank that dies — there is no way to tell them without a collective)
        HIP_TRY(c, hipMemcpyAsync(tail, sums + D, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));           // the job's row count and error count are on the host now — on every rank
        const double Jtot = tail[0];
        if (tail[1] != 0.0) {
            c->err = local_err != 0.0 ? "dhmc_update_metric_dense: this rank could not stage its draws / allocate its buffers (all ranks return)"
                                      : "dhmc_update_metric_dense: another rank of the job failed before the estimate (all ranks return)";
            stage_free(c, &s);
            return DHMC_ERR_HIP;
        }
        if (!(Jtot >= 2.0)) { stage_free(c, &s); return DHMC_ERR_INVALID_ARGUMENT; }
        hipLaunchKernelGGL(pooled_mean_finish_kernel, dim3((D + 255) / 256), dim3(256), 0, c->stream, D, sums);
        HIP_TRY(c, hipMemcpyAsync(mean, sums, sizeof(double) * (size_t)D, hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(pooled_cov_kernel, dim3(ld / 64, ld / 64), dim3(256), 0, c->stream, D, J, x, mean, S, ld, (size_t)0, (size_t)0, (size_t)0);
        if (c->metric_allreduce(c->metric_allreduce_user, S, (int64_t)ld * ld, (void*)c->stream) != 0) {
            c->err = "dhmc_update_metric_dense: the all-reduce callback failed (scatter matrix)"; stage_free(c, &s); return DHMC_ERR_CALLBACK;
        }
        // from here on every rank holds the same S: regularisation and factorisation succeed or fail on all of them alike
        hipLaunchKernelGGL(cov_regularize_kernel, dim3((unsigned)(((size_t)D * D + 255) / 256)), dim3(256), 0, c->stream, D, ld, (int64_t)Jtot, lambda, S, (size_t)0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { c->err = std::string("dhmc_update_metric_dense: ") + hipGetErrorString(e); rc = DHMC_ERR_HIP; }
        else rc = device_dense_metric(c, S, ld, -1);   // DHMC_ERR_INVALID_ARGUMENT: the estimate is not positive definite
        stage_free(c, &s);
        return rc;
    }
    rc = stage_in(c, draws, sizeof(double) * (size_t)c->cfg.chains * n * D, on_device, &s);
    if (rc) return rc;
    if (!c->per_chain_dense) {
        DevBuf bmean, bS;
        HIP_TRY(c, hipMalloc(&bmean.p, sizeof(double) * ld));
        HIP_TRY(c, hipMalloc(&bS.p, sizeof(double) * (size_t)ld * ld));
        double* const mean = (double*)bmean.p;
        double* const S = (double*)bS.p;
        const double* x = (const double*)s.dev;
        const double Jtot = (double)J;
        hipLaunchKernelGGL(pooled_mean_kernel, dim3((D + 255) / 256), dim3(256), 0, c->stream, D, J, x, mean, (size_t)0, (size_t)0, 0);
        hipLaunchKernelGGL(pooled_cov_kernel, dim3(ld / 64, ld / 64), dim3(256), 0, c->stream, D, J, x, mean, S, ld, (size_t)0, (size_t)0, (size_t)0);
        hipLaunchKernelGGL(cov_regularize_kernel, dim3((unsigned)(((size_t)D * D + 255) / 256)), dim3(256), 0, c->stream, D, ld, (int64_t)Jtot, lambda, S, (size_t)0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { c->err = std::string("dhmc_update_metric_dense: ") + hipGetErrorString(e); rc = DHMC_ERR_HIP; }
        else rc = device_dense_metric(c, S, ld, -1);   // DHMC_ERR_INVALID_ARGUMENT: the estimate is not positive definite
    } else {
        // per-chain metrics (mcmc.jl:281-285 runs per chain): estimate, regularise and factorise a BATCH of chains per launch
        // (blockIdx.z = chain; the same kernels, so the same bits as chain by chain), ≈ 1 GiB of work space at a time.  Every chain
        // stands for itself: one whose estimate is refused keeps its metric, the others are updated all the same.
        const size_t n = (size_t)ld * ld;
        const int Bmax = (int)std::max<size_t>(1, std::min<size_t>((size_t)nest, ((size_t)1 << 30) / (5 * n * sizeof(double))));
        DevBuf bmean, bS, bwork, bflags;
        HIP_TRY(c, hipMalloc(&bmean.p, sizeof(double) * (size_t)Bmax * ld));
        HIP_TRY(c, hipMalloc(&bS.p, sizeof(double) * (size_t)Bmax * n * 2));          // the estimates, and Symmetric(estimate)
        HIP_TRY(c, hipMalloc(&bwork.p, sizeof(double) * (size_t)Bmax * n * 3));
        HIP_TRY(c, hipMalloc(&bflags.p, sizeof(int) * 2 * (size_t)Bmax));
        double* const mean = (double*)bmean.p;
        double* const S = (double*)bS.p;
        double* const Ssym = S + (size_t)Bmax * n;
        double* const work = (double*)bwork.p;
        int* const flags = (int*)bflags.p;
        std::vector<int> hflags(2 * (size_t)Bmax);
        for (int k0 = 0; k0 < nest && rc == DHMC_OK; k0 += Bmax) {
            const int B = std::min(Bmax, nest - k0);
            const unsigned Bz = (unsigned)B;
            const double* x = (const double*)s.dev + (size_t)k0 * J * D;
            HIP_TRY(c, hipMemsetAsync(flags, 0, sizeof(int) * 2 * (size_t)B, c->stream));
            hipLaunchKernelGGL(pooled_mean_kernel, dim3((D + 255) / 256, 1, Bz), dim3(256), 0, c->stream, D, J, x, mean, (size_t)J * D, (size_t)ld, 0);
            hipLaunchKernelGGL(pooled_cov_kernel, dim3(ld / 64, ld / 64, Bz), dim3(256), 0, c->stream, D, J, x, mean, S, ld, (size_t)J * D, (size_t)ld, n);
            hipLaunchKernelGGL(cov_regularize_kernel, dim3((unsigned)(((size_t)D * D + 255) / 256), 1, Bz), dim3(256), 0, c->stream, D, ld, J, lambda, S, n);
            hipLaunchKernelGGL(df_symmetrize_kernel, dim3((unsigned)((n + 255) / 256), 1, Bz), dim3(256), 0, c->stream, (const double*)S, ld, D, Ssym, ld,
                               flags, n, n);
            double* const WTb = work + (size_t)B * n;                                  // the X buffers: free again once M = XᵀX exists
            df_dense_metric(Ssym, Ssym, WTb, D, ld, work, flags, c->stream, B);
            hipLaunchKernelGGL(df_commit_kernel, dim3((unsigned)((n + 255) / 256), 1, Bz), dim3(256), 0, c->stream, (const double*)Ssym, (const double*)WTb,
                               (const int*)flags, c->d_Minv + (size_t)k0 * n, c->d_WT + (size_t)k0 * n, n);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { c->err = std::string("dhmc_update_metric_dense: ") + hipGetErrorString(e); rc = DHMC_ERR_HIP; break; }
            HIP_TRY(c, hipMemcpyAsync(hflags.data(), flags, sizeof(int) * 2 * (size_t)B, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            for (int b2 = 0; b2 < B; ++b2)
                if (hflags[2 * b2] || hflags[2 * b2 + 1]) { if (refused++ == 0) first_refused = k0 + b2; }
        }
    }
    stage_free(c, &s);
    if (rc == DHMC_OK && refused) {
        c->err = "dhmc_update_metric_dense: the covariance estimate of " + std::to_string(refused) + " chain(s) (first: chain " +
                 std::to_string(first_refused) + ") is not finite / positive definite; those chains keep their metric";
        return DHMC_ERR_INVALID_ARGUMENT;
    }
    return rc;
}

int dhmc_set_metric_allreduce(dhmc_ctx* c, dhmc_allreduce_fn fn, void* user) {
    if (!c || c->cfg.metric != DHMC_METRIC_DENSE || c->per_chain_dense) return DHMC_ERR_INVALID_ARGUMENT;   // a shared dense metric is what is pooled
    c->metric_allreduce = fn;
    c->metric_allreduce_user = fn ? user : nullptr;
    return DHMC_OK;
}

}  // extern "C"
