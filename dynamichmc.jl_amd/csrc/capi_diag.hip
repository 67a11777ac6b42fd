// Post-hoc diagnostics on the device, where the draws and the tree statistics lie (include/dhmc.h): ESS / R-hat (plain, bulk,
// tail; ess_kernels.hpp) and summarize_tree_statistics / EBFMI (treestat_kernels.hpp).  No context: device + stream are arguments.
#include <hipcub/hipcub.hpp>
#include "capi_util.hpp"
#include "ess_kernels.hpp"
#include "treestat_kernels.hpp"

using namespace dhmc;

extern "C" {
// ESS and R-hat of `ncoords` series sets laid out as draws[C][n][dim] (ess_kernels.hpp): n <= ESS_LDS_MAX_N with the whole series in LDS,
// longer series from HBM a chunk of lags at a time.  de / dr: device [ncoords].  DHMC_ESS_LONG=1 forces the long path (tests).
namespace {
struct EssWork {
    DevBuf da, dm, dx, dst;
    bool long_series = false;
    int prepare(int64_t chains, int64_t n, int ncoords) {
        const char* e = std::getenv("DHMC_ESS_LONG");
        const bool force_long = e && std::atoi(e) != 0;
        long_series = n > ESS_LDS_MAX_N || force_long;
        const size_t nseries = (size_t)ncoords * chains;
        if (nseries > 0x7fffffffull) return DHMC_ERR_UNSUPPORTED;
        if (hipMalloc(&dm.p, sizeof(double) * nseries) != hipSuccess) return DHMC_ERR_HIP;
        if (!long_series) return hipMalloc(&da.p, sizeof(double) * nseries * n) == hipSuccess ? DHMC_OK : DHMC_ERR_HIP;
        if (hipMalloc(&dx.p, sizeof(double) * nseries * n) != hipSuccess || hipMalloc(&da.p, sizeof(double) * nseries * ESS_LAG_CHUNK) != hipSuccess ||
            hipMalloc(&dst.p, sizeof(EssState) * ncoords) != hipSuccess)
            return DHMC_ERR_HIP;
        return DHMC_OK;
    }
};
// the short path only enqueues work on s; the long path returns with the stream drained
int ess_estimate(EssWork& w, hipStream_t s, const double* draws, int64_t chains, int64_t n, int64_t dim, const int32_t* d_coords, int ncoords,
                 double* de, double* dr) {
    if (!w.long_series) {
        hipLaunchKernelGGL(ess_acov_kernel, dim3(ncoords, (unsigned)chains), dim3(ESS_THREADS), sizeof(double) * n, s, draws, n, dim,
                           d_coords, chains, (double*)w.da.p, (double*)w.dm.p);
        hipLaunchKernelGGL(ess_finish_kernel, dim3(ncoords), dim3(ESS_THREADS), sizeof(double) * n, s, (const double*)w.da.p,
                           (const double*)w.dm.p, n, chains, de, dr);
        return hipGetLastError() == hipSuccess ? DHMC_OK : DHMC_ERR_HIP;
    }
    const size_t nseries = (size_t)ncoords * chains;
    hipLaunchKernelGGL(ess_center_kernel, dim3(ncoords, (unsigned)chains), dim3(ESS_THREADS), 0, s, draws, n, dim, d_coords, chains,
                       (double*)w.dx.p, (double*)w.dm.p);
    std::vector<EssState> st(ncoords);
    for (int64_t t0 = 0; t0 < n; t0 += ESS_LAG_CHUNK) {
        hipLaunchKernelGGL(ess_acov_lags_kernel, dim3((unsigned)nseries, ESS_LAG_CHUNK / ESS_THREADS), dim3(ESS_THREADS), 0, s,
                           (const double*)w.dx.p, n, t0, (double*)w.da.p);
        hipLaunchKernelGGL(ess_finish_chunk_kernel, dim3(ncoords), dim3(ESS_THREADS), 0, s, (const double*)w.da.p, (const double*)w.dm.p, n,
                           chains, t0, (EssState*)w.dst.p, de, dr);
        if (hipGetLastError() != hipSuccess) return DHMC_ERR_HIP;
        if (hipMemcpyAsync(st.data(), w.dst.p, sizeof(EssState) * ncoords, hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
        if (hipStreamSynchronize(s) != hipSuccess) return DHMC_ERR_HIP;
        bool all = true;
        for (const EssState& e : st) all = all && e.done;
        if (all) break;
    }
    return DHMC_OK;
}
}  // namespace

int dhmc_ess_rhat(int32_t device, void* stream, const double* draws, int64_t chains, int64_t n, int64_t dim,
                  const int32_t* coords, int32_t ncoords, double* ess, double* rhat) {
    if (!draws || !coords || !ess || !rhat || chains < 1 || n < 4 || dim < 1 || ncoords < 1) return DHMC_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < ncoords; ++i)
        if (coords[i] < 0 || coords[i] >= dim) return DHMC_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(device) != hipSuccess) return DHMC_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    DevBuf dc, de, dr;
    EssWork work;
    if (int rc = work.prepare(chains, n, ncoords)) return rc;
    if (hipMalloc(&dc.p, sizeof(int32_t) * ncoords) != hipSuccess || hipMalloc(&de.p, sizeof(double) * ncoords) != hipSuccess ||
        hipMalloc(&dr.p, sizeof(double) * ncoords) != hipSuccess)
        return DHMC_ERR_HIP;
    if (hipMemcpyAsync(dc.p, coords, sizeof(int32_t) * ncoords, hipMemcpyHostToDevice, s) != hipSuccess) return DHMC_ERR_HIP;
    if (int rc = ess_estimate(work, s, draws, chains, n, dim, (const int32_t*)dc.p, ncoords, (double*)de.p, (double*)dr.p)) return rc;
    if (hipMemcpyAsync(ess, de.p, sizeof(double) * ncoords, hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
    if (hipMemcpyAsync(rhat, dr.p, sizeof(double) * ncoords, hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
    if (hipStreamSynchronize(s) != hipSuccess) return DHMC_ERR_HIP;
    return DHMC_OK;
}

int dhmc_ess_bulk(int32_t device, void* stream, const double* draws, int64_t chains, int64_t n, int64_t dim,
                  const int32_t* coords, int32_t ncoords, double* ess, double* rhat) {
    if (!draws || !coords || !ess || !rhat || chains < 1 || n < 8 || dim < 1 || ncoords < 1) return DHMC_ERR_INVALID_ARGUMENT;
    const int64_t half = n / 2, N2 = 2 * half, S = chains * N2, C2 = 2 * chains;
    if (S > 0x7fffffffll) return DHMC_ERR_UNSUPPORTED;
    for (int i = 0; i < ncoords; ++i)
        if (coords[i] < 0 || coords[i] >= dim) return DHMC_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(device) != hipSuccess) return DHMC_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    DevBuf dk, dk2, di, di2, dz, de, dr, dtmp, dc0;
    EssWork work;
    if (int rc = work.prepare(C2, half, 1)) return rc;
    size_t tmp_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const double*)nullptr, (double*)nullptr, (const int32_t*)nullptr,
                                           (int32_t*)nullptr, (int)S, 0, 64, s) != hipSuccess) return DHMC_ERR_HIP;
    const int32_t zero = 0;
    if (hipMalloc(&dk.p, sizeof(double) * S) != hipSuccess || hipMalloc(&dk2.p, sizeof(double) * S) != hipSuccess ||
        hipMalloc(&di.p, sizeof(int32_t) * S) != hipSuccess || hipMalloc(&di2.p, sizeof(int32_t) * S) != hipSuccess ||
        hipMalloc(&dz.p, sizeof(double) * S) != hipSuccess || hipMalloc(&de.p, sizeof(double)) != hipSuccess ||
        hipMalloc(&dr.p, sizeof(double)) != hipSuccess || hipMalloc(&dtmp.p, tmp_bytes ? tmp_bytes : 8) != hipSuccess ||
        hipMalloc(&dc0.p, sizeof(int32_t)) != hipSuccess)
        return DHMC_ERR_HIP;
    if (hipMemcpyAsync(dc0.p, &zero, sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess) return DHMC_ERR_HIP;
    const unsigned nb = (unsigned)((S + 255) / 256);
    for (int j = 0; j < ncoords; ++j) {
        hipLaunchKernelGGL(ess_gather_kernel, dim3(nb), dim3(256), 0, s, draws, n, dim, coords[j], chains, N2, (double*)dk.p, (int32_t*)di.p);
        if (hipcub::DeviceRadixSort::SortPairs(dtmp.p, tmp_bytes, (const double*)dk.p, (double*)dk2.p, (const int32_t*)di.p,
                                               (int32_t*)di2.p, (int)S, 0, 64, s) != hipSuccess) return DHMC_ERR_HIP;
        hipLaunchKernelGGL(ess_rank_kernel, dim3(nb), dim3(256), 0, s, (const double*)dk2.p, (const int32_t*)di2.p, S, (double*)dz.p);
        // z is [2C][N'][1]: the estimator of dhmc_ess_rhat on one "coordinate"
        if (hipGetLastError() != hipSuccess) return DHMC_ERR_HIP;
        if (int rc = ess_estimate(work, s, (const double*)dz.p, C2, half, 1, (const int32_t*)dc0.p, 1, (double*)de.p, (double*)dr.p)) return rc;
        if (hipMemcpyAsync(ess + j, de.p, sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
        if (hipMemcpyAsync(rhat + j, dr.p, sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return DHMC_ERR_HIP;
    return DHMC_OK;
}

int dhmc_ess_tail(int32_t device, void* stream, const double* draws, int64_t chains, int64_t n, int64_t dim,
                  const int32_t* coords, int32_t ncoords, double* ess) {
    if (!draws || !coords || !ess || chains < 1 || n < 8 || dim < 1 || ncoords < 1) return DHMC_ERR_INVALID_ARGUMENT;
    const int64_t half = n / 2, N2 = 2 * half, S = chains * N2, C2 = 2 * chains;
    if (S > 0x7fffffffll) return DHMC_ERR_UNSUPPORTED;
    for (int i = 0; i < ncoords; ++i)
        if (coords[i] < 0 || coords[i] >= dim) return DHMC_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(device) != hipSuccess) return DHMC_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    DevBuf dk, dk2, di, di2, dz, de, dr, dtmp, dc0;
    EssWork work;
    if (int rc = work.prepare(C2, half, 1)) return rc;
    size_t tmp_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const double*)nullptr, (double*)nullptr, (const int32_t*)nullptr,
                                           (int32_t*)nullptr, (int)S, 0, 64, s) != hipSuccess) return DHMC_ERR_HIP;
    const int32_t zero = 0;
    if (hipMalloc(&dk.p, sizeof(double) * S) != hipSuccess || hipMalloc(&dk2.p, sizeof(double) * S) != hipSuccess ||
        hipMalloc(&di.p, sizeof(int32_t) * S) != hipSuccess || hipMalloc(&di2.p, sizeof(int32_t) * S) != hipSuccess ||
        hipMalloc(&dz.p, sizeof(double) * S) != hipSuccess || hipMalloc(&de.p, sizeof(double) * 2) != hipSuccess ||
        hipMalloc(&dr.p, sizeof(double)) != hipSuccess || hipMalloc(&dtmp.p, tmp_bytes ? tmp_bytes : 8) != hipSuccess ||
        hipMalloc(&dc0.p, sizeof(int32_t)) != hipSuccess)
        return DHMC_ERR_HIP;
    if (hipMemcpyAsync(dc0.p, &zero, sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess) return DHMC_ERR_HIP;
    const unsigned nb = (unsigned)((S + 255) / 256);
    std::vector<double> both((size_t)ncoords * 2);
    for (int j = 0; j < ncoords; ++j) {
        hipLaunchKernelGGL(ess_gather_kernel, dim3(nb), dim3(256), 0, s, draws, n, dim, coords[j], chains, N2, (double*)dk.p, (int32_t*)di.p);
        if (hipcub::DeviceRadixSort::SortPairs(dtmp.p, tmp_bytes, (const double*)dk.p, (double*)dk2.p, (const int32_t*)di.p,
                                               (int32_t*)di2.p, (int)S, 0, 64, s) != hipSuccess) return DHMC_ERR_HIP;
        for (int upper = 0; upper < 2; ++upper) {
            hipLaunchKernelGGL(ess_tail_indicator_kernel, dim3(nb), dim3(256), 0, s, (const double*)dk2.p, (const int32_t*)di2.p, S, upper, (double*)dz.p);
            if (int rc = ess_estimate(work, s, (const double*)dz.p, C2, half, 1, (const int32_t*)dc0.p, 1, (double*)de.p + upper, (double*)dr.p)) return rc;
        }
        if (hipGetLastError() != hipSuccess) return DHMC_ERR_HIP;
        if (hipMemcpyAsync(both.data() + 2 * j, de.p, 2 * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
        if (hipStreamSynchronize(s) != hipSuccess) return DHMC_ERR_HIP;     // (de is reused by the next coordinate)
    }
    for (int j = 0; j < ncoords; ++j) ess[j] = std::min(both[2 * j], both[2 * j + 1]);
    return DHMC_OK;
}

int dhmc_summarize_tree_statistics(int32_t device, void* stream, const double* pi, const double* acceptance_rate,
                                   const int64_t* term_left, const int64_t* term_right, const int32_t* depth,
                                   int64_t chains, int64_t n, int on_device, dhmc_tree_statistics_summary* summary,
                                   double* ebfmi) {
    if (!pi || !acceptance_rate || !term_left || !term_right || !depth || !summary || chains < 1 || n < 1) return DHMC_ERR_INVALID_ARGUMENT;
    const int64_t total = chains * n;
    if (total > 0x7fffffffll) return DHMC_ERR_UNSUPPORTED;
    if (hipSetDevice(device) != hipSuccess) return DHMC_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    DevBuf in[5], dsum, deb, dcnt, dsorted, dtmp, dout;
    const void* src[5] = {pi, acceptance_rate, term_left, term_right, depth};
    const size_t esz[5] = {8, 8, 8, 8, 4};
    const void* dev[5];
    for (int i = 0; i < 5; ++i) {
        dev[i] = src[i];
        if (!on_device) {
            if (hipMalloc(&in[i].p, esz[i] * total) != hipSuccess) return DHMC_ERR_HIP;
            if (hipMemcpyAsync(in[i].p, src[i], esz[i] * total, hipMemcpyHostToDevice, s) != hipSuccess) return DHMC_ERR_HIP;
            dev[i] = in[i].p;
        }
    }
    size_t tmp_bytes = 0;
    if (hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, (const double*)nullptr, (double*)nullptr, (int)total, 0, 64, s) != hipSuccess)
        return DHMC_ERR_HIP;
    const size_t ncnt = 3 + TS_DEPTH_BINS;
    if (hipMalloc(&dsum.p, sizeof(double) * chains) != hipSuccess || hipMalloc(&deb.p, sizeof(double) * chains) != hipSuccess ||
        hipMalloc(&dcnt.p, sizeof(unsigned long long) * ncnt) != hipSuccess || hipMalloc(&dsorted.p, sizeof(double) * total) != hipSuccess ||
        hipMalloc(&dtmp.p, tmp_bytes ? tmp_bytes : 8) != hipSuccess || hipMalloc(&dout.p, sizeof(double) * 6) != hipSuccess)
        return DHMC_ERR_HIP;
    if (hipMemsetAsync(dcnt.p, 0, sizeof(unsigned long long) * ncnt, s) != hipSuccess) return DHMC_ERR_HIP;
    hipLaunchKernelGGL(treestat_chain_kernel, dim3((unsigned)chains), dim3(WAVE), 0, s, (const double*)dev[0], (const double*)dev[1],
                       (const int64_t*)dev[2], (const int64_t*)dev[3], (const int32_t*)dev[4], n, (double*)deb.p, (double*)dsum.p,
                       (unsigned long long*)dcnt.p);
    if (hipcub::DeviceRadixSort::SortKeys(dtmp.p, tmp_bytes, (const double*)dev[1], (double*)dsorted.p, (int)total, 0, 64, s) != hipSuccess)
        return DHMC_ERR_HIP;
    hipLaunchKernelGGL(treestat_finish_kernel, dim3(1), dim3(WAVE), 0, s, (const double*)dsum.p, chains, total, (const double*)dsorted.p,
                       (double*)dout.p);
    if (hipGetLastError() != hipSuccess) return DHMC_ERR_HIP;
    double out6[6];
    unsigned long long cnt[3 + TS_DEPTH_BINS];
    if (hipMemcpyAsync(out6, dout.p, sizeof(out6), hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
    if (hipMemcpyAsync(cnt, dcnt.p, sizeof(cnt), hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
    if (ebfmi && hipMemcpyAsync(ebfmi, deb.p, sizeof(double) * chains, hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
    if (hipStreamSynchronize(s) != hipSuccess) return DHMC_ERR_HIP;
    summary->n = total;
    summary->a_mean = out6[0];
    for (int i = 0; i < 5; ++i) summary->a_quantiles[i] = out6[1 + i];
    summary->max_depth = (int64_t)cnt[0];
    summary->divergence = (int64_t)cnt[1];
    summary->turning = (int64_t)cnt[2];
    for (int d = 0; d < TS_DEPTH_BINS; ++d) summary->depth_counts[d] = (int64_t)cnt[3 + d];
    return DHMC_OK;
}

}  // extern "C"
