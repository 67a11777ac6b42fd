"""Reference ESS / R-hat estimators for the tests (numpy / scipy / torch): what `dhmc_ess_rhat` and `dhmc_ess_bulk`
(dynamichmc.jl_amd/csrc/ess_kernels.hpp) are checked against.  The reference's own tests call MCMCDiagnosticTools.ess_rhat
(test/sample-correctness_utilities.jl:40-43), which is not vendored: parity with that package is unpinned."""
import numpy as np


def ess_rhat(x):
    """Multi-chain bulk ESS and R-hat of one scalar, x [C][N] (Vehtari et al. 2021 estimator with
    Geyer's initial monotone sequence; no rank normalisation).  The reference's tests use
    MCMCDiagnosticTools.ess_rhat (test/sample-correctness_utilities.jl:40-43), not vendored."""
    x = np.asarray(x, np.float64)
    C, N = x.shape
    xm = x - x.mean(axis=1, keepdims=True)
    nfft = 1 << (2 * N - 1).bit_length()
    f = np.fft.rfft(xm, n=nfft, axis=1)
    acov = np.fft.irfft(f * np.conj(f), n=nfft, axis=1)[:, :N] / N
    W = (acov[:, 0] * N / (N - 1)).mean()
    B = x.mean(axis=1).var(ddof=1) * N if C > 1 else 0.0
    var_plus = W * (N - 1) / N + B / N
    rho = 1 - (W - acov.mean(axis=0)) / var_plus
    rho[0] = 1
    T = N // 2
    pair = rho[0:2 * T:2] + rho[1:2 * T:2]
    k = np.argmax(pair <= 0) if (pair <= 0).any() else len(pair)
    pair = np.minimum.accumulate(np.clip(pair[:k], 0, None))
    tau = max(-1 + 2 * pair.sum(), 1 / np.log10(C * N))
    return C * N / tau, float(np.sqrt(var_plus / W))


def ess_bulk(x):
    """Bulk ESS and rank-normalised split-R-hat of one scalar, x [C][N] (Vehtari et al. 2021; the default kind of
    MCMCDiagnosticTools.ess_rhat, which the reference's tests call): split every chain in two, replace the draws by the
    normal scores of their average ranks, then the estimator of ess_rhat.  Host flavour (scipy) of `dhmc_ess_bulk`."""
    from scipy.special import ndtri
    from scipy.stats import rankdata
    x = np.asarray(x, np.float64)
    C, N = x.shape
    h = N // 2
    xs = x[:, :2 * h].reshape(2 * C, h)
    r = rankdata(xs.ravel(), method="average").reshape(xs.shape)
    return ess_rhat(ndtri((r - 0.375) / (xs.size + 0.25)))



def ess_tail(x):
    """Tail ESS of one scalar, x [C][N] (Vehtari et al. 2021 §4.3; MCMCDiagnosticTools ess(kind = :tail)): split every chain in
    two, the indicators of the pooled 5 % and 95 % quantiles, the plain estimator on each, the smaller of the two.  Host
    flavour of `dhmc_ess_tail`."""
    x = np.asarray(x, np.float64)
    C, N = x.shape
    h = N // 2
    xs = x[:, :2 * h].reshape(2 * C, h)
    q05, q95 = np.quantile(xs.ravel(), [0.05, 0.95])
    return min(ess_rhat((xs <= q05).astype(np.float64))[0], ess_rhat((xs >= q95).astype(np.float64))[0])


def ess_bulk_torch(draws, coords=None):
    """The same estimator with torch FFTs — an independent cross-check of the HIP kernels (tests only)."""
    import torch
    C, N, D = draws.shape
    idx = torch.arange(D, device=draws.device) if coords is None else torch.as_tensor(coords, device=draws.device)
    x = draws[:, :, idx].permute(2, 0, 1).contiguous()                  # [k][C][N]
    xm = x - x.mean(dim=2, keepdim=True)
    nfft = 1 << (2 * N - 1).bit_length()
    f = torch.fft.rfft(xm, n=nfft, dim=2)
    acov = torch.fft.irfft(f * f.conj(), n=nfft, dim=2)[:, :, :N] / N
    W = (acov[:, :, 0] * N / (N - 1)).mean(dim=1)
    B = x.mean(dim=2).var(dim=1, unbiased=True) * N if C > 1 else torch.zeros_like(W)
    var_plus = W * (N - 1) / N + B / N
    rho = 1 - (W[:, None] - acov.mean(dim=1)) / var_plus[:, None]
    rho[:, 0] = 1
    T = N // 2
    pair = rho[:, 0:2 * T:2] + rho[:, 1:2 * T:2]
    keep = torch.cumprod((pair > 0).to(pair.dtype), dim=1)
    pair = torch.cummin(pair.clamp(min=0) * keep, dim=1).values * keep
    tau = torch.clamp(-1 + 2 * pair.sum(dim=1), min=1.0 / np.log10(C * N))
    return C * N / tau, torch.sqrt(var_plus / W)


