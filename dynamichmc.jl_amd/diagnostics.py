"""Post-hoc NUTS diagnostics over the SoA tree statistics, after DynamicHMC.Diagnostics
(src/diagnostics.jl) — cold path, host side, per chain."""
from collections import Counter

import numpy as np

ACCEPTANCE_QUANTILES = [0.05, 0.25, 0.5, 0.75, 0.95]      # diagnostics.jl:35


def EBFMI(tree_statistics):
    """Energy Bayesian fraction of missing information (diagnostics.jl:29): mean(abs2, diff(π)) / var(π), per chain."""
    pi = np.asarray(tree_statistics.pi)
    return (np.diff(pi, axis=1) ** 2).mean(axis=1) / pi.var(axis=1, ddof=1)


def count_terminations(tree_statistics):
    """diagnostics.jl:65-82: counts of divergence / max depth / turning, pooled over chains."""
    l, r = np.asarray(tree_statistics.termination_left), np.asarray(tree_statistics.termination_right)
    maxd = (l == 1) & (r == 0)
    div = (l == r)
    return dict(max_depth=int(maxd.sum()), divergence=int(div.sum()), turning=int((~maxd & ~div).sum()))


def count_depths(tree_statistics):
    """diagnostics.jl:87-95."""
    return dict(sorted(Counter(np.asarray(tree_statistics.depth).ravel().tolist()).items()))


def summarize_tree_statistics(tree_statistics):
    """diagnostics.jl:100-106."""
    a = np.asarray(tree_statistics.acceptance_rate)
    return dict(N=a.size, a_mean=float(a.mean()), a_quantiles=np.quantile(a, ACCEPTANCE_QUANTILES).tolist(),
                termination_counts=count_terminations(tree_statistics), depth_counts=count_depths(tree_statistics))


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


def ess_bulk_device(draws, coords=None, kind="bulk"):
    """ESS and R-hat per coordinate for draws [C][N][D] held in HBM (a CUDA torch tensor), computed where they lie by
    the library's HIP kernels (csrc/ess_kernels.hpp), so ESS/s can be reported without shipping the draws to the host
    (SURVEY.md §8 f-3).  kind = "bulk": rank-normalised split-chain bulk ESS (`dhmc_ess_bulk`, the estimator of
    ess_bulk above); "plain": no split, no rank normalisation (`dhmc_ess_rhat`, ess_rhat above).
    Returns numpy arrays (ess [k], rhat [k])."""
    import ctypes as C_
    from . import _abi as abi
    Cn, N, D = draws.shape
    if not (draws.is_cuda and draws.is_contiguous() and str(draws.dtype) == "torch.float64"):
        raise ValueError("draws must be a contiguous float64 CUDA tensor [C][N][D]")
    idx = np.arange(D, dtype=np.int32) if coords is None else np.ascontiguousarray(
        coords.cpu().numpy() if hasattr(coords, "cpu") else coords, np.int32)
    ess = np.zeros(idx.size); rhat = np.zeros(idx.size)
    import torch
    fn = abi.lib().dhmc_ess_bulk if kind == "bulk" else abi.lib().dhmc_ess_rhat
    rc = fn(C_.c_int32(draws.device.index or 0), C_.c_void_p(torch.cuda.current_stream(draws.device).cuda_stream),
                                 C_.c_void_p(draws.data_ptr()), C_.c_int64(Cn), C_.c_int64(N), C_.c_int64(D),
                                 C_.c_void_p(idx.ctypes.data), C_.c_int32(idx.size), C_.c_void_p(ess.ctypes.data),
                                 C_.c_void_p(rhat.ctypes.data))
    if rc != abi.OK:
        raise RuntimeError(f"dhmc_ess_{'bulk' if kind == 'bulk' else 'rhat'}: {abi.ERROR_NAMES.get(rc, rc)}")
    return ess, rhat


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


# ---- diagnostics that call the hot path (device) -------------------------------------------------------------

def _probe_context(l, q, kappa, rng, device):
    from . import _abi as abi
    from .api import GaussianKineticEnergy, _argcheck, _as_rng
    from .context import DeviceContext
    q = np.asarray(q, np.float64)
    q2 = q[None, :] if q.ndim == 1 else q                       # [C][D]: one start point per chain
    kappa = GaussianKineticEnergy(l.dimension()) if kappa is None else kappa   # diagnostics.jl:146,216
    _argcheck(kappa.size() == l.dimension(), "dimension(ℓ) == size(κ, 1)")    # hamiltonian.jl:147
    rng = _as_rng(0x23EF614D if rng is None else rng)
    ctx = DeviceContext(l.dimension(), q2.shape[0], target=l.family, target_params=l.params(), seed=rng.seed,
                        chain_offset=rng.chain_offset, device=device,
                        metric=abi.METRIC_DENSE if kappa.dense else abi.METRIC_DIAG)
    ctx.init(q2, allow_failure=True)                            # evaluate_ℓ(ℓ, q), non-strict (diagnostics.jl:148,219)
    if (ctx.status() & abi.ST_NONFINITE_POSITION).any():        # hamiltonian.jl:203 throws also when non-strict
        ctx._raise(abi.ERR_CHAIN_FAILURE, "evaluate_ℓ")
    if kappa.dense:
        ctx.set_metric_dense(kappa.Minv)
    else:
        ctx.set_metric_diag(kappa.Minv)
    return ctx, q.ndim == 1


def explore_log_acceptance_ratios(l, q, log2eps, *, rng=None, kappa=None, N=20, ps=None, device=0):
    """diagnostics.jl:144-152: the uncapped log acceptance ratios of one leapfrog step from `q` for step sizes
    2.0 .^ log2ϵs and `N` random momenta (or the given `ps` [N][D]); returns [len(log2eps), N] as the reference
    (with a leading chain axis when `q` is [C][D])."""
    ctx, single = _probe_context(l, q, kappa, rng, device)
    try:
        eps = np.power(2.0, np.asarray(log2eps, np.float64))
        out = ctx.explore_log_acceptance_ratios(eps, n_momenta=N, ps=ps)       # [C][N][n_eps]
    finally:
        ctx.close()
    out = np.swapaxes(out, 1, 2)
    return out[0] if single else out


def leapfrog_trajectory(l, q, eps, positions, *, rng=None, kappa=None, p=None, device=0):
    """diagnostics.jl:214-227: the leapfrog trajectory through `positions` (a range containing 0) relative to
    `q`, tracked in each direction up to the first non-finite log density.  Returns a list of dicts
    (z = dict(q, lq, p), position, Δ) as the reference's vector of NamedTuples (a list per chain for [C][D] q)."""
    positions = list(positions)
    A, B = positions[0], positions[-1]
    if not (positions == list(range(A, B + 1)) and A <= 0 <= B):
        raise ValueError("ArgumentError: Positions has to contain `0`.")     # diagnostics.jl:218
    ctx, single = _probe_context(l, q, kappa, rng, device)
    try:
        r = ctx.leapfrog_trajectory(eps, A, B, p=p)
    finally:
        ctx.close()
    res = []
    for c in range(r["delta"].shape[0]):
        lo, hi = r["range"][c]
        res.append([dict(z=dict(q=r["q"][c, i - A], lq=float(r["logdensity"][c, i - A]), p=r["p"][c, i - A]),
                         position=i, Δ=float(r["delta"][c, i - A])) for i in range(lo, hi + 1)])
    return res[0] if single else res
