"""Post-hoc NUTS diagnostics over the SoA tree statistics, after DynamicHMC.Diagnostics (src/diagnostics.jl): host
flavours (numpy, per chain) and the device entry points that read statistics and draws where they lie in HBM
(`summarize_tree_statistics_device`, `ess_bulk_device`).  The reference estimators the ESS kernels are checked
against live with the tests (tests/ess_reference.py), not here."""
from collections import Counter

import numpy as np

ACCEPTANCE_QUANTILES = [0.05, 0.25, 0.5, 0.75, 0.95]      # diagnostics.jl:35


class InvalidTree:
    """trees.jl:180-200: positions relative to the starting node.  left == right: a divergent node; (1, 0): the sentinel
    REACHED_MAX_DEPTH; anything else: the tree was turning between the two positions.  Like the reference's struct, the
    constructor validates nothing: a subtree that turns while the trajectory is extended BACKWARD is recorded in build order,
    InvalidTree(i′, i₊) with i′ > i₊ (trees.jl:255 — e.g. (-1, -8); about 7 % of the transitions of the funnel golden), so
    left > right is ordinary data here although the reference's docstring reserves it for the sentinel.  The statistics arrays hold
    the two integers (`termination_left`, `termination_right`); `termination(tree_statistics, chain, i)` builds this view of one entry."""
    __slots__ = ("left", "right")

    def __init__(self, left, right=None):                  # InvalidTree(i) = InvalidTree(i, i) (trees.jl:185)
        self.left = int(left)
        self.right = int(left if right is None else right)

    def __eq__(self, other):
        return isinstance(other, InvalidTree) and (self.left, self.right) == (other.left, other.right)

    def __hash__(self):
        return hash((self.left, self.right))

    def __repr__(self):                                    # Base.show (trees.jl:189-199)
        if is_divergent(self):
            return f"divergence at position {self.left}"
        if self == REACHED_MAX_DEPTH:
            return "reached maximum depth without divergence or turning"
        return f"turning at positions {self.left}:{self.right}"


def is_divergent(invalid_tree):                            # trees.jl:187
    return invalid_tree.left == invalid_tree.right


REACHED_MAX_DEPTH = InvalidTree(1, 0)                      # trees.jl:202


def termination(tree_statistics, chain, i):
    """`tree_statistics[i].termination` of one chain (NUTS.jl:208-221) as an InvalidTree."""
    return InvalidTree(int(tree_statistics.termination_left[chain, i]), int(tree_statistics.termination_right[chain, i]))


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
    """diagnostics.jl:87-95: element d is the number of trees of depth d (0 … the deepest seen), as the reference's Vector{Int}
    and as the device flavour returns it."""
    return np.bincount(np.asarray(tree_statistics.depth).ravel().astype(np.int64)).tolist()


def summarize_tree_statistics(tree_statistics):
    """diagnostics.jl:100-106."""
    a = np.asarray(tree_statistics.acceptance_rate)
    return dict(N=a.size, a_mean=float(a.mean()), a_quantiles=np.quantile(a, ACCEPTANCE_QUANTILES).tolist(),
                termination_counts=count_terminations(tree_statistics), depth_counts=count_depths(tree_statistics))


class TreeStatisticsSummary(dict):
    """summarize_tree_statistics' result (diagnostics.jl:47-58): N, a_mean, a_quantiles, termination_counts, depth_counts."""


def summarize_tree_statistics_device(pi, acceptance_rate, term_left, term_right, depth, device=None, stream=None):
    """EBFMI per chain (diagnostics.jl:29-32) and summarize_tree_statistics (:100-106: mean and ACCEPTANCE_QUANTILES of the
    acceptance rates, count_terminations :65-82, count_depths :87-95; chains pooled) computed by `dhmc_summarize_tree_statistics`
    (csrc/treestat_kernels.hpp).  The five [C][N] arrays are the SoA tree statistics of a run: CUDA torch tensors are read
    where they lie in HBM, numpy arrays are staged by the library.  Returns (summary, ebfmi [C])."""
    import ctypes as C_
    from . import _abi as abi
    arrs = [pi, acceptance_rate, term_left, term_right, depth]
    on_dev = all(hasattr(a, "is_cuda") and a.is_cuda for a in arrs)
    if on_dev:
        import torch
        want = [torch.float64, torch.float64, torch.int64, torch.int64, torch.int32]
        if not all(a.is_contiguous() and a.dtype == w for a, w in zip(arrs, want)):
            raise ValueError("tree statistics must be contiguous [C][N] CUDA tensors (float64, float64, int64, int64, int32)")
        Cn, N = pi.shape
        ptrs = [C_.c_void_p(a.data_ptr()) for a in arrs]
        dev = pi.device.index or 0
        strm = torch.cuda.current_stream(pi.device).cuda_stream if stream is None else stream
    else:
        want = [np.float64, np.float64, np.int64, np.int64, np.int32]
        arrs = [np.ascontiguousarray(np.asarray(a), w) for a, w in zip(arrs, want)]
        Cn, N = arrs[0].shape
        ptrs = [C_.c_void_p(a.ctypes.data) for a in arrs]
        dev, strm = (0 if device is None else device), stream
    out = abi.TreeStatisticsSummaryABI()
    ebfmi = np.zeros(Cn)
    rc = abi.lib().dhmc_summarize_tree_statistics(C_.c_int32(dev), C_.c_void_p(strm), *ptrs, C_.c_int64(Cn), C_.c_int64(N),
                                                  C_.c_int(1 if on_dev else 0), C_.byref(out), C_.c_void_p(ebfmi.ctypes.data))
    if rc != abi.OK:
        raise RuntimeError(f"dhmc_summarize_tree_statistics: {abi.ERROR_NAMES.get(rc, rc)}")
    dc = list(out.depth_counts)
    last = max((i for i, v in enumerate(dc) if v), default=-1)
    return TreeStatisticsSummary(N=int(out.n), a_mean=float(out.a_mean), a_quantiles=list(out.a_quantiles),
                                 termination_counts=dict(max_depth=int(out.max_depth), divergence=int(out.divergence),
                                                         turning=int(out.turning)),
                                 depth_counts=dc[:last + 1]), ebfmi


def ess_bulk_device(draws, coords=None, kind="bulk"):
    """ESS and R-hat per coordinate for draws [C][N][D] held in HBM (a CUDA torch tensor), computed where they lie by
    the library's HIP kernels (csrc/ess_kernels.hpp), so ESS/s can be reported without shipping the draws to the host
    (SURVEY.md §8 f-3).  kind = "bulk": rank-normalised split-chain bulk ESS (`dhmc_ess_bulk`, the estimator of
    tests/ess_reference.py ess_bulk); "plain": no split, no rank normalisation (`dhmc_ess_rhat`); "tail": the smaller of the
    ESS of the 5 % and 95 % quantile indicators over the split chains (`dhmc_ess_tail`; R-hat is NaN for this kind).
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
    if kind == "tail":
        rc = abi.lib().dhmc_ess_tail(C_.c_int32(draws.device.index or 0), C_.c_void_p(torch.cuda.current_stream(draws.device).cuda_stream),
                                     C_.c_void_p(draws.data_ptr()), C_.c_int64(Cn), C_.c_int64(N), C_.c_int64(D),
                                     C_.c_void_p(idx.ctypes.data), C_.c_int32(idx.size), C_.c_void_p(ess.ctypes.data))
        if rc != abi.OK:
            raise RuntimeError(f"dhmc_ess_tail: {abi.ERROR_NAMES.get(rc, rc)}")
        return ess, np.full(idx.size, np.nan)
    fn = abi.lib().dhmc_ess_bulk if kind == "bulk" else abi.lib().dhmc_ess_rhat
    rc = fn(C_.c_int32(draws.device.index or 0), C_.c_void_p(torch.cuda.current_stream(draws.device).cuda_stream),
                                 C_.c_void_p(draws.data_ptr()), C_.c_int64(Cn), C_.c_int64(N), C_.c_int64(D),
                                 C_.c_void_p(idx.ctypes.data), C_.c_int32(idx.size), C_.c_void_p(ess.ctypes.data),
                                 C_.c_void_p(rhat.ctypes.data))
    if rc != abi.OK:
        raise RuntimeError(f"dhmc_ess_{'bulk' if kind == 'bulk' else 'rhat'}: {abi.ERROR_NAMES.get(rc, rc)}")
    return ess, rhat


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
    if l.family == abi.TARGET_EXTERNAL:                         # the caller's batched model: the probes step all chains around it
        ctx.set_logdensity_callback(l.callback())
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
