"""Host-side mirror of DynamicHMC.jl's calling surface for the many-chain HIP path.

Same names, keyword arguments, defaults and result fields as the reference's `src/mcmc.jl`,
`src/NUTS.jl`, `src/stepsize.jl` and `src/hamiltonian.jl`, with one addition: every array has
a leading chain dimension (the reference runs one chain per call and leaves multi-chain runs
to the caller, docs/src/worked_example.md:95-104).  All sampling work happens inside
libdhmc_amd.so; this module only orchestrates stages, exactly as the reference's `_warmup`
fold (mcmc.jl:450-457) does.

The reference's host language is Julia, which is not installed in this environment, so the
tested host wrapper is this Python twin; INTEGRATION.md shows the equivalent `ccall` shim.
"""
import math
import time
from dataclasses import dataclass, field
from typing import Any, Optional

import numpy as np

from . import _abi as abi
from .context import DeviceContext, DynamicHMCError

__all__ = [
    "NUTS", "InitialStepsizeSearch", "DualAveraging", "FixedStepsize", "TuningNUTS",
    "default_warmup_stages", "fixed_stepsize_warmup_stages", "GaussianKineticEnergy",
    "mcmc_with_warmup", "mcmc_keep_warmup", "mcmc_steps", "mcmc_next_step",
    "stack_posterior_matrices", "pool_posterior_matrices", "TreeStatisticsNUTS",
    "StandardNormal", "DiagNormal", "TridiagNormal", "MvNormal", "Funnel", "LogisticRegression", "AlwaysDivergent", "TorchLogDensity", "DeviceFunctorLogDensity",
    "NoProgressReport", "LogProgressReport", "ProgressMeterReport", "report", "make_mcmc_reporter", "default_reporter", "DynamicHMCError",
    "Diagonal", "Symmetric", "PhiloxRNG", "WarmupState", "EvaluatedLogDensity", "PhasePoint",
]

Diagonal = "Diagonal"
Symmetric = "Symmetric"
MAX_DIRECTIONS_DEPTH = 32          # trees.jl:10
DEFAULT_MAX_TREE_DEPTH = 10        # NUTS.jl:166


def _argcheck(cond, msg):
    if not cond:
        raise ValueError(f"ArgumentError: {msg}")   # the reference's @argcheck


# ---- algorithm / adaptation parameter objects ------------------------------------------------

@dataclass(frozen=True)
class NUTS:
    """NUTS(; max_depth, min_Δ, turn_statistic_configuration) — NUTS.jl:178-195."""
    max_depth: int = DEFAULT_MAX_TREE_DEPTH
    min_delta: float = -1000.0
    turn_statistic_configuration: str = "generalized"

    def __post_init__(self):
        _argcheck(0 < self.max_depth <= MAX_DIRECTIONS_DEPTH, "0 < max_depth ≤ MAX_DIRECTIONS_DEPTH")
        _argcheck(self.min_delta < 0, "min_Δ < 0")
        _argcheck(self.turn_statistic_configuration == "generalized", "only Val(:generalized) is supported")


@dataclass(frozen=True)
class InitialStepsizeSearch:
    """stepsize.jl:23-36."""
    initial_eps: float = 0.1
    log_threshold: float = math.log(0.8)
    maxiter_crossing: int = 400

    def __post_init__(self):
        _argcheck(math.isfinite(self.log_threshold) and self.log_threshold < 0, "isfinite(log_threshold) && log_threshold < 0")
        _argcheck(math.isfinite(self.initial_eps) and 0 < self.initial_eps, "isfinite(initial_ϵ) && 0 < initial_ϵ")
        _argcheck(self.maxiter_crossing >= 50, "maxiter_crossing ≥ 50")


@dataclass(frozen=True)
class DualAveraging:
    """stepsize.jl:98-118."""
    delta: float = 0.8
    gamma: float = 0.05
    kappa: float = 0.75
    t0: int = 10

    def __post_init__(self):
        _argcheck(0 < self.delta < 1, "0 < δ < 1")
        _argcheck(self.gamma > 0, "γ > 0")
        _argcheck(0.5 < self.kappa <= 1, "0.5 < κ ≤ 1")
        _argcheck(self.t0 >= 0, "t₀ ≥ 0")


@dataclass(frozen=True)
class FixedStepsize:
    """stepsize.jl:181-189."""


@dataclass(frozen=True)
class TuningNUTS:
    """TuningNUTS{M}(N, stepsize_adaptation, λ = 5/N) — mcmc.jl:178-195.  M is None, Diagonal or Symmetric."""
    N: int
    stepsize_adaptation: Any = field(default_factory=DualAveraging)
    M: Optional[str] = None
    lam: Optional[float] = None

    def __post_init__(self):
        _argcheck(self.M in (None, Diagonal, Symmetric), "M <: Union{Nothing,Diagonal,Symmetric}")
        _argcheck(self.N >= 20, "N ≥ 20")
        if self.lam is None:
            object.__setattr__(self, "lam", 5.0 / self.N)
        _argcheck(self.lam >= 0, "λ ≥ 0")


def _doubling_warmup_stages(M, stepsize_adaptation, middle_steps, doubling_stages):   # mcmc.jl:400-403
    return tuple(TuningNUTS(middle_steps * 2 ** i, stepsize_adaptation, M) for i in range(doubling_stages))


def default_warmup_stages(stepsize_search=InitialStepsizeSearch(), M=Diagonal, stepsize_adaptation=DualAveraging(),
                          init_steps=75, middle_steps=25, doubling_stages=5, terminating_steps=50):
    """mcmc.jl:415-425: 75 + 25·(1+2+4+8+16) + 50 = 900 transitions by default."""
    _argcheck(M in (Diagonal, Symmetric), "M <: Union{Diagonal,Symmetric}")
    return (stepsize_search, TuningNUTS(init_steps, stepsize_adaptation, None),
            *_doubling_warmup_stages(M, stepsize_adaptation, middle_steps, doubling_stages),
            TuningNUTS(terminating_steps, stepsize_adaptation, None))


def fixed_stepsize_warmup_stages(M=Diagonal, middle_steps=25, doubling_stages=5):
    """mcmc.jl:436-440."""
    _argcheck(M in (Diagonal, Symmetric), "M <: Union{Diagonal,Symmetric}")
    return _doubling_warmup_stages(M, FixedStepsize(), middle_steps, doubling_stages)


# ---- kinetic energy, phase-space containers ----------------------------------------------------

class GaussianKineticEnergy:
    """hamiltonian.jl:56-87.  `GaussianKineticEnergy(N, m_inv=1.0)` (:87) or
    `GaussianKineticEnergy(diag)` with the diagonal of M⁻¹ as [D] (shared) or [C][D] (:80)."""

    def __init__(self, Minv, m_inv=1.0, dense=False):
        if isinstance(Minv, (int, np.integer)):
            Minv = np.full(int(Minv), float(m_inv))
        Minv = np.asarray(Minv, np.float64)
        self.dense = bool(dense)
        if self.dense:                           # GaussianKineticEnergy(M⁻¹::AbstractMatrix) (:73), shared by all chains
            _argcheck(Minv.ndim == 2 and Minv.shape[0] == Minv.shape[1], "checksquare(M⁻¹)")
            S = np.triu(Minv) + np.triu(Minv, 1).T      # Symmetric(M⁻¹)
            try:
                self.W = np.linalg.cholesky(np.linalg.inv(S))
            except np.linalg.LinAlgError:
                raise ValueError("ArgumentError: M⁻¹ is not positive definite")
            self.Minv = S
            return
        _argcheck(Minv.ndim in (1, 2), "diag(M⁻¹) is [D] (shared) or [C][D] (per chain)")
        _argcheck(np.all(Minv > 0), "diagonal of M⁻¹ must be positive")
        self.Minv = Minv
        self.W = np.sqrt(1.0 / Minv)          # W such that W*W' = M (:80)

    def size(self):
        return self.Minv.shape[-1]

    def __repr__(self):                           # hamiltonian.jl:89-91
        if self.dense:
            d = np.diagonal(self.Minv, axis1=-2, axis2=-1)       # [D], or [C][D] for a per-chain κ
            return f"Gaussian kinetic energy (Symmetric), √diag(M⁻¹): {np.sqrt(d)}"
        return f"Gaussian kinetic energy (Diagonal), √diag(M⁻¹): {np.sqrt(self.Minv)}"


@dataclass
class EvaluatedLogDensity:
    """hamiltonian.jl:165-186, for C chains: q [C][D], ℓq [C], ∇ℓq [C][D]."""
    q: np.ndarray
    lq: np.ndarray
    grad: np.ndarray


@dataclass
class PhasePoint:
    """hamiltonian.jl:225-234: a point in phase space, z = (Q, p) — for C chains p is [C][D]."""
    Q: EvaluatedLogDensity
    p: np.ndarray


@dataclass
class WarmupState:
    """mcmc.jl:72-79."""
    Q: EvaluatedLogDensity
    kappa: GaussianKineticEnergy
    eps: Optional[np.ndarray]

    def __repr__(self):
        e = "unspecified" if self.eps is None else f"≈ {np.round(np.median(self.eps), 3)} (median over chains)"
        return f"adapted sampling parameters: stepsize (ϵ) {e}, {self.kappa!r}"


@dataclass
class TreeStatisticsNUTS:
    """NUTS.jl:208-221 as a structure of arrays, each [C][N]; `termination` is (left, right)
    (trees.jl:180-202: left == right divergence, left < right turning, (1, 0) reached max depth)."""
    pi: np.ndarray
    depth: np.ndarray
    termination_left: np.ndarray
    termination_right: np.ndarray
    acceptance_rate: np.ndarray
    steps: np.ndarray
    directions: np.ndarray

    @property
    def is_divergent(self):
        return self.termination_left == self.termination_right


# ---- log densities (the LogDensityProblems side of the boundary) -------------------------------

class _Target:
    family = None

    def capabilities(self):      # LogDensityProblems.capabilities ≥ LogDensityOrder(1) (hamiltonian.jl:146)
        return 1

    def dimension(self):         # LogDensityProblems.dimension (hamiltonian.jl:147)
        return self.D

    def params(self):
        return None


class StandardNormal(_Target):
    """ℓ(q) = -½ Σ q², ∇ℓ = -q."""
    family = abi.TARGET_STD_NORMAL

    def __init__(self, D):
        self.D = int(D)


class DiagNormal(_Target):
    """ℓ(q) = -½ Σ prec_i (q_i - μ_i)²  (the reference tests' multivariate_normal(μ, I·v), test/utilities.jl:67)."""
    family = abi.TARGET_DIAG_NORMAL

    def __init__(self, mu, prec):
        self.mu = np.asarray(mu, np.float64); self.prec = np.broadcast_to(np.asarray(prec, np.float64), self.mu.shape).copy()
        self.D = self.mu.size

    def params(self):
        return np.concatenate([self.mu, self.prec])


class TridiagNormal(_Target):
    """ℓ(q) = -½ q'Pq with symmetric tridiagonal precision P (diag [D], off [D-1])."""
    family = abi.TARGET_TRIDIAG_NORMAL

    def __init__(self, diag, off):
        self.diag = np.asarray(diag, np.float64); self.D = self.diag.size
        self.off = np.zeros(self.D); self.off[:self.D - 1] = np.asarray(off, np.float64)[:self.D - 1]

    def params(self):
        return np.concatenate([self.diag, self.off])


class MvNormal(_Target):
    """ℓ(q) = -½ (q-μ)'Σ⁻¹(q-μ) with a full covariance Σ — the reference tests' multivariate_normal(μ, L)
    (test/utilities.jl:64).  Intended for small correlated targets (the precision is streamed per chain)."""
    family = abi.TARGET_DENSE_NORMAL

    def __init__(self, mu, Sigma):
        self.mu = np.asarray(mu, np.float64); self.Sigma = np.asarray(Sigma, np.float64)
        self.D = self.mu.size
        _argcheck(self.Sigma.shape == (self.D, self.D), "Σ is D×D")
        self.P = np.linalg.inv(self.Sigma); self.P = (self.P + self.P.T) / 2

    def params(self):
        return np.concatenate([self.mu, self.P.ravel()])


class Funnel(_Target):
    """Neal's funnel: v = q₀ ~ N(0, 3²), q_i | v ~ N(0, eᵛ)."""
    family = abi.TARGET_FUNNEL

    def __init__(self, D):
        self.D = int(D)


class LogisticRegression(_Target):
    """Bernoulli-logit regression, β ~ N(0, I):  ℓ(β) = Σ_n [y_n x_n·β - log(1 + e^{x_n·β})] - ½ β·β.
    X [N][D] and y [N] are copied to the GPU once and shared by all chains."""
    family = abi.TARGET_LOGISTIC

    def __init__(self, X, y):
        self.X = np.ascontiguousarray(X, np.float64); self.y = np.ascontiguousarray(y, np.float64)
        _argcheck(self.X.ndim == 2 and self.y.shape == (self.X.shape[0],), "X is [N][D], y is [N]")
        self.D = self.X.shape[1]

    def params(self):
        return np.concatenate([np.array([self.X.shape[0]], np.int64).view(np.float64), self.X.ravel(), self.y])


class TorchLogDensity(_Target):
    """The user's own model (the reference accepts any LogDensityProblems object, hamiltonian.jl:146-147,204): a batched
    PyTorch function evaluated on the GPU for all chains at once, once per leapfrog round.  Either
    `logdensity_and_gradient(q) -> (lq [C], grad [C][D])`, or just `logdensity(q) -> lq [C]` (any differentiable torch
    code; the gradient then comes from autograd).  q is a float64 CUDA tensor [C][D].  Diagonal or (shared) dense
    metric, D <= 4096."""
    family = abi.TARGET_EXTERNAL

    def __init__(self, dimension, logdensity=None, logdensity_and_gradient=None):
        _argcheck((logdensity is None) != (logdensity_and_gradient is None), "give logdensity or logdensity_and_gradient")
        self.D = int(dimension)
        self._f, self._fg = logdensity, logdensity_and_gradient

    def callback(self):
        if self._fg is not None:
            return self._fg
        import torch
        f = self._f

        def fg(q):
            with torch.enable_grad():
                x = q.detach().clone().requires_grad_(True)
                lq = f(x)
                (g,) = torch.autograd.grad(lq.sum(), x)
            return lq.detach(), g
        return fg


class DeviceFunctorLogDensity(_Target):
    """The user's own model as a DEVICE FUNCTOR: HIP C++ source defining `struct <name>` in namespace dhmc with the interface
    of the built-in families (include/dhmc.h dhmc_register_target_source), compiled at run time into the
    library's own per-draw / initialisation / step-size-search kernels — no host round trip per leapfrog, as `north_star`
    asks ("the user ∇log π is supplied as a device function").  `params`: doubles handed to the functor's constructor.
    Diagonal or shared dense metric; up to 1024 coordinates inside the per-draw kernels, up to 4096 evaluated for all chains between the
    streaming round engine's kernels (only eval() is compiled then)."""

    def __init__(self, dimension, source, name, params=None):
        self.D = int(dimension)
        self._params = None if params is None else np.ascontiguousarray(params, np.float64).ravel()
        import ctypes
        h = ctypes.c_int32(-1)
        rc = abi.lib().dhmc_register_target_source(source.encode(), name.encode(), ctypes.byref(h))
        _argcheck(rc == abi.OK, "dhmc_register_target_source")
        self.family = abi.TARGET_USER_BASE + h.value

    def params(self):
        return self._params

    @staticmethod
    def check(dimension, source, name, metric=abi.METRIC_DIAG):
        """Compile only (no GPU needed) the kernels of `metric`; returns (ok, compiler log)."""
        import ctypes
        log = ctypes.create_string_buffer(1 << 16)
        rc = abi.lib().dhmc_check_target_source(source.encode(), name.encode(), ctypes.c_int32(dimension), ctypes.c_int32(metric), log,
                                                ctypes.c_uint64(len(log)))
        return rc == abi.OK, log.value.decode(errors="replace")


class AlwaysDivergent(_Target):
    """The reference's AlwaysDivergentTest (test/test_NUTS.jl:58-73)."""
    family = abi.TARGET_ALWAYS_DIVERGENT

    def __init__(self, K):
        self.D = int(K)


# ---- rng, reporters -----------------------------------------------------------------------------

@dataclass
class PhiloxRNG:
    """Stands in for the reference's `rng::AbstractRNG` argument: the seed of the ABI's
    counter-based stream (include/dhmc.h); chain c uses key (seed, chain_offset + c)."""
    seed: int = 0x23EF614D
    chain_offset: int = 0


REPORT_SIGDIGITS = 3       # reporting.jl:8


def _sig(x, digits=REPORT_SIGDIGITS):
    x = float(x)
    return x if x == 0 or not np.isfinite(x) else float(f"{x:.{digits}g}")


class NoProgressReport:
    """reporting.jl:14-46: nothing is reported; its MCMC reporter is itself."""

    def report(self, message_or_step=None, **meta):
        pass

    def make_mcmc_reporter(self, total_steps, currently_warmup=False, **meta):
        return self

    step_chunk = 0            # transitions between step reports (0: the stage runs as one call)


class LogProgressReport:
    """reporting.jl:62-76: progress as log lines (`@info` there, `printer` here).  `report(message; meta...)` prints the message;
    `make_mcmc_reporter(total_steps)` returns a LogMCMCReport whose `report(step)` prints "MCMC progress" when `step_interval`
    steps or `time_interval_s` seconds have passed since the last report (a *step* is a NUTS transition of ALL chains here)."""

    def __init__(self, chain_id=None, step_interval=100, time_interval_s=1000.0, printer=print):
        self.chain_id, self.step_interval, self.time_interval_s, self.printer = chain_id, int(step_interval), float(time_interval_s), printer

    def _emit(self, message, meta):
        if self.chain_id is not None:                                         # _log_meta (reporting.jl:83-85)
            meta = dict(chain_id=self.chain_id, **meta)
        extra = ", ".join(f"{k} = {v}" for k, v in meta.items())
        self.printer(f"[ Info: {message}" + (f"  {extra}" if extra else ""))

    def report(self, message, **meta):                                        # reporting.jl:87-90
        self._emit(str(message), meta)

    def make_mcmc_reporter(self, total_steps, currently_warmup=False, **meta):   # reporting.jl:115-118
        self._emit("Starting MCMC", dict(total_steps=int(total_steps), **meta))
        return LogMCMCReport(self, int(total_steps))

    step_chunk = 0


class LogMCMCReport:
    """reporting.jl:100-137: the state of the last emitted progress line for a stage with a known number of steps."""

    def __init__(self, log_progress_report, total_steps):
        self.log_progress_report, self.total_steps = log_progress_report, total_steps
        self.last_reported_step, self.last_reported_time = -1, time.perf_counter()

    @property
    def step_chunk(self):
        return max(1, self.log_progress_report.step_interval)

    def report(self, message_or_step, **meta):
        r = self.log_progress_report
        if isinstance(message_or_step, str):                                  # reporting.jl:110-113
            r._emit(message_or_step, meta)
            return
        step = int(message_or_step)
        _argcheck(1 <= step <= self.total_steps, "1 ≤ step ≤ total_steps")    # reporting.jl:123
        d_steps = step - self.last_reported_step
        now = time.perf_counter()
        d_time = now - self.last_reported_time
        if self.last_reported_step < 0 or d_steps >= r.step_interval or d_time >= r.time_interval_s:
            sps = d_time / d_steps
            progress = dict(step=step, seconds_per_step=_sig(sps, 2), estimated_seconds_left=_sig((self.total_steps - step) * sps, 2))
            r._emit("MCMC progress", dict(progress, **meta))
            self.last_reported_step, self.last_reported_time = step, now


class ProgressMeterReport:
    """reporting.jl:140-175: a progress bar per stage (ProgressMeter.jl there; tqdm here when it is installed, else a plain
    counter line per update); messages are not shown."""

    def __init__(self, updates=100, stream=None):
        self.updates, self.stream = max(1, int(updates)), stream

    def report(self, message_or_step=None, **meta):
        pass

    def make_mcmc_reporter(self, total_steps, currently_warmup=False, **meta):
        return ProgressMeterReportMCMC(bool(currently_warmup), int(total_steps), self.updates, self.stream)

    step_chunk = 0


class ProgressMeterReportMCMC:
    def __init__(self, currently_warmup, total_steps, updates, stream):
        self.currently_warmup, self.total_steps, self.last = currently_warmup, total_steps, 0
        self.step_chunk = max(1, total_steps // updates)
        desc = "Warmup: " if currently_warmup else "MCMC: "
        try:
            from tqdm import tqdm
            self.bar = tqdm(total=total_steps, desc=desc, file=stream, leave=True)
        except ImportError:                                                   # pragma: no cover
            self.bar, self.desc, self.stream = None, desc, stream

    def report(self, message_or_step=None, **meta):
        if isinstance(message_or_step, str) or message_or_step is None:
            return
        step = int(message_or_step)
        if self.bar is not None:
            self.bar.update(step - self.last)                                 # ProgressMeter.next! once per step there
            if step >= self.total_steps:
                self.bar.close()
        else:                                                                 # pragma: no cover
            print(f"{self.desc}{step}/{self.total_steps}", file=self.stream)
        self.last = step


def report(reporter, message_or_step, **meta):
    """reporting.jl:32: `report(reporter, message::AbstractString; meta...)` / `report(reporter, step::Integer; meta...)`."""
    return reporter.report(message_or_step, **meta)


def make_mcmc_reporter(reporter, total_steps, currently_warmup=False, **meta):
    """reporting.jl:49: a reporter for a stage with a known number of steps."""
    return reporter.make_mcmc_reporter(total_steps, currently_warmup=currently_warmup, **meta)


def default_reporter():
    """reporting.jl:184-190: log when interactive, otherwise none."""
    import sys
    return LogProgressReport() if hasattr(sys, "ps1") else NoProgressReport()


# ---- the driver ---------------------------------------------------------------------------------

@dataclass
class SamplingLogDensity:
    """mcmc.jl:41-58 plus the device context that holds the chains."""
    rng: PhiloxRNG
    l: _Target
    algorithm: NUTS
    reporter: Any
    ctx: DeviceContext
    on_device: bool = False      # results stay in HBM as torch CUDA tensors instead of numpy arrays
    keep_warmup: bool = True     # False: warmup-stage draws are consumed on the device and never copied out


def _as_rng(rng):
    if isinstance(rng, PhiloxRNG):
        return rng
    if isinstance(rng, (int, np.integer)):
        return PhiloxRNG(int(rng))
    raise TypeError("rng must be a PhiloxRNG or an integer seed")


def _device_buffers(ctx, N):
    """Output buffers for dhmc_run in HBM (torch is plumbing here: an allocator for device memory)."""
    try:
        import torch
    except ImportError:
        return None
    if not torch.cuda.is_available():
        return None
    dev = torch.device("cuda", ctx.cfg.device)
    tdt = {np.float64: torch.float64, np.int64: torch.int64, np.int32: torch.int32, np.uint32: torch.int32}
    return {name: torch.empty((ctx.C, N, ctx.D) if name == "draws" else (ctx.C, N), dtype=tdt[dt], device=dev)
            for name, dt in abi.OUTPUT_FIELDS}


def _host(arrs):
    out = {}
    for k, v in arrs.items():
        a = v.cpu().numpy() if hasattr(v, "is_cuda") else v
        out[k] = a.view(np.uint32) if k == "directions" else a
    return out


def _run_discarding(slogd, N, da, mcmc_reporter):
    """A stage whose draws nobody keeps (mcmc_with_warmup discards its warmup, mcmc.jl:579-583) and whose metric update — if any —
    reads a metric window: dhmc_run without outputs.  Step reports as in _run."""
    ctx = slogd.ctx
    chunk = int(getattr(mcmc_reporter, "step_chunk", 0) or 0)
    if chunk <= 0 or chunk >= N:
        ctx.run_into(N, {}, da=da)
        if mcmc_reporter is not None and N > 0 and chunk > 0:
            mcmc_reporter.report(N, **({"ϵ": _sig(np.median(ctx.stepsize()))} if da is not None else {}))
        return
    for n0 in range(0, N, chunk):
        L = min(chunk, N - n0)
        ctx.run_into(L, {}, da=None if da is None else dict(da, init=int(n0 == 0), finalize=int(n0 + L >= N)))
        mcmc_reporter.report(n0 + L, **({"ϵ": _sig(np.median(ctx.stepsize()))} if da is not None else {}))


def _run(slogd, N, da=None, keep=True, mcmc_reporter=None):
    """The N transitions of a stage for all chains.  The outputs are written to HBM and only come to the host when the caller
    keeps them as numpy arrays (`keep` and not `on_device`); returns (arrays, draws usable for the metric update).
    One dhmc_run — unless the stage's reporter wants step reports (reporting.jl:120-137: `report(mcmc_reporter, i; ϵ)` per draw,
    mcmc.jl:279,378): then the stage runs as calls of `mcmc_reporter.step_chunk` transitions (the chains resume where they stand:
    the same transitions, the same bits; dual averaging initialised by the first call and finalised by the last) with a report
    after each."""
    ctx = slogd.ctx
    bufs = _device_buffers(ctx, N)
    chunk = int(getattr(mcmc_reporter, "step_chunk", 0) or 0)
    if chunk <= 0 or chunk >= N:
        if bufs is None:
            arrs = ctx.run(N, da=da)
        else:
            ctx.run_into(N, bufs, da=da)
        if mcmc_reporter is not None and N > 0 and chunk > 0:
            mcmc_reporter.report(N, **({"ϵ": _sig(np.median(ctx.stepsize()))} if da is not None else {}))
    else:
        if bufs is None:
            arrs = {name: np.zeros((ctx.C, N, ctx.D) if name == "draws" else (ctx.C, N), dt) for name, dt in abi.OUTPUT_FIELDS}
        big = arrs if bufs is None else bufs
        for n0 in range(0, N, chunk):
            L = min(chunk, N - n0)
            dk = None if da is None else dict(da, init=int(n0 == 0), finalize=int(n0 + L >= N))
            if bufs is None:
                part = ctx.run(L, da=dk)
            else:
                part = _device_buffers(ctx, L)
                ctx.run_into(L, part, da=dk)
            for k, v in part.items():
                big[k][:, n0:n0 + L] = v
            mcmc_reporter.report(n0 + L, **({"ϵ": _sig(np.median(ctx.stepsize()))} if da is not None else {}))
    if bufs is None:
        return arrs, arrs["draws"]
    if not keep:
        return None, bufs["draws"]
    return (bufs if slogd.on_device else _host(bufs)), bufs["draws"]


def _collect(arrs):
    ts = TreeStatisticsNUTS(arrs["pi"], arrs["depth"], arrs["term_left"], arrs["term_right"],
                            arrs["acceptance_rate"], arrs["steps"], arrs["directions"])
    return arrs["draws"], ts, arrs["logdensities"], arrs["eps"]


def _state(ctx):
    q, lq, g = ctx.position()
    eps = ctx.stepsize()
    kappa = GaussianKineticEnergy.__new__(GaussianKineticEnergy)
    kappa.dense = ctx.cfg.metric == abi.METRIC_DENSE
    if kappa.dense and ctx.cfg.dense_per_chain:      # the reference's semantics: every chain its own Symmetric M⁻¹ — [C][D][D]
        mw = [ctx.metric_dense(c) for c in range(ctx.C)]
        kappa.Minv = np.stack([m for m, _ in mw]); kappa.W = np.stack([w for _, w in mw])
    elif kappa.dense:
        kappa.Minv, kappa.W = ctx.metric_dense()
    else:
        kappa.Minv = ctx.metric_diag(); kappa.W = np.sqrt(1.0 / kappa.Minv)
    return WarmupState(EvaluatedLogDensity(q, lq, g), kappa, None if np.isnan(eps).all() else eps)


def initialize_warmup_state(slogd, q=None, kappa=None, eps=None, **unknown):
    """mcmc.jl:129-132: q default random_position (mcmc.jl:108), κ default unit metric, ϵ default nothing."""
    for k in unknown:
        if k not in ("κ", "ϵ"):
            raise ValueError(f"ArgumentError: unknown initialization field {k}")
    kappa = unknown.get("κ", kappa)
    eps = unknown.get("ϵ", eps)
    ctx = slogd.ctx
    ctx.init(None if q is None else np.asarray(q, np.float64))
    if kappa is not None:
        _argcheck(kappa.size() == ctx.D, "dimension(ℓ) == size(κ, 1)")    # hamiltonian.jl:147
        if kappa.dense:
            _argcheck(kappa.Minv.ndim == 2, "a dense κ given at initialization is one matrix (every chain starts from it)")
            ctx.set_metric_dense(kappa.Minv)
        else:
            ctx.set_metric_diag(kappa.Minv)
    if eps is not None:
        ctx.set_stepsize(eps)
    return _state(ctx)


def warmup(slogd, stage, warmup_state):
    """The three `warmup` methods of mcmc.jl:99,134-148,258-286.  Returns (results, warmup_state′)."""
    ctx = slogd.ctx
    if stage is None:                                         # mcmc.jl:99
        return None, warmup_state
    if isinstance(stage, InitialStepsizeSearch):              # mcmc.jl:134-148
        _argcheck(warmup_state.eps is None, "stepsize ϵ manually specified, won't perform initial search")
        ctx.find_initial_stepsize(stage.initial_eps, stage.log_threshold, stage.maxiter_crossing)
        st = _state(ctx)
        slogd.reporter.report("found initial stepsize", **{"ϵ": _sig(np.median(st.eps))})              # mcmc.jl:141-142
        return None, st
    if isinstance(stage, TuningNUTS):                         # mcmc.jl:258-286
        if stage.M == Symmetric and ctx.cfg.metric != abi.METRIC_DENSE:
            raise ValueError("ArgumentError: a Symmetric metric stage needs a context with a dense κ")
        if stage.M == Diagonal and ctx.cfg.metric != abi.METRIC_DIAG:
            raise ValueError("ArgumentError: a Diagonal metric stage needs a context with a diagonal κ")
        _argcheck(warmup_state.eps is not None, "ϵ > 0")       # stepsize.jl:135
        ad = stage.stepsize_adaptation
        da = None if isinstance(ad, FixedStepsize) else dict(delta=ad.delta, gamma=ad.gamma, kappa=ad.kappa, t0=ad.t0)
        mcmc_reporter = make_mcmc_reporter(slogd.reporter, stage.N, currently_warmup=True,         # mcmc.jl:268-270
                                           tuning="stepsize" if stage.M is None else "stepsize and " + str(stage.M) + " metric")
        # A Diagonal stage's variance (mcmc.jl:281-284 with sample_M⁻¹(Diagonal, ·), :209) comes from running moments that the kernels
        # update with every draw (include/dhmc.h dhmc_metric_window_begin) — whether the stage's posterior matrix is kept or not, so
        # that mcmc_with_warmup and mcmc_keep_warmup sample the same chains bit for bit, as they do in the reference (mcmc.jl:579-583).
        if stage.M == Diagonal:
            ctx.metric_window_begin()
        try:
            if not slogd.keep_warmup and stage.M != Symmetric:
                _run_discarding(slogd, stage.N, da, mcmc_reporter)     # nobody reads this posterior matrix: no [C][N][D] buffer at all
                arrs = dev_draws = None
            else:
                arrs, dev_draws = _run(slogd, stage.N, da=da, keep=slogd.keep_warmup, mcmc_reporter=mcmc_reporter)
            if stage.M == Diagonal:
                ctx.update_metric_diag_window(stage.lam)
        finally:
            if stage.M == Diagonal and ctx.metric_window_count() >= 0:
                ctx.metric_window_end()
        if stage.M == Symmetric:                               # mcmc.jl:281-284 with sample_M⁻¹(Symmetric, ·) (:210): pooled over the
            ctx.update_metric_dense(dev_draws, stage.lam)      # context's chains (shared M⁻¹), or chain by chain (per_chain_metric)
        st = _state(ctx)
        if stage.M is not None:
            mcmc_reporter.report("adaptation finished", adapted_kinetic_energy=repr(st.kappa))        # mcmc.jl:283
        if arrs is None:
            return None, st
        draws, ts, lds, epss = _collect(arrs)
        return dict(posterior_matrix=draws, tree_statistics=ts, eps=epss, logdensities=lds), st
    raise TypeError(f"unknown warmup stage {stage!r}")


def _warmup(slogd, stages, initial_warmup_state):             # mcmc.jl:450-457
    acc, st = [], initial_warmup_state
    for stage in stages:
        results, st = warmup(slogd, stage, st)
        acc.append(dict(stage=stage, results=results, warmup_state=st))
    return acc, st


def mcmc(slogd, N, warmup_state):
    """mcmc.jl:366-381."""
    _argcheck(warmup_state.eps is not None, "ϵ > 0")
    mcmc_reporter = make_mcmc_reporter(slogd.reporter, N, currently_warmup=False)                  # mcmc.jl:372
    arrs, _ = _run(slogd, N, mcmc_reporter=mcmc_reporter)
    draws, ts, lds, _ = _collect(arrs)
    return dict(posterior_matrix=draws, tree_statistics=ts, logdensities=lds)


PER_CHAIN_DENSE_AUTO_DIM = 256              # per_chain_metric=None: every chain its own Symmetric metric up to this dimension …
PER_CHAIN_DENSE_AUTO_BYTES = 2 << 30        # … while the C pairs (M⁻¹, Wᵀ) of padded matrices stay below this


def _per_chain_metric_default(l, chains, metric_allreduce, max_depth=10):
    """What `per_chain_metric=None` means for a Symmetric warmup: the reference's semantics — every chain adapts its own M⁻¹ from its
    own draws (mcmc.jl:281-285) — whenever that is affordable: a built-in functor family (the wave-per-chain dense kernels; a caller's
    functor and callback models run the shared dense metric only), at most 256
    coordinates (beyond, a matvec per chain streams 8·D² bytes per leapfrog and the pooled GEMM engine is the practical choice),
    2·C·Dpad² doubles of matrices plus the chains' workspace within 2 GiB, no job-wide pooling requested.  Otherwise ONE M⁻¹ pooled over the context's chains (the batched
    engine's design; a stated deviation from the reference)."""
    D = l.dimension()
    dpad = 64 * max(1, -(-D // 64))
    nvec = 18 + 7 * max_depth                              # workspace rows per chain of the dense kernels (csrc/nuts_dense_kernel.hpp wd_nvec)
    footprint = 16 * chains * dpad * dpad + 8 * chains * nvec * dpad          # the C pairs (M⁻¹, Wᵀ) and the chains' workspace
    return (metric_allreduce is None and l.family not in (abi.TARGET_EXTERNAL, abi.TARGET_LOGISTIC) and l.family < abi.TARGET_USER_BASE
            and D <= PER_CHAIN_DENSE_AUTO_DIM and footprint <= PER_CHAIN_DENSE_AUTO_BYTES)


def mcmc_keep_warmup(rng, l, N, *, chains=1, initialization=(), warmup_stages=None, algorithm=NUTS(),
                     reporter=None, device=0, on_device=False, per_chain_metric=None, metric_allreduce=None, _keep_warmup=True):
    """mcmc.jl:521-532.  `chains` independent chains run at once on one GPU.  `on_device=True` returns the
    posterior matrices and statistics as torch CUDA tensors (no PCIe copy of the draws).
    `per_chain_metric` concerns Symmetric (dense) metrics only — a Diagonal κ is always per chain: True gives every chain its own
    M⁻¹ adapted from its own draws, exactly what C separate calls of the reference do (mcmc.jl:281-284) — the wave-per-chain dense
    kernels, 2·C·D² doubles of HBM; κ.M⁻¹ then comes back as [C][D][D]; False adapts ONE M⁻¹ from the pooled draws of all chains,
    which is what lets the leapfrog's products run as one GEMM over the batch; None (default, round 5): the reference's per-chain
    semantics where that is affordable (`_per_chain_metric_default`: at most 256 coordinates, a functor family, ≤ 2 GiB of
    matrices), the pooled metric otherwise.
    `metric_allreduce` (a job sharded over several GPUs, one process each, `rng.chain_offset` = the block's first chain): an
    in-place SUM over the ranks (sharding.TorchAllReduce(torch.distributed)) — the shared Symmetric M⁻¹ is then adapted from the
    draws of ALL ranks, so every rank samples with the matrix one GPU holding all chains would have adapted (to rounding)."""
    warmup_stages = default_warmup_stages() if warmup_stages is None else warmup_stages
    reporter = default_reporter() if reporter is None else reporter
    rng = _as_rng(rng)
    _argcheck(l.capabilities() >= 1, "capabilities(ℓ) ≥ LogDensityOrder(1)")   # hamiltonian.jl:146
    init = dict(initialization)
    k0 = init.get("κ", init.get("kappa"))
    wants_dense = any(isinstance(s, TuningNUTS) and s.M == Symmetric for s in warmup_stages)
    metric = abi.METRIC_DENSE if ((k0 is not None and k0.dense) or wants_dense) else abi.METRIC_DIAG
    if per_chain_metric is None:
        per_chain_metric = metric == abi.METRIC_DENSE and _per_chain_metric_default(l, chains, metric_allreduce, algorithm.max_depth)
        if k0 is not None and k0.dense and np.ndim(k0.Minv) == 2:
            per_chain_metric = False               # a caller who hands over ONE matrix for all chains asks for the shared metric
        if per_chain_metric:                       # said once per call, where the reporter says things: κ.M⁻¹ will be [C][D][D]
            reporter.report("Symmetric metric: one M⁻¹ per chain (κ.M⁻¹ is [chains][D][D]); per_chain_metric=False pools one")
    ctx = DeviceContext(l.dimension(), chains, target=l.family, target_params=l.params(), seed=rng.seed,
                        max_depth=algorithm.max_depth, min_delta=algorithm.min_delta,
                        chain_offset=rng.chain_offset, device=device, metric=metric,
                        dense_per_chain=bool(per_chain_metric) and metric == abi.METRIC_DENSE)
    if l.family == abi.TARGET_EXTERNAL:
        ctx.set_logdensity_callback(l.callback())
    if metric_allreduce is not None:
        _argcheck(metric == abi.METRIC_DENSE and not per_chain_metric, "metric_allreduce pools a shared Symmetric metric")
        ctx.set_metric_allreduce(metric_allreduce)
    slogd = SamplingLogDensity(rng, l, algorithm, reporter, ctx, on_device=on_device, keep_warmup=_keep_warmup)
    initial = initialize_warmup_state(slogd, **dict(initialization))
    wu, final = _warmup(slogd, warmup_stages, initial)
    inference = mcmc(slogd, N, final)
    return dict(initial_warmup_state=initial, warmup=wu, final_warmup_state=final, inference=inference,
                sampling_logdensity=slogd)


def mcmc_with_warmup(rng, l, N, *, chains=1, initialization=(), warmup_stages=None, algorithm=NUTS(),
                     reporter=None, device=0, on_device=False, per_chain_metric=None, metric_allreduce=None):
    """mcmc.jl:575-584: returns posterior_matrix [C][N][D], tree_statistics, logdensities [C][N], κ, ϵ [C].
    The warmup stages' draws never leave the GPU (the reference discards them too, mcmc.jl:579-583)."""
    r = mcmc_keep_warmup(rng, l, N, chains=chains, initialization=initialization, warmup_stages=warmup_stages,
                         algorithm=algorithm, reporter=reporter, device=device, on_device=on_device,
                         per_chain_metric=per_chain_metric, metric_allreduce=metric_allreduce, _keep_warmup=False)
    out = dict(r["inference"])
    out["kappa"] = r["final_warmup_state"].kappa
    out["eps"] = r["final_warmup_state"].eps
    r["sampling_logdensity"].ctx.close()
    return out


@dataclass
class MCMCSteps:
    """mcmc.jl:295-300, bound to the context that holds the chains."""
    slogd: SamplingLogDensity


def mcmc_steps(sampling_logdensity, warmup_state=None):
    """mcmc.jl:335-340 (constructor 2: from `mcmc_keep_warmup` results): steps with the κ and ϵ of `warmup_state`, from its
    Q — pushed into the context when they are not what it already holds."""
    ctx = sampling_logdensity.ctx
    if warmup_state is not None:
        _argcheck(warmup_state.eps is not None, "warmup_state.ϵ ≢ nothing")         # mcmc.jl:336
        k = warmup_state.kappa
        if k.dense and k.Minv.ndim == 3:                       # per-chain Symmetric κ: only as the context's own state
            _argcheck(bool(ctx.cfg.dense_per_chain) and all(np.array_equal(ctx.metric_dense(c)[0], k.Minv[c]) for c in range(ctx.C)),
                      "a per-chain dense κ can only be continued on the context that adapted it")
        elif k.dense:
            if not np.array_equal(ctx.metric_dense()[0], k.Minv):
                ctx.set_metric_dense(k.Minv)
        elif not np.array_equal(ctx.metric_diag(), np.broadcast_to(k.Minv, (ctx.C, ctx.D))):
            ctx.set_metric_diag(k.Minv)
        if not np.array_equal(ctx.stepsize(), np.broadcast_to(warmup_state.eps, (ctx.C,))):
            ctx.set_stepsize(warmup_state.eps)
        if not np.array_equal(ctx.position()[0], warmup_state.Q.q):
            ctx.set_position(warmup_state.Q.q)
    return MCMCSteps(sampling_logdensity)


def mcmc_next_step(steps, Q=None):
    """mcmc.jl:348-351: one transition of every chain from Q (default: where the chains are); returns (Q′, tree statistics
    of that transition)."""
    ctx = steps.slogd.ctx
    # A Q that is the context's own last state (the object this function returned, or equal arrays) costs no copy; a Q of
    # the caller's own is evaluated on the device — strictly, as dhmc_init does, where the reference would trust its ℓq and
    # ∇ℓq (mcmc.jl:348-351): a stated deviation, the position is what defines the step.
    # The cached state counts only while nobody else moved the chains (two MCMCSteps over one context, a run or set_position in
    # between): the context's position epoch says so.
    last = getattr(steps, "_last_Q", None)
    if last is not None and getattr(steps, "_last_epoch", None) != ctx.position_epoch:
        last = None
    if Q is not None and Q is not last and not (last is not None and np.array_equal(last.q, np.asarray(Q.q))):
        if last is not None or not np.array_equal(ctx.position()[0], np.asarray(Q.q)):
            ctx.set_position(Q.q)
    draws, ts, lds, _ = _collect(ctx.run(1))
    q, lq, g = ctx.position()
    out = EvaluatedLogDensity(q, lq, g)
    try:
        steps._last_Q = out
        steps._last_epoch = ctx.position_epoch
    except AttributeError:
        pass
    return out, ts


def stack_posterior_matrices(results):
    """mcmc.jl:602-604: [draw, chain, parameter].  Accepts one multi-chain result or a list of results."""
    rs = results if isinstance(results, (list, tuple)) else [results]
    pm = np.concatenate([np.asarray(r["posterior_matrix"]) for r in rs], axis=0)   # [C][N][D]
    return np.transpose(pm, (1, 0, 2))


def pool_posterior_matrices(results):
    """mcmc.jl:615-617: [parameter, pooled draw]."""
    rs = results if isinstance(results, (list, tuple)) else [results]
    pm = np.concatenate([np.asarray(r["posterior_matrix"]) for r in rs], axis=0)
    return pm.reshape(-1, pm.shape[-1]).T
