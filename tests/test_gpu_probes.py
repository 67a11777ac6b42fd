"""GPU: the Diagnostics functions that call the hot path — leapfrog_trajectory and
explore_log_acceptance_ratios (src/diagnostics.jl:144-152, 214-227) — through the C ABI, bit for bit against
the oracle, plus the reference's own tests for them (test/test_diagnostics.jl:42-76) through the host API."""
import numpy as np

import ess_reference
import pytest

import oracle_lib as ol
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_package()


def _pair(pkg, D, C, target=ol.TARGET_STD_NORMAL, params=None, **kw):
    return (pkg.DeviceContext(D, C, target=target, target_params=params, **kw),
            ol.Oracle(D, C, target=target, params=params, threads=4, **kw))


def _same(a, b, what):
    for k in ("delta", "logdensity", "q", "p", "range", "status"):
        assert np.array_equal(a[k], b[k], equal_nan=True), f"{what}: {k}"


CASES = [
    ("std3", 3, ol.TARGET_STD_NORMAL, None),
    ("std100", 100, ol.TARGET_STD_NORMAL, None),
    ("std1000", 1000, ol.TARGET_STD_NORMAL, None),
    ("diag70", 70, ol.TARGET_DIAG_NORMAL, lambda D: ol.target_params_blob(
        ol.TARGET_DIAG_NORMAL, D, mu=np.linspace(-1, 1, D), prec=np.linspace(0.5, 4, D))),
    ("funnel30", 30, ol.TARGET_FUNNEL, None),
    ("tridiag200", 200, ol.TARGET_TRIDIAG_NORMAL, lambda D: ol.target_params_blob(
        ol.TARGET_TRIDIAG_NORMAL, D, diag=np.full(D, 2.5), off=np.full(D - 1, -1.0))),
    # logistic regression: ℓ, ∇ℓ of all chains by the GEMMs of its round engine between the probe's kernels (2 500 observations: two blocks)
    ("logistic40", 40, ol.TARGET_LOGISTIC, lambda D: ol.target_params_blob(
        ol.TARGET_LOGISTIC, D, X=np.random.default_rng(2).normal(size=(2500, D)) / 6,
        y=(np.random.default_rng(3).random(2500) < 0.4).astype(float))),
]


@pytest.mark.parametrize("name,D,target,mk", CASES, ids=[c[0] for c in CASES])
def test_trajectory_and_ratios_match_oracle(pkg, name, D, target, mk):
    C = 5
    params = mk(D) if mk else None
    dev, ora = _pair(pkg, D, C, target=target, params=params, seed=7)
    dev.init(); ora.init()
    minv = np.random.default_rng(3).uniform(0.5, 2.0, (C, D))
    dev.set_metric_diag(minv); ora.set_metric_diag(minv)
    for eps, first, last, idx in ((0.1, -6, 9, 0), (0.37, 0, 5, 3), (0.05, -4, 0, 1)):
        _same(dev.leapfrog_trajectory(eps, first, last, momentum_index=idx),
              ora.leapfrog_trajectory(eps, first, last, momentum_index=idx), f"{name} eps={eps}")
    p = np.random.default_rng(5).normal(size=(C, D))
    _same(dev.leapfrog_trajectory(0.2, -2, 2, p=p), ora.leapfrog_trajectory(0.2, -2, 2, p=p), f"{name} given p")
    eps = 2.0 ** np.arange(-5, 3)
    a = dev.explore_log_acceptance_ratios(eps, n_momenta=7, momentum_index=2, allow_failure=True)
    b = ora.explore_log_acceptance_ratios(eps, n_momenta=7, momentum_index=2, allow_failure=True)
    assert np.array_equal(a, b, equal_nan=True), name
    ps = np.random.default_rng(6).normal(size=(C, 3, D))
    assert np.array_equal(dev.explore_log_acceptance_ratios(eps, ps=ps, allow_failure=True),
                          ora.explore_log_acceptance_ratios(eps, ps=ps, allow_failure=True), equal_nan=True)
    # the chains themselves are untouched
    for x, y in zip(dev.position(), ora.position()):
        assert np.array_equal(x, y)
    assert not dev.status().any()


def test_probes_through_the_batched_evaluation_path_match_oracle(pkg):
    """Beyond 1024 coordinates the density is evaluated for all chains at once between kernels (the engine of
    DHMC_TARGET_EXTERNAL, with builtin_normal_eval_kernel where the host's callback stands): the probes run as lock-step
    leapfrogs around that evaluation (external_rounds.hpp ext_probe_*) — bit-equal to the oracle, diagonal and dense metric."""
    D, C = 1500, 4
    params = ol.target_params_blob(ol.TARGET_TRIDIAG_NORMAL, D, diag=np.full(D, 2.5), off=np.full(D - 1, -1.0))
    for metric in (ol.METRIC_DIAG, ol.METRIC_DENSE):
        dev, ora = _pair(pkg, D, C, target=ol.TARGET_TRIDIAG_NORMAL, params=params, metric=metric, seed=23)
        dev.init(); ora.init()
        if metric == ol.METRIC_DENSE:
            idx = np.arange(D)
            S = 0.6 ** np.abs(idx[:, None] - idx[None, :]) * 0.7
            dev.set_metric_dense(S); ora.set_metric_dense(S)
        else:
            minv = np.random.default_rng(3).uniform(0.5, 2.0, (C, D))
            dev.set_metric_diag(minv); ora.set_metric_diag(minv)
        tag = f"big metric={metric}"
        _same(dev.leapfrog_trajectory(0.1, -3, 4, momentum_index=2), ora.leapfrog_trajectory(0.1, -3, 4, momentum_index=2), tag)
        p = np.random.default_rng(5).normal(size=(C, D))
        _same(dev.leapfrog_trajectory(0.2, -2, 0, p=p), ora.leapfrog_trajectory(0.2, -2, 0, p=p), tag + " given p")
        eps = 2.0 ** np.arange(-4, 1)
        assert np.array_equal(dev.explore_log_acceptance_ratios(eps, n_momenta=3, momentum_index=1),
                              ora.explore_log_acceptance_ratios(eps, n_momenta=3, momentum_index=1)), tag
        ps = np.random.default_rng(6).normal(size=(C, 2, D))
        assert np.array_equal(dev.explore_log_acceptance_ratios(eps, ps=ps), ora.explore_log_acceptance_ratios(eps, ps=ps)), tag
        for x, y in zip(dev.position(), ora.position()):
            assert np.array_equal(x, y)
        dev.close()


def test_probes_of_a_torch_model(pkg):
    """Diagnostics.leapfrog_trajectory / explore_log_acceptance_ratios (diagnostics.jl:144-227) for the caller's own batched model:
    with D = 1 (nothing to sum) bit-equal to the built-in functor; a trajectory that runs into a non-finite density stops there."""
    import torch
    ext = pkg.TorchLogDensity(1, logdensity_and_gradient=lambda q: (-0.5 * (q * q).sum(1), -q))
    q = np.array([[0.3], [-1.2], [2.0]])
    a = pkg.diagnostics.leapfrog_trajectory(ext, q, 0.25, range(-4, 6), rng=9)
    b = pkg.diagnostics.leapfrog_trajectory(pkg.StandardNormal(1), q, 0.25, range(-4, 6), rng=9)
    assert len(a) == 3 and [len(x) for x in a] == [len(x) for x in b] == [10, 10, 10]
    for ca, cb in zip(a, b):
        for ta, tb in zip(ca, cb):
            assert ta["position"] == tb["position"] and ta["Δ"] == tb["Δ"] and ta["z"]["lq"] == tb["z"]["lq"]
            assert np.array_equal(ta["z"]["q"], tb["z"]["q"]) and np.array_equal(ta["z"]["p"], tb["z"]["p"])
    ra = pkg.diagnostics.explore_log_acceptance_ratios(ext, q, np.arange(-6, 2), rng=9, N=5)
    rb = pkg.diagnostics.explore_log_acceptance_ratios(pkg.StandardNormal(1), q, np.arange(-6, 2), rng=9, N=5)
    assert ra.shape == (3, 8, 5) and np.array_equal(ra, rb)
    # a density that is −Inf beyond |q| > 3: the trajectory is tracked up to the first such point (diagnostics.jl:179)
    def walled(qq):
        lq = -0.5 * (qq * qq).sum(1)
        return torch.where((qq.abs() > 3).any(1), torch.full_like(lq, -float("inf")), lq), -qq
    w = pkg.TorchLogDensity(2, logdensity_and_gradient=walled)
    tr = pkg.diagnostics.leapfrog_trajectory(w, np.array([2.5, 0.0]), 0.5, range(0, 12), p=np.array([3.0, 0.1]))
    assert 2 <= len(tr) < 12 and tr[-1]["z"]["lq"] == -np.inf and all(np.isfinite(t["z"]["lq"]) for t in tr[:-1])


def test_dense_metric_probes_match_oracle(pkg):
    D, C = 40, 4
    params = ol.target_params_blob(ol.TARGET_TRIDIAG_NORMAL, D, diag=np.full(D, 2.5), off=np.full(D - 1, -1.0))
    dev, ora = _pair(pkg, D, C, target=ol.TARGET_TRIDIAG_NORMAL, params=params, metric=ol.METRIC_DENSE, seed=21)
    A = np.random.default_rng(1).normal(size=(D, D))
    S = A @ A.T / D + np.eye(D)
    dev.init(); ora.init()
    dev.set_metric_dense(S); ora.set_metric_dense(S)
    _same(dev.leapfrog_trajectory(0.15, -5, 7, momentum_index=4), ora.leapfrog_trajectory(0.15, -5, 7, momentum_index=4), "dense")
    p = np.random.default_rng(2).normal(size=(C, D))
    _same(dev.leapfrog_trajectory(0.15, -1, 3, p=p), ora.leapfrog_trajectory(0.15, -1, 3, p=p), "dense given p")
    eps = 2.0 ** np.arange(-4, 2)
    assert np.array_equal(dev.explore_log_acceptance_ratios(eps, n_momenta=5), ora.explore_log_acceptance_ratios(eps, n_momenta=5))


def test_trajectory_stops_at_first_nonfinite_density(pkg):   # diagnostics.jl:176-186
    D = 30
    dev, ora = _pair(pkg, D, 6, target=ol.TARGET_FUNNEL, seed=5)
    q0 = np.zeros((6, D)); q0[:, 0] = np.linspace(-6, 6, 6)
    dev.init(q0); ora.init(q0)
    a = dev.leapfrog_trajectory(40.0, -4, 4, allow_failure=True)     # an absurd step: overflow within a few steps
    b = ora.leapfrog_trajectory(40.0, -4, 4, allow_failure=True)
    for k in ("delta", "logdensity", "range"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert (a["range"][:, 1] < 4).any() or (a["range"][:, 0] > -4).any()
    assert np.isnan(a["delta"]).any()
    dv, oa = _pair(pkg, 4, 2, target=ol.TARGET_ALWAYS_DIVERGENT)
    dv.init(np.zeros((2, 4)), allow_failure=True); oa.init(np.zeros((2, 4)), allow_failure=True)
    x = dv.leapfrog_trajectory(0.1, -3, 4, allow_failure=True); y = oa.leapfrog_trajectory(0.1, -3, 4, allow_failure=True)
    assert np.array_equal(x["range"], y["range"]) and np.array_equal(x["delta"], y["delta"], equal_nan=True)


def test_argument_checks(pkg):   # diagnostics.jl:218
    dev = pkg.DeviceContext(3, 2)
    dev.init()
    with pytest.raises(ValueError):
        dev.leapfrog_trajectory(0.1, 1, 3)
    with pytest.raises(ValueError):
        dev.leapfrog_trajectory(0.1, -3, -1)


def test_reference_log_acceptance_ratios(pkg):   # test_diagnostics.jl:42-49
    l = pkg.DiagNormal(np.ones(5), np.ones(5))
    log2eps = range(-5, 6)
    logA = pkg.diagnostics.explore_log_acceptance_ratios(l, np.zeros(5), log2eps, N=13)
    assert np.isfinite(logA).all()
    assert logA.shape == (len(log2eps), 13)


def test_reference_leapfrog_trajectory(pkg):   # test_diagnostics.jl:51-76
    K, eps, ix0 = 2, 0.1, 5
    l = pkg.DiagNormal(np.ones(K), np.ones(K))
    kappa = pkg.GaussianKineticEnergy(K)
    p = np.ones(K) * 0.98
    # manual trajectory zs1[1..15] (0..14 steps from q = 0), by the closed-form Gaussian leapfrog
    zs, q, pp = [], np.zeros(K), p.copy()
    for _ in range(15):
        zs.append((q.copy(), pp.copy(), -0.5 * ((q - 1) ** 2).sum() - 0.5 * (pp ** 2).sum()))
        pm = pp + eps / 2 * (-(q - 1.0)); q = q + eps * pm; pp = pm + eps / 2 * (-(q - 1.0))
    pis = np.array([z[2] for z in zs])
    traj = pkg.diagnostics.leapfrog_trajectory(l, zs[ix0 - 1][0], eps, range(1 - ix0, 15 - ix0 + 1), kappa=kappa, p=zs[ix0 - 1][1])
    assert [t["position"] for t in traj] == list(range(1 - ix0, 15 - ix0 + 1))
    assert np.allclose([t["Δ"] for t in traj], pis - pis[ix0 - 1], atol=1e-5)
    assert all(np.allclose(t["z"]["q"], z[0]) and np.allclose(t["z"]["p"], z[1]) for t, z in zip(traj, zs))
    with pytest.raises(ValueError):
        pkg.diagnostics.leapfrog_trajectory(l, np.zeros(K), eps, range(1, 4))


def test_ess_rhat_kernels_match_host_estimator(pkg):
    """dhmc_ess_rhat (HIP, draws where they lie in HBM) against diagnostics.ess_rhat (numpy FFT) and the torch flavour."""
    import torch
    rng = np.random.default_rng(11)
    for C, N, D, phi in ((6, 500, 5, 0.6), (64, 100, 40, 0.0), (1, 1001, 3, 0.9), (3, 37, 2, -0.4)):
        e = rng.normal(size=(C, N, D))
        x = np.zeros_like(e)
        x[:, 0] = e[:, 0]
        for n in range(1, N):
            x[:, n] = phi * x[:, n - 1] + e[:, n]          # AR(1): ESS well below / above C·N
        x += rng.normal(size=(C, 1, D)) * 0.05              # slightly different chain means: R-hat > 1
        coords = np.arange(D, dtype=np.int32)[:: max(1, D // 7)]
        t = torch.from_numpy(x).cuda()
        ess, rhat = pkg.diagnostics.ess_bulk_device(t, coords, kind="plain")
        e2, r2 = ess_reference.ess_bulk_torch(t, torch.from_numpy(coords).long().cuda())
        for k, j in enumerate(coords):
            eh, rh = ess_reference.ess_rhat(x[:, :, j])
            assert np.isclose(ess[k], eh, rtol=1e-9), (C, N, j, ess[k], eh)
            assert np.isclose(rhat[k], rh, rtol=1e-12)
        assert np.allclose(ess, e2.cpu().numpy(), rtol=1e-9) and np.allclose(rhat, r2.cpu().numpy(), rtol=1e-12)
        # rank-normalised split-chain bulk ESS (Vehtari et al. 2021) against the scipy flavour; ties included
        xt = x.copy(); xt[:, 1::7] = xt[:, 0:-1:7][:, :xt[:, 1::7].shape[1]]      # repeated draws, as NUTS produces
        tt = torch.from_numpy(xt).cuda()
        eb, rb = pkg.diagnostics.ess_bulk_device(tt, coords)
        for k, j in enumerate(coords):
            eh, rh = ess_reference.ess_bulk(xt[:, :, j])
            assert np.isclose(eb[k], eh, rtol=1e-7), (C, N, j, eb[k], eh)
            assert np.isclose(rb[k], rh, rtol=1e-9)
        # tail ESS: indicators of the pooled 5 % / 95 % quantiles over the split chains (dhmc_ess_tail)
        if N >= 100:
            et, _ = pkg.diagnostics.ess_bulk_device(tt, coords, kind="tail")
            for k, j in enumerate(coords):
                eh = ess_reference.ess_tail(xt[:, :, j])
                assert np.isclose(et[k], eh, rtol=1e-7), (C, N, j, et[k], eh)
    with pytest.raises(RuntimeError):
        pkg.diagnostics.ess_bulk_device(torch.zeros((2, 3, 2), dtype=torch.float64, device="cuda"), kind="plain")   # n < 4


def _ar1(rng, C, N, D, phi):
    e = rng.normal(size=(C, N, D))
    x = np.zeros_like(e)
    x[:, 0] = e[:, 0]
    for n in range(1, N):
        x[:, n] = phi * x[:, n - 1] + e[:, n]
    return x + rng.normal(size=(C, 1, D)) * 0.05


def test_ess_long_series_path_gives_the_bits_of_the_lds_path(pkg, monkeypatch):
    """Series up to 7680 draws are held in LDS; longer ones lie in HBM and their autocovariances are computed a chunk of 1024
    lags at a time until Geyer's truncation (csrc/ess_kernels.hpp).  Same arithmetic order: forced onto short series
    (DHMC_ESS_LONG=1) the long path returns the bits of the LDS path — truncation inside the first chunk (φ = 0.6), beyond
    it (φ = 0.9995: pairs stay positive for thousands of lags) and a series that ends inside a chunk (N = 1500)."""
    import torch
    rng = np.random.default_rng(5)
    for C, N, D, phi in ((6, 3000, 3, 0.6), (4, 7000, 2, 0.9995), (3, 1500, 2, 0.999), (2, 37, 1, -0.4)):
        t = torch.from_numpy(_ar1(rng, C, N, D, phi)).cuda()
        res = {}
        for flag in ("0", "1"):
            monkeypatch.setenv("DHMC_ESS_LONG", flag)
            res[flag] = [pkg.diagnostics.ess_bulk_device(t, kind=k) for k in ("plain", "bulk")] + \
                        ([pkg.diagnostics.ess_bulk_device(t, kind="tail")] if N >= 100 else [])
        for a, b in zip(res["0"], res["1"]):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1], equal_nan=True), (C, N, phi)
    monkeypatch.delenv("DHMC_ESS_LONG")


def test_ess_rhat_series_longer_than_lds(pkg):
    """n = 7680 is the longest series one workgroup holds in LDS; n = 30 000 (split halves of 15 000 for the bulk / tail
    kinds) goes through the long path, against the host estimators."""
    import torch
    x = torch.randn((2, 7680, 3), dtype=torch.float64, device="cuda")
    ess, rhat = pkg.diagnostics.ess_bulk_device(x, np.array([0, 2], np.int32), kind="plain")
    assert (ess > 0.5 * 2 * 7680).all() and (np.abs(rhat - 1) < 0.01).all()
    rng = np.random.default_rng(8)
    xs = _ar1(rng, 4, 30000, 2, 0.8)
    t = torch.from_numpy(xs).cuda()
    ess, rhat = pkg.diagnostics.ess_bulk_device(t, kind="plain")
    eb, rb = pkg.diagnostics.ess_bulk_device(t)
    et, _ = pkg.diagnostics.ess_bulk_device(t, kind="tail")
    for j in range(2):
        eh, rh = ess_reference.ess_rhat(xs[:, :, j])
        assert np.isclose(ess[j], eh, rtol=1e-9) and np.isclose(rhat[j], rh, rtol=1e-12)
        eh, rh = ess_reference.ess_bulk(xs[:, :, j])
        assert np.isclose(eb[j], eh, rtol=1e-7) and np.isclose(rb[j], rh, rtol=1e-9)
        assert np.isclose(et[j], ess_reference.ess_tail(xs[:, :, j]), rtol=1e-7)
    assert (np.abs(ess / (4 * 30000 * (1 - 0.8) / (1 + 0.8)) - 1) < 0.15).all()      # AR(1): ESS = S (1 − φ)/(1 + φ)


def test_tree_statistics_summary_on_device(pkg):
    """dhmc_summarize_tree_statistics (HIP: EBFMI per chain, count_terminations, count_depths, mean and quantiles of the
    acceptance rates, src/diagnostics.jl:29-106) bit for bit against the oracle's restatement — on the statistics of a
    real funnel run (divergences, several depths), from device tensors where dhmc_run left them and from host arrays."""
    import torch
    C, N, D = 96, 120, 30
    dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=6, max_depth=6)
    dev.init(); dev.find_initial_stepsize()
    dev.run(40, da={}, fields=[])
    out = {"pi": torch.empty((C, N), dtype=torch.float64, device="cuda"), "acceptance_rate": torch.empty((C, N), dtype=torch.float64, device="cuda"),
           "term_left": torch.empty((C, N), dtype=torch.int64, device="cuda"), "term_right": torch.empty((C, N), dtype=torch.int64, device="cuda"),
           "depth": torch.empty((C, N), dtype=torch.int32, device="cuda")}
    dev.run_into(N, out)
    S, eb = pkg.diagnostics.summarize_tree_statistics_device(out["pi"], out["acceptance_rate"], out["term_left"], out["term_right"], out["depth"])
    host = {k: v.cpu().numpy() for k, v in out.items()}
    So, ebo = ol.summarize_tree_statistics(host["pi"], host["acceptance_rate"], host["term_left"], host["term_right"], host["depth"])
    assert dict(S) == So                                    # counts and floating point alike: same summation orders
    assert np.array_equal(eb, ebo)
    S2, eb2 = pkg.diagnostics.summarize_tree_statistics_device(host["pi"], host["acceptance_rate"], host["term_left"], host["term_right"], host["depth"])
    assert dict(S2) == So and np.array_equal(eb2, ebo)      # host arrays staged by the library
    assert So["termination_counts"]["divergence"] + So["termination_counts"]["max_depth"] > 0 and len(So["depth_counts"]) >= 4
    ts = type("TS", (), dict(pi=host["pi"], acceptance_rate=host["acceptance_rate"], termination_left=host["term_left"],
                             termination_right=host["term_right"], depth=host["depth"]))
    assert np.allclose(eb, pkg.diagnostics.EBFMI(ts), rtol=1e-10)                       # the host numpy flavour agrees
    assert pkg.diagnostics.count_terminations(ts) == So["termination_counts"]
