"""GPU: the reference's statistical end-to-end tests (test/sample-correctness_tests.jl through NUTS_tests,
test/sample-correctness_utilities.jl:65-126) replayed through the host wrapper.  Same fail thresholds:
R̂ ≤ 2(1.01-1)+1 = 1.02, τ = ESS/N ≥ 0.5, E-BFMI ≥ 0.25, per-coordinate k-sample Anderson–Darling p ≥ 0.01
against 1000 exact samples (with the Bonferroni correction the reference computes, 0.01/d).  Targets are built
here (LogDensityTestSuite is not vendored): random correlated MVNs via MvNormal, and the reference's literal
cases — three isolated ill-conditioned MVNs, 1-D huge/tiny variance, scaled diagonal, kept 2/3/8-dim
(sample-correctness_tests.jl:25-87) — from the numeric fixture tests/golden/reference_mvn_cases.json."""
import warnings

import numpy as np

import ess_reference
import pytest
from scipy import stats

import oracle_lib as ol
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(0x121AA2F4)


@pytest.fixture(scope="module")
def pkg():
    return load_package()


def nuts_tests(pkg, l, exact_sampler, N, K=5, seed=1, R_fail=1.02, tau_fail=0.5, p_fail=0.01, ebfmi_fail=0.25, **mcmc_args):
    r = pkg.mcmc_with_warmup(seed, l, N, chains=K, reporter=pkg.NoProgressReport(), **mcmc_args)
    pm = r["posterior_matrix"]                                   # [K][N][d]
    d = pm.shape[2]
    stat = [ess_reference.ess_bulk(pm[:, :, k]) for k in range(d)]     # MCMCDiagnosticTools.ess_rhat default: bulk, split chains
    rhat = max(s[1] for s in stat); tau = min(s[0] for s in stat) / N
    assert rhat <= R_fail, f"R̂ = {rhat}"
    assert tau >= tau_fail, f"τ = {tau}"
    assert pkg.diagnostics.EBFMI(r["tree_statistics"]).min() >= ebfmi_fail
    Z = pm.reshape(-1, d); Z1 = exact_sampler(1000)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                          # scipy caps/floors the p-value and says so
        ps = [stats.anderson_ksamp([Z[:, k], Z1[:, k]]).pvalue for k in range(d)]
    assert min(ps) >= p_fail / d, f"AD p = {min(ps)}"
    return r


def mvn_case(pkg, mu, Sigma):
    L = np.linalg.cholesky(Sigma)
    return pkg.MvNormal(mu, Sigma), (lambda n: mu + RNG.normal(size=(n, len(mu))) @ L.T)


def rand_C(K):   # a random correlation matrix (the reference draws a CorrCholeskyFactor, utilities.jl:23-26)
    A = RNG.normal(size=(K, K)) / 4 + np.eye(K)
    S = A @ A.T
    s = np.sqrt(np.diag(S))
    return S / np.outer(s, s)


def test_random_correlated_mvns_dense_adaptation(pkg):
    """sample-correctness_tests.jl:12-23: random K ∈ 2:8, μ, scales d, correlation C; Symmetric adaptation."""
    for i in range(4):
        K = int(RNG.integers(2, 9))
        mu = RNG.normal(size=K); dsc = np.abs(RNG.normal(size=K)) * 2 + 0.1
        Sigma = rand_C(K) * np.outer(dsc, dsc)
        l, samp = mvn_case(pkg, mu, Sigma)
        nuts_tests(pkg, l, samp, 1000, seed=10 + i, warmup_stages=pkg.default_warmup_stages(M=pkg.Symmetric))


def _reference_cases():
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_mvn_cases.json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("case", _reference_cases(), ids=lambda c: c["name"])
def test_reference_literal_cases(pkg, case):
    """sample-correctness_tests.jl:25-87: the three isolated ill-conditioned MVNs (dense adaptation, as there),
    the 1-D huge / tiny variance cases, the mildly scaled diagonal and the kept 2/3/8-dim cases, with the
    reference's literal μ and L (tests/golden/reference_mvn_cases.json, extracted by make_reference_cases.py)."""
    mu = np.array(case["mu"]); L = np.array(case["L"])
    l = pkg.MvNormal(mu, L @ L.T)
    samp = lambda n: mu + RNG.normal(size=(n, len(mu))) @ L.T
    stages = pkg.default_warmup_stages(M=pkg.Symmetric) if case["metric"] == "Symmetric" else pkg.default_warmup_stages()
    nuts_tests(pkg, l, samp, 1000, seed=100 + len(mu), warmup_stages=stages)


def test_reference_mixture_of_two_normals_through_a_device_functor(pkg):
    """sample-correctness_tests.jl:89-98: mix(0.2, N(0, I₃), N(1, (0.4 C₂)(0.4 C₂)ᵀ)), 1000 draws × 5 chains, with the call's own
    alert levels (τ_alert = 0.15 → τ_fail = 0.075, p_alert = 0.005 → p_fail = 0.0005; R̂ ≤ 1.02, E-BFMI ≥ 0.25 by default).  Not a
    built-in family: the log-sum-exp of the two components is a caller's device functor (tests/user_functors.py MIXTURE3),
    compiled into the per-draw kernels — the reference's LogDensityTestSuite.mix on the device.  Literals:
    tests/golden/reference_mixture_case.json (make_reference_cases.py)."""
    import json, os
    import user_functors as uf
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_mixture_case.json")) as fh:
        c = json.load(fh)
    a = c["alpha"]; mu2 = np.array(c["mu2"]); L2 = np.array(c["L2"])
    S2 = L2 @ L2.T
    P2 = np.linalg.inv(S2); P2 = (P2 + P2.T) / 2
    params = np.concatenate([[a, -np.log(abs(np.linalg.det(L2)))], mu2, P2.ravel()])
    l = pkg.DeviceFunctorLogDensity(3, uf.MIXTURE3, "Mixture3", params=params)

    def exact(n):
        first = RNG.random(n) < a
        z = RNG.normal(size=(n, 3))
        return np.where(first[:, None], z, mu2 + z @ L2.T)
    r = nuts_tests(pkg, l, exact, c["N"], seed=41, tau_fail=c["tau_alert"] * 0.5, p_fail=c["p_alert"] * 0.1)
    x = r["posterior_matrix"].reshape(-1, 3)
    mean = (1 - a) * mu2
    assert np.abs(x.mean(0) - mean).max() < 0.1                 # E x = (1 - α) μ₂
    # the density itself at a few points, against numpy (the functor's ℓ up to the shared -3/2 log 2π)
    ctx = pkg.DeviceContext(3, 4, target=l.family, target_params=l.params(), seed=2)
    q0 = RNG.normal(size=(4, 3)) * 1.5 + 0.5
    ctx.init(q0)
    q, lq, g = ctx.position()
    n1 = np.log(a) - 0.5 * (q0 ** 2).sum(1)
    d = q0 - mu2
    n2 = np.log(1 - a) - np.log(abs(np.linalg.det(L2))) - 0.5 * np.einsum("ci,ij,cj->c", d, P2, d)
    want = np.logaddexp(n1, n2)
    assert np.allclose(lq, want, rtol=1e-13, atol=1e-13)
    w1 = np.exp(n1 - want)[:, None]
    assert np.allclose(g, -w1 * q0 - (1 - w1) * (d @ P2), rtol=1e-12, atol=1e-12)


def test_dense_normal_target_parity(pkg):
    """The dense-precision target itself: HIP == oracle, bit for bit."""
    K = 7
    mu = RNG.normal(size=K); A = RNG.normal(size=(K, K)); Sigma = A @ A.T + 0.1 * np.eye(K)
    P = np.linalg.inv(Sigma); P = (P + P.T) / 2
    params = ol.target_params_blob(ol.TARGET_DENSE_NORMAL, K, mu=mu, P=P)
    dev = pkg.DeviceContext(K, 5, target=ol.TARGET_DENSE_NORMAL, target_params=params, seed=77)
    ora = ol.Oracle(K, 5, target=ol.TARGET_DENSE_NORMAL, params=params, seed=77)
    for e in (dev, ora):
        e.init(); e.find_initial_stepsize()
    a, b = dev.run(30, da={}), ora.run(30, da={})
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def _tail_cases():
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_tail_cases.json"), encoding="utf-8") as fh:
        return json.load(fh)


def _levels(lv):
    """NUTS_tests' keyword defaults (sample-correctness_utilities.jl:65-69) around the levels a call sets."""
    R_alert = lv.get("R̂_alert", 1.01); tau_alert = lv.get("τ_alert", 1.0); p_alert = lv.get("p_alert", 0.1); e_alert = lv.get("EBFMI_alert", 0.5)
    return dict(R_fail=lv.get("R̂_fail", 2 * (R_alert - 1) + 1), tau_fail=lv.get("τ_fail", tau_alert * 0.5),
                p_fail=lv.get("p_fail", p_alert * 0.1), ebfmi_fail=lv.get("EBFMI_fail", e_alert / 2))


def _check_density(pkg, l, D, logpdf, grad):
    """The functor's ℓ and ∇ℓ at a few points against numpy."""
    ctx = pkg.DeviceContext(D, 6, target=l.family, target_params=l.params(), seed=2)
    q0 = RNG.normal(size=(6, D)) * 1.3 + 0.3
    ctx.init(q0)
    _, lq, g = ctx.position()
    assert np.allclose(lq, [logpdf(y) for y in q0], rtol=1e-12, atol=1e-12)
    assert np.allclose(g, [grad(y) for y in q0], rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("which", [0, 1])
def test_reference_heavier_tails_and_skewness_through_device_functors(pkg, which):
    """sample-correctness_tests.jl:100-112: elongate(1.1)(N(0, I₅)) and (elongate(1.1) ∘ shift(1))(N(0, I₅)), 10 000 draws × 5
    chains, each call's own relaxed bars.  LogDensityTestSuite's transformations are not under /root/reference (parity unpinned):
    the definitions are this repository's, written down in tests/user_functors.py (ELONGATED) — the exact sampler below is the
    same map applied to normal draws.  Numbers: tests/golden/reference_tail_cases.json (make_reference_cases.py)."""
    import user_functors as uf
    c = _tail_cases()
    K, k = c["K"], c["elongate"]
    b = np.zeros(K) if which == 0 else np.array(c["shift"])
    call = c["calls"][which]
    l = pkg.DeviceFunctorLogDensity(K, uf.ELONGATED, "Elongated", params=np.concatenate([[k], b]))
    a, c1 = 1 / k - 1, (k - 1) * K / k

    def logpdf(y):
        s = np.linalg.norm(y); u = y * s ** a
        return -0.5 * ((u - b) ** 2).sum() - c1 * np.log(s) - np.log(k)

    def grad(y):
        s = np.linalg.norm(y); u = y * s ** a; d = u - b
        return -(s ** a * d + a * s ** (a - 2) * y * (y @ d)) - c1 * y / s ** 2

    def exact(n):
        x = RNG.normal(size=(n, K)) + b
        return x * np.linalg.norm(x, axis=1, keepdims=True) ** (k - 1)
    _check_density(pkg, l, K, logpdf, grad)
    r = nuts_tests(pkg, l, exact, call["N"], seed=61 + which, **_levels(call["levels"]))
    # radial moment: E‖y‖² = E‖x‖^(2k) against the exact sampler's
    x = r["posterior_matrix"].reshape(-1, K)
    want = (np.linalg.norm(exact(200000), axis=1) ** 2).mean()
    assert abs((x ** 2).sum(1).mean() / want - 1) < 0.05


def test_reference_funnel_mixed_with_a_normal_through_a_device_functor(pkg):
    """sample-correctness_tests.jl:114-117: mix(0.8, funnel()(N(0, I₅)), N(0, I₅)), 10 000 draws × 5 chains, the call's bars
    (E-BFMI ≥ 0.1, τ ≥ 0.05, AD p ≥ 0.005 / d, R̂ ≤ 1.05).  funnel() as defined in tests/user_functors.py (FUNNEL_MIX)."""
    import user_functors as uf
    c = _tail_cases()
    K, al = c["K"], c["funnel_mix_alpha"]
    call = c["calls"][2]
    l = pkg.DeviceFunctorLogDensity(K, uf.FUNNEL_MIX, "FunnelMix", params=np.array([al]))

    def parts(y):
        v, S = y[0], (y[1:] ** 2).sum()
        return np.log(al) - 0.5 * v * v - 0.5 * np.exp(-v) * S - 0.5 * (K - 1) * v, np.log(1 - al) - 0.5 * (v * v + S)

    def logpdf(y):
        return np.logaddexp(*parts(y))

    def grad(y):
        lf, ln = parts(y); lt = np.logaddexp(lf, ln)
        v, S = y[0], (y[1:] ** 2).sum()
        gf = np.concatenate([[-v + 0.5 * np.exp(-v) * S - 0.5 * (K - 1)], -np.exp(-v) * y[1:]])
        return np.exp(lf - lt) * gf + np.exp(ln - lt) * (-y)

    def exact(n):
        x = RNG.normal(size=(n, K))
        f = RNG.random(n) < al
        x[f, 1:] *= np.exp(x[f, :1] / 2)
        return x
    _check_density(pkg, l, K, logpdf, grad)
    nuts_tests(pkg, l, exact, call["N"], seed=71, **_levels(call["levels"]))
