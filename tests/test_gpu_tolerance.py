"""GPU: the tolerance clause of `north_star` at PRODUCTION size, for every BASELINE config.

"Results match the reference CPU path on the same RNG seeds within a stated fp64 tolerance on posterior moments and
per-step Hamiltonian error."  The parity suite compares the HIP path with the oracle bit for bit — but both sides then share
the ABI's summation order (256-coordinate blocks, 64 interleaved fma chains, butterfly), the deterministic transcendentals of
include/dhmc_detmath.h and, for the shared dense metric, the one-product recurrence.  None of those three is what
Julia + OpenBLAS + libm do (src/hamiltonian.jl:103,110,277-280; src/NUTS.jl:130 — their orders are unpinned).  So here the
oracle runs in its OTHER flavour: every sum a plain left-to-right loop with separately rounded products
(oracle/mathops.hpp sequential_sums), glibc's exp/log/log1p/sincos (det=False), and the reference's two-product dense
leapfrog (set_dense_products(2)) — same algorithm, same random stream, none of the ABI's arithmetic choices — and the
device must stay within:

    positions after the first transition      rtol 1e-12 (atol 1e-13)
    per-step Hamiltonian error Δ = π − π₀     |difference| < 1e-9 along a 16-step trajectory, and for every transition's π
    integer tree outputs                      identical over the first transitions (depth, steps, termination, directions)
    posterior moments                         within 3 Monte-Carlo standard errors

at D = 1000 diagonal (configs[1]), D = 1000 dense (configs[2]), the funnel (configs[3]) and logistic regression with
N = 10⁵ observations (configs[4]).
"""
import numpy as np
import pytest

import oracle_lib as ol
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu

TREE_KEYS = ("depth", "steps", "term_left", "term_right", "directions")


@pytest.fixture(scope="module")
def pkg():
    return load_package()


@pytest.fixture()
def plain_sums():
    ol.set_sequential_sums(True)
    try:
        yield
    finally:
        ol.set_sequential_sums(False)


def _first_transitions(dev, ora, n, q_rtol=1e-12):
    a, b = dev.run(n), ora.run(n)
    for k in TREE_KEYS:
        assert np.array_equal(a[k], b[k]), k
    assert np.allclose(a["draws"][:, 0], b["draws"][:, 0], rtol=q_rtol, atol=1e-13)
    assert np.abs(a["pi"] - b["pi"]).max() < 1e-9                          # joint log density of every accepted point
    assert np.abs(a["acceptance_rate"] - b["acceptance_rate"]).max() < 1e-9
    return a, b


def _per_step_energy(dev, ora, eps, steps=16):
    """Δ = logdensity(H, z_i) − logdensity(H, z₀) along one trajectory from every chain's position, both directions
    (Diagnostics.leapfrog_trajectory, diagnostics.jl:214-227): the per-step Hamiltonian error."""
    da = dev.leapfrog_trajectory(eps, -steps // 2, steps // 2, momentum_index=3, with_points=False)
    oa = ora.leapfrog_trajectory(eps, -steps // 2, steps // 2, momentum_index=3)
    d_dev, d_ora = da["delta"], oa["delta"]
    assert d_dev.shape == d_ora.shape and np.array_equal(da["range"], oa["range"])
    assert np.abs(d_dev - d_ora).max() < 1e-9
    return np.abs(d_dev).max()


def test_config2_1000dim_diagonal(pkg, plain_sums):
    D, C = 1000, 32
    dev = pkg.DeviceContext(D, C, seed=101)
    ora = ol.Oracle(D, C, seed=101, det=False, threads=16)
    dev.init(); ora.init()
    dev.find_initial_stepsize(); ora.find_initial_stepsize()
    assert np.array_equal(dev.stepsize(), ora.stepsize())
    _first_transitions(dev, ora, 5)
    assert _per_step_energy(dev, ora, 0.3) > 1e-3                          # a real trajectory: the energy does move
    # adaptation + a metric window, then moments of further draws.  (Dual averaging amplifies last-place differences of
    # the acceptance rate — tests/test_oracle_nuts.py::test_dense_one_product_… — so after it the two runs are compared
    # as what they are: two samplers of the same target.)
    a, b = dev.run(60, da={}), ora.run(60, da={})
    dev.update_metric_diag(a["draws"][:, 30:]); ora.update_metric_diag(b["draws"][:, 30:])
    assert np.allclose(dev.metric_diag(), ora.metric_diag(), rtol=0.05)
    a, b = dev.run(40, da={}), ora.run(40, da={})
    assert np.allclose(dev.stepsize(), ora.stepsize(), rtol=0.05)
    a, b = dev.run(60), ora.run(60)
    n_eff = C * 60 * D / 2.0                                               # q² over coordinates, chains and draws
    assert abs((a["draws"] ** 2).mean() - (b["draws"] ** 2).mean()) < 3 * np.sqrt(2.0 / n_eff)
    assert np.abs(a["draws"].mean((0, 1)) - b["draws"].mean((0, 1))).max() < 3 / np.sqrt(C * 60 / 2.0)
    assert abs((a["draws"] ** 2).mean() - 1.0) < 0.02                      # ... and it is the target's second moment


def _config3(D):
    rho = 0.5
    sig = np.logspace(-1, 1, D)
    Pc = np.zeros(D) + (1 + rho ** 2) / (1 - rho ** 2); Pc[0] = Pc[-1] = 1 / (1 - rho ** 2)
    diag = Pc / sig ** 2
    off = np.zeros(D); off[:D - 1] = -rho / (1 - rho ** 2) / (sig[:-1] * sig[1:])
    idx = np.arange(D)
    Sigma = np.outer(sig, sig) * rho ** np.abs(idx[:, None] - idx[None, :])
    return sig, diag, off, Sigma


def test_config3_1000dim_dense_one_product_against_the_references_recurrence(pkg, plain_sums):
    """The device runs the ONE-product recurrence with MFMA chains; the oracle the reference's two products as plain loops."""
    D, C = 1000, 8
    sig, diag, off, Sigma = _config3(D)
    params = ol.target_params_blob(ol.TARGET_TRIDIAG_NORMAL, D, diag=diag, off=off)
    dev = pkg.DeviceContext(D, C, target=ol.TARGET_TRIDIAG_NORMAL, target_params=params, metric=ol.METRIC_DENSE, seed=77)
    ora = ol.Oracle(D, C, target=ol.TARGET_TRIDIAG_NORMAL, params=params, metric=ol.METRIC_DENSE, seed=77, det=False, threads=8)
    assert dev.dense_products() == 1
    ora.set_dense_products(2)
    dev.set_metric_dense(Sigma); ora.set_metric_dense(Sigma)
    q0 = np.random.default_rng(5).normal(size=(C, D)) * sig
    dev.init(q0); ora.init(q0)
    dev.set_stepsize(0.3); ora.set_stepsize(0.3)
    a, b = dev.run(6), ora.run(6)
    for k in TREE_KEYS:
        assert np.array_equal(a[k], b[k]), k
    assert np.allclose(a["draws"][:, 0] / sig, b["draws"][:, 0] / sig, rtol=1e-11, atol=1e-12)
    assert np.abs(a["pi"] - b["pi"]).max() < 1e-9
    assert np.abs(a["acceptance_rate"] - b["acceptance_rate"]).max() < 1e-9
    assert (a["steps"] >= 7).all()                                         # real trees (depth 3-4 with the perfect metric)


def test_dense_one_product_and_two_product_recurrences_agree(pkg):
    """the deviation of the one-product recurrence from the reference's, at D = 1000, over 60 transitions
    of 64 chains — identical trees, per-transition energy within 1e-9, positions within 1e-9 relative (the two runs see
    the same random stream; rounding differences of ≈1e-16 per step do not grow beyond this over 900 leapfrogs)."""
    D, C, N = 1000, 64, 60
    sig, diag, off, Sigma = _config3(D)
    params = ol.target_params_blob(ol.TARGET_TRIDIAG_NORMAL, D, diag=diag, off=off)
    q0 = np.random.default_rng(6).normal(size=(C, D)) * sig
    out = {}
    for products in (1, 2):
        dev = pkg.DeviceContext(D, C, target=ol.TARGET_TRIDIAG_NORMAL, target_params=params, metric=ol.METRIC_DENSE, seed=78)
        dev.set_dense_products(products)
        dev.set_metric_dense(Sigma)
        dev.init(q0); dev.set_stepsize(0.3)
        out[products] = dev.run(N)
        out[products]["rounds"] = dev.last_run_rounds()
        dev.close()
    a, b = out[1], out[2]
    for k in TREE_KEYS:
        assert np.array_equal(a[k], b[k]), k
    assert np.abs(a["pi"] - b["pi"]).max() < 1e-9
    assert np.abs((a["pi"] - a["logdensities"]) - (b["pi"] - b["logdensities"])).max() < 1e-9   # kinetic energies
    assert np.allclose(a["draws"] / sig, b["draws"] / sig, rtol=1e-9, atol=1e-10)
    assert a["steps"].sum() >= 15 * C * N // 2


def test_config4_funnel(pkg, plain_sums):
    D, C = 30, 128
    dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=55)
    ora = ol.Oracle(D, C, target=ol.TARGET_FUNNEL, seed=55, det=False, threads=16)
    dev.init(); ora.init()
    dev.find_initial_stepsize(); ora.find_initial_stepsize()
    assert np.array_equal(dev.stepsize(), ora.stepsize())
    a, b = _first_transitions(dev, ora, 4, q_rtol=1e-11)
    _per_step_energy(dev, ora, 0.05)
    # moments: the funnel's neck makes trajectories sensitive, so the two flavours may part ways after a while — that is
    # what the Monte-Carlo criterion is for.  v = q₀ ~ N(0, 9).
    a, b = dev.run(150, da={}), ora.run(150, da={})
    a, b = dev.run(200), ora.run(200)
    va, vb = a["draws"][:, :, 0], b["draws"][:, :, 0]
    se = 3.0 / np.sqrt(C * 200 / 20.0)                                     # sd 3, ≈ 1 effective draw in 20 for v
    assert abs(va.mean() - vb.mean()) < 3 * se * np.sqrt(2)                # (NUTS under-explores the neck with either
    assert abs(va.std() - vb.std()) < 3 * se * np.sqrt(2)                  # arithmetic: the two are compared with each other)


def test_config5_logistic_with_1e5_observations(pkg, plain_sums):
    N, D, C = 100000, 256, 8
    rng = np.random.default_rng(0)
    X = rng.normal(size=(N, D)) / 16
    y = (rng.random(N) < 1 / (1 + np.exp(-X @ rng.normal(size=D)))).astype(float)
    params = ol.target_params_blob(ol.TARGET_LOGISTIC, D, X=X, y=y)
    dev = pkg.DeviceContext(D, C, target=ol.TARGET_LOGISTIC, target_params=params, seed=9)
    ora = ol.Oracle(D, C, target=ol.TARGET_LOGISTIC, params=params, seed=9, det=False, threads=8)
    q0 = np.random.default_rng(1).normal(size=(C, D)) * 0.1
    dev.init(q0); ora.init(q0)
    qd, lqd, gd = dev.position(); qo, lqo, go = ora.position()
    assert np.allclose(lqd, lqo, rtol=1e-12)                               # Σ over 10⁵ observations: blocks vs one loop
    assert np.allclose(gd, go, rtol=1e-9, atol=1e-9)
    dev.set_stepsize(0.02); ora.set_stepsize(0.02)
    a, b = dev.run(2), ora.run(2)
    for k in TREE_KEYS:
        assert np.array_equal(a[k], b[k]), k
    assert np.allclose(a["draws"][:, 0], b["draws"][:, 0], rtol=1e-10, atol=1e-12)
    assert np.abs(a["pi"] - b["pi"]).max() < 1e-7                          # π ≈ −7·10⁴ here: 1e-7 is 1e-12 relative
    assert np.abs(a["acceptance_rate"] - b["acceptance_rate"]).max() < 1e-8
