"""CPU: the oracle's restatement of the two Diagnostics functions that call the hot path
(src/diagnostics.jl:144-152, 214-227), against the reference's own tests for them
(test/test_diagnostics.jl:42-76)."""
import numpy as np
import pytest

import oracle_lib as ol


def _mvn_ones(K, chains=1):
    # multivariate_normal(ones(K)) of test/utilities.jl:64-67: mean 1, unit covariance
    return ol.Oracle(K, chains, target=ol.TARGET_DIAG_NORMAL,
                     params=ol.target_params_blob(ol.TARGET_DIAG_NORMAL, K, mu=np.ones(K), prec=np.ones(K)))


@pytest.mark.parametrize("det", [True, False])
def test_log_acceptance_ratios(det):   # test_diagnostics.jl:42-49
    K, N = 5, 13
    o = ol.Oracle(K, 1, target=ol.TARGET_DIAG_NORMAL, det=det,
                  params=ol.target_params_blob(ol.TARGET_DIAG_NORMAL, K, mu=np.ones(K), prec=np.ones(K)))
    o.init(np.zeros((1, K)))
    log2eps = np.arange(-5, 6)
    logA = o.explore_log_acceptance_ratios(2.0 ** log2eps, n_momenta=N)[0].T     # the reference's [log2ϵ, A] matrix
    assert logA.shape == (len(log2eps), N)
    assert np.isfinite(logA).all()
    # small steps conserve the Hamiltonian: |Δ| shrinks like ϵ³ for one step
    assert np.abs(logA[0]).max() < 1e-3 < np.abs(logA[-1]).max()


def test_leapfrog_trajectory():   # test_diagnostics.jl:51-76
    K, eps, n, ix0 = 2, 0.1, 15, 5
    o = _mvn_ones(K)
    o.init(np.zeros((1, K)))
    p = np.full(K, 0.98)
    # manual trajectory: zs1[k] = k-1 leapfrog steps from (q, p), k = 1..15
    fwd = o.leapfrog_trajectory(eps, 0, n - 1, p=p)
    qs, ps, ds = fwd["q"][0], fwd["p"][0], fwd["delta"][0]
    # the same trajectory re-centred on its 5th point
    o2 = _mvn_ones(K)
    o2.init(qs[ix0 - 1][None, :])
    tr = o2.leapfrog_trajectory(eps, 1 - ix0, n - ix0, p=ps[ix0 - 1])
    assert tr["range"][0].tolist() == [1 - ix0, n - ix0]
    assert np.allclose(tr["delta"][0], ds - ds[ix0 - 1], atol=1e-5)
    assert np.allclose(tr["q"][0], qs) and np.allclose(tr["p"][0], ps)
    assert tr["delta"][0][ix0 - 1] == 0.0
    # the independent Gaussian leapfrog formula of test_hamiltonian.jl:69-109 (unit metric): q, p along the path
    q, pp = np.zeros(K), p.copy()
    for k in range(1, n):
        pm = pp + eps / 2 * (-(q - 1.0))
        q = q + eps * pm
        pp = pm + eps / 2 * (-(q - 1.0))
        assert np.allclose(qs[k], q, rtol=1e-12) and np.allclose(ps[k], pp, rtol=1e-12)


def test_trajectory_stops_at_first_nonfinite_density():   # diagnostics.jl:176-186
    D = 4
    o = ol.Oracle(D, 2, target=ol.TARGET_ALWAYS_DIVERGENT)
    o.init(np.zeros((2, D)), allow_failure=True)
    st_q, st_lq, _ = o.position()
    tr = o.leapfrog_trajectory(0.1, -3, 4, allow_failure=True)
    # either the start is already non-finite (no steps at all) or the first step in each direction is the last
    lo, hi = tr["range"][0]
    assert (lo, hi) in ((0, 0), (-1, 1))
    assert np.isnan(tr["delta"][0][:3 + lo]).all() and np.isnan(tr["delta"][0][3 + hi + 1:]).all()


def test_positions_must_contain_zero():   # diagnostics.jl:218
    o = _mvn_ones(2)
    o.init(np.zeros((1, 2)))
    with pytest.raises(ol.OracleError):
        o.leapfrog_trajectory(0.1, 1, 3)
    with pytest.raises(ol.OracleError):
        o.leapfrog_trajectory(0.1, -3, -1)


def test_tree_statistics_summary_against_numpy():
    """oracle/diagnostics.hpp summarize_tree_statistics + ebfmi (src/diagnostics.jl:29-106 in the ABI's summation orders)
    against the plain numpy formulas: counts exact, floating point to rounding."""
    rng = np.random.default_rng(5)
    C, N = 7, 333
    pi = rng.normal(size=(C, N)).cumsum(axis=1) * 0.3 - 500
    acc = rng.beta(5, 1.2, size=(C, N))
    depth = rng.integers(0, 7, size=(C, N)).astype(np.int32)
    kind = rng.integers(0, 3, size=(C, N))
    tl = np.where(kind == 0, 1, np.where(kind == 1, 5, -3)).astype(np.int64)
    tr = np.where(kind == 0, 0, np.where(kind == 1, 5, 4)).astype(np.int64)
    S, eb = ol.summarize_tree_statistics(pi, acc, tl, tr, depth)
    assert S["N"] == C * N
    assert S["termination_counts"] == dict(max_depth=int((kind == 0).sum()), divergence=int((kind == 1).sum()), turning=int((kind == 2).sum()))
    assert S["depth_counts"] == np.bincount(depth.ravel()).tolist()
    assert np.isclose(S["a_mean"], acc.mean(), rtol=1e-13)
    assert np.allclose(S["a_quantiles"], np.quantile(acc, [0.05, 0.25, 0.5, 0.75, 0.95]), rtol=1e-12)   # Julia's default = numpy's "linear"
    ref = (np.diff(pi, axis=1) ** 2).mean(axis=1) / pi.var(axis=1, ddof=1)                              # diagnostics.jl:29-32
    assert np.allclose(eb, ref, rtol=1e-11)
