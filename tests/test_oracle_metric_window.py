"""The metric window of include/dhmc.h (dhmc_metric_window_begin … dhmc_update_metric_diag_window) in the oracle: the running moments
(dhmc_detmath.h dm_window_update) give the reference's sample_M⁻¹(Diagonal, posterior_matrix) = var(posterior_matrix; dims = 2)
(src/mcmc.jl:209) without the posterior matrix."""
import numpy as np
import pytest
import oracle_lib as ol


def _welford(draws):
    """dm_window_update restated in numpy, one draw at a time (fp64; numpy does not contract a*b+c)."""
    mean = np.zeros(draws.shape[1:]); m2 = np.zeros(draws.shape[1:])
    for n, x in enumerate(draws, 1):
        d = x - mean
        mean = mean + d * (1.0 / n)
        m2 = m2 + d * (x - mean)
    return m2 / (len(draws) - 1)


@pytest.mark.parametrize("D", [3, 70])
def test_window_estimate_is_the_sample_variance(D):
    C = 3
    win = ol.Oracle(D, C, seed=17, threads=2); two = ol.Oracle(D, C, seed=17, threads=2)
    for o in (win, two):
        o.init(); o.find_initial_stepsize()
        o.run(10, da={})
    win.metric_window_begin()
    a = win.run(12, da={}); a2 = win.run(18, da=dict(init=0))      # a window spans calls
    b = two.run(12, da={}); b2 = two.run(18, da=dict(init=0))
    draws = np.concatenate([b["draws"], b2["draws"]], axis=1)
    assert np.array_equal(np.concatenate([a["draws"], a2["draws"]], axis=1), draws)     # the window changes no transition
    win.update_metric_diag_window(); two.update_metric_diag(draws)
    got, ref = win.metric_diag(), two.metric_diag()
    for c in range(C):
        assert np.array_equal(got[c], _welford(draws[c]))                               # the pinned order, bit for bit
    assert np.allclose(got, draws.var(axis=1, ddof=1), rtol=1e-12, atol=0)              # Statistics.var, to rounding
    assert np.allclose(got, ref, rtol=1e-12, atol=0)                                    # the two-pass update of the draws
    assert not np.array_equal(got, ref) or D < 4                                        # (different summation orders)


def test_window_needs_two_draws_and_closes():
    o = ol.Oracle(4, 2, seed=1, threads=1)
    o.init(); o.find_initial_stepsize()
    with pytest.raises(RuntimeError):
        o.update_metric_diag_window()          # none open
    o.metric_window_begin(); o.run(1)
    with pytest.raises(RuntimeError):
        o.update_metric_diag_window()          # one draw
    o.run(1)
    o.update_metric_diag_window()
    with pytest.raises(RuntimeError):
        o.update_metric_diag_window()          # closed by the update
