"""CPU suite: the packed per-draw kernel's body (csrc/packed_body.inc), compiled by g++ with one lane per chain
(tests/hostsim), against the oracle — every output of whole multi-stage runs with np.array_equal.  What this pins without a
GPU: the packed engine's tree logic (iterative adjacent_tree with per-chain scalars, one leaf per trip, the gate), its RNG
bookkeeping and the packed evaluators of the four families; the GPU suite adds the DPP group operations."""
import numpy as np
import pytest

import hostsim_lib as hs
import oracle_lib as ol


def _pair(D, C, target, seed, eps, minv=None, params=None, max_depth=10, min_delta=-1000.0, q0=None, align=4, lds_levels=3, blob=None):
    ora = ol.Oracle(D, C, target=target, seed=seed, max_depth=max_depth, min_delta=min_delta, params=blob)
    ora.init(q0)
    ora.set_stepsize(eps)
    if minv is not None:
        ora.set_metric_diag(minv)
    q, lq, g = ora.position()
    sim = hs.HostSim(D, C, target, q, lq, g, eps, minv=minv, seed=seed, max_depth=max_depth, min_delta=min_delta, params=params,
                     align=align, lds_levels=lds_levels)
    return ora, sim


def _same(a, b, what=""):
    for k in a:
        assert np.array_equal(a[k], b[k]), f"{what}: field {k} differs"


@pytest.mark.parametrize("D,align,lds_levels", [(30, 4, 3), (30, 1, 0), (30, 8, 9), (5, 4, 2), (64, 2, 1), (17, 16, 4), (2, 4, 3)])
def test_funnel_stages_match_oracle(D, align, lds_levels):
    C = 6
    ora, sim = _pair(D, C, ol.TARGET_FUNNEL, seed=11 + D, eps=0.25, align=align, lds_levels=lds_levels)
    for stage, (N, da) in enumerate([(30, dict()), (25, None), (12, dict(init=1, finalize=0)), (9, dict(init=0, finalize=1))]):
        _same(sim.run(N, da=da), ora.run(N, da=da), f"stage {stage}")
        assert np.array_equal(sim.eps, ora.stepsize())
    q, lq, g = ora.position()
    assert np.array_equal(sim.q[:, :D], q) and np.array_equal(sim.lq, lq) and np.array_equal(sim.g[:, :D], g)
    d = ora.da_state()
    for k in ("mu", "Hbar", "logeps", "logeps_bar", "m"):
        assert np.array_equal(sim.da[k], d[k]), k


def test_funnel_deep_trees_and_divergences():
    # a large step in the funnel's neck: divergent leaves, turning subtrees, depth-limited trees, −Inf densities
    D, C = 30, 8
    rng = np.random.default_rng(3)
    q0 = rng.normal(size=(C, D)) * 0.05
    q0[:, 0] = np.linspace(-6.0, 2.0, C)
    for eps, md in ((0.9, 6), (0.02, 5), (3.0, 10)):
        ora, sim = _pair(D, C, ol.TARGET_FUNNEL, seed=5, eps=eps, q0=q0, max_depth=md)
        a, b = sim.run(40), ora.run(40)
        _same(a, b, f"eps {eps}")
        assert np.array_equal(sim.status, ora.status())
    # the last case must have exercised divergences
    assert (b["term_left"] == b["term_right"]).any()


def test_window_metric_and_adaptation_schedule():
    D, C = 30, 4
    ora, sim = _pair(D, C, ol.TARGET_FUNNEL, seed=77, eps=0.1)
    _same(sim.run(20, da=dict()), ora.run(20, da=dict()), "initial")
    for n in (25, 40):
        ora.metric_window_begin(); sim.window_begin()
        _same(sim.run(n, da=dict()), ora.run(n, da=dict()), f"window {n}")
        ora.update_metric_diag_window(); sim.window_update_metric()
        assert np.array_equal(sim.minv[:, :D], ora.metric_diag())
    _same(sim.run(30), ora.run(30), "inference")


@pytest.mark.parametrize("D", [3, 10, 32, 50])
def test_normal_families_match_oracle(D):
    C = 5
    rng = np.random.default_rng(D)
    minv = rng.uniform(0.3, 3.0, size=(C, D))
    ora, sim = _pair(D, C, ol.TARGET_STD_NORMAL, seed=2, eps=0.4, minv=minv)
    _same(sim.run(25, da=dict()), ora.run(25, da=dict()), "std normal, adaptive")
    _same(sim.run(25), ora.run(25), "std normal")
    mu, prec = rng.normal(size=D), rng.uniform(0.2, 5.0, size=D)
    ora, sim = _pair(D, C, ol.TARGET_DIAG_NORMAL, seed=4, eps=0.3, params=(mu, prec),
                     blob=ol.target_params_blob(ol.TARGET_DIAG_NORMAL, D, mu=mu, prec=prec))
    _same(sim.run(25, da=dict()), ora.run(25, da=dict()), "diag normal, adaptive")
    _same(sim.run(25), ora.run(25), "diag normal")
    # tridiagonal-precision normal (round 6: the packed engine's fifth family): an AR(1)-like precision, diagonally dominant
    diag, off = rng.uniform(1.5, 3.0, size=D), rng.uniform(-0.6, 0.6, size=max(D - 1, 0))
    ora, sim = _pair(D, C, ol.TARGET_TRIDIAG_NORMAL, seed=6, eps=0.3, params=(diag, np.concatenate([off, [0.0]])),
                     blob=ol.target_params_blob(ol.TARGET_TRIDIAG_NORMAL, D, diag=diag, off=off))
    _same(sim.run(25, da=dict()), ora.run(25, da=dict()), "tridiagonal normal, adaptive")
    _same(sim.run(25), ora.run(25), "tridiagonal normal")
    # dense-precision normal (round 6: the sixth family): a well-conditioned SPD matrix, every (Pd)_i one k-ascending fma chain
    A = rng.normal(size=(D, D)) * 0.3
    Pm = A @ A.T + np.diag(rng.uniform(1.0, 2.0, size=D))
    mu = rng.normal(size=D)
    ora, sim = _pair(D, C, ol.TARGET_DENSE_NORMAL, seed=8, eps=0.25, params=(mu, Pm),
                     blob=ol.target_params_blob(ol.TARGET_DENSE_NORMAL, D, mu=mu, P=Pm))
    _same(sim.run(25, da=dict()), ora.run(25, da=dict()), "dense-precision normal, adaptive")
    _same(sim.run(25), ora.run(25), "dense-precision normal")


def test_always_divergent_matches_oracle():
    # the reference's fault-injection density (test/test_NUTS.jl:58-85): every transition ends at its first leaf
    D, C = 3, 2
    ora, sim = _pair(D, C, ol.TARGET_ALWAYS_DIVERGENT, seed=9, eps=0.5, q0=np.zeros((C, D)))
    a, b = sim.run(6), ora.run(6)
    _same(a, b, "always divergent")
    assert (a["depth"] == 0).all() and (a["steps"] == 1).all() and (a["acceptance_rate"] == 0).all()


@pytest.mark.parametrize("target,D", [(ol.TARGET_FUNNEL, 30), (ol.TARGET_STD_NORMAL, 7)])
def test_queue_of_places_changes_no_result(target, D):
    # a lane group that takes chain after chain from the launch's queue (any order) gives every chain the result of its own launch
    C = 9
    rng = np.random.default_rng(11)
    q0 = rng.normal(size=(C, D)) * 0.3
    ora, sim = _pair(D, C, target, seed=21, eps=0.3, q0=q0)
    order = rng.permutation(C)
    for stage, da in enumerate((dict(), dict(init=0), None)):
        _same(sim.run(15, da=da, queue=True, order=order), ora.run(15, da=da), f"stage {stage}")
    assert np.array_equal(sim.status, ora.status())
    assert np.array_equal(sim.transition, np.full(C, 45, np.uint32))


@pytest.mark.parametrize("target,D", [(ol.TARGET_FUNNEL, 30), (ol.TARGET_STD_NORMAL, 7)])
def test_end_game_hand_over_changes_no_result(target, D):
    # a launch that gives its chains up once few lane groups still have one (here: always), continued by later launches
    C, N = 7, 23
    rng = np.random.default_rng(8)
    q0 = rng.normal(size=(C, D)) * 0.2
    ora, sim = _pair(D, C, target, seed=31, eps=0.3, q0=q0)
    for stage, da in enumerate((dict(), None)):
        a, launches = sim.run_handover(N, da=da)
        _same(a, ora.run(N, da=da), f"stage {stage}")
        assert launches > 2
    assert np.array_equal(sim.eps, ora.stepsize()) and np.array_equal(sim.status, ora.status())
