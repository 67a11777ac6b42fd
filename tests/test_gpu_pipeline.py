"""GPU: the pipeline kernel (csrc/nuts_pipeline_kernel.hpp: a short chain as four wavefronts — integrator, turn-statistic builder,
visited-statistic builder, proposal builder — joined by a ring of leaf records in LDS) against the oracle and the wave-per-chain kernel, bit for bit: every family it serves, dimensions
1 … 64, divergences / depth limits / −Inf densities (the integrator runs ahead of trees that end early), metric windows, launch
order, host outputs in chunks, resumed calls, and the engine choice of dhmc_run after a launch that a few chains held open."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_package()


@pytest.fixture(autouse=True)
def _always_pipeline():
    old = {k: os.environ.get(k) for k in ("DHMC_PIPELINE", "DHMC_PACKED")}
    os.environ["DHMC_PIPELINE"] = "1"
    os.environ.pop("DHMC_PACKED", None)
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _same(a, b, what=""):
    for k in a:
        assert np.array_equal(a[k], b[k]), f"{what}: field {k} differs"


def _stages(dev, ora, what):
    for i, (N, da) in enumerate([(24, dict()), (16, None), (9, dict(init=1, finalize=0)), (7, dict(init=0, finalize=1))]):
        _same(dev.run(N, da=da), ora.run(N, da=da), f"{what} stage {i}")
        assert np.array_equal(dev.stepsize(), ora.stepsize()), what
    for x, y in zip(dev.position(), ora.position()):
        assert np.array_equal(x, y), what
    assert np.array_equal(dev.status(), ora.status())


@pytest.mark.parametrize("D,C", [(2, 5), (7, 33), (30, 40), (33, 9), (64, 6)])
def test_funnel_matches_oracle(pkg, D, C):
    dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=200 + D)
    ora = ol.Oracle(D, C, target=ol.TARGET_FUNNEL, seed=200 + D, threads=8)
    for e in (dev, ora):
        e.init(); e.set_stepsize(0.2)
    _stages(dev, ora, f"funnel D={D}")


@pytest.mark.parametrize("family", ["std", "diag", "tridiag", "mvnormal"])
def test_normal_families_match_oracle(pkg, family):
    D, C = 20, 12
    rng = np.random.default_rng(7)
    if family == "std":
        tgt, params = ol.TARGET_STD_NORMAL, None
    elif family == "diag":
        tgt = ol.TARGET_DIAG_NORMAL
        params = ol.target_params_blob(tgt, D, mu=rng.normal(size=D), prec=rng.uniform(0.2, 5.0, size=D))
    elif family == "tridiag":
        tgt = ol.TARGET_TRIDIAG_NORMAL
        params = ol.target_params_blob(tgt, D, diag=np.full(D, 2.5), off=np.full(D - 1, -1.0))
    else:
        tgt = ol.TARGET_DENSE_NORMAL
        A = rng.normal(size=(D, D)); P = A @ A.T / D + np.eye(D)
        params = ol.target_params_blob(tgt, D, mu=rng.normal(size=D), P=(P + P.T) / 2)
    minv = rng.uniform(0.3, 3.0, size=(C, D))
    dev = pkg.DeviceContext(D, C, target=tgt, target_params=params, seed=3)
    ora = ol.Oracle(D, C, target=tgt, params=params, seed=3, threads=8)
    for e in (dev, ora):
        e.init(); e.set_metric_diag(minv); e.find_initial_stepsize()
    _stages(dev, ora, family)


@pytest.mark.parametrize("D,C", [(65, 6), (100, 4), (128, 9), (129, 5), (200, 7), (256, 5)])
def test_wider_chains_match_oracle(pkg, D, C):
    """Rows of 128 and 256 doubles (two and four slots per lane): the funnel (deep trees, suspended levels in LDS and in the HBM
    workspace), a diagonal normal under a per-chain metric with a metric window, and the reference's own case — 100 dimensions, 4 chains."""
    rng = np.random.default_rng(D)
    dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=400 + D)
    ora = ol.Oracle(D, C, target=ol.TARGET_FUNNEL, seed=400 + D, threads=8)
    q0 = rng.normal(size=(C, D)) * 0.2
    q0[:, 0] = np.linspace(-3.0, 1.5, C)
    for e in (dev, ora):
        e.init(q0); e.set_stepsize(0.15)
    _stages(dev, ora, f"funnel D={D}")
    tgt = ol.TARGET_DIAG_NORMAL
    params = ol.target_params_blob(tgt, D, mu=rng.normal(size=D), prec=rng.uniform(0.2, 5.0, size=D))
    dev = pkg.DeviceContext(D, C, target=tgt, target_params=params, seed=7)
    ora = ol.Oracle(D, C, target=tgt, params=params, seed=7, threads=8)
    minv = rng.uniform(0.3, 3.0, size=(C, D))
    for e in (dev, ora):
        e.init(); e.set_metric_diag(minv); e.find_initial_stepsize()
        e.metric_window_begin()
    _same(dev.run(30, da={}), ora.run(30, da={}), "diag normal, window")
    for e in (dev, ora):
        e.update_metric_diag_window()
    assert np.array_equal(dev.metric_diag(), ora.metric_diag())
    _stages(dev, ora, f"diag normal D={D}")


def test_trees_that_end_early(pkg):
    """Divergent leaves, turning subtrees, depth limits, −Inf densities: the integrator has run ahead when the builder stops."""
    D, C = 30, 24
    rng = np.random.default_rng(3)
    q0 = rng.normal(size=(C, D)) * 0.05
    q0[:, 0] = np.linspace(-6.0, 2.0, C)
    for eps, md in ((0.9, 6), (0.02, 5), (3.0, 10), (0.3, 1)):
        dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=5, max_depth=md)
        ora = ol.Oracle(D, C, target=ol.TARGET_FUNNEL, seed=5, max_depth=md, threads=8)
        for e in (dev, ora):
            e.init(q0); e.set_stepsize(eps)
        a, b = dev.run(40), ora.run(40)
        _same(a, b, f"eps {eps} max_depth {md}")
        assert np.array_equal(dev.status(), ora.status())
    dev = pkg.DeviceContext(3, 5, target=ol.TARGET_ALWAYS_DIVERGENT, seed=9)
    ora = ol.Oracle(3, 5, target=ol.TARGET_ALWAYS_DIVERGENT, seed=9)
    for e in (dev, ora):
        e.init(np.zeros((5, 3))); e.set_stepsize(0.5)
    a = dev.run(6)
    _same(a, ora.run(6), "always divergent")
    assert (a["depth"] == 0).all() and (a["steps"] == 1).all()


def test_pipeline_equals_wave_and_packed_kernels_over_a_warmup(pkg):
    """300 funnel chains through the three engines: adaptation, two metric windows, launch order, host outputs in chunks."""
    D, C = 30, 300
    res = []
    for env in (dict(), dict(DHMC_PIPELINE="0", DHMC_PACKED="0"), dict(DHMC_PIPELINE="0", DHMC_PACKED="1")):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env); os.environ["DHMC_HOST_CHUNK"] = "7"
        try:
            dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=8)
        finally:
            os.environ.pop("DHMC_HOST_CHUNK")
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        dev.init(); dev.find_initial_stepsize()
        out = [dev.run(30, da={})]
        for n in (25, 40):
            dev.metric_window_begin()
            out.append(dev.run(n, da={}))
            dev.update_metric_diag_window()
        out.append(dev.run(33))
        res.append((out, dev.metric_diag(), dev.stepsize(), dev.position(), dev.last_run_leapfrogs()))
    for r in res[1:]:
        for a, b in zip(res[0][0], r[0]):
            _same(a, b, "engine vs engine")
        assert np.array_equal(res[0][1], r[1]) and np.array_equal(res[0][2], r[2]) and res[0][4] == r[4]
        for x, y in zip(res[0][3], r[3]):
            assert np.array_equal(x, y)


def test_engine_choice_after_a_tail_bound_launch(pkg):
    """Without DHMC_PIPELINE / DHMC_PACKED: packed until a launch is held open by a few chains, the pipeline kernel after it; the bits of
    the wave-per-chain kernel throughout."""
    D, C = 30, 512
    res = []
    for env in (dict(), dict(DHMC_PIPELINE="0", DHMC_PACKED="0")):
        os.environ.pop("DHMC_PIPELINE", None)
        os.environ.update(env)
        try:
            dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=31)
        finally:
            for k in env:
                os.environ.pop(k, None)
        dev.init(); dev.find_initial_stepsize()
        res.append([dev.run(60, da={}), dev.run(50), dev.run(40, da=dict(init=1, finalize=1)), dev.run(35)])
    for a, b in zip(*res):
        _same(a, b, "engine choice")
    work = res[0][1]["steps"].sum(axis=1)
    assert work.max() > 3 * work.mean()
