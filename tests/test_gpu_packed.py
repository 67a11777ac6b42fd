"""GPU: the packed small-D engine (csrc/packed_core.hpp: several chains per wavefront, L lanes × 4 coordinates per chain) against
the oracle and against the wave-per-chain kernel, bit for bit — every group width (L = 1, 2, 4, 8, 16), every packed family,
chain counts that leave idle groups in the last wave, metric windows, launch order, host outputs in chunks, gate widths and LDS
level counts.  The CPU suite runs the same body through tests/hostsim (L = 1); this file adds the DPP group operations."""
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
def _always_packed():
    """Every context of this file runs the packed kernel (DHMC_PACKED=1) unless a test says otherwise: without the variable the
    library picks the engine per launch from the previous launch's work."""
    old = os.environ.get("DHMC_PACKED")
    os.environ["DHMC_PACKED"] = "1"
    os.environ.pop("DHMC_PIPELINE", None)
    yield
    if old is None:
        os.environ.pop("DHMC_PACKED", None)
    else:
        os.environ["DHMC_PACKED"] = old


class _env:
    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        for k, v in self.old.items():
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


@pytest.mark.parametrize("cpl", [2, 4])
@pytest.mark.parametrize("D,C", [(2, 5), (4, 70), (7, 33), (13, 17), (30, 11), (32, 64), (33, 9), (64, 6)])
def test_funnel_every_group_width_matches_oracle(pkg, D, C, cpl):
    """Both layouts (2 and 4 coordinates per lane: L = 1 … 16 lanes per chain), chain counts that leave idle groups."""
    with _env(DHMC_PK="cpl=%s" % cpl):
        dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=100 + D)
    ora = ol.Oracle(D, C, target=ol.TARGET_FUNNEL, seed=100 + D, threads=8)
    for e in (dev, ora):
        e.init(); e.set_stepsize(0.2)
    _stages(dev, ora, f"funnel D={D} C={C}")


@pytest.mark.parametrize("D", [3, 10, 30, 50])
def test_normal_families_match_oracle(pkg, D):
    C = 21
    rng = np.random.default_rng(D)
    minv = rng.uniform(0.3, 3.0, size=(C, D))
    dev = pkg.DeviceContext(D, C, seed=2)
    ora = ol.Oracle(D, C, seed=2, threads=8)
    for e in (dev, ora):
        e.init(); e.set_metric_diag(minv); e.find_initial_stepsize()
    assert np.array_equal(dev.stepsize(), ora.stepsize())
    _stages(dev, ora, f"std normal D={D}")
    mu, prec = rng.normal(size=D), rng.uniform(0.2, 5.0, size=D)
    blob = ol.target_params_blob(ol.TARGET_DIAG_NORMAL, D, mu=mu, prec=prec)
    dev = pkg.DeviceContext(D, C, target=ol.TARGET_DIAG_NORMAL, target_params=blob, seed=4)
    ora = ol.Oracle(D, C, target=ol.TARGET_DIAG_NORMAL, params=blob, seed=4, threads=8)
    for e in (dev, ora):
        e.init(); e.set_stepsize(0.3)
    _stages(dev, ora, f"diag normal D={D}")
    # the tridiagonal-precision normal (round 6): neighbour coordinates across the lanes of a group (DPP row shifts)
    diag, off = rng.uniform(1.5, 3.0, size=D), rng.uniform(-0.6, 0.6, size=max(D - 1, 0))
    blob = ol.target_params_blob(ol.TARGET_TRIDIAG_NORMAL, D, diag=diag, off=off)
    dev = pkg.DeviceContext(D, C, target=ol.TARGET_TRIDIAG_NORMAL, target_params=blob, seed=6)
    ora = ol.Oracle(D, C, target=ol.TARGET_TRIDIAG_NORMAL, params=blob, seed=6, threads=8)
    for e in (dev, ora):
        e.init(); e.set_stepsize(0.3)
    _stages(dev, ora, f"tridiagonal normal D={D}")
    # the dense-precision normal (round 6): a D x D matvec inside the lane group (the group's lanes pass d_k around: Grp::pick)
    A = rng.normal(size=(D, D)) * 0.3
    Pm = A @ A.T + np.diag(rng.uniform(1.0, 2.0, size=D))
    mu = rng.normal(size=D)
    blob = ol.target_params_blob(ol.TARGET_DENSE_NORMAL, D, mu=mu, P=Pm)
    for cpl in (2, 4):
        if D > 32 and cpl == 2:
            continue
        with _env(DHMC_PK="cpl=%d" % cpl, DHMC_PACKED="1"):     # (the library takes the packed kernel for this family from 16 chains per CU on, D <= 32)
            dev = pkg.DeviceContext(D, C, target=ol.TARGET_DENSE_NORMAL, target_params=blob, seed=8)
        ora = ol.Oracle(D, C, target=ol.TARGET_DENSE_NORMAL, params=blob, seed=8, threads=8)
        for e in (dev, ora):
            e.init(); e.set_stepsize(0.25)
        _stages(dev, ora, f"dense-precision normal D={D} cpl={cpl}")


def test_divergences_depth_limits_and_always_divergent(pkg):
    D, C = 30, 24
    rng = np.random.default_rng(3)
    q0 = rng.normal(size=(C, D)) * 0.05
    q0[:, 0] = np.linspace(-6.0, 2.0, C)
    for eps, md in ((0.9, 6), (0.02, 5), (3.0, 10)):
        dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=5, max_depth=md)
        ora = ol.Oracle(D, C, target=ol.TARGET_FUNNEL, seed=5, max_depth=md, threads=8)
        for e in (dev, ora):
            e.init(q0); e.set_stepsize(eps)
        a, b = dev.run(40), ora.run(40)
        _same(a, b, f"eps {eps}")
        assert np.array_equal(dev.status(), ora.status())
    assert (b["term_left"] == b["term_right"]).any()
    dev = pkg.DeviceContext(3, 5, target=ol.TARGET_ALWAYS_DIVERGENT, seed=9)
    ora = ol.Oracle(3, 5, target=ol.TARGET_ALWAYS_DIVERGENT, seed=9)
    for e in (dev, ora):
        e.init(np.zeros((5, 3))); e.set_stepsize(0.5)
    a = dev.run(6)
    _same(a, ora.run(6), "always divergent")
    assert (a["depth"] == 0).all() and (a["steps"] == 1).all()


def test_packed_equals_wave_kernel_windows_and_launch_order(pkg):
    """The same 300 chains through both engines: a warmup with two metric windows (the chains' work diverges, so the second and
    later launches run in launch order), then inference with host outputs in chunks."""
    D, C = 30, 300
    res = []
    for packed in (1, 0):
        with _env(DHMC_PACKED=packed, DHMC_PIPELINE=0, DHMC_HOST_CHUNK=7):
            dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=8)
        dev.init(); dev.find_initial_stepsize()
        out = [dev.run(30, da={})]
        for n in (25, 40):
            dev.metric_window_begin()
            out.append(dev.run(n, da={}))
            dev.update_metric_diag_window()
        out.append(dev.run(33))
        res.append((out, dev.metric_diag(), dev.stepsize(), dev.position(), dev.last_run_leapfrogs()))
    for a, b in zip(res[0][0], res[1][0]):
        _same(a, b, "packed vs wave")
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    for x, y in zip(res[0][3], res[1][3]):
        assert np.array_equal(x, y)
    assert res[0][4] == res[1][4] == int(res[0][0][-1]["steps"].sum())


def test_engine_choice_follows_the_previous_launchs_work(pkg):
    """Without DHMC_PACKED the engine is chosen per launch (dhmc_run): packed until a launch is held open by a few chains with many
    times the mean's work, the wave-per-chain kernel after such a launch.  Whatever is chosen: the bits of either engine alone."""
    D, C = 30, 512
    res = []
    for env in (dict(), dict(DHMC_PACKED=1), dict(DHMC_PACKED=0)):
        old = os.environ.pop("DHMC_PACKED", None)
        try:
            with _env(**env):
                dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=31)
        finally:
            if old is not None:
                os.environ["DHMC_PACKED"] = old
        dev.init(); dev.find_initial_stepsize()
        res.append([dev.run(60, da={}), dev.run(50), dev.run(40, da=dict(init=1, finalize=1)), dev.run(35)])
    for r in res[1:]:
        for a, b in zip(res[0], r):
            _same(a, b, "engine choice")
    work = res[0][1]["steps"].sum(axis=1)
    assert work.max() > 3 * work.mean(), "the case should have a launch that is held open by a few chains"


@pytest.mark.parametrize("align,levels,cpl", [(1, 0, 2), (2, 1, 4), (8, 2, 2), (16, 6, 4), (4, 9, 2)])
def test_gate_width_and_lds_levels_change_no_result(pkg, align, levels, cpl):
    D, C = 30, 40
    with _env(DHMC_PK="align=%s,lds_levels=%s,cpl=%s" % (align, levels, cpl)):
        dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=21)
    ora = ol.Oracle(D, C, target=ol.TARGET_FUNNEL, seed=21, threads=8)
    for e in (dev, ora):
        e.init(); e.set_stepsize(0.15)
    _same(dev.run(30, da={}), ora.run(30, da={}), f"align {align} levels {levels}")
    _same(dev.run(30), ora.run(30), f"align {align} levels {levels}")


def test_chain_offset_is_the_rng_key(pkg):
    """Chains 40..79 of a job run as their own context: the same draws as in the whole job (sharding over GPUs)."""
    D = 30
    whole = pkg.DeviceContext(D, 80, target=ol.TARGET_FUNNEL, seed=6)
    part = pkg.DeviceContext(D, 40, target=ol.TARGET_FUNNEL, seed=6, chain_offset=40)
    for e in (whole, part):
        e.init(); e.set_stepsize(0.2)
    a, b = whole.run(25, da={}), part.run(25, da={})
    for k in a:
        assert np.array_equal(a[k][40:], b[k]), k


@pytest.mark.parametrize("cpl", [2, 4])
@pytest.mark.parametrize("D,C,waves", [(30, 41, 1), (7, 300, 2), (2, 500, 3), (64, 23, 4)])
def test_queue_of_places_matches_oracle(pkg, D, C, cpl, waves):
    """A launch of fewer waves than the chains need (DHMC_PK="max_waves=…"): the groups take chain after chain from the launch's queue of
    places — every chain the bits of its own launch, in any launch order (the second and later calls run in the order of the
    previous call's work)."""
    with _env(DHMC_PK="cpl=%s,max_waves=%s,queue=1" % (cpl, waves)):
        dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=300 + D)
    ora = ol.Oracle(D, C, target=ol.TARGET_FUNNEL, seed=300 + D, threads=8)
    rng = np.random.default_rng(D)
    q0 = rng.normal(size=(C, D)) * 0.3
    q0[:, 0] = np.linspace(-4.0, 2.0, C)          # trees of very different sizes
    for e in (dev, ora):
        e.init(q0); e.set_stepsize(0.3)
    for i, (N, da) in enumerate([(40, dict()), (40, None), (33, dict(init=0, finalize=1))]):
        _same(dev.run(N, da=da), ora.run(N, da=da), f"queue D={D} C={C} stage {i}")
    assert np.array_equal(dev.status(), ora.status())
    for x, y in zip(dev.position(), ora.position()):
        assert np.array_equal(x, y)


def test_end_game_of_a_tail_bound_packed_launch(pkg, capfd):
    """Many chains, a heavy-tailed tree size: the packed launch gives up the chains that are still running when few lane groups have
    one left, and the pipeline kernel finishes them (dhmc_run, RunParams::pk_live).  Thresholds lowered so that 600 chains do it; the
    bits of a context without the hand-over and of the oracle."""
    D, C = 30, 600
    os.environ.pop("DHMC_PACKED", None)
    res = []
    for env in (dict(DHMC_PK="many_chains=100,handover=60,max_waves=24", DHMC_DEBUG_ORDER=1, DHMC_HOST_CHUNK=1000000),
                dict(DHMC_PK="many_chains=100,handover=0,max_waves=24", DHMC_DEBUG_ORDER=1, DHMC_HOST_CHUNK=1000000)):
        with _env(**env):
            dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=17)
            dev.init(); dev.find_initial_stepsize()
            out = [dev.run(40, da={})]
            dev.metric_window_begin()
            out.append(dev.run(48, da={}))
            dev.update_metric_diag_window()
            out.append(dev.run(64))
            out.append(dev.run(35, da=dict(init=0, finalize=1)))
        res.append((out, dev.metric_diag(), dev.stepsize(), dev.position(), capfd.readouterr().err))
    handed = [int(l.split("end game: ")[1].split()[0]) for l in res[0][4].splitlines() if "end game:" in l]
    assert handed and max(handed) > 0, res[0][4][-600:]
    assert "end game:" not in res[1][4] and "engine: packed" in res[1][4]
    for a, b in zip(res[0][0], res[1][0]):
        _same(a, b, "hand-over vs one kernel")
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    for x, y in zip(res[0][3], res[1][3]):
        assert np.array_equal(x, y)
    ora = ol.Oracle(D, 24, target=ol.TARGET_FUNNEL, seed=17, threads=8)
    ora.init(); ora.find_initial_stepsize()
    b0 = ora.run(40, da={})
    ora.metric_window_begin(); b1 = ora.run(48, da={}); ora.update_metric_diag_window()
    b2 = ora.run(64)
    b3 = ora.run(35, da=dict(init=0, finalize=1))
    for a, b in zip(res[0][0], (b0, b1, b2, b3)):
        for k in a:
            assert np.array_equal(a[k][:24], b[k]), k


def test_end_game_as_shipped(pkg, capfd):
    """The end game with the library's own thresholds (until now only with lowered ones): 8192 funnel chains (more than 24
    per CU), outputs on the device, no engine switch set — after a tail-bound launch a call of N >= 32 transitions ends with the
    pipeline kernel finishing the chains the packed launch gave up (DHMC_DEBUG_ORDER only prints).  Same bits as a context whose
    packed launches never hand over, and as the oracle on the first 24 chains."""
    import torch
    D, C = 30, 8192
    os.environ.pop("DHMC_PACKED", None)
    fields = ("draws", "logdensities", "eps", "pi", "acceptance_rate", "steps", "term_left", "term_right", "depth", "directions")
    dts = dict(draws=torch.float64, logdensities=torch.float64, eps=torch.float64, pi=torch.float64, acceptance_rate=torch.float64,
               steps=torch.int64, term_left=torch.int64, term_right=torch.int64, depth=torch.int32, directions=torch.int32)
    sched = ((40, {}), (64, {}), (48, None), (40, dict(init=0, finalize=1)))
    res = []
    for env in (dict(DHMC_DEBUG_ORDER=1), dict(DHMC_DEBUG_ORDER=1, DHMC_PK="handover=0")):
        with _env(**env):
            dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=23)
            dev.init(); dev.find_initial_stepsize()
            outs = []
            for N, da in sched:
                bufs = {k: torch.empty((C, N, D) if k == "draws" else (C, N), dtype=dts[k], device="cuda") for k in fields}
                dev.run_into(N, bufs, da=da)
                torch.cuda.synchronize()
                outs.append({k: v[:64].cpu().numpy().copy() if k == "draws" else v.cpu().numpy().copy() for k, v in bufs.items()})
                outs[-1]["directions"] = outs[-1]["directions"].view(np.uint32)
                del bufs
            res.append((outs, dev.stepsize(), capfd.readouterr().err))
            del dev
    handed = [int(l.split("end game: ")[1].split()[0]) for l in res[0][2].splitlines() if "end game:" in l]
    assert handed and max(handed) > 0, res[0][2][-800:]
    assert "end game:" not in res[1][2]
    for a, b in zip(res[0][0], res[1][0]):
        _same(a, b, "end game vs one packed launch")
    assert np.array_equal(res[0][1], res[1][1])
    ora = ol.Oracle(D, 24, target=ol.TARGET_FUNNEL, seed=23, threads=8)
    ora.init(); ora.find_initial_stepsize()
    for (N, da), a in zip(sched, res[0][0]):
        b = ora.run(N, da=da)
        for k in b:
            assert np.array_equal(a[k][:24], b[k]), k
