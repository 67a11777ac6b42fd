"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeds.

Everything is compared BIT FOR BIT (np.array_equal): integer outputs (depth, steps,
termination, directions) and floating-point outputs alike, because the ABI pins the random
stream, the summation order and the scalar math (include/dhmc.h, include/dhmc_detmath.h).
The looser tolerance north_star allows (fp64 tolerance on posterior moments and per-step
Hamiltonian error) is exercised in test_gpu_statistics.py against the libm flavour of the oracle.
"""
import numpy as np
import pytest

import oracle_lib as ol
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_package()


def assert_same(a, b, what=""):
    for k in a:
        if not np.array_equal(a[k], b[k]):
            bad = np.argwhere(a[k] != b[k])
            raise AssertionError(f"{what}: field {k} differs at {bad[:5].tolist()} "
                                 f"gpu={a[k][tuple(bad[0])]!r} oracle={b[k][tuple(bad[0])]!r} ({len(bad)} mismatches)")


def make_pair(pkg, D, C, target=ol.TARGET_STD_NORMAL, params=None, **kw):
    dev = pkg.DeviceContext(D, C, target=target, target_params=params, **kw)
    ora = ol.Oracle(D, C, target=target, params=params, threads=8, **kw)
    return dev, ora


def test_init_random_positions(pkg):
    for D in (3, 30, 64, 100, 1000):
        dev, ora = make_pair(pkg, D, 5, seed=11)
        dev.init(); ora.init()
        for x, y in zip(dev.position(), ora.position()):
            assert np.array_equal(x, y)
        q = dev.position()[0]
        assert q.min() >= -2 and q.max() < 2 and q.std() > 0.5   # mcmc.jl:108


@pytest.mark.parametrize("D", [3, 30, 100, 200, 500, 1000])
def test_stepsize_search_and_fixed_run(pkg, D):
    C = 6
    dev, ora = make_pair(pkg, D, C, seed=D)
    dev.init(); ora.init()
    dev.find_initial_stepsize(); ora.find_initial_stepsize()
    assert np.array_equal(dev.stepsize(), ora.stepsize())
    a, b = dev.run(12), ora.run(12)
    assert_same(a, b, f"D={D}")
    assert (a["steps"] >= 1).all() and (a["depth"] >= 0).all()


@pytest.mark.parametrize("D", [5, 100, 1000])
def test_adaptive_stages_and_metric_update(pkg, D):
    """The shape of the reference's default warmup (mcmc.jl:415-425), shortened: step size
    search, a step-size-only stage, two metric stages, a final step-size stage, inference."""
    C = 4
    dev, ora = make_pair(pkg, D, C, seed=99)
    dev.init(); ora.init()
    dev.find_initial_stepsize(); ora.find_initial_stepsize()
    for n, metric in ((15, False), (25, True), (30, True), (10, False)):
        a, b = dev.run(n, da={}), ora.run(n, da={})
        assert_same(a, b, f"D={D} stage {n}")
        assert np.array_equal(dev.stepsize(), ora.stepsize())
        if metric:
            dev.update_metric_diag(a["draws"]); ora.update_metric_diag(b["draws"])
            assert np.array_equal(dev.metric_diag(), ora.metric_diag())
    a, b = dev.run(20), ora.run(20)
    assert_same(a, b, f"D={D} inference")


@pytest.mark.parametrize("case", ["D=3", "D=100", "D=1000", "D=1000 chunked host outputs", "D=1100 batched evaluation",
                                  "logistic rounds D=70", "logistic rounds D=600"])
def test_metric_window_matches_oracle(pkg, case, monkeypatch):
    """a tuning stage's metric update without its posterior matrix (include/dhmc.h dhmc_metric_window_begin). The
    kernels add every transition's draw to per-chain running moments where they store it — the per-draw kernel, K3 and K3b of the
    round engines — in the order dhmc_detmath.h pins (dm_window_update), so the oracle's window gives the same M⁻¹ bit for bit;
    the window spans two calls (one of them without any output), and changes no transition."""
    C = 5
    kw = {}
    if case.startswith("logistic"):
        D = int(case.split("=")[1]); N = 900
        rng = np.random.default_rng(D)
        X = rng.normal(size=(N, D)) / 8; y = (rng.random(N) < 0.5).astype(float)
        kw = dict(target=ol.TARGET_LOGISTIC, params=ol.target_params_blob(ol.TARGET_LOGISTIC, D, X=X, y=y))
        monkeypatch.setenv("DHMC_LOGISTIC_ROUNDS", "1")
    else:
        D = int(case.split()[0].split("=")[1])
    if "chunked" in case:
        monkeypatch.setenv("DHMC_HOST_CHUNK", "4")
    dev, ora = make_pair(pkg, D, C, seed=41, **kw)
    dev.init(); ora.init()
    dev.find_initial_stepsize(); ora.find_initial_stepsize()
    assert_same(dev.run(6, da={}), ora.run(6, da={}), case + " stepsize stage")
    assert dev.metric_window_count() == -1
    dev.metric_window_begin(); ora.metric_window_begin()
    a, b = dev.run(11, da={}), ora.run(11, da={})
    assert_same(a, b, case + " window, first call")
    assert dev.metric_window_count() == 11
    dev.run_into(9, {}, da=dict(init=0)); ora.run(9, da=dict(init=0), fields=[])           # no outputs at all
    assert dev.metric_window_count() == 20
    dev.update_metric_diag_window(); ora.update_metric_diag_window()
    assert dev.metric_window_count() == -1
    assert np.array_equal(dev.metric_diag(), ora.metric_diag())
    assert_same(dev.run(8), ora.run(8), case + " inference")
    # … and the estimate is the sample variance of the window's draws, to rounding
    ref = pkg.DeviceContext(D, C, seed=41, target=kw.get("target", ol.TARGET_STD_NORMAL), target_params=kw.get("params"))
    ref.init(); ref.find_initial_stepsize(); ref.run(6, da={})
    d1 = ref.run(11, da={}); d2 = ref.run(9, da=dict(init=0))
    draws = np.concatenate([d1["draws"], d2["draws"]], axis=1)
    assert np.array_equal(d1["draws"], a["draws"])
    assert np.allclose(dev.metric_diag(), draws.var(axis=1, ddof=1), rtol=1e-11, atol=0)
    with pytest.raises(ValueError):
        dev.update_metric_diag_window()                          # no window open
    dense = pkg.DeviceContext(8, 2, metric=ol.METRIC_DENSE)
    with pytest.raises(ValueError):
        dense.metric_window_begin()                              # the diagonal metric's estimate


def test_split_stage_equals_whole_stage(pkg):
    """One TuningNUTS stage issued as two dhmc_run calls (init on the first, finalize on the
    second) equals one call: the adaptation state persists in the context."""
    D, C = 50, 4
    dev1 = pkg.DeviceContext(D, C, seed=5); dev2 = pkg.DeviceContext(D, C, seed=5)
    for d in (dev1, dev2):
        d.init(); d.find_initial_stepsize()
    whole = dev1.run(30, da={})
    p1 = dev2.run(18, da=dict(init=1, finalize=0))
    p2 = dev2.run(12, da=dict(init=0, finalize=1))
    for k in whole:
        assert np.array_equal(whole[k], np.concatenate([p1[k], p2[k]], axis=1)), k
    assert np.array_equal(dev1.stepsize(), dev2.stepsize())


def test_chain_offset_sharding_is_partition_independent(pkg):
    """Chains [0,8) in one context == chains [0,4) and [4,8) in two contexts (multi-GPU shards)."""
    D = 40
    full = pkg.DeviceContext(D, 8, seed=3)
    lo = pkg.DeviceContext(D, 4, seed=3, chain_offset=0)
    hi = pkg.DeviceContext(D, 4, seed=3, chain_offset=4)
    outs = []
    for d in (full, lo, hi):
        d.init(); d.find_initial_stepsize()
        outs.append(d.run(10, da={}))
    for k in outs[0]:
        assert np.array_equal(outs[0][k], np.concatenate([outs[1][k], outs[2][k]], axis=0)), k


def test_diag_normal_target(pkg):
    rng = np.random.default_rng(0)
    D, C = 70, 5
    mu = rng.normal(size=D); prec = 1 / (rng.normal(size=D) ** 2 + 0.1)
    params = np.concatenate([mu, prec])
    dev, ora = make_pair(pkg, D, C, target=ol.TARGET_DIAG_NORMAL, params=params, seed=21)
    dev.init(); ora.init()
    dev.find_initial_stepsize(); ora.find_initial_stepsize()
    a, b = dev.run(25, da={}), ora.run(25, da={})
    assert_same(a, b, "diag normal")


def test_tridiag_normal_target(pkg):
    # AR(1)-type precision: correlated MVN with O(D) gradient (BASELINE config 3's target)
    for D in (7, 130):
        rho = 0.5
        diag = np.full(D, (1 + rho ** 2) / (1 - rho ** 2)); diag[0] = diag[-1] = 1 / (1 - rho ** 2)
        off = np.full(D, -rho / (1 - rho ** 2))
        params = np.concatenate([diag, off])
        dev, ora = make_pair(pkg, D, 4, target=ol.TARGET_TRIDIAG_NORMAL, params=params, seed=8)
        dev.init(); ora.init()
        dev.find_initial_stepsize(); ora.find_initial_stepsize()
        a, b = dev.run(20, da={}), ora.run(20, da={})
        assert_same(a, b, f"tridiag D={D}")


def test_funnel_target_divergences(pkg):
    """Neal's funnel D=30 (BASELINE config 4): per-chain divergent tree depths and divergences."""
    D, C = 30, 64
    dev, ora = make_pair(pkg, D, C, target=ol.TARGET_FUNNEL, seed=4)
    dev.init(); ora.init()
    dev.find_initial_stepsize(); ora.find_initial_stepsize()
    a, b = dev.run(40, da={}), ora.run(40, da={})
    assert_same(a, b, "funnel warmup")
    a, b = dev.run(40), ora.run(40)
    assert_same(a, b, "funnel inference")
    assert len(np.unique(a["depth"])) >= 3          # depths really differ across chains


def test_always_divergent(pkg):  # test_NUTS.jl:75-85 through the HIP path
    dev = pkg.DeviceContext(3, 4, target=ol.TARGET_ALWAYS_DIVERGENT)
    dev.init(np.zeros((4, 3)))
    dev.set_stepsize(1.0)
    r = dev.run(2)
    assert (r["term_left"] == r["term_right"]).all()
    assert (r["acceptance_rate"] == 0).all() and (r["depth"] == 0).all() and (r["steps"] == 1).all()
    assert (r["draws"] == 0).all()


def test_max_depth_and_min_delta_config(pkg):
    D, C = 20, 6
    for md in (1, 2, 3):
        dev, ora = make_pair(pkg, D, C, seed=1, max_depth=md)
        dev.init(); ora.init()
        dev.set_stepsize(0.01); ora.set_stepsize(0.01)   # tiny ϵ: every tree hits max_depth
        a, b = dev.run(6), ora.run(6)
        assert_same(a, b, f"max_depth={md}")
        assert (a["depth"] == md).all() and (a["term_left"] == 1).all() and (a["term_right"] == 0).all()
        assert (a["steps"] == 2 ** md - 1).all()


def test_error_mapping(pkg):
    with pytest.raises(ValueError):
        pkg.DeviceContext(10, 2, max_depth=0)         # NUTS.jl:190
    with pytest.raises(ValueError):
        pkg.DeviceContext(10, 2, max_depth=33)
    with pytest.raises(ValueError):
        pkg.DeviceContext(10, 2, min_delta=1.0)       # NUTS.jl:191
    dev = pkg.DeviceContext(10, 2)
    with pytest.raises(pkg.DynamicHMCError):          # hamiltonian.jl:203
        dev.init(np.full((2, 10), np.nan))
    dev.init()
    with pytest.raises(ValueError):                   # stepsize.jl:135: ϵ unspecified
        dev.run(1)
    with pytest.raises(ValueError):
        dev.set_stepsize(-1.0)
    dev.set_stepsize(0.5)
    with pytest.raises(ValueError):                   # mcmc.jl:137: ϵ given, no search
        dev.find_initial_stepsize()
    with pytest.raises(ValueError):
        dev.run(1, da=dict(delta=1.5))                # stepsize.jl:108


def test_state_export_import_resumes_exactly(pkg):
    D, C = 33, 3
    a = pkg.DeviceContext(D, C, seed=12); b = pkg.DeviceContext(D, C, seed=12)
    a.init(); a.find_initial_stepsize(); a.run(10, da={})
    b.import_state(a.export_state())
    ra, rb = a.run(8), b.run(8)
    for k in ra:
        assert np.array_equal(ra[k], rb[k]), k


def _logistic_problem(N, D, seed):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, D)) / np.sqrt(D)
    beta = rng.normal(size=D)
    y = (rng.random(N) < 1 / (1 + np.exp(-X @ beta))).astype(float)
    return X, y


@pytest.mark.parametrize("N,D", [(100, 3), (333, 40), (500, 256)])
def test_logistic_regression_target(pkg, N, D):
    """BASELINE config 5's model (device-side ∇log π over a resident design matrix), small sizes.
    Also the only family that keeps ∇ℓ with stored proposals instead of recomputing it."""
    X, y = _logistic_problem(N, D, N + D)
    params = ol.target_params_blob(ol.TARGET_LOGISTIC, D, X=X, y=y)
    dev, ora = make_pair(pkg, D, 4, target=ol.TARGET_LOGISTIC, params=params, seed=17)
    dev.init(); ora.init()
    for x, z in zip(dev.position(), ora.position()):
        assert np.array_equal(x, z)
    dev.find_initial_stepsize(); ora.find_initial_stepsize()
    assert np.array_equal(dev.stepsize(), ora.stepsize())
    a, b = dev.run(12, da={}), ora.run(12, da={})
    assert_same(a, b, f"logistic N={N} D={D} warmup")
    dev.update_metric_diag(a["draws"]); ora.update_metric_diag(b["draws"])
    a, b = dev.run(8), ora.run(8)
    assert_same(a, b, f"logistic N={N} D={D}")
    for x, z in zip(dev.position(), ora.position()):
        assert np.array_equal(x, z)            # ∇ℓ carried with the proposal == ∇ℓ re-evaluated


@pytest.mark.parametrize("D", [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024])
def test_ragged_dimensions_at_slot_boundaries(pkg, D):
    """Every boundary of the lane/slot layout (Dpad = 64·NPL, NPL ∈ {1,2,4,8,16}), single and few chains."""
    for C in (1, 3):
        dev, ora = make_pair(pkg, D, C, seed=1000 + D)
        dev.init(); ora.init()
        dev.find_initial_stepsize(); ora.find_initial_stepsize()
        a, b = dev.run(6, da={}), ora.run(6, da={})
        assert_same(a, b, f"D={D} C={C}")
        dev.update_metric_diag(a["draws"]); ora.update_metric_diag(b["draws"])
        assert np.array_equal(dev.metric_diag(), ora.metric_diag())
        assert_same(dev.run(4), ora.run(4), f"D={D} C={C} fixed")


def test_empty_and_unrecorded_runs(pkg):
    dev, ora = make_pair(pkg, 10, 2, seed=3)
    dev.init(); ora.init(); dev.find_initial_stepsize(); ora.find_initial_stepsize()
    assert dev.run(0)["draws"].shape == (2, 0, 10)                 # N = 0 is a no-op
    dev.run(5, fields=[]); ora.run(5, fields=[])                   # nothing recorded: state still advances identically
    assert np.array_equal(dev.position()[0], ora.position()[0])
    with pytest.raises(RuntimeError, match="unsupported"):
        pkg.DeviceContext(4097, 1)                                 # D > 4096 is outside this build (DHMC_ERR_UNSUPPORTED)
    big, obig = make_pair(pkg, 1030, 2, target=ol.TARGET_ALWAYS_DIVERGENT, seed=3)      # beyond 1024 coordinates every family runs through
    q0 = np.zeros((2, 1030))                                                         # the batched-evaluation engine: the reference's
    big.init(q0); obig.init(q0); big.set_stepsize(0.3); obig.set_stepsize(0.3)       # AlwaysDivergentTest (test_NUTS.jl:58-85) there
    a, b = big.run(3), obig.run(3)
    assert_same(a, b, "always divergent, D = 1030")
    assert (a["steps"] == 1).all() and (a["depth"] == 0).all() and (a["acceptance_rate"] == 0).all()
    with pytest.raises(ValueError):
        dev.run(-1)


def _shard_worker(rank, world, port, total, D, N, outdir):
    import os, sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # two ranks share the box's one GPU: RCCL needs one device per rank
    off, cnt = pkg.sharding.shard_chains(total, world, rank)
    dev = pkg.DeviceContext(D, cnt, seed=77, chain_offset=off)
    dev.init(); dev.find_initial_stepsize()
    r = dev.run(N, da={})
    draws = pkg.sharding.gather_chain_major(torch.from_numpy(r["draws"]), dist, total, world)
    steps = pkg.sharding.gather_chain_major(torch.from_numpy(r["steps"]), dist, total, world)
    rate = pkg.sharding.job_throughput(int(r["steps"].sum()), 2.0 + rank, dist)
    if rank == 0:
        np.savez(os.path.join(outdir, "gathered.npz"), draws=draws.numpy(), steps=steps.numpy(), rate=rate)
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_hip_shards_equal_single_context(pkg, tmp_path):
    """The multi-GPU layout with the HIP path as the per-rank engine: two PROCESSES (both on this box's GPU), contiguous
    chain blocks with the block offset as RNG chain_offset, results gathered through sharding.gather_chain_major — equal,
    chain for chain, to one context holding all chains (the HIP counterpart of tests/test_distributed_gloo.py)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    total, D, N, world = 37, 1000, 6, 2                         # ragged blocks of 19 and 18 chains, the headline width
    mp.spawn(_shard_worker, args=(world, port, total, D, N, str(tmp_path)), nprocs=world, join=True)
    g = np.load(tmp_path / "gathered.npz")
    single = pkg.DeviceContext(D, total, seed=77)
    single.init(); single.find_initial_stepsize()
    r = single.run(N, da={})
    assert np.array_equal(g["draws"], r["draws"]) and np.array_equal(g["steps"], r["steps"])
    assert np.isclose(float(g["rate"]), r["steps"].sum() / 3.0)


def test_set_position_keeps_metric_stepsize_and_streams(pkg):
    """dhmc_set_position: Q := evaluate_ℓ(ℓ, q) at positions of the caller's (what mcmc_next_step(steps, Q) needs for a
    foreign Q, mcmc.jl:348-351) without touching κ, ϵ or the random-stream counters."""
    D, C = 70, 4
    a = pkg.DeviceContext(D, C, seed=5); b = pkg.DeviceContext(D, C, seed=5)
    for d in (a, b):
        d.init(); d.find_initial_stepsize()
        r = d.run(20, da={}); d.update_metric_diag(r["draws"]); d.run(2)
    b.set_position(a.position()[0])                       # its own position handed back in: nothing may change
    ra, rb = a.run(3), b.run(3)
    for k in ra:
        assert np.array_equal(ra[k], rb[k]), k
    qx = np.random.default_rng(1).normal(size=(C, D))
    m0, e0 = a.metric_diag(), a.stepsize()
    a.set_position(qx)
    fresh = pkg.DeviceContext(D, C, seed=99); fresh.init(qx)
    for x, y in zip(a.position(), fresh.position()):
        assert np.array_equal(x, y)                       # (q, ℓq, ∇ℓq) of the new point
    assert np.array_equal(a.metric_diag(), m0) and np.array_equal(a.stepsize(), e0)
    with pytest.raises(ValueError):
        a.set_stepsize(np.ones(3))                        # neither a scalar nor one value per chain
    import torch
    with pytest.raises(ValueError):                       # hamiltonian.jl:63 for a device array too
        a.set_metric_diag(torch.full((C, D), -1.0, dtype=torch.float64, device="cuda"))
    assert np.array_equal(a.metric_diag(), m0)


def test_dense_context_keeps_its_metric_across_init(pkg):
    """include/dhmc.h: dhmc_init resets the per-chain diagonal metric to the unit one, a dense context keeps its shared
    M⁻¹ (initialize_warmup_state takes κ as a keyword, mcmc.jl:129) — as the oracle's restatement does."""
    D = 6
    rng = np.random.default_rng(2)
    A = rng.normal(size=(D, D)); S = A @ A.T + D * np.eye(D)
    d = pkg.DeviceContext(D, 3, metric=ol.METRIC_DENSE, seed=1)
    assert np.array_equal(d.metric_dense()[0], np.eye(D))          # GaussianKineticEnergy(N) at creation
    d.set_metric_dense(S); before = d.metric_dense()
    d.init()
    after = d.metric_dense()
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    e = pkg.DeviceContext(D, 3, seed=1)
    e.init(); e.set_metric_diag(np.full(D, 3.0)); e.init()
    assert np.array_equal(e.metric_diag(), np.ones((3, D)))        # the diagonal one is reset


@pytest.mark.parametrize("family,D,metric", [("std", 1500, "diag"), ("diag", 2500, "diag"), ("tridiag", 1100, "diag"), ("tridiag", 1100, "dense"),
                                             ("funnel", 1300, "diag"), ("mvnormal", 1100, "diag"), ("logistic", 1100, "diag")])
def test_builtin_normal_families_beyond_1024_dimensions(pkg, family, D, metric):
    """The reference has no dimension limit (hamiltonian.jl:56-87).  Beyond the register-resident kernels' 1024 coordinates
    the built-in normal families and the funnel run through the streaming round engine (32 / 64 slots per lane, K3 as 8 / 16
    waves per chain) with their density evaluated for all chains between the kernels (builtin_normal_eval_kernel; the full-
    precision normal as one product (q − μ)·P over the chains; the logistic regression's gradient as the two GEMMs of its round
    engine over all chains): init, step-size search, adaptive stage with a metric update and
    a fixed stage, bit for bit against the oracle."""
    rng = np.random.default_rng(D)
    C = 3
    if family == "std":
        tgt, params = ol.TARGET_STD_NORMAL, None
    elif family == "diag":
        tgt = ol.TARGET_DIAG_NORMAL
        params = ol.target_params_blob(tgt, D, mu=rng.normal(size=D), prec=1 / (rng.normal(size=D) ** 2 + 0.1))
    elif family == "funnel":
        tgt, params = ol.TARGET_FUNNEL, None
    elif family == "logistic":                                 # 2 500 observations: two blocks of the ABI's Σ over observations
        tgt = ol.TARGET_LOGISTIC
        X = rng.normal(size=(2500, D)) / 30
        y = (rng.random(2500) < 1 / (1 + np.exp(-X @ rng.normal(size=D)))).astype(float)
        params = ol.target_params_blob(tgt, D, X=X, y=y)
    elif family == "mvnormal":
        tgt = ol.TARGET_DENSE_NORMAL
        idx = np.arange(D)
        P = np.diag(np.linspace(0.5, 2.0, D)) + 0.3 * 0.5 ** np.abs(idx[:, None] - idx[None, :])
        params = ol.target_params_blob(tgt, D, mu=rng.normal(size=D), P=P)
    else:
        tgt = ol.TARGET_TRIDIAG_NORMAL
        params = ol.target_params_blob(tgt, D, diag=np.full(D, 2.5), off=np.full(D - 1, -1.0))
    m = ol.METRIC_DENSE if metric == "dense" else ol.METRIC_DIAG
    dev, ora = make_pair(pkg, D, C, target=tgt, params=params, seed=4, metric=m)
    if metric == "dense":
        idx = np.arange(D)
        S = 0.5 ** np.abs(idx[:, None] - idx[None, :]) + 0.5 * np.eye(D)
        dev.set_metric_dense(S); ora.set_metric_dense(S)
    dev.init(); ora.init()
    for x, y in zip(dev.position(), ora.position()):
        assert np.array_equal(x, y)
    dev.find_initial_stepsize(); ora.find_initial_stepsize()
    assert np.array_equal(dev.stepsize(), ora.stepsize())
    a, b = dev.run(8, da={}), ora.run(8, da={})
    assert_same(a, b, f"{family} D={D} {metric} adaptive")
    if metric == "diag":
        dev.update_metric_diag(a["draws"]); ora.update_metric_diag(b["draws"])
        assert np.array_equal(dev.metric_diag(), ora.metric_diag())
    assert_same(dev.run(5), ora.run(5), f"{family} D={D} {metric} fixed")


def test_host_outputs_in_chunks_equal_device_outputs(pkg, monkeypatch):
    """dhmc_run with host result buffers leaves in chunks of transitions over a copy stream, two staging buffers deep
    (csrc/dhmc_capi.hip); the chunks are the same transitions of the same kernel: pageable and page-locked destinations, a
    chunk length that does not divide N, adaptation across the chunks — all equal to one launch into device buffers."""
    import torch
    from dynamichmc_jl_amd.context import pinned_empty
    D, C, N = 70, 9, 11
    fields = pkg.abi.OUTPUT_FIELDS
    tdt = {np.float64: torch.float64, np.int64: torch.int64, np.int32: torch.int32, np.uint32: torch.int32}

    def run(kind, chunk, engine="diag"):
        if chunk:
            monkeypatch.setenv("DHMC_HOST_CHUNK", str(chunk))
        else:
            monkeypatch.delenv("DHMC_HOST_CHUNK", raising=False)
        if engine == "diag":
            ctx = pkg.DeviceContext(D, C, seed=12)
        elif engine == "dense":                                # the GEMM round engine (one-product recurrence): a call per chunk
            ctx = pkg.DeviceContext(D, C, seed=12, metric=ol.METRIC_DENSE)
            ctx.set_metric_dense(np.diag(np.linspace(0.5, 2.0, D)) + 0.05)
        else:                                                  # the batched-evaluation engine (external models, D > 1024)
            ctx = pkg.DeviceContext(1100, C, seed=12)
        ctx.init(); ctx.find_initial_stepsize()
        Dd = ctx.D
        res = []
        for da in ({}, None):
            shape = lambda k: (C, N, Dd) if k == "draws" else (C, N)
            if kind == "device":
                arrs = {k: torch.zeros(shape(k), dtype=tdt[dt], device="cuda") for k, dt in fields}
            elif kind == "pinned":
                arrs = {k: pinned_empty(shape(k), dt) for k, dt in fields}
            else:
                arrs = {k: np.zeros(shape(k), dt) for k, dt in fields}
            ctx.run_into(N, arrs, da=da)
            res.append({k: (a.cpu().numpy().view(dt) if kind == "device" else np.array(a)) for (k, dt), a in zip(fields, arrs.values())})
        res.append({"eps": ctx.stepsize(), "q": ctx.position()[0]})
        ctx.close()
        return res

    ref = run("device", 0)
    for kind, chunk in (("pageable", 0), ("pageable", 4), ("pinned", 4), ("pinned", 1), ("pinned", 11), ("pinned", 50)):
        got = run(kind, chunk)
        for a, b in zip(ref, got):
            for k in a:
                assert np.array_equal(a[k], b[k]), (kind, chunk, k)
    # the round engines: with host outputs a long call runs as calls of L transitions (dhmc_run), chunk k leaving while k + 1 computes
    for engine in ("dense", "big"):
        ref = run("device", 0, engine)
        for kind, chunk in (("pageable", 4), ("pinned", 3), ("pinned", 1)):
            got = run(kind, chunk, engine)
            for a, b in zip(ref, got):
                for k in a:
                    assert np.array_equal(a[k], b[k]), (engine, kind, chunk, k)


def test_step_reports_do_not_change_the_run(pkg):
    """test/test_logging.jl runs mcmc_with_warmup with every reporter.  A reporter that wants step reports (reporting.jl:120-137,
    mcmc.jl:279,378) makes a stage run as calls of `step_interval` transitions with a report after each: the same transitions, the
    same bits as one call per stage — diagonal and Symmetric warmup — and the lines the reference's log would show."""
    import io
    l = pkg.DiagNormal(np.linspace(-1, 1, 7), np.linspace(0.5, 2, 7))
    for M in (pkg.Diagonal, pkg.Symmetric):
        stages = lambda: pkg.default_warmup_stages(M=M, middle_steps=20, doubling_stages=3)
        ref = pkg.mcmc_with_warmup(4, l, 230, chains=5, warmup_stages=stages(), reporter=pkg.NoProgressReport())
        lines = []
        got = pkg.mcmc_with_warmup(4, l, 230, chains=5, warmup_stages=stages(), reporter=pkg.LogProgressReport(step_interval=60, printer=lines.append))
        bar = pkg.mcmc_with_warmup(4, l, 230, chains=5, warmup_stages=stages(), reporter=pkg.ProgressMeterReport(updates=7, stream=io.StringIO()))
        for r in (got, bar):
            assert np.array_equal(ref["posterior_matrix"], r["posterior_matrix"]) and np.array_equal(ref["eps"], r["eps"])
            assert np.array_equal(ref["tree_statistics"].steps, r["tree_statistics"].steps)
            assert np.array_equal(ref["kappa"].Minv, r["kappa"].Minv)
        assert any("found initial stepsize" in x for x in lines) and any("adaptation finished" in x for x in lines)
        starts = [x for x in lines if "Starting MCMC" in x]
        assert len(starts) == len(stages()) - 1 + 1 and "total_steps = 230" in starts[-1]           # every TuningNUTS stage + inference
        inf = [x for x in lines[lines.index(starts[-1]):] if "MCMC progress" in x]
        assert [int(x.split("step = ")[1].split(",")[0]) for x in inf] == [60, 120, 180]        # (230 − 180 < step_interval)
        assert any("MCMC progress" in x and "ϵ = " in x for x in lines)                                # warmup steps carry ϵ
