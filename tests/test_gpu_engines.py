"""GPU: the alternative engines of the library produce the same bits.  The production paths (LDS-resident
diagonal kernel, dense round engine with MFMA GEMMs, GEMM-gradient round engine for logistic regression) are
checked against the independent wave-per-chain kernels selected by the documented tuning knobs."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_package()


def _run_with_env(pkg, env, make, steps):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = make()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return steps(ctx)


def _same(a, b):
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_diag_kernel_lds_variants_agree(pkg):
    def steps(ctx):
        ctx.init(); ctx.find_initial_stepsize()
        r = ctx.run(30, da={})
        ctx.update_metric_diag(r["draws"])
        return ctx.run(20)
    make = lambda: pkg.DeviceContext(1000, 32, seed=5)
    _same(_run_with_env(pkg, {"DHMC_L1_LDS": "1"}, make, steps), _run_with_env(pkg, {"DHMC_L1_LDS": "0"}, make, steps))


def test_lds_trajectory_layout_with_a_target_that_is_not_coordinate_wise(pkg):
    """VERDICT r2 #5 / docs/DESIGN_history_rounds1-4.md §10: the layout of the wide per-draw kernel that keeps the trajectory's edge momenta in LDS
    and M⁻¹ in registers was restricted to coordinate-wise targets in round 2, after a fault in a fuzz sweep with the
    tridiagonal-precision normal at D = 1000.  The fault does not reproduce (19 000 tridiagonal cases at 700 <= D <= 1024,
    the whole GPU suite and the fuzz sweep with the layout forced on for every family: tools/experiments/tpl_fault_repro.py,
    profiles/r03_lds_layout_fault.txt), the restriction is gone, and this is the regression test: that family, D = 1000 and
    1024, through adaptation, a metric update, divergent and depth-limited trees — against the oracle and against the
    other (DHMC_L1_LDS=0) layout."""
    for D in (1000, 1024):
        rng = np.random.default_rng(D)
        params = ol.target_params_blob(ol.TARGET_TRIDIAG_NORMAL, D, diag=np.full(D, 2.0 + rng.random()), off=np.full(D - 1, -0.9 * rng.random()))

        def steps(ctx):
            out = {}
            ctx.init(); ctx.find_initial_stepsize()
            a = ctx.run(12, da={})
            ctx.update_metric_diag(a["draws"])
            out.update({"w_" + k: v for k, v in a.items()})
            out.update({"i_" + k: v for k, v in ctx.run(6).items()})
            ctx.set_stepsize(3.0)                       # most leaves diverge
            out.update({"d_" + k: v for k, v in ctx.run(4).items()})
            ctx.set_stepsize(2e-3)                      # every tree is cut at max_depth
            out.update({"m_" + k: v for k, v in ctx.run(2).items()})
            return out
        make = lambda: pkg.DeviceContext(D, 9, target=ol.TARGET_TRIDIAG_NORMAL, target_params=params, seed=33, max_depth=6)
        a = _run_with_env(pkg, {"DHMC_L1_LDS": "1"}, make, steps)
        b = _run_with_env(pkg, {"DHMC_L1_LDS": "0"}, make, steps)
        o = steps(ol.Oracle(D, 9, target=ol.TARGET_TRIDIAG_NORMAL, params=params, seed=33, max_depth=6, threads=9))
        _same(a, b)
        _same(a, o)
        assert (a["m_depth"] == 6).all()


@pytest.mark.parametrize("products", [1, 2])
def test_dense_round_engine_equals_wave_kernel(pkg, products):
    """Both dense engines run both recurrences (the reference's two products per leapfrog, or one with u = M⁻¹∇ℓ carried) with the
    same bits, so which of them serves a context — the wave-per-chain kernel for narrow chains, the GEMM rounds for wide ones or very
    many — is a matter of speed only."""
    rng = np.random.default_rng(3)
    K = 96
    A = rng.normal(size=(K, K)); Minv = np.linalg.inv(A.T @ A / K + 0.1 * np.eye(K))

    def steps(ctx):
        ctx.set_dense_products(products)
        ctx.set_metric_dense(Minv); ctx.init(); ctx.find_initial_stepsize()
        a = ctx.run(25, da={})
        b = ctx.run(15)
        return {**{"w_" + k: v for k, v in a.items()}, **b}
    make = lambda: pkg.DeviceContext(K, 300, metric=ol.METRIC_DENSE, seed=9)        # 300 chains: two half-batches of 150
    _same(_run_with_env(pkg, {"DHMC_DENSE": "rounds=1"}, make, steps), _run_with_env(pkg, {"DHMC_DENSE": "rounds=0"}, make, steps))


@pytest.mark.parametrize("N,D,C", [(777, 70, 40), (6000, 70, 40), (4100, 200, 70), (2048, 64, 33), (2049, 64, 130)],
                         ids=["one block of observations", "three blocks", "three blocks, two row tiles", "exactly one block",
                              "one block and one observation, three row tiles"])
def test_logistic_round_engine_equals_functor_kernel(pkg, N, D, C):
    """The round engine (Q′·Xᵀ and the split-K R·X over the rows of the chains still running, link and block sums, the
    blocks folded in order by K2) against the wave-per-chain functor, which walks the same blocks of DHMC_LOGISTIC_BLOCK
    observations one chain at a time.  Chains finish their transitions at different rounds, so the row list shrinks."""
    rng = np.random.default_rng(4)
    X = rng.normal(size=(N, D)) / 8; y = (rng.random(N) < 0.5).astype(float)
    params = ol.target_params_blob(ol.TARGET_LOGISTIC, D, X=X, y=y)

    def steps(ctx):
        ctx.init(); ctx.find_initial_stepsize()
        a = ctx.run(20, da={})
        ctx.update_metric_diag(a["draws"])
        return {**{"w_" + k: v for k, v in a.items()}, **ctx.run(12)}
    make = lambda: pkg.DeviceContext(D, C, target=ol.TARGET_LOGISTIC, target_params=params, seed=2)
    _same(_run_with_env(pkg, {"DHMC_LOGISTIC_ROUNDS": "1"}, make, steps), _run_with_env(pkg, {"DHMC_LOGISTIC_ROUNDS": "0"}, make, steps))


@pytest.mark.parametrize("metric", ["diag", "dense"])
def test_launch_order_changes_no_result(pkg, metric):
    """The per-draw kernels start their chains in the order of the previous launch's work, longest first (csrc/nuts_kernels.hpp
    RunParams::launch_order; dhmc_capi.hip run_call), when a chain did more than 3 % above the mean — Neal's funnel, whose chains
    adapt to very different step sizes, always does.  Chains are independent: the order must not show in any output."""
    D, C = 30, 300

    def steps(ctx):
        if metric == "dense":
            ctx.set_metric_dense(np.diag(np.linspace(0.5, 2.0, D)) + 0.02)
        ctx.init(); ctx.find_initial_stepsize()
        a = ctx.run(40, da={})                         # identity order; its work decides the next launch's order
        b = ctx.run(40)
        # … through dual averaging (continued from the first stage), an open metric window, and host outputs that leave in chunks
        c = ctx.run(100, da=dict(init=0))
        res = {**{"a_" + k: v for k, v in a.items()}, **{"b_" + k: v for k, v in b.items()}, **{"c_" + k: v for k, v in c.items()}}
        if metric == "diag":
            ctx.metric_window_begin()
            d = ctx.run(90)
            ctx.update_metric_diag_window()
            res.update({"d_" + k: v for k, v in d.items()}, minv=ctx.metric_diag())
        e = ctx.run(150)
        res.update({"e_" + k: v for k, v in e.items()})
        return {**res, "q": ctx.position()[0], "eps": ctx.stepsize(), "work_spread": np.array([a["steps"].sum(1).max() / a["steps"].sum(1).mean()])}
    kw = dict(metric=ol.METRIC_DENSE) if metric == "dense" else {}
    make = lambda: pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=6, **kw)
    env = {"DHMC_DENSE": "rounds=0", "DHMC_HOST_CHUNK": "70"}
    on = _run_with_env(pkg, {**env, "DHMC_LAUNCH_ORDER": "1"}, make, steps)
    off = _run_with_env(pkg, {**env, "DHMC_LAUNCH_ORDER": "0"}, make, steps)
    assert on["work_spread"][0] > 1.03                 # (the reordering did take place)
    _same(on, off)


@pytest.mark.parametrize("D", [500, 1000])
def test_block_per_chain_k3_equals_wave_per_chain_k3(pkg, D):
    """Round engines, chains of 512+ coordinates: K3 as a 4-wave workgroup per chain (dots chained from wave to wave
    in the ABI's order) against the one-wave-per-chain K3 (DHMC_DENSE="k3_block=0")."""
    rng = np.random.default_rng(3)
    A = rng.normal(size=(D, 40))
    S = A @ A.T / 40 + np.eye(D)
    params = ol.target_params_blob(ol.TARGET_TRIDIAG_NORMAL, D, diag=np.full(D, 2.5), off=np.full(D - 1, -1.0))

    def steps(ctx):
        ctx.set_metric_dense(S); ctx.init(); ctx.find_initial_stepsize()
        a = ctx.run(12, da={})
        return {**{"w_" + k: v for k, v in a.items()}, **ctx.run(8)}
    make = lambda: pkg.DeviceContext(D, 24, target=ol.TARGET_TRIDIAG_NORMAL, target_params=params, metric=ol.METRIC_DENSE, seed=5)
    _same(_run_with_env(pkg, {"DHMC_DENSE": "rounds=1,k3_block=1"}, make, steps), _run_with_env(pkg, {"DHMC_DENSE": "rounds=1,k3_block=0"}, make, steps))


def test_position_overflow_matches_oracle(pkg):
    """ϵ so large that q′ overflows: ℓq is non-finite, the kernel scans the position (hamiltonian.jl:203), the
    chain's status word records it, the leaf is divergent — against the oracle."""
    D, C = 1000, 3
    dev = pkg.DeviceContext(D, C, seed=8); ora = ol.Oracle(D, C, seed=8, threads=4)
    dev.init(); ora.init()
    dev.set_stepsize(1e308); ora.set_stepsize(1e308)
    a, b = dev.run(3, allow_failure=True), ora.run(3, allow_failure=True)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert np.array_equal(dev.status(), ora.status()) and (dev.status() != 0).any()
    for x, y in zip(dev.position(), ora.position()):
        assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("parts", ["1", "2", "4"])
def test_dense_rounds_row_lists_change_nothing_but_the_rows_multiplied(pkg, parts):
    """Dense round engine with unequal trees (a metric without the target's correlation): from the moment the first
    chains of a call have finished, the products run over the row list of the chains still running.  Same bits as
    multiplying every row every round (DHMC_DENSE="row_lists=0"), with the batch run as 1, 2 or 4 parts."""
    D, C = 96, 512
    rho = 0.8
    sig = np.logspace(-0.5, 0.5, D)
    Pc = np.zeros(D) + (1 + rho ** 2) / (1 - rho ** 2); Pc[0] = Pc[-1] = 1 / (1 - rho ** 2)
    params = ol.target_params_blob(ol.TARGET_TRIDIAG_NORMAL, D, diag=Pc / sig ** 2, off=-rho / (1 - rho ** 2) / (sig[:-1] * sig[1:]))

    def steps(ctx):
        ctx.set_metric_dense(np.diag(sig ** 2)); ctx.init(); ctx.find_initial_stepsize()
        a = ctx.run(12, da={})
        b = ctx.run(8)
        assert b["steps"].sum(1).min() < 0.7 * b["steps"].sum(1).max()        # the chains do finish at different rounds
        return {**{"w_" + k: v for k, v in a.items()}, **b}
    make = lambda: pkg.DeviceContext(D, C, metric=ol.METRIC_DENSE, target=ol.TARGET_TRIDIAG_NORMAL, target_params=params, seed=21)
    env = "rounds=1,parts=%s," % parts
    _same(_run_with_env(pkg, {"DHMC_DENSE": env + "row_lists=1"}, make, steps),
          _run_with_env(pkg, {"DHMC_DENSE": env + "row_lists=0"}, make, steps))
