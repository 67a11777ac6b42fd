"""GPU: `north_star` — "the user ∇log π is supplied as a device function".  A functor a caller wrote (tests/user_functors.py) is
compiled at run time into nuts_run_kernel / init_kernel / stepsize_search_kernel (dhmc_register_target_source) and runs with no
host round trip per leapfrog."""
import numpy as np
import pytest

import oracle_lib as ol
import user_functors as uf
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_package()


@pytest.mark.parametrize("D", [5, 200, 1000, 1500, 2500])
def test_user_written_diag_normal_is_bit_equal_to_the_builtin_family_and_the_oracle(pkg, D):
    """(1500, 2500: beyond the register-resident kernels — the functor's eval() compiled into functor_eval_kernel, which the streaming
    round engine calls where an external model's callback would stand; the reference puts no limit on the dimension,
    hamiltonian.jl:146-147.)"""
    rng = np.random.default_rng(D)
    mu = rng.normal(size=D); prec = np.exp(rng.normal(size=D))
    user = pkg.DeviceFunctorLogDensity(D, uf.DIAG_NORMAL, "MyDiagNormal", params=np.concatenate([mu, prec]))
    C = 6 if D <= 1024 else 3
    n1, n2 = (25, 12) if D <= 1024 else (8, 5)

    def steps(ctx):
        out = {}
        ctx.init(); ctx.find_initial_stepsize()
        out["eps0"] = ctx.stepsize()
        a = ctx.run(n1, da={})
        ctx.update_metric_diag(a["draws"])
        out.update({"w_" + k: v for k, v in a.items()})
        out.update({"i_" + k: v for k, v in ctx.run(n2).items()})
        q, lq, g = ctx.position()
        out.update(q=q, lq=lq, g=g)
        return out
    a = steps(pkg.DeviceContext(D, C, target=user.family, target_params=user.params(), seed=3))
    b = steps(pkg.DeviceContext(D, C, target=ol.TARGET_DIAG_NORMAL, target_params=ol.target_params_blob(ol.TARGET_DIAG_NORMAL, D, mu=mu, prec=prec), seed=3))
    o = steps(ol.Oracle(D, C, target=ol.TARGET_DIAG_NORMAL, params=ol.target_params_blob(ol.TARGET_DIAG_NORMAL, D, mu=mu, prec=prec), seed=3, threads=6))
    if D > 1024:          # … and with the shared dense metric: the engine's products are the MFMA GEMMs, the density still the functor's
        idx = np.arange(D)
        S = 0.5 ** np.abs(idx[:, None] - idx[None, :]) + 0.5 * np.eye(D)
        res = []
        for mk in (lambda: pkg.DeviceContext(D, C, target=user.family, target_params=user.params(), seed=5, metric=ol.METRIC_DENSE),
                   lambda: ol.Oracle(D, C, target=ol.TARGET_DIAG_NORMAL, params=ol.target_params_blob(ol.TARGET_DIAG_NORMAL, D, mu=mu, prec=prec), seed=5,
                                     metric=ol.METRIC_DENSE, threads=6)):
            e = mk()
            e.set_metric_dense(S); e.init(); e.find_initial_stepsize()
            res.append((e.stepsize(), e.run(5, da={}), e.run(3)))
        assert np.array_equal(res[0][0], res[1][0])
        for x, y in ((res[0][1], res[1][1]), (res[0][2], res[1][2])):
            for k in x:
                assert np.array_equal(x[k], y[k]), k
    for k in a:
        assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(a[k], o[k]), k


def test_a_model_of_the_callers_own_through_mcmc_with_warmup(pkg):
    """Independent Student-t(5) coordinates with scales 0.5 … 4: no built-in family; posterior moments from the drop-in API."""
    D, nu = 12, 5.0
    scale = np.linspace(0.5, 4.0, D)
    l = pkg.DeviceFunctorLogDensity(D, uf.STUDENT_T, "StudentT", params=np.concatenate([[nu], scale]))
    r = pkg.mcmc_with_warmup(7, l, 1500, chains=64, reporter=pkg.NoProgressReport())
    x = r["posterior_matrix"].reshape(-1, D)
    sd = scale * np.sqrt(nu / (nu - 2))
    assert np.abs(x.mean(0) / sd).max() < 0.06
    assert np.abs(x.std(0) / sd - 1).max() < 0.12
    assert 0.6 < r["tree_statistics"].acceptance_rate.mean() < 0.97
    r2 = pkg.mcmc_with_warmup(7, l, 1500, chains=64, reporter=pkg.NoProgressReport())
    assert np.array_equal(r["posterior_matrix"], r2["posterior_matrix"])


def test_a_broken_functor_fails_at_create_with_the_compilers_log(pkg):
    l = pkg.DeviceFunctorLogDensity(4, uf.BROKEN, "Broken")
    with pytest.raises(Exception):
        pkg.DeviceContext(4, 2, target=l.family)
    assert b"undeclared_thing" in pkg.abi.lib().dhmc_target_source_log()


def test_diagnostics_probes_of_a_user_functor_equal_the_builtin_family(pkg):
    """Diagnostics.leapfrog_trajectory / explore_log_acceptance_ratios (diagnostics.jl:144-227) for the caller's functor: the probe
    kernels are compiled from its source as well."""
    D, C = 130, 4
    rng = np.random.default_rng(1)
    mu = rng.normal(size=D); prec = np.exp(rng.normal(size=D))
    user = pkg.DeviceFunctorLogDensity(D, uf.DIAG_NORMAL, "MyDiagNormal", params=np.concatenate([mu, prec]))
    a = pkg.DeviceContext(D, C, target=user.family, target_params=user.params(), seed=9)
    b = pkg.DeviceContext(D, C, target=ol.TARGET_DIAG_NORMAL, target_params=ol.target_params_blob(ol.TARGET_DIAG_NORMAL, D, mu=mu, prec=prec), seed=9)
    for ctx in (a, b):
        ctx.init(); ctx.find_initial_stepsize(); ctx.run(5, da={})
    ta, tb = a.leapfrog_trajectory(0.1, -5, 7, momentum_index=2), b.leapfrog_trajectory(0.1, -5, 7, momentum_index=2)
    for k in ("delta", "logdensity", "q", "p", "range"):
        assert np.array_equal(ta[k], tb[k]), k
    ra, rb = a.explore_log_acceptance_ratios([0.05, 0.2, 0.8], n_momenta=6), b.explore_log_acceptance_ratios([0.05, 0.2, 0.8], n_momenta=6)
    assert np.array_equal(ra, rb)


@pytest.mark.parametrize("D,C", [(40, 6), (200, 160), (1000, 136), (520, 256)])
def test_user_functor_with_the_dense_metric_is_bit_equal_to_the_builtin_family(pkg, D, C):
    """DHMC_METRIC_DENSE for the caller's functor: 6 chains run the wave-per-chain dense kernels, 136+ the GEMM round engine
    (K0 / K2 / K3 compiled around the functor; at D = 1000 the workgroup-per-chain K3b with the fused position update) — the same
    kernels as the built-in family's, so the same bits; the small case also against the oracle.  256 chains (an even count from
    256 up) run as two half-batches on two streams: every launch must cover its own half only (round 3 launched the functor's
    kernels over all chains of the context for each half)."""
    rng = np.random.default_rng(D)
    mu = rng.normal(size=D); prec = np.exp(rng.normal(size=D))
    A = rng.normal(size=(D, D)) / np.sqrt(D)
    Sigma = np.diag(1 / prec) + 0.1 * (A @ A.T) * np.sqrt(np.outer(1 / prec, 1 / prec))
    user = pkg.DeviceFunctorLogDensity(D, uf.DIAG_NORMAL, "MyDiagNormal", params=np.concatenate([mu, prec]))
    blob = ol.target_params_blob(ol.TARGET_DIAG_NORMAL, D, mu=mu, prec=prec)

    def steps(ctx, n):
        out = {}
        ctx.set_metric_dense(Sigma)
        ctx.init(); ctx.find_initial_stepsize()
        out["eps0"] = ctx.stepsize()
        out.update({"w_" + k: v for k, v in ctx.run(n, da={}).items()})
        out.update({"i_" + k: v for k, v in ctx.run(n // 2).items()})
        q, lq, g = ctx.position()
        out.update(q=q, lq=lq, g=g)
        t = ctx.leapfrog_trajectory(0.1, -3, 4, momentum_index=1)
        out.update(t_delta=t["delta"], t_q=t["q"])
        return out
    n = 12 if D == 1000 else 20
    a = steps(pkg.DeviceContext(D, C, target=user.family, target_params=user.params(), metric=ol.METRIC_DENSE, seed=4), n)
    b = steps(pkg.DeviceContext(D, C, target=ol.TARGET_DIAG_NORMAL, target_params=blob, metric=ol.METRIC_DENSE, seed=4), n)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert (a["i_steps"] >= 3).all()
    if D == 40:
        o = steps(ol.Oracle(D, C, target=ol.TARGET_DIAG_NORMAL, params=blob, metric=ol.METRIC_DENSE, seed=4, threads=6), n)
        for k in a:
            assert np.array_equal(a[k], o[k]), k


def test_user_functor_dense_warmup_through_the_drop_in_api(pkg):
    """mcmc_with_warmup(…; warmup_stages = default_warmup_stages(; M = Symmetric)) (mcmc.jl:475-497) with the caller's functor."""
    D, nu = 12, 5.0
    scale = np.linspace(0.5, 4.0, D)
    l = pkg.DeviceFunctorLogDensity(D, uf.STUDENT_T, "StudentT", params=np.concatenate([[nu], scale]))
    r = pkg.mcmc_with_warmup(11, l, 1200, chains=160, reporter=pkg.NoProgressReport(),
                             warmup_stages=pkg.default_warmup_stages(M=pkg.Symmetric))
    x = r["posterior_matrix"].reshape(-1, D)
    sd = scale * np.sqrt(nu / (nu - 2))
    assert np.abs(x.mean(0) / sd).max() < 0.06
    assert np.abs(x.std(0) / sd - 1).max() < 0.12


def test_code_objects_are_reused_from_the_disk_cache(pkg, tmp_path, monkeypatch):
    """DHMC_RTC_CACHE=<dir>: the compiled module of a functor is kept on disk and the next registration of the same source (the next
    process, in practice) loads it instead of compiling — same kernels, same bits; a corrupt file is ignored and replaced."""
    import glob, time
    monkeypatch.setenv("DHMC_RTC_CACHE", str(tmp_path))
    D = 70
    params = np.concatenate([np.linspace(-1, 1, D), np.linspace(0.5, 2, D)])

    def run():
        user = pkg.DeviceFunctorLogDensity(D, uf.DIAG_NORMAL, "MyDiagNormal", params=params)       # a new handle every time
        t0 = time.perf_counter()
        ctx = pkg.DeviceContext(D, 4, target=user.family, target_params=user.params(), seed=8)
        dt = time.perf_counter() - t0
        log = pkg.abi.lib().dhmc_target_source_log().decode()
        ctx.init(); ctx.find_initial_stepsize()
        out = ctx.run(6)["draws"]
        ctx.close()
        return out, log, dt
    a, log_a, t_a = run()
    files = glob.glob(str(tmp_path / "dhmc_rtc_*.co"))
    assert len(files) == 1 and "loaded from" not in log_a
    b, log_b, t_b = run()
    assert "loaded from" in log_b and np.array_equal(a, b) and t_b < t_a
    with open(files[0], "r+b") as fh:                       # damage the file: it is not trusted, the functor is compiled again
        fh.seek(0); fh.write(b"garbage!")
    c, log_c, _ = run()
    assert "loaded from" not in log_c and np.array_equal(a, c)
    d, log_d, _ = run()
    assert "loaded from" in log_d and np.array_equal(a, d)


@pytest.mark.parametrize("D", [5, 64, 100, 128])
def test_user_functor_through_the_pipeline_kernel(pkg, D, monkeypatch, capfd):
    """A caller's functor (kRecomputeGrad: the gradient is recomputed from a stored position) through the four-wavefront pipeline kernel
    (DHMC_PIPELINE=1; rows of 64 and 128 doubles): the bits of the wave-per-chain kernel and of the oracle."""
    rng = np.random.default_rng(D)
    mu = rng.normal(size=D); prec = np.exp(rng.normal(size=D))
    user = pkg.DeviceFunctorLogDensity(D, uf.DIAG_NORMAL, "MyDiagNormal", params=np.concatenate([mu, prec]))
    C = 5

    def steps(ctx):
        out = {}
        ctx.init(); ctx.set_stepsize(0.05)            # small steps: trees of 31 … 255 leapfrogs
        out["eps0"] = ctx.stepsize()
        ctx.metric_window_begin()
        a = ctx.run(25, da={})
        ctx.update_metric_diag_window()
        out.update({"w_" + k: v for k, v in a.items()})
        out.update({"i_" + k: v for k, v in ctx.run(12).items()})
        q, lq, g = ctx.position()
        out.update(q=q, lq=lq, g=g, minv=ctx.metric_diag())
        return out
    monkeypatch.setenv("DHMC_DEBUG_ORDER", "1")
    monkeypatch.setenv("DHMC_PIPELINE", "1")
    a = steps(pkg.DeviceContext(D, C, target=user.family, target_params=user.params(), seed=3))
    assert "engine: pipeline" in capfd.readouterr().err
    monkeypatch.setenv("DHMC_PIPELINE", "0")
    monkeypatch.setenv("DHMC_PACKED", "0")            # (since round 6 such a functor would otherwise run packed)
    b = steps(pkg.DeviceContext(D, C, target=user.family, target_params=user.params(), seed=3))
    assert "engine: wave" in capfd.readouterr().err
    o = steps(ol.Oracle(D, C, target=ol.TARGET_DIAG_NORMAL, params=ol.target_params_blob(ol.TARGET_DIAG_NORMAL, D, mu=mu, prec=prec), seed=3, threads=5))
    assert a["w_steps"].max() >= 31
    for k in a:
        assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(a[k], o[k]), k


@pytest.mark.parametrize("D", [7, 30, 50])
def test_user_functor_through_the_packed_kernel(pkg, D, monkeypatch, capfd):
    """A caller's functor whose ℓ is a sum of per-coordinate terms (kElementwise, kDeferred) through the PACKED kernel (several chains per
    wavefront; csrc/packed_kernels.hpp PackedFunctor, one more hiprtc module per lane-group shape): the bits of the wave-per-chain
    kernel (DHMC_PACKED=0) and of the oracle, over adaptive and fixed stages with a metric window; a second functor (Student-t: no
    built-in family, its own deterministic log) packed against the wave kernel."""
    rng = np.random.default_rng(D)
    mu = rng.normal(size=D); prec = np.exp(rng.normal(size=D))
    user = pkg.DeviceFunctorLogDensity(D, uf.DIAG_NORMAL, "MyDiagNormal", params=np.concatenate([mu, prec]))
    C = 37

    def steps(ctx, eps):
        out = {}
        ctx.init(); ctx.set_stepsize(eps)
        ctx.metric_window_begin()
        a = ctx.run(30, da={})
        ctx.update_metric_diag_window()
        out.update({"w_" + k: v for k, v in a.items()})
        out.update({"i_" + k: v for k, v in ctx.run(17).items()})
        q, lq, g = ctx.position()
        out.update(q=q, lq=lq, g=g, minv=ctx.metric_diag(), eps=ctx.stepsize())
        return out
    monkeypatch.setenv("DHMC_DEBUG_ORDER", "1")
    monkeypatch.setenv("DHMC_PACKED", "1")
    a = steps(pkg.DeviceContext(D, C, target=user.family, target_params=user.params(), seed=3), 0.2)
    assert "engine: packed" in capfd.readouterr().err
    monkeypatch.setenv("DHMC_PACKED", "0")
    b = steps(pkg.DeviceContext(D, C, target=user.family, target_params=user.params(), seed=3), 0.2)
    assert "engine: packed" not in capfd.readouterr().err
    o = steps(ol.Oracle(D, C, target=ol.TARGET_DIAG_NORMAL, params=ol.target_params_blob(ol.TARGET_DIAG_NORMAL, D, mu=mu, prec=prec), seed=3, threads=8), 0.2)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(a[k], o[k]), k
    scale = np.exp(rng.normal(size=D) * 0.3)
    t = pkg.DeviceFunctorLogDensity(D, uf.STUDENT_T, "StudentT", params=np.concatenate([[5.0], scale]))
    monkeypatch.setenv("DHMC_PACKED", "1")
    a = steps(pkg.DeviceContext(D, C, target=t.family, target_params=t.params(), seed=4), 0.3)
    assert "engine: packed" in capfd.readouterr().err
    monkeypatch.setenv("DHMC_PACKED", "0")
    b = steps(pkg.DeviceContext(D, C, target=t.family, target_params=t.params(), seed=4), 0.3)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
