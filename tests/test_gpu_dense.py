"""GPU: dense (Symmetric) Gaussian kinetic energy — GaussianKineticEnergy(M⁻¹) with
W = cholesky(inv(M⁻¹)).L (hamiltonian.jl:73) — through the C ABI, bit for bit against the oracle,
plus the reference's own dense tests (test_hamiltonian.jl:20-32, test_NUTS.jl:87-111)."""
import numpy as np
import pytest

import oracle_lib as ol
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(0x3C574111)


@pytest.fixture(scope="module")
def pkg():
    return load_package()


def rand_sigma(n):  # test/utilities.jl:6-9
    A = RNG.normal(size=(n, n))
    return A.T @ A + 0.01


def pair(pkg, D, C, **kw):
    dev = pkg.DeviceContext(D, C, metric=ol.METRIC_DENSE, **kw)
    ora = ol.Oracle(D, C, metric=ol.METRIC_DENSE, threads=8, **kw)
    return dev, ora


def same(a, b, what):
    for k in a:
        assert np.array_equal(a[k], b[k]), f"{what}: {k}"


def test_dense_kinetic_energy_construction(pkg):  # test_hamiltonian.jl:20-32
    for K in (2, 5, 10, 70):
        S = rand_sigma(K)
        dev, ora = pair(pkg, K, 2)
        dev.set_metric_dense(np.linalg.inv(S)); ora.set_metric_dense(np.linalg.inv(S))
        minv, W = dev.metric_dense()
        assert np.array_equal(W, ora.metric_dense_W())
        assert np.allclose(np.triu(W, 1), 0)                       # W isa LowerTriangular
        assert np.allclose(minv @ W @ W.T, np.eye(K), atol=1e-8)   # M⁻¹ W Wᵀ ≈ I
    with pytest.raises(ValueError):                                # not positive definite
        dev.set_metric_dense(-np.eye(70))
    with pytest.raises(ValueError):                                # diagonal context refuses a dense metric
        pkg.DeviceContext(5, 1).set_metric_dense(np.eye(5))


@pytest.mark.parametrize("K", [2, 3, 8, 33, 100])
def test_dense_parity_random_metric(pkg, K):
    """Random dense metric unrelated to the target (as rand_Hz, test/utilities.jl:85-96)."""
    mu = RNG.normal(size=K); prec = 1 / (RNG.normal(size=K) ** 2 + 0.1)
    params = np.concatenate([mu, prec])
    dev = pkg.DeviceContext(K, 5, metric=ol.METRIC_DENSE, target=ol.TARGET_DIAG_NORMAL, target_params=params, seed=K)
    ora2 = ol.Oracle(K, 5, metric=ol.METRIC_DENSE, target=ol.TARGET_DIAG_NORMAL, params=params, seed=K, threads=8)
    Minv = np.linalg.inv(rand_sigma(K))
    dev.set_metric_dense(Minv); ora2.set_metric_dense(Minv)
    dev.init(); ora2.init()
    dev.find_initial_stepsize(); ora2.find_initial_stepsize()
    assert np.array_equal(dev.stepsize(), ora2.stepsize())
    same(dev.run(15, da={}), ora2.run(15, da={}), f"K={K} adaptive")
    same(dev.run(10), ora2.run(10), f"K={K} fixed")


def test_perfect_metric_correlated_normal(pkg):
    """test_NUTS.jl:87-111 in BASELINE config 3's form: correlated MVN (tridiagonal precision) with
    the perfect dense metric M⁻¹ = Σ, fixed ϵ = 0.5: mean and covariance are recovered."""
    K, C, N = 8, 16, 2500
    rho = 0.6
    diag = np.full(K, (1 + rho ** 2) / (1 - rho ** 2)); diag[0] = diag[-1] = 1 / (1 - rho ** 2)
    off = np.full(K, -rho / (1 - rho ** 2))
    P = np.diag(diag) + np.diag(off[:K - 1], 1) + np.diag(off[:K - 1], -1)
    Sigma = np.linalg.inv(P)
    params = np.concatenate([diag, off])
    dev = pkg.DeviceContext(K, C, metric=ol.METRIC_DENSE, target=ol.TARGET_TRIDIAG_NORMAL, target_params=params, seed=2)
    ora = ol.Oracle(K, C, metric=ol.METRIC_DENSE, target=ol.TARGET_TRIDIAG_NORMAL, params=params, seed=2, threads=8)
    for e in (dev, ora):
        e.set_metric_dense(Sigma); e.init(); e.set_stepsize(0.5)
    a, b = dev.run(N, fields=["draws", "depth", "acceptance_rate"]), ora.run(N, fields=["draws", "depth", "acceptance_rate"])
    same(a, b, "perfect metric")
    q = a["draws"].reshape(-1, K)
    Cov = np.cov(q.T)
    tol = np.diag(Cov).max() / 50 * 4          # 40000 pooled draws of 16 chains; reference uses 1e4 of one chain
    assert np.abs(q.mean(0)).sum() < tol * K
    assert np.allclose(Cov, Sigma, atol=0.1, rtol=0.1)
    assert a["depth"].mean() < 3.5             # a perfect metric decorrelates: short trees


def test_dense_metric_adaptation_pooled(pkg):
    """TuningNUTS{Symmetric}: κ := GaussianKineticEnergy(regularize(Symmetric(cov(pm)), λ)) (mcmc.jl:210,218-222,
    281-284), pooled over the chains that share the dense M⁻¹; device (MFMA covariance) == oracle, bit for bit,
    and the estimate approaches the target's covariance."""
    K, C = 12, 16
    rho = 0.6
    diag = np.full(K, (1 + rho ** 2) / (1 - rho ** 2)); diag[0] = diag[-1] = 1 / (1 - rho ** 2)
    off = np.full(K, -rho / (1 - rho ** 2))
    P = np.diag(diag) + np.diag(off[:K - 1], 1) + np.diag(off[:K - 1], -1)
    params = np.concatenate([diag, off])
    dev = pkg.DeviceContext(K, C, metric=ol.METRIC_DENSE, target=ol.TARGET_TRIDIAG_NORMAL, target_params=params, seed=6)
    ora = ol.Oracle(K, C, metric=ol.METRIC_DENSE, target=ol.TARGET_TRIDIAG_NORMAL, params=params, seed=6, threads=8)
    for e in (dev, ora):
        e.init(); e.find_initial_stepsize()
    for n in (40, 75, 150):
        a, b = dev.run(n, da={}), ora.run(n, da={})
        same(a, b, f"stage {n}")
        lam = 5.0 / n
        dev.update_metric_dense(a["draws"], lam); ora.update_metric_dense(b["draws"], lam)
        md, Wd = dev.metric_dense(); mo, Wo = ora.metric_dense()
        assert np.array_equal(md, mo) and np.array_equal(Wd, Wo)
    same(dev.run(30), ora.run(30), "after adaptation")
    assert np.allclose(md, np.linalg.inv(P), atol=0.35, rtol=0.3)      # 2400 pooled draws


def test_default_warmup_with_symmetric_metric(pkg):
    """mcmc_with_warmup(...; warmup_stages = default_warmup_stages(; M = Symmetric)) (mcmc.jl docstring :566-569).  Since round 5 the
    default for a small model is the reference's own: every chain adapts ITS metric from ITS draws (mcmc.jl:281-285; api.py
    _per_chain_metric_default); per_chain_metric=False is the pooled metric of the batched engine."""
    K = 6
    rho = 0.7
    diag = np.full(K, (1 + rho ** 2) / (1 - rho ** 2)); diag[0] = diag[-1] = 1 / (1 - rho ** 2)
    off = np.full(K - 1, -rho / (1 - rho ** 2))
    l = pkg.TridiagNormal(diag, off)
    r = pkg.mcmc_with_warmup(5, l, 1500, chains=8, warmup_stages=pkg.default_warmup_stages(M=pkg.Symmetric),
                             reporter=pkg.NoProgressReport())
    assert r["kappa"].dense and r["kappa"].Minv.shape == (8, K, K)                      # one Symmetric κ per chain
    with pytest.raises(TypeError):                                                      # code written for ONE [D][D] matrix fails loudly:
        float(r["kappa"].Minv[0, 1])                                                    # an entry is a row of K numbers now
    lines = []
    pkg.mcmc_with_warmup(5, l, 20, chains=8, warmup_stages=pkg.default_warmup_stages(M=pkg.Symmetric),
                         reporter=pkg.LogProgressReport(printer=lines.append))
    assert any("one M⁻¹ per chain" in x for x in lines)                                 # … and the reporter says which metric it is
    P = np.diag(diag) + np.diag(off, 1) + np.diag(off, -1)
    q = r["posterior_matrix"].reshape(-1, K)
    assert np.allclose(np.cov(q.T), np.linalg.inv(P), atol=0.15, rtol=0.15)
    assert np.allclose(r["kappa"].Minv.mean(0), np.linalg.inv(P), atol=0.3, rtol=0.3)   # each from 400 draws: their mean is close
    assert not np.array_equal(r["kappa"].Minv[0], r["kappa"].Minv[1])
    assert r["tree_statistics"].acceptance_rate.mean() >= 0.7
    r = pkg.mcmc_with_warmup(5, l, 1500, chains=8, warmup_stages=pkg.default_warmup_stages(M=pkg.Symmetric),
                             reporter=pkg.NoProgressReport(), per_chain_metric=False)
    assert r["kappa"].dense and r["kappa"].Minv.shape == (K, K)                         # the pooled metric on request
    assert np.allclose(r["kappa"].Minv, np.linalg.inv(P), atol=0.3, rtol=0.3)
    assert r["tree_statistics"].acceptance_rate.mean() >= 0.7


def test_per_chain_symmetric_warmup_through_the_api_is_what_separate_runs_give(pkg):
    """mcmc_with_warmup(…; per_chain_metric = True): every chain adapts its own Symmetric κ from its own draws (mcmc.jl:281-284),
    so chain c of a 4-chain call is, bit for bit, the 1-chain call whose stream starts at chain c."""
    K = 5
    l = pkg.MvNormal(np.arange(K) * 0.5, rand_sigma(K))
    stages = lambda: pkg.default_warmup_stages(M=pkg.Symmetric, middle_steps=20, doubling_stages=3)
    r = pkg.mcmc_with_warmup(pkg.PhiloxRNG(31), l, 200, chains=4, warmup_stages=stages(), reporter=pkg.NoProgressReport(),
                             per_chain_metric=True)
    assert r["kappa"].dense and r["kappa"].Minv.shape == (4, K, K)
    assert not np.array_equal(r["kappa"].Minv[0], r["kappa"].Minv[1])
    repr(r["kappa"])
    for c in (0, 3):
        one = pkg.mcmc_with_warmup(pkg.PhiloxRNG(31, chain_offset=c), l, 200, chains=1, warmup_stages=stages(),
                                   reporter=pkg.NoProgressReport(), per_chain_metric=True)
        assert np.array_equal(one["posterior_matrix"][0], r["posterior_matrix"][c])
        assert np.array_equal(one["kappa"].Minv[0], r["kappa"].Minv[c])
        assert np.array_equal(one["eps"], r["eps"][c:c + 1])


def test_dense_state_export_import(pkg):
    """The resume blob of a dense context carries the shared M⁻¹ / W as well."""
    K = 10
    Minv = np.linalg.inv(rand_sigma(K))
    a = pkg.DeviceContext(K, 3, metric=ol.METRIC_DENSE, seed=4); b = pkg.DeviceContext(K, 3, metric=ol.METRIC_DENSE, seed=4)
    a.set_metric_dense(Minv); a.init(); a.find_initial_stepsize(); a.run(10, da={})
    b.import_state(a.export_state())
    assert np.array_equal(a.metric_dense()[1], b.metric_dense()[1])
    ra, rb = a.run(6), b.run(6)
    for k in ra:
        assert np.array_equal(ra[k], rb[k]), k


@pytest.mark.parametrize("K", [33, 100, 257])
def test_device_factorisation_matches_the_oracle_at_block_boundaries(pkg, K):
    """GaussianKineticEnergy(M⁻¹) (hamiltonian.jl:73) built on the device (blocked right-looking Cholesky / triangular
    inverse in blocks of 32, XᵀX on fp64 MFMA, csrc/dense_factor.hpp): M⁻¹ and W bit-equal to the oracle's unblocked
    loops, for sizes around the block width; host and device inputs; a matrix that is not positive definite is refused
    and leaves the metric in place."""
    import torch
    S = rand_sigma(K)
    dev, ora = pair(pkg, K, 2)
    dev.set_metric_dense(S); ora.set_metric_dense(S)
    md, Wd = dev.metric_dense(); mo, Wo = ora.metric_dense()
    assert np.array_equal(md, mo) and np.array_equal(Wd, Wo)
    assert np.allclose(Wd @ Wd.T, np.linalg.inv(S), rtol=1e-8, atol=1e-10)
    dev.set_metric_dense(torch.from_numpy(2 * S).cuda()); ora.set_metric_dense(2 * S)     # a device matrix
    md, Wd = dev.metric_dense(); mo, Wo = ora.metric_dense()
    assert np.array_equal(md, mo) and np.array_equal(Wd, Wo)
    bad = S.copy(); bad[K // 2, K // 2] = -1.0
    with pytest.raises(ValueError):
        dev.set_metric_dense(bad)
    assert np.array_equal(dev.metric_dense()[0], mo)


def test_dense_metric_adaptation_at_1000_dimensions(pkg):
    """SURVEY.md §8 f-2 at BASELINE config 3's width: a TuningNUTS{Symmetric} stage on a 1000-dim correlated normal —
    pooled covariance (MFMA), regularisation, Cholesky, triangular inverse, XᵀX and the second Cholesky all on the
    device — gives M⁻¹ and W bit-equal to the oracle's, and the chains continue identically afterwards."""
    D, C, n = 1000, 8, 50
    rho = 0.5
    sig = np.logspace(-0.2, 0.2, D)
    Pc = np.zeros(D) + (1 + rho ** 2) / (1 - rho ** 2); Pc[0] = Pc[-1] = 1 / (1 - rho ** 2)
    diag = Pc / sig ** 2
    off = np.zeros(D); off[:D - 1] = -rho / (1 - rho ** 2) / (sig[:-1] * sig[1:])
    params = np.concatenate([diag, off])
    dev = pkg.DeviceContext(D, C, metric=ol.METRIC_DENSE, target=ol.TARGET_TRIDIAG_NORMAL, target_params=params, seed=16)
    ora = ol.Oracle(D, C, metric=ol.METRIC_DENSE, target=ol.TARGET_TRIDIAG_NORMAL, params=params, seed=16, threads=8)
    q0 = np.random.default_rng(3).normal(size=(C, D)) * sig
    for e in (dev, ora):
        e.init(q0); e.find_initial_stepsize()
    a, b = dev.run(n, da={}), ora.run(n, da={})
    same(a, b, "adaptive stage")
    lam = 5.0 / n
    dev.update_metric_dense(a["draws"], lam); ora.update_metric_dense(b["draws"], lam)
    md, Wd = dev.metric_dense(); mo, Wo = ora.metric_dense()
    assert np.array_equal(md, mo), "M⁻¹"
    assert np.array_equal(Wd, Wo), "W"
    same(dev.run(3, da={}), ora.run(3, da={}), "after the metric update")


def test_logistic_regression_with_a_dense_metric_matches_oracle(pkg):
    """Symmetric warmup of a logistic regression: the dense round engine with ℓ, ∇ℓ of all chains evaluated by the family's two
    GEMMs between the kernels (the functor would re-read X twice per gradient and chain) — search, adaptive stage, dense metric
    update, fixed stage and a probe, bit for bit against the oracle; 2 500 observations = two blocks of the ABI's Σ over them."""
    D, C, N = 24, 5, 2500
    rng = np.random.default_rng(12)
    X = rng.normal(size=(N, D)) / 5
    y = (rng.random(N) < 1 / (1 + np.exp(-X @ rng.normal(size=D)))).astype(float)
    params = ol.target_params_blob(ol.TARGET_LOGISTIC, D, X=X, y=y)
    dev = pkg.DeviceContext(D, C, metric=ol.METRIC_DENSE, target=ol.TARGET_LOGISTIC, target_params=params, seed=6)
    ora = ol.Oracle(D, C, metric=ol.METRIC_DENSE, target=ol.TARGET_LOGISTIC, params=params, seed=6, threads=5)
    for e in (dev, ora):
        e.init(); e.find_initial_stepsize()
    assert np.array_equal(dev.stepsize(), ora.stepsize())
    a, b = dev.run(40, da={}), ora.run(40, da={})
    same(a, b, "adaptive stage")
    dev.update_metric_dense(a["draws"], 5.0 / 40); ora.update_metric_dense(b["draws"], 5.0 / 40)
    assert np.array_equal(dev.metric_dense()[0], ora.metric_dense()[0])
    same(dev.run(10), ora.run(10), "fixed stage")
    ta, tb = dev.leapfrog_trajectory(0.05, -2, 3, momentum_index=1), ora.leapfrog_trajectory(0.05, -2, 3, momentum_index=1)
    for k in ("delta", "logdensity", "q", "p"):
        assert np.array_equal(ta[k], tb[k]), k


def test_per_chain_dense_metric_is_the_references_semantics(pkg):
    """dense_per_chain = 1: every chain has its own M⁻¹ and adapts it from its OWN draws (mcmc.jl:281-285, what C independent
    reference runs do), instead of one shared matrix from the pooled draws.  Per-chain M⁻¹ / W and the chains' continuation
    bit-equal to the oracle; the chains' matrices differ from each other; the resume blob carries all of them."""
    K, C = 9, 5
    rho = 0.6
    diag = np.full(K, (1 + rho ** 2) / (1 - rho ** 2)); diag[0] = diag[-1] = 1 / (1 - rho ** 2)
    off = np.full(K, -rho / (1 - rho ** 2))
    params = np.concatenate([diag, off])
    kw = dict(metric=ol.METRIC_DENSE, target=ol.TARGET_TRIDIAG_NORMAL, seed=21, dense_per_chain=True)
    dev = pkg.DeviceContext(K, C, target_params=params, **kw)
    ora = ol.Oracle(K, C, params=params, threads=4, **kw)
    for e in (dev, ora):
        e.init(); e.find_initial_stepsize()
    for n in (60, 120):
        a, b = dev.run(n, da={}), ora.run(n, da={})
        same(a, b, f"stage {n}")
        lam = 5.0 / n
        dev.update_metric_dense(a["draws"], lam); ora.update_metric_dense(b["draws"], lam)
        for c in range(C):
            md, Wd = dev.metric_dense(c); mo, Wo = ora.metric_dense(c)
            assert np.array_equal(md, mo) and np.array_equal(Wd, Wo), c
    assert not np.array_equal(dev.metric_dense(0)[0], dev.metric_dense(1)[0])
    same(dev.run(25), ora.run(25), "after adaptation")
    twin = pkg.DeviceContext(K, C, target_params=params, **kw)
    twin.import_state(dev.export_state())
    same(dev.run(5), twin.run(5), "resumed")
    S = rand_sigma(K)
    dev.set_metric_dense(S); ora.set_metric_dense(S)              # one matrix for every chain
    for c in (0, C - 1):
        assert np.array_equal(dev.metric_dense(c)[0], ora.metric_dense(c)[0])


def test_per_chain_dense_metric_takes_the_one_product_recurrence_on_request(pkg):
    K, C = 20, 4
    kw = dict(metric=ol.METRIC_DENSE, seed=5, dense_per_chain=True)
    dev = pkg.DeviceContext(K, C, **kw); ora = ol.Oracle(K, C, threads=4, **kw)
    assert dev.dense_products() == 2
    dev.set_dense_products(1); ora.set_dense_products(1)
    for e in (dev, ora):
        e.init(); e.find_initial_stepsize()
    a, b = dev.run(60, da={}), ora.run(60, da={})
    same(a, b, "stage")
    dev.update_metric_dense(a["draws"], 5.0 / 60); ora.update_metric_dense(b["draws"], 5.0 / 60)
    same(dev.run(20), ora.run(20), "after the per-chain update")
    assert not np.array_equal(dev.metric_dense(0)[0], dev.metric_dense(1)[0])


def test_per_chain_dense_update_in_batches_across_block_boundaries(pkg):
    """The per-chain update estimates and factorises all chains of a batch per launch (blockIdx.z = chain): D = 70 spans three
    32-step Cholesky blocks and two 64-column covariance tiles; every chain's M⁻¹ and W bit-equal to the oracle's."""
    K, C, n = 70, 6, 150
    sig = np.logspace(-0.5, 0.5, K)
    params = ol.target_params_blob(ol.TARGET_DIAG_NORMAL, K, mu=np.zeros(K), prec=1 / sig ** 2)
    kw = dict(metric=ol.METRIC_DENSE, target=ol.TARGET_DIAG_NORMAL, seed=33, dense_per_chain=True)
    dev = pkg.DeviceContext(K, C, target_params=params, **kw)
    ora = ol.Oracle(K, C, params=params, threads=6, **kw)
    for e in (dev, ora):
        e.init(); e.find_initial_stepsize()
    a, b = dev.run(n, da={}), ora.run(n, da={})
    same(a, b, "stage")
    dev.update_metric_dense(a["draws"], 5.0 / n); ora.update_metric_dense(b["draws"], 5.0 / n)
    for c in range(C):
        md, Wd = dev.metric_dense(c); mo, Wo = ora.metric_dense(c)
        assert np.array_equal(md, mo) and np.array_equal(Wd, Wo), c
    same(dev.run(4, da={}), ora.run(4, da={}), "after the update")


def test_per_chain_dense_update_lets_every_chain_stand_for_itself(pkg):
    """with dense_per_chain a chain whose covariance estimate is refused (here: a chain whose window draws are all the
    same point — zero covariance, λ = 0) keeps its metric, while every other chain is updated; the call reports
    DHMC_ERR_INVALID_ARGUMENT and names the chain.  (The reference fails per chain: each chain is its own mcmc_with_warmup.)"""
    K, C, n = 6, 4, 40
    dev = pkg.DeviceContext(K, C, metric=ol.METRIC_DENSE, seed=3, dense_per_chain=True)
    dev.init(); dev.find_initial_stepsize()
    a = dev.run(n, da={})
    draws = a["draws"].copy()
    draws[2] = draws[2, :1]                                   # chain 2: a degenerate window
    before = [dev.metric_dense(c)[0] for c in range(C)]
    with pytest.raises(Exception) as ei:
        dev.update_metric_dense(draws, 0.0)
    assert "chain 2" in str(ei.value) or "1 chain" in str(ei.value)
    after = [dev.metric_dense(c)[0] for c in range(C)]
    assert np.array_equal(after[2], before[2])                # refused: unchanged
    for c in (0, 1, 3):
        assert not np.array_equal(after[c], before[c])        # the others did adapt
        assert np.allclose(after[c], np.cov(draws[c].T), rtol=1e-10)
    dev.run(5)                                                # and the context goes on


def test_metric_pooled_over_ranks_with_one_rank_is_the_plain_update(pkg):
    """include/dhmc.h dhmc_set_metric_allreduce: with ONE rank the job-wide estimate (column sums / row count and the scatter
    matrix passed through the all-reduce callback) is bit-identical to dhmc_update_metric_dense without a callback, and close to
    the host statement of the protocol (sharding.pooled_covariance); the callback really ran (two collectives per update)."""
    D, C, N = 37, 6, 30
    rng = np.random.default_rng(8)
    draws = rng.normal(size=(C, N, D)) * np.linspace(0.5, 2, D) + 1.0
    calls = []
    a = pkg.DeviceContext(D, C, metric=ol.METRIC_DENSE, seed=1)
    b = pkg.DeviceContext(D, C, metric=ol.METRIC_DENSE, seed=1)
    b.set_metric_allreduce(lambda t: calls.append(int(t.numel())))          # one rank: the sum over ranks is the tensor itself
    for ctx in (a, b):
        ctx.init()
        ctx.update_metric_dense(draws, 0.05)
    Ma, Wa = a.metric_dense(); Mb, Wb = b.metric_dense()
    assert np.array_equal(Ma, Mb) and np.array_equal(Wa, Wb)
    assert calls == [D + 2, 64 * 64]                                       # column sums + count + error slot, then the padded scatter matrix
    S, _, J = pkg.sharding.pooled_covariance(draws)
    want = 0.95 * S + 0.05 * np.diag(np.diag(S))
    assert J == C * N and np.allclose(Mb, want, rtol=1e-11, atol=1e-13)
    b.set_metric_allreduce(None)
    b.update_metric_dense(draws, 0.05)
    assert len(calls) == 2 and np.array_equal(b.metric_dense()[0], Ma)


def _pooled_worker(rank, world, port, total, D, N, outdir):
    import os, sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    import oracle_lib as ol2
    from __graft_entry__ import load_package
    pkg = load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # two ranks on this box's one GPU: RCCL wants one device per rank
    off, cnt = pkg.sharding.shard_chains(total, world, rank)
    draws = np.random.default_rng(4).normal(size=(total, N, D)) * np.linspace(0.5, 2, D)
    ctx = pkg.DeviceContext(D, cnt, metric=ol2.METRIC_DENSE, seed=2, chain_offset=off)
    ctx.init()
    ctx.set_metric_allreduce(pkg.sharding.TorchAllReduce(dist))
    ctx.update_metric_dense(draws[off:off + cnt], 0.1)
    M, W = ctx.metric_dense()
    np.savez(os.path.join(outdir, f"metric{rank}.npz"), M=M, W=W)
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_adapt_one_shared_dense_metric(pkg, tmp_path):
    """Two PROCESSES, each a context over its block of chains (ragged: 4 and 3), adapt the shared dense metric from the draws of
    both (all-reduce over gloo: they share this box's GPU): the two ranks end with the same M⁻¹ and W bit for bit, and that matrix
    agrees with one context holding all seven chains to rounding (a collective's summation order is its own: rtol 1e-12)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    total, D, N, world = 7, 70, 25, 2
    mp.spawn(_pooled_worker, args=(world, port, total, D, N, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "metric0.npz"), np.load(tmp_path / "metric1.npz")
    assert np.array_equal(a["M"], b["M"]) and np.array_equal(a["W"], b["W"])
    draws = np.random.default_rng(4).normal(size=(total, N, D)) * np.linspace(0.5, 2, D)
    one = pkg.DeviceContext(D, total, metric=ol.METRIC_DENSE, seed=2)
    one.init(); one.update_metric_dense(draws, 0.1)
    M1, W1 = one.metric_dense()
    assert np.allclose(a["M"], M1, rtol=1e-12, atol=1e-15) and np.allclose(a["W"], W1, rtol=1e-10, atol=1e-13)
