"""GPU: DHMC_TARGET_EXTERNAL — the user's own batched log density (a PyTorch function on the device) behind the
round engine: the downward plugin API of the reference (LogDensityProblems.logdensity_and_gradient, hamiltonian.jl:204)."""
import numpy as np
import pytest

from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_package()


def test_leapfrogs_of_an_external_density_match_the_builtin_functor(pkg):
    """A standard normal given as torch code walks the same trees as the built-in functor wherever ℓ agrees to the
    last bit (∇ℓ = -q is exact; ℓ differs only in summation order): with D = 1 there is nothing to sum."""
    import torch
    l = pkg.TorchLogDensity(1, logdensity_and_gradient=lambda q: (-0.5 * (q * q).sum(1), -q))
    a = pkg.mcmc_with_warmup(5, l, 300, chains=16, reporter=pkg.NoProgressReport())
    b = pkg.mcmc_with_warmup(5, pkg.StandardNormal(1), 300, chains=16, reporter=pkg.NoProgressReport())
    assert np.array_equal(a["posterior_matrix"], b["posterior_matrix"])
    assert np.array_equal(a["tree_statistics"].steps, b["tree_statistics"].steps)
    assert np.array_equal(a["eps"], b["eps"])
    assert torch.cuda.is_available()


def test_autograd_model_posterior_moments(pkg):
    """A correlated Gaussian written as plain torch code, gradient by autograd: posterior mean and covariance."""
    import torch
    D, C = 6, 64
    rng = np.random.default_rng(2)
    A = rng.normal(size=(D, D)); Sigma = A @ A.T / D + np.eye(D); mu = rng.normal(size=D)
    P = torch.tensor(np.linalg.inv(Sigma), device="cuda"); m = torch.tensor(mu, device="cuda")
    l = pkg.TorchLogDensity(D, logdensity=lambda q: -0.5 * torch.einsum("ci,ij,cj->c", q - m, P, q - m))
    r = pkg.mcmc_with_warmup(11, l, 500, chains=C, reporter=pkg.NoProgressReport())
    x = r["posterior_matrix"].reshape(-1, D)
    assert np.abs(x.mean(0) - mu).max() < 0.08
    assert np.abs(np.cov(x.T) - Sigma).max() < 0.25
    assert 0.6 < r["tree_statistics"].acceptance_rate.mean() < 0.95
    # same seed, same model: the run is reproducible bit for bit
    r2 = pkg.mcmc_with_warmup(11, l, 500, chains=C, reporter=pkg.NoProgressReport())
    assert np.array_equal(r["posterior_matrix"], r2["posterior_matrix"])


def test_nonfinite_outputs_follow_evaluate_l_rules(pkg):
    """hamiltonian.jl:202-217: ℓ = -Inf (or a non-finite gradient) at a trial point is a rejection, not an error;
    an invalid INITIAL point is an error (strict)."""
    import torch

    def fg(q):      # half-line support: ℓ = -x for x > 0, -Inf otherwise
        x = q[:, 0]
        return torch.where(x > 0, -x, torch.full_like(x, -float("inf"))), -torch.ones_like(q)
    l = pkg.TorchLogDensity(1, logdensity_and_gradient=fg)
    r = pkg.mcmc_with_warmup(3, l, 400, chains=32, initialization=dict(q=np.full((32, 1), 1.0)), reporter=pkg.NoProgressReport())
    x = r["posterior_matrix"]
    assert (x > 0).all() and abs(x.mean() - 1.0) < 0.15          # Exp(1)
    with pytest.raises(pkg.DynamicHMCError):
        pkg.mcmc_with_warmup(3, l, 10, chains=4, initialization=dict(q=np.full((4, 1), -1.0)), reporter=pkg.NoProgressReport())


def test_callback_errors_surface_as_python_exceptions(pkg):
    def bad(q):
        raise KeyError("model blew up")
    with pytest.raises(KeyError):
        pkg.mcmc_with_warmup(1, pkg.TorchLogDensity(3, logdensity_and_gradient=bad), 5, chains=2, reporter=pkg.NoProgressReport())
    ctx = pkg.DeviceContext(3, 2, target=pkg.abi.TARGET_EXTERNAL)
    with pytest.raises(RuntimeError):
        ctx.init()                                               # no callback registered
    with pytest.raises(ValueError):
        pkg.DeviceContext(3, 2).set_logdensity_callback(lambda q: None)   # not an external-target context


def _with_env(env, f):
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return f()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_wide_chain_kernels_reproduce_the_16_slot_path(pkg):
    """External models are served up to D = 4096 (32 / 64 slots per lane) by the streaming round-engine kernels; forced
    onto a 1000-dim model (DHMC_FORCE_NPL) they must give the bits of the 16-slot path: padding adds exact zeros."""
    import torch
    D, C = 1000, 8
    scale = torch.linspace(0.5, 2.0, D, dtype=torch.float64, device="cuda")
    l = pkg.TorchLogDensity(D, logdensity_and_gradient=lambda q: (-0.5 * ((q / scale) ** 2).sum(1), -q / scale ** 2))
    run = lambda: pkg.mcmc_with_warmup(7, l, 30, chains=C, reporter=pkg.NoProgressReport(),
                                       warmup_stages=pkg.default_warmup_stages(middle_steps=20, doubling_stages=2))
    base = run()
    for npl in ("32", "64"):
        r = _with_env({"DHMC_FORCE_NPL": npl}, run)
        assert np.array_equal(r["posterior_matrix"], base["posterior_matrix"]), npl
        assert np.array_equal(r["tree_statistics"].steps, base["tree_statistics"].steps), npl
        assert np.array_equal(r["eps"], base["eps"]) and np.array_equal(r["kappa"].Minv, base["kappa"].Minv), npl


def test_model_with_more_than_1024_dimensions(pkg):
    import torch
    D, C = 3000, 48
    sd = torch.linspace(0.5, 3.0, D, dtype=torch.float64, device="cuda")
    l = pkg.TorchLogDensity(D, logdensity=lambda q: -0.5 * ((q / sd) ** 2).sum(1))
    r = pkg.mcmc_with_warmup(3, l, 120, chains=C, reporter=pkg.NoProgressReport())
    x = r["posterior_matrix"].reshape(-1, D)
    assert np.abs(x.mean(0) / sd.cpu().numpy()).max() < 0.25
    assert np.abs(x.std(0) / sd.cpu().numpy() - 1).max() < 0.2
    assert 0.6 < r["tree_statistics"].acceptance_rate.mean() < 0.95
    with pytest.raises((ValueError, RuntimeError)):
        pkg.mcmc_with_warmup(3, pkg.StandardNormal(4097), 5, chains=2, reporter=pkg.NoProgressReport())   # every family: D <= 4096
    with pytest.raises((ValueError, RuntimeError)):
        pkg.mcmc_with_warmup(3, pkg.TorchLogDensity(4097, logdensity=lambda q: -0.5 * (q * q).sum(1)), 5, chains=2, reporter=pkg.NoProgressReport())


def test_dense_metric_for_external_models(pkg):
    """Symmetric (dense) adaptation for the caller's model (mcmc.jl:210,218-222; hamiltonian.jl:73): the dense round
    engine with the callback as the density.  D = 1 reproduces the built-in functor bit for bit (ℓ has nothing to sum);
    a strongly correlated Gaussian gets a usable step size only with the dense metric."""
    import torch
    sym = lambda: pkg.default_warmup_stages(M=pkg.Symmetric, middle_steps=20, doubling_stages=3)
    a = pkg.mcmc_with_warmup(5, pkg.TorchLogDensity(1, logdensity_and_gradient=lambda q: (-0.5 * (q * q).sum(1), -q)), 200, chains=16,
                             warmup_stages=sym(), reporter=pkg.NoProgressReport())
    b = pkg.mcmc_with_warmup(5, pkg.StandardNormal(1), 200, chains=16, warmup_stages=sym(), reporter=pkg.NoProgressReport(),
                             per_chain_metric=False)       # a callback model adapts the pooled metric: the built-in family likewise here
    assert np.array_equal(a["posterior_matrix"], b["posterior_matrix"]) and np.array_equal(a["eps"], b["eps"])
    assert np.array_equal(a["kappa"].Minv, b["kappa"].Minv) and a["kappa"].dense

    D, C = 8, 64
    rng = np.random.default_rng(4)
    A = rng.normal(size=(D, D)); Sigma = A @ A.T + 0.05 * np.eye(D)           # condition number in the hundreds
    P = torch.tensor(np.linalg.inv(Sigma), device="cuda")
    l = pkg.TorchLogDensity(D, logdensity=lambda q: -0.5 * torch.einsum("ci,ij,cj->c", q, P, q))
    dense = pkg.mcmc_with_warmup(2, l, 400, chains=C, warmup_stages=pkg.default_warmup_stages(M=pkg.Symmetric), reporter=pkg.NoProgressReport())
    diag = pkg.mcmc_with_warmup(2, l, 400, chains=C, reporter=pkg.NoProgressReport())
    x = dense["posterior_matrix"].reshape(-1, D)
    assert np.abs(np.cov(x.T) - Sigma).max() < 0.25 * np.abs(Sigma).max()
    assert np.median(dense["eps"]) > 2 * np.median(diag["eps"])
    assert dense["tree_statistics"].steps.mean() < diag["tree_statistics"].steps.mean()


def test_context_is_unusable_after_a_callback_failure_until_reinitialised(pkg):
    """A callback that fails in the middle of dhmc_run leaves trial positions in the chain state: the context refuses to go
    on (DHMC_ERR_CALLBACK) until dhmc_init gives it a consistent state again."""
    calls = {"n": 0}

    def lg(q):
        calls["n"] += 1
        if calls["n"] == 6:
            raise KeyError("model blew up in the middle of a run")
        return -0.5 * (q * q).sum(1), -q

    ctx = pkg.DeviceContext(3, 2, target=pkg.abi.TARGET_EXTERNAL)
    ctx.set_logdensity_callback(lg)
    ctx.init(); ctx.set_stepsize(0.5)
    with pytest.raises(KeyError):
        ctx.run(20)
    with pytest.raises(RuntimeError, match="callback"):
        ctx.run(1)
    with pytest.raises(RuntimeError, match="callback"):
        ctx.position()
    calls["n"] = 100
    ctx.init(); ctx.set_stepsize(0.5)
    assert ctx.run(3)["draws"].shape == (2, 3, 3)
