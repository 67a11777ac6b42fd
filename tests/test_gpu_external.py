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
