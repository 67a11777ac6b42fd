"""BASELINE config 3 as a context (correlated normal with tridiagonal precision, dense M⁻¹ = Σ, ϵ fixed): for tools/pcie_rate.py."""
import numpy as np


def context(pkg, D, C, seed=1, eps=0.3):
    rho = 0.5
    sig = np.logspace(-1, 1, D)
    Pc = np.zeros(D) + (1 + rho ** 2) / (1 - rho ** 2); Pc[0] = Pc[-1] = 1 / (1 - rho ** 2)
    diag = Pc / sig ** 2
    off = np.zeros(D); off[:D - 1] = -rho / (1 - rho ** 2) / (sig[:-1] * sig[1:])
    idx = np.arange(D)
    Sigma = np.outer(sig, sig) * rho ** np.abs(idx[:, None] - idx[None, :])
    ctx = pkg.DeviceContext(D, C, target=pkg.abi.TARGET_TRIDIAG_NORMAL, target_params=np.concatenate([diag, off]), metric=pkg.abi.METRIC_DENSE,
                            seed=seed)
    ctx.set_metric_dense(Sigma)
    ctx.init(np.random.default_rng(5).normal(size=(C, D)) * sig)
    ctx.set_stepsize(eps)
    ctx.run(3, fields=[])
    return ctx
