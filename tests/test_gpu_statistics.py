"""GPU: the reference's tolerance-based end-to-end tests through the host wrapper
(test/test_mcmc.jl), agreement with the libm flavour of the oracle inside the tolerance
north_star states, and size-independent properties at BASELINE.json's full size."""
import numpy as np
import pytest

import oracle_lib as ol
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_package()


def test_mcmc_with_warmup_5dim_normal(pkg):
    """test_mcmc.jl:18-26: 5-dim N(1, I), 10⁴ draws: logdensities, mean, std, acceptance, ϵ."""
    l = pkg.DiagNormal(np.ones(5), 1.0)
    r = pkg.mcmc_with_warmup(pkg.PhiloxRNG(0x23EF614D), l, 10000, chains=4, reporter=pkg.NoProgressReport())
    pm, ts = r["posterior_matrix"], r["tree_statistics"]
    assert pm.shape == (4, 10000, 5)
    assert np.allclose(r["logdensities"], -0.5 * ((pm - 1) ** 2).sum(-1), rtol=1e-12, atol=1e-12)
    for c in range(4):                                   # every chain alone meets the reference's bars
        assert np.abs(pm[c].mean(0) - 1).max() < 0.04
        assert np.abs(pm[c].std(0, ddof=1) - 1).max() < 0.04
        assert ts.acceptance_rate[c].mean() >= 0.8
        assert 0.5 <= r["eps"][c] <= 2
    assert np.all(r["kappa"].Minv > 0)


def test_fixed_stepsize_and_explicit_eps(pkg):
    """test_mcmc.jl:28-48."""
    l = pkg.DiagNormal(np.ones(5), 1.0)
    r = pkg.mcmc_with_warmup(1, l, 2000, chains=3, initialization=dict(eps=1.0), warmup_stages=pkg.fixed_stepsize_warmup_stages(),
                             reporter=pkg.NoProgressReport())
    assert np.all(r["eps"] == 1.0)
    assert np.abs(r["posterior_matrix"].mean((0, 1)) - 1).max() < 0.06
    r = pkg.mcmc_with_warmup(2, l, 2000, chains=3, initialization=dict(eps=1.0),
                             warmup_stages=pkg.default_warmup_stages(stepsize_search=None), reporter=pkg.NoProgressReport())
    assert np.all((0.5 <= r["eps"]) & (r["eps"] <= 2))
    with pytest.raises(ValueError):     # mcmc.jl:137: ϵ given AND a search stage
        pkg.mcmc_with_warmup(2, l, 10, initialization=dict(eps=1.0), reporter=pkg.NoProgressReport())


def test_stepwise_api_and_keep_warmup(pkg):
    """test_mcmc.jl:50-57."""
    l = pkg.DiagNormal(np.ones(5), 1.0)
    res = pkg.mcmc_keep_warmup(3, l, 0, chains=2, reporter=pkg.NoProgressReport())
    assert len(res["warmup"]) == 8 and res["warmup"][0]["results"] is None
    assert res["warmup"][2]["results"]["posterior_matrix"].shape == (2, 25, 5)
    steps = pkg.mcmc_steps(res["sampling_logdensity"], res["final_warmup_state"])
    Q = res["final_warmup_state"].Q
    qs = []
    for _ in range(1000):
        Q, ts = pkg.mcmc_next_step(steps, Q)
        qs.append(Q.q)
    qs = np.stack(qs, 1)
    assert np.abs(qs.mean((0, 1)) - 1).max() < 0.15


def test_two_step_objects_over_one_context_do_not_trust_a_stale_position(pkg):
    """mcmc.jl:348-351: the step starts from the Q it is GIVEN.  Two MCMCSteps over one context: after B moved the chains, A's
    remembered last state is stale — a Q equal to it must be pushed back into the context, not assumed to be there."""
    l = pkg.DiagNormal(np.zeros(4), 1.0)
    res = pkg.mcmc_keep_warmup(5, l, 0, chains=3, reporter=pkg.NoProgressReport())
    sl, ws = res["sampling_logdensity"], res["final_warmup_state"]
    a = pkg.mcmc_steps(sl, ws)
    qa, _ = pkg.mcmc_next_step(a, ws.Q)                 # A remembers qa
    b = pkg.mcmc_steps(sl)
    for _ in range(3):
        pkg.mcmc_next_step(b)                           # B moves the chains elsewhere
    assert not np.array_equal(sl.ctx.position()[0], qa.q)
    before = sl.ctx.position_epoch
    pkg.mcmc_next_step(a, qa)                           # from qa again: the context must be put back there first
    assert sl.ctx.position_epoch >= before + 2          # set_position + run, not run alone


def test_200dim_never_reaches_depth_12(pkg):
    """test_mcmc.jl:60-72 (issue #115): 200-dim standard normal, max_depth = 12, 20×1000 draws."""
    r = pkg.mcmc_with_warmup(4, pkg.StandardNormal(200), 1000, chains=20, algorithm=pkg.NUTS(max_depth=12),
                             reporter=pkg.NoProgressReport())
    assert (r["tree_statistics"].depth < 12).all()
    s = pkg.diagnostics.summarize_tree_statistics(r["tree_statistics"])
    assert s["termination_counts"]["max_depth"] == 0
    assert (pkg.diagnostics.EBFMI(r["tree_statistics"]) >= 0.25).all()      # sample-correctness_utilities.jl:66


def test_agreement_with_libm_oracle_within_fp64_tolerance(pkg):
    """north_star: "results match the reference CPU path on the same RNG seeds within a stated
    fp64 tolerance on posterior moments and per-step Hamiltonian error".  The libm flavour of
    the oracle uses glibc's exp/log/sincos where the HIP path uses include/dhmc_detmath.h, so
    the two differ by last-place roundings only.  Tolerances: first transition 1e-12 relative
    on positions and 1e-9 absolute on the Hamiltonian error; trees (integers) identical over
    the first 5 transitions; posterior moments of 300 further draws × 64 chains within 3 MC
    standard errors."""
    D, C = 100, 64
    dev = pkg.DeviceContext(D, C, seed=31)
    ora = ol.Oracle(D, C, seed=31, det=False, threads=8)
    dev.init(); ora.init()
    dev.find_initial_stepsize(); ora.find_initial_stepsize()
    assert np.array_equal(dev.stepsize(), ora.stepsize())
    a, b = dev.run(5), ora.run(5)
    for k in ("depth", "steps", "term_left", "term_right", "directions"):
        assert np.array_equal(a[k], b[k]), k
    assert np.allclose(a["draws"][:, 0], b["draws"][:, 0], rtol=1e-12, atol=1e-13)
    herr_a = a["pi"] - a["logdensities"]; herr_b = b["pi"] - b["logdensities"]
    assert np.abs(herr_a[:, 0] - herr_b[:, 0]).max() < 1e-9
    assert np.abs(a["acceptance_rate"][:, 0] - b["acceptance_rate"][:, 0]).max() < 1e-12
    a, b = dev.run(300, da={}), ora.run(300, da={})
    se = 1 / np.sqrt(C * 300 / 2)
    assert np.abs(a["draws"].mean((0, 1)) - b["draws"].mean((0, 1))).max() < 6 * se
    assert abs(a["draws"].var() - b["draws"].var()) < 0.02
    assert abs(a["acceptance_rate"].mean() - b["acceptance_rate"].mean()) < 0.02


def test_full_size_properties(pkg):
    """BASELINE.json configs[1] at full size (D=1000, 4096 chains): size-independent properties."""
    import torch
    D, C, N = 1000, 4096, 30
    dev = pkg.DeviceContext(D, C, seed=0x23EF614D)
    dev.init(); dev.find_initial_stepsize()
    dev.run(40, da={}, fields=[])
    out = {k: torch.empty((C, N, D) if k == "draws" else (C, N), dtype=dt, device="cuda")
           for k, dt in (("draws", torch.float64), ("logdensities", torch.float64), ("pi", torch.float64),
                         ("acceptance_rate", torch.float64), ("steps", torch.int64), ("depth", torch.int32),
                         ("term_left", torch.int64), ("term_right", torch.int64))}
    dev.run_into(N, out)
    q = out["draws"]
    # the stored log density is ℓ of the stored draw
    assert torch.allclose(out["logdensities"], -0.5 * (q * q).sum(-1), rtol=1e-12, atol=0)
    # a tree of depth d visited at most 2^(d+1) - 1 leaves and at least 2^d - 1 ... and a valid top-level tree exactly
    steps, depth = out["steps"], out["depth"].long()
    assert (steps >= 1).all() and (steps <= 2 ** (depth + 1) - 1).all() and (steps >= 2 ** depth - 1).all()
    l, r = out["term_left"], out["term_right"]
    top_turn = (l <= 0) & (r >= 0) & (l < r)
    assert (steps[top_turn] == (2 ** depth[top_turn] - 1)).all()
    assert (r[top_turn] - l[top_turn] == steps[top_turn]).all()
    a = out["acceptance_rate"]
    assert (a >= 0).all() and (a <= 1).all() and 0.6 < float(a.mean()) < 0.95
    # energy error of the accepted point is small for a well-adapted chain
    assert float((out["pi"] - out["logdensities"]).mean()) < 0        # π = ℓ - K
    # posterior moments (standard normal) over 4096 × 30 draws × 1000 coordinates
    assert abs(float(q.mean())) < 2e-3 and abs(float(q.var()) - 1) < 5e-3
    # partition independence at full size: chains 100..163 of the big job == a 64-chain job at offset 100
    small = pkg.DeviceContext(D, 64, seed=0x23EF614D, chain_offset=100)
    small.init(); small.find_initial_stepsize()
    small.run(40, da={}, fields=[])
    s = small.run(N, fields=["draws", "steps"])
    assert np.array_equal(s["draws"], q[100:164].cpu().numpy())
    assert np.array_equal(s["steps"], steps[100:164].cpu().numpy())


def test_results_on_device_equal_host_results(pkg):
    """on_device=True hands back torch CUDA tensors (no PCIe copy of the draws); same bits as the numpy path, and
    the warmup stages — whose draws only feed the metric update — never leave the GPU either way."""
    import torch
    l = pkg.DiagNormal(np.linspace(-1, 1, 40), np.linspace(0.5, 3, 40))
    a = pkg.mcmc_with_warmup(9, l, 150, chains=6, reporter=pkg.NoProgressReport())
    b = pkg.mcmc_with_warmup(9, l, 150, chains=6, reporter=pkg.NoProgressReport(), on_device=True)
    assert isinstance(a["posterior_matrix"], np.ndarray) and b["posterior_matrix"].is_cuda
    assert np.array_equal(a["posterior_matrix"], b["posterior_matrix"].cpu().numpy())
    assert np.array_equal(a["logdensities"], b["logdensities"].cpu().numpy())
    for f in ("pi", "depth", "termination_left", "termination_right", "acceptance_rate", "steps"):
        assert np.array_equal(getattr(a["tree_statistics"], f), getattr(b["tree_statistics"], f).cpu().numpy()), f
    assert np.array_equal(a["tree_statistics"].directions, b["tree_statistics"].directions.cpu().numpy().view(np.uint32))
    assert np.array_equal(a["eps"], b["eps"])
    k = pkg.mcmc_keep_warmup(9, l, 150, chains=6, reporter=pkg.NoProgressReport())
    assert np.array_equal(k["inference"]["posterior_matrix"], a["posterior_matrix"])
    assert k["warmup"][2]["results"]["posterior_matrix"].shape == (6, 25, 40)      # first metric window kept on request
