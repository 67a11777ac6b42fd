"""GPU: the five BASELINE.json configs as parity-test cases (configs[1] is also the bench workload).
Where the oracle cannot cover the full size in seconds, a slice of the chains is compared bit for bit
(chains are independent and partition independent) and the rest is checked through properties."""
import numpy as np

import ess_reference
import pytest

import oracle_lib as ol
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_package()


def test_config1_reference_path(pkg):
    """configs[0]: 100-dim isotropic standard MVN, 4 chains, 1000 draws via mcmc_with_warmup with the
    default 900-transition warmup — the reference's own CPU-runnable case, here HIP vs oracle, bit for bit."""
    D, C, N = 100, 4, 1000
    res = pkg.mcmc_keep_warmup(pkg.PhiloxRNG(0x23EF614D), pkg.StandardNormal(D), N, chains=C, reporter=pkg.NoProgressReport())
    ora = ol.Oracle(D, C, seed=0x23EF614D, threads=4)
    ora.init(); ora.find_initial_stepsize()
    stages = pkg.default_warmup_stages()[1:]
    assert sum(s.N for s in stages) == 900
    for st, got in zip(stages, res["warmup"][1:]):
        if st.M is not None:
            ora.metric_window_begin()          # the API's Diagonal stages adapt from a metric window (include/dhmc.h)
        r = ora.run(st.N, da={})
        assert np.array_equal(r["draws"], got["results"]["posterior_matrix"])
        assert np.array_equal(r["eps"], got["results"]["eps"])
        assert np.array_equal(r["depth"], got["results"]["tree_statistics"].depth)
        if st.M is not None:
            ora.update_metric_diag_window()
            assert np.array_equal(ora.metric_diag(), got["warmup_state"].kappa.Minv)
        assert np.array_equal(ora.stepsize(), got["warmup_state"].eps)
    r = ora.run(N)
    inf = res["inference"]
    assert np.array_equal(r["draws"], inf["posterior_matrix"])
    assert np.array_equal(r["logdensities"], inf["logdensities"])
    assert np.array_equal(r["acceptance_rate"], inf["tree_statistics"].acceptance_rate)
    assert np.array_equal(r["steps"], inf["tree_statistics"].steps)
    assert np.array_equal(r["term_left"], inf["tree_statistics"].termination_left)
    pm = inf["posterior_matrix"]
    assert abs(pm.mean()) < 0.01 and abs(pm.var() - 1) < 0.02
    ess = min(ess_reference.ess_rhat(pm[:, :, k])[0] for k in range(0, D, 10))
    assert ess / (C * N) >= 0.5                      # τ = ESS/N ≥ 0.5 (sample-correctness_utilities.jl:67)


def test_config3_dense_metric_1000dim_slice(pkg):
    """configs[2]: 1000-dim correlated MVN (Σ_ij = σ_i σ_j ρ^|i-j| through its tridiagonal precision),
    dense M⁻¹ = Σ shared by all chains; 4 chains × 3 transitions against the oracle."""
    D, C = 1000, 4
    rho = 0.5
    sig = np.logspace(-1, 1, D)
    Pc = np.zeros(D) + (1 + rho ** 2) / (1 - rho ** 2); Pc[0] = Pc[-1] = 1 / (1 - rho ** 2)
    diag = Pc / sig ** 2
    off = np.zeros(D); off[:D - 1] = -rho / (1 - rho ** 2) / (sig[:-1] * sig[1:])
    idx = np.arange(D)
    Sigma = np.outer(sig, sig) * rho ** np.abs(idx[:, None] - idx[None, :])
    params = np.concatenate([diag, off])
    import os
    os.environ["DHMC_DENSE"] = "rounds=1"          # the production engine (MFMA GEMM rounds) even for this 4-chain slice
    try:
        dev = pkg.DeviceContext(D, C, metric=ol.METRIC_DENSE, target=ol.TARGET_TRIDIAG_NORMAL, target_params=params, seed=3)
    finally:
        del os.environ["DHMC_DENSE"]
    ora = ol.Oracle(D, C, metric=ol.METRIC_DENSE, target=ol.TARGET_TRIDIAG_NORMAL, params=params, seed=3, threads=4)
    q0 = np.random.default_rng(5).normal(size=(C, D)) * sig
    for e in (dev, ora):
        e.set_metric_dense(Sigma); e.init(q0); e.set_stepsize(0.4)
    a, b = dev.run(3), ora.run(3)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert (a["steps"] >= 3).all()
    assert dev.last_run_rounds() > 0


def test_config4_funnel_4096_chains(pkg):
    """configs[3] on one GPU's share: Neal's 30-dim funnel, 4096 chains; the first 48 chains bit for bit
    against the oracle, all chains through properties (depth spread, divergences recorded, v marginal)."""
    D, C, NW, N = 30, 4096, 150, 100
    dev = pkg.DeviceContext(D, C, target=ol.TARGET_FUNNEL, seed=44)
    dev.init(); dev.find_initial_stepsize()
    w = dev.run(NW, da={}, fields=["draws", "depth"])
    dev.update_metric_diag(w["draws"])
    r = dev.run(N)
    ora = ol.Oracle(D, 48, target=ol.TARGET_FUNNEL, seed=44, threads=8)
    ora.init(); ora.find_initial_stepsize()
    wo = ora.run(NW, da={}, fields=["draws", "depth"])
    assert np.array_equal(wo["draws"], w["draws"][:48])
    ora.update_metric_diag(wo["draws"])
    ro = ora.run(N)
    for k in ro:
        assert np.array_equal(ro[k], r[k][:48]), k
    assert len(np.unique(r["depth"])) >= 5                            # per-chain divergent tree depths
    div = r["term_left"] == r["term_right"]
    assert 0 < div.mean() < 0.2                                       # the funnel's neck produces divergences
    assert np.allclose(r["logdensities"], -r["draws"][..., 0] ** 2 / 18 - 0.5 * np.exp(-r["draws"][..., 0]) * (r["draws"][..., 1:] ** 2).sum(-1)
                       - 14.5 * r["draws"][..., 0], rtol=1e-10, atol=1e-9)
    v = r["draws"][:, -1, 0]
    # v ~ N(0, 3²) in truth; NUTS with a fixed step size cannot enter the neck (v << 0), so the sampled
    # marginal is shifted upward — the well-known funnel pathology, flagged by the divergences above
    assert -0.5 < v.mean() < 1.8 and 1.2 < v.std() < 3.6


def test_config5_logistic_p256_slice(pkg):
    """configs[4]'s model at p = 256 with N = 2000 observations for the oracle comparison, and at the
    full N = 10⁵ (X resident in HBM: 2 × 205 MB) through a gradient check against numpy."""
    rng = np.random.default_rng(7)
    D = 256
    beta = rng.normal(size=D)

    def data(N):
        X = rng.normal(size=(N, D)) / 16
        y = (rng.random(N) < 1 / (1 + np.exp(-X @ beta))).astype(float)
        return X, y

    X, y = data(2000)
    params = ol.target_params_blob(ol.TARGET_LOGISTIC, D, X=X, y=y)
    import os
    os.environ["DHMC_LOGISTIC_ROUNDS"] = "1"       # the production engine (GEMM gradients) even for 6 chains
    try:
        dev = pkg.DeviceContext(D, 6, target=ol.TARGET_LOGISTIC, target_params=params, seed=8)
    finally:
        del os.environ["DHMC_LOGISTIC_ROUNDS"]
    ora = ol.Oracle(D, 6, target=ol.TARGET_LOGISTIC, params=params, seed=8, threads=6)
    for e in (dev, ora):
        e.init(); e.find_initial_stepsize()
    a, b = dev.run(10, da={}), ora.run(10, da={})
    for k in a:
        assert np.array_equal(a[k], b[k]), k

    X, y = data(100000)
    big = pkg.DeviceContext(D, 64, target=ol.TARGET_LOGISTIC, target_params=ol.target_params_blob(ol.TARGET_LOGISTIC, D, X=X, y=y), seed=9)
    big.init(); big.find_initial_stepsize()
    r = big.run(3, da={}, fields=["draws", "logdensities", "steps"])
    q, lq, g = big.position()
    eta = q @ X.T
    ref_lq = (y * eta - np.logaddexp(0, eta)).sum(1) - 0.5 * (q * q).sum(1)
    ref_g = (y - 1 / (1 + np.exp(-eta))) @ X - q
    assert np.allclose(lq, ref_lq, rtol=1e-11)
    assert np.allclose(g, ref_g, rtol=1e-9, atol=1e-9)
    assert (r["steps"] >= 1).all()


def _config3_problem(D=1000, rho=0.5):
    sig = np.logspace(-1, 1, D)
    Pc = np.zeros(D) + (1 + rho ** 2) / (1 - rho ** 2); Pc[0] = Pc[-1] = 1 / (1 - rho ** 2)
    diag = Pc / sig ** 2
    off = np.zeros(D); off[:D - 1] = -rho / (1 - rho ** 2) / (sig[:-1] * sig[1:])
    idx = np.arange(D)
    return np.concatenate([diag, off]), np.outer(sig, sig) * rho ** np.abs(idx[:, None] - idx[None, :])


def test_config3_two_stream_engine_at_production_width(pkg):
    """configs[2] through the PRODUCTION branch of the dense round engine: 256 chains at D = 1000 run as two
    half-batches on two streams (dhmc_capi.hip `nh = 2`); chains from both halves (0..3 and 128..131, reproduced by
    oracles at those chain offsets) must match bit for bit through an adaptive stage and a fixed one."""
    D, C = 1000, 256
    params, Sigma = _config3_problem(D)
    dev = pkg.DeviceContext(D, C, metric=ol.METRIC_DENSE, target=ol.TARGET_TRIDIAG_NORMAL, target_params=params, seed=31)
    q0 = np.random.default_rng(5).normal(size=(C, D)) * np.sqrt(np.diag(Sigma))
    dev.set_metric_dense(Sigma); dev.init(q0); dev.find_initial_stepsize()
    eps0 = dev.stepsize()
    a1 = dev.run(3, da={}); a2 = dev.run(2)
    assert dev.last_run_rounds() > 0 and (a2["steps"] >= 3).all()
    for off in (0, 128):
        ora = ol.Oracle(D, 4, metric=ol.METRIC_DENSE, target=ol.TARGET_TRIDIAG_NORMAL, params=params, seed=31, chain_offset=off, threads=4)
        ora.set_metric_dense(Sigma); ora.init(q0[off:off + 4]); ora.find_initial_stepsize()
        assert np.array_equal(eps0[off:off + 4], ora.stepsize())
        b1 = ora.run(3, da={}); b2 = ora.run(2)
        for k in b1:
            assert np.array_equal(a1[k][off:off + 4], b1[k]), (off, k)
            assert np.array_equal(a2[k][off:off + 4], b2[k]), (off, k)


def test_logistic_skinny_gemm_engine_against_oracle(pkg):
    """The GEMM-gradient round engine with N >= 4096 observations, where G = R·X runs through
    gemm_skinny_pc_f64_kernel (producer/consumer MFMA waves): 128 chains on the device, chains 0..2 and 125..127 against
    the oracle (every output of an adaptive stage with a metric update, and of a fixed one)."""
    rng = np.random.default_rng(12)
    N, D, C = 4500, 64, 128
    X = rng.normal(size=(N, D)) / 8
    y = (rng.random(N) < 1 / (1 + np.exp(-X @ rng.normal(size=D)))).astype(float)
    params = ol.target_params_blob(ol.TARGET_LOGISTIC, D, X=X, y=y)
    dev = pkg.DeviceContext(D, C, target=ol.TARGET_LOGISTIC, target_params=params, seed=14)
    dev.init(); dev.find_initial_stepsize()
    a1 = dev.run(8, da={}); dev.update_metric_diag(a1["draws"]); a2 = dev.run(5)
    assert dev.last_run_rounds() > 0
    for off in (0, 125):
        ora = ol.Oracle(D, 3, target=ol.TARGET_LOGISTIC, params=params, seed=14, chain_offset=off, threads=3)
        ora.init(); ora.find_initial_stepsize()
        b1 = ora.run(8, da={}); ora.update_metric_diag(b1["draws"]); b2 = ora.run(5)
        for k in b1:
            assert np.array_equal(a1[k][off:off + 3], b1[k]), (off, k)
            assert np.array_equal(a2[k][off:off + 3], b2[k]), (off, k)


def test_config5_full_size_bit_exact_against_oracle(pkg):
    """configs[4] at its full N = 10⁵ observations, p = 256, through the GEMM-gradient round engine (128 chains): ℓ, ∇ℓ
    and every output of two transitions of chains 0 and 127 are BIT-EQUAL to the oracle's (one k-ascending fma chain
    of 10⁵ terms per gradient coordinate), not merely close."""
    rng = np.random.default_rng(7)
    N, D, C = 100000, 256, 128
    X = rng.normal(size=(N, D)) / 16
    y = (rng.random(N) < 1 / (1 + np.exp(-X @ rng.normal(size=D)))).astype(float)
    params = ol.target_params_blob(ol.TARGET_LOGISTIC, D, X=X, y=y)
    dev = pkg.DeviceContext(D, C, target=ol.TARGET_LOGISTIC, target_params=params, seed=9)
    dev.init(); dev.set_stepsize(0.02)
    a = dev.run(2)
    assert dev.last_run_rounds() > 0
    q, lq, g = dev.position()
    for off in (0, 127):
        ora = ol.Oracle(D, 1, target=ol.TARGET_LOGISTIC, params=params, seed=9, chain_offset=off, threads=1)
        ora.init(); ora.set_stepsize(0.02)
        b = ora.run(2)
        for k in b:
            assert np.array_equal(a[k][off:off + 1], b[k]), (off, k)
        qo, lqo, go = ora.position()
        assert np.array_equal(q[off:off + 1], qo) and np.array_equal(lq[off:off + 1], lqo) and np.array_equal(g[off:off + 1], go)


def test_bench_launches_its_own_ranks():
    """a plain `python bench.py --gpus N` (no WORLD_SIZE in the environment) starts N ranks itself, one per
    GPU (torch.distributed.run on 127.0.0.1), and rank 0 prints the one JSON line with n_gpus = N.  On this 1-GPU box the two
    ranks share the device: the collectives then run over gloo (RCCL refuses two ranks on one GPU), everything else — chain
    sharding by chain_offset, barrier + max-over-ranks timing, the gather of the last draws — is the 8-GPU path."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--chains", "2048", "--steps", "2", "--warmup", "1",
                        "--transitions", "40", "--short-warmup", "--no-cpu-baseline", "--allow-shared-gpu"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["chains_per_gpu"] == 2048
    assert d["config"]["collective_backend"] in ("gloo", "nccl")
    assert "2048 chains" in d["config"]["workload"]
    assert d["value"] > 1e7 and d["scaling"] == "weak"
    import torch
    if torch.cuda.device_count() < 2:
        assert d["config"]["shared_gpu"] is True
        # without the explicit flag a run that finds fewer devices than ranks REFUSES: no n_gpus line for GPUs that are not there
        q = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--chains", "512", "--steps", "1", "--warmup", "0",
                            "--transitions", "10", "--short-warmup", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
        assert q.returncode != 0 and not [l for l in q.stdout.splitlines() if l.startswith("{")]
        assert "distinct device" in q.stderr


def test_bench_collectives_run_over_rccl():
    """The branch of bench.py that an 8-GPU node takes — RCCL process group, barrier with device ids, all-reduce of time and
    leapfrogs, all-gather of the last draws — executed for real: one rank under torch.distributed.run owns the GPU alone, so
    the device identities are distinct and the collectives go over RCCL (`collective_backend: "nccl"`)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", "29671", os.path.join(root, "bench.py"), "--gpus", "1", "--chains", "1024", "--steps", "2",
                        "--warmup", "1", "--transitions", "40", "--short-warmup", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["collective_backend"] == "nccl"
    assert d["value"] > 1e7


def test_bench_default_line_carries_the_other_configs_and_a_live_traffic_figure():
    """the default single-GPU invocation puts configs 3 / 4 / 5 under the same clock as the headline
    (`other_configs` in the one JSON line) and measures roofline.traffic afresh (two rocprofv3 counter passes of a probe run) instead
    of quoting a committed profile.  Short sizes here; the line's shape is what is checked."""
    import json, os, shutil, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--transitions", "50", "--short-warmup",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 1e7
    oc = d["other_configs"]
    assert set(oc) == {"c3", "c4", "c4_32768", "c5"}
    for k, floor in (("c3", 1e6), ("c4", 1e7), ("c4_32768", 1e8), ("c5", 1e4)):
        assert "error" not in oc[k], oc[k]
        assert oc[k]["value"] > floor and oc[k]["steps"] == 2 and 0 < oc[k]["roofline"]["frac"] < 1.5
        n = oc[k]["at_config_n"]                        # one call of the config's own N beside the 20-transition steps
        assert n["value"] > floor and n["steps"] == 1 and n["transitions_per_step"] == (200 if k == "c5" else 1000)
    if shutil.which("rocprofv3"):
        assert d["roofline"]["traffic_source"] == "live", d["roofline"]["traffic_source"]
        per_leapfrog = d["roofline"]["traffic"] / d["roofline"]["leapfrogs_per_launch"]
        assert 2e3 < per_leapfrog < 1e5                 # ≈ 11 KB per leapfrog in rounds 1-3


def _tree_properties(out):
    """Size-independent invariants of every transition (trees.jl:283-319, NUTS.jl:59-89): a tree of depth d visited between
    2^d - 1 and 2^(d+1) - 1 leaves; a trajectory that ended by turning at the top level spans exactly its 2^d - 1 new points;
    acceptance rates are probabilities; π = ℓ - K <= ℓ."""
    steps, depth = out["steps"], out["depth"].long()
    assert (steps >= 1).all() and (steps <= 2 ** (depth + 1) - 1).all() and (steps >= 2 ** depth - 1).all()
    l, r = out["term_left"], out["term_right"]
    top_turn = (l <= 0) & (r >= 0) & (l < r)
    assert (steps[top_turn] == (2 ** depth[top_turn] - 1)).all()
    assert (r[top_turn] - l[top_turn] == steps[top_turn]).all()
    a = out["acceptance_rate"]
    assert (a >= 0).all() and (a <= 1).all()
    assert (out["pi"] <= out["logdensities"]).all()


def test_config3_full_size_properties(pkg):
    """BASELINE.json configs[2] at FULL size — D = 1000 correlated normal, dense M⁻¹ = Σ, 4096 chains through the two-stream GEMM
    round engine: the stored log density is ℓ of the stored draw (tridiagonal precision, recomputed in torch), the tree
    invariants hold for every transition, the whitened draws have unit variance, and chains 2048..2051 — the first chains of the
    SECOND half-batch — equal a 4-chain oracle at that chain offset bit for bit."""
    import torch
    D, C, N = 1000, 4096, 12
    params, Sigma = _config3_problem(D)
    sig = np.sqrt(np.diag(Sigma))
    dev = pkg.DeviceContext(D, C, metric=ol.METRIC_DENSE, target=ol.TARGET_TRIDIAG_NORMAL, target_params=params, seed=77)
    q0 = np.random.default_rng(6).normal(size=(C, D)) * sig
    dev.set_metric_dense(Sigma); dev.init(q0); dev.find_initial_stepsize()
    eps0 = dev.stepsize()
    dev.run(6, da={}, fields=[])
    out = {k: torch.empty((C, N, D) if k == "draws" else (C, N), dtype=dt, device="cuda")
           for k, dt in (("draws", torch.float64), ("logdensities", torch.float64), ("pi", torch.float64), ("acceptance_rate", torch.float64),
                         ("steps", torch.int64), ("depth", torch.int32), ("term_left", torch.int64), ("term_right", torch.int64))}
    dev.run_into(N, out)
    assert dev.last_run_rounds() > 0
    q = out["draws"]
    diag = torch.tensor(params[:D], device="cuda"); off = torch.tensor(params[D:2 * D - 1], device="cuda")
    quad = (diag * q * q).sum(-1) + 2 * (off * q[..., :-1] * q[..., 1:]).sum(-1)
    assert torch.allclose(out["logdensities"], -0.5 * quad, rtol=1e-11, atol=1e-8)
    _tree_properties(out)
    assert 0.6 < float(out["acceptance_rate"].mean()) < 0.97
    w = q / torch.tensor(sig, device="cuda")
    assert abs(float(w.var()) - 1) < 0.02 and abs(float(w.mean())) < 5e-3
    off0 = 2048
    ora = ol.Oracle(D, 4, metric=ol.METRIC_DENSE, target=ol.TARGET_TRIDIAG_NORMAL, params=params, seed=77, chain_offset=off0, threads=4)
    ora.set_metric_dense(Sigma); ora.init(q0[off0:off0 + 4]); ora.find_initial_stepsize()
    assert np.array_equal(eps0[off0:off0 + 4], ora.stepsize())
    ora.run(6, da={}, fields=[])
    b = ora.run(N, fields=["draws", "steps", "acceptance_rate"])
    assert np.array_equal(b["draws"], q[off0:off0 + 4].cpu().numpy())
    assert np.array_equal(b["steps"], out["steps"][off0:off0 + 4].cpu().numpy())
    assert np.array_equal(b["acceptance_rate"], out["acceptance_rate"][off0:off0 + 4].cpu().numpy())


def test_config5_full_size_properties(pkg):
    """BASELINE.json configs[4], one GPU's share at FULL size — N = 10⁵ observations, p = 256, 1024 chains through the
    GEMM-gradient round engine: the stored log density is ℓ of the stored draw (recomputed in torch as one fp64 matmul), the tree
    invariants hold, the position the context ends on is the last draw, and its gradient is Xᵀ(y - σ(Xβ)) - β."""
    import torch
    rng = np.random.default_rng(0)
    Nobs, D, C, N = 100000, 256, 1024, 3
    X = rng.normal(size=(Nobs, D)) / 16
    y = (rng.random(Nobs) < 1 / (1 + np.exp(-X @ rng.normal(size=D)))).astype(float)
    dev = pkg.DeviceContext(D, C, target=ol.TARGET_LOGISTIC, target_params=ol.target_params_blob(ol.TARGET_LOGISTIC, D, X=X, y=y), seed=5)
    dev.init(); dev.set_stepsize(0.02)
    dev.run(3, da={}, fields=[])
    out = {k: torch.empty((C, N, D) if k == "draws" else (C, N), dtype=dt, device="cuda")
           for k, dt in (("draws", torch.float64), ("logdensities", torch.float64), ("pi", torch.float64), ("acceptance_rate", torch.float64),
                         ("steps", torch.int64), ("depth", torch.int32), ("term_left", torch.int64), ("term_right", torch.int64))}
    dev.run_into(N, out)
    assert dev.last_run_rounds() > 0
    Xt = torch.tensor(X, device="cuda"); yt = torch.tensor(y, device="cuda")
    q = out["draws"].reshape(-1, D)
    eta = q @ Xt.T
    lq = (yt * eta - torch.nn.functional.softplus(eta)).sum(1) - 0.5 * (q * q).sum(1)
    assert torch.allclose(out["logdensities"].reshape(-1), lq, rtol=1e-11)
    _tree_properties(out)
    assert (out["steps"] >= 3).all()
    qf, lqf, gf = dev.position()
    assert np.array_equal(qf, out["draws"][:, -1].cpu().numpy()) and np.array_equal(lqf, out["logdensities"][:, -1].cpu().numpy())
    ql = out["draws"][:, -1]
    el = ql @ Xt.T
    g = (yt - torch.sigmoid(el)) @ Xt - ql
    assert np.allclose(gf, g.cpu().numpy(), rtol=1e-9, atol=1e-9)
