#!/usr/bin/env python
"""bench.py — leapfrog-steps/sec (all chains) + ESS/sec of the many-chain NUTS hot path.

Workload (BASELINE.json configs[1]): 1000-dim standard MVN, diagonal mass matrix, 4096 chains
per MI355X.  Untimed setup is the reference's default warmup (random positions mcmc.jl:108, initial
step size search, 900 dual-averaging transitions with per-chain diagonal metric updates after the
25/50/100/200/400 windows, mcmc.jl:415-425) so the timed region runs at adapted per-chain ϵ and M⁻¹.

One "step" = one dhmc_run call = one pass of the per-draw loop (mcmc.jl:374-379) of
`--transitions` NUTS transitions (default 1000: ≈0.2 s of GPU work, so that K = 10..20 timed steps
are several seconds of GPU activity) for every chain, draws and tree statistics written to buffers already
resident in HBM.  value = Σ leapfrog steps of all chains and ranks ÷ wall time of the K timed steps
(barrier + synchronize on both sides, max over ranks).  The warmup phase (adaptation on, metric
updates included) is timed separately during setup and reported as `warmup_phase`.

roofline: the chain state is register/LDS-resident, so the bound that applies is the fp64 vector
unit (≈30·D useful flops per leapfrog against 78.6 TFLOP/s); the 48·D-byte streaming
model and the measured HBM traffic are kept as secondary keys inside `roofline`.

After the headline's timed region and the CPU baseline, the default single-GPU run also takes short measurements of
BASELINE.json configs[2..4] (dense metric / funnel share / logistic share: `other_configs` in the same JSON line, so that they
are under the driver's clock too; --no-other-configs skips them) and, when rocprofv3 is on PATH, measures the per-draw kernel's
HBM traffic afresh in two counter passes of a small probe run (`roofline.traffic`, `traffic_source: "live"`; --traffic profile
falls back to the committed profiles/*traffic.json and names the commit it was taken at).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

D = 1000
CHAINS_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
ALGO_BYTES_PER_LEAPFROG = 48 * D   # read q,p,∇ℓ + write q',p',∇ℓ' in fp64
VALU_PEAK_TFLOPS = 78.6        # MI355X_MICROARCH.md: fp64 vector peak
ALGO_FLOPS_PER_LEAPFROG = 30 * D   # 6D integrator + 3D density + 3D kinetic + D p♯ + D ρ + ≈15D amortised turn checks


ALL_RUN_LEAPFROGS = [0]   # every dhmc_run of this process (setup + warmup + timed + ESS), for the PMC summaries


def _run(ctx, n, arrays, **kw):
    ctx.run_into(n, arrays, **kw)
    ALL_RUN_LEAPFROGS[0] += ctx.last_run_leapfrogs()


def measured_traffic_per_leapfrog():
    """HBM bytes per leapfrog from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE, separate runs; tools/profile.sh -> profiles/*traffic.json), with the commit the file was
    last touched at (a kernel changed since then makes the figure stale), or None."""
    import glob
    import subprocess
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        val = json.load(fh).get("hbm_bytes_per_leapfrog")
    src = os.path.basename(files[-1])
    try:
        c = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h %cs", "--", files[-1]], capture_output=True, text=True, timeout=10).stdout.strip()
        if c:
            src += f" (committed {c})"
    except Exception:
        pass
    return val, src


def live_traffic_per_leapfrog(seed):
    """HBM bytes per leapfrog of the per-draw kernel measured NOW: two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE
    cannot share a pass; counters only — no trace domains beside them) over a small probe run of this file
    (--traffic-probe: the same workload with the short setup and one 100-transition step), summed over its nuts_run_kernel
    dispatches and divided by the leapfrogs the probe reports.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for
    gfx950.  None when rocprofv3 is missing or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    tot, leapfrogs = {}, None
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        env = dict(os.environ, TMPDIR="/tmp")
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--output-format", "csv", "--pmc", counter, "-d", d, "-o", "t", "--", sys.executable, os.path.abspath(__file__),
                   "--traffic-probe", "--seed", str(seed)]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd="/tmp", env=env)
            except Exception as e:      # noqa
                return None, f"rocprofv3 pass failed: {e!r}"
            try:
                leapfrogs = json.loads(r.stdout.strip().splitlines()[-1])["probe_leapfrogs"]
            except Exception:
                return None, "probe run printed no leapfrog count: " + (r.stderr or r.stdout)[-200:]
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if "nuts_run_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                            tot[counter] = tot.get(counter, 0.0) + float(row["Counter_Value"])
    if "FETCH_SIZE" not in tot or "WRITE_SIZE" not in tot or not leapfrogs:
        return None, "counter rows missing"
    return (tot["FETCH_SIZE"] * 1024 * 2 + tot["WRITE_SIZE"] * 1024) / leapfrogs, "live"


def traffic_probe(pkg, torch, seed):
    """The run the counter passes wrap: short adaptive setup, then one step of 100 transitions; prints the leapfrogs of ALL its
    dhmc_run calls (the counters are summed over all dispatches of the kernel too)."""
    ctx, _ = setup_context(pkg, torch, 0, CHAINS_PER_GPU, seed, True)
    out = {"draws": torch.empty((CHAINS_PER_GPU, 100, D), dtype=torch.float64, device="cuda")}
    _run(ctx, 100, out)
    torch.cuda.synchronize()
    print(json.dumps({"probe_leapfrogs": ALL_RUN_LEAPFROGS[0]}))


def setup_context(pkg, torch, rank, chains, seed, short, keep_draws=False):
    stream = torch.cuda.current_stream().cuda_stream
    ctx = pkg.DeviceContext(D, chains, seed=seed, chain_offset=rank * chains, device=torch.cuda.current_device(),
                            stream=stream)
    ctx.init()
    ctx.find_initial_stepsize()
    # default_warmup_stages (mcmc.jl:415-425): 75 stepsize-only, metric windows 25/50/100/200/400, 50 stepsize-only
    stages = [(20, False), (25, True), (20, False)] if short else \
        [(75, False), (25, True), (50, True), (100, True), (200, True), (400, True), (50, False)]
    torch.cuda.synchronize()
    t0 = time.perf_counter(); lf0 = ALL_RUN_LEAPFROGS[0]; kms = 0.0
    bufs = {n: torch.empty((chains, n, D), dtype=torch.float64, device="cuda") for n, metric in stages if metric and keep_draws}
    for n, metric in stages:
        if metric and not keep_draws:
            ctx.metric_window_begin()          # the stage's draws go into per-chain running moments, not into a [C][n][D] buffer
        _run(ctx, n, {"draws": bufs[n]} if metric and keep_draws else {}, da={})
        kms += ctx.last_run_kernel_ms()
        if metric and keep_draws:              # --warmup-draws: round 3's path, the two-pass update of stored windows (25 GB at 8192 chains)
            ctx.update_metric_diag(bufs[n])
        elif metric:
            ctx.update_metric_diag_window()
    del bufs
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lf = ALL_RUN_LEAPFROGS[0] - lf0
    warm = {"value": lf / dt, "unit": "leapfrog-steps/s", "seconds": dt, "leapfrogs": lf, "transitions": sum(n for n, _ in stages),
            "kernel_ms": kms, "note": "TuningNUTS stages with dual averaging on, the metric windows' draws accumulated into per-chain "
                                       "running moments by the kernel (dhmc_metric_window_begin: no posterior matrix of a warmup "
                                       "stage is stored), per-chain diagonal metric updates included; this rank"}
    return ctx, warm


def bulk_ess_min(pkg, torch, draws, ncoord=16):
    """min bulk ESS over a spread of `ncoord` coordinates, computed on the GPU (diagnostics.ess_bulk_device)."""
    idx = torch.linspace(0, draws.shape[2] - 1, ncoord, device=draws.device).long()
    ess, _ = pkg.diagnostics.ess_bulk_device(draws, idx)
    return float(ess.min())


def cpu_baseline(transitions, threads):
    """The C++ oracle (restatement of the reference; Julia is not installed anywhere in this
    environment) on the host cores, same workload shape on a bounded sample of chains."""
    import oracle_lib as ol
    chains = 8 * threads
    o = ol.Oracle(D, chains, seed=1234, threads=threads)
    o.init(); o.find_initial_stepsize()
    o.run(20, da={}, fields=[])
    o.metric_window_begin()
    o.run(25, da={}, fields=[])
    o.update_metric_diag_window()
    o.run(20, da={}, fields=[])
    t0 = time.perf_counter()
    r = o.run(transitions, fields=["steps"])
    dt = time.perf_counter() - t0
    nsteps = int(r["steps"].sum())
    return {"value": nsteps / dt, "unit": "leapfrog-steps/s", "cores": threads, "kind": "port",
            "sample": f"{chains} chains x {transitions} transitions, D={D}, adapted eps/diag metric, "
                      f"{nsteps} leapfrog steps in {dt:.1f} s (C++ oracle, -O3, one chain per thread)"}


_WORK_STREAMS = []


def _work_stream(torch):
    """A non-default stream for a library context (kept alive here).  The round engines capture their rounds into a
    hipGraph, which the legacy default stream does not allow; dhmc_run synchronises its stream before it returns, and the
    timed regions are bracketed by device-wide synchronisation, so torch's own stream needs no extra ordering."""
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    _WORK_STREAMS.append(s)
    return s.cuda_stream


def bench_config3(args, pkg, torch):
    """BASELINE.json configs[2]: 1000-dim correlated MVN (tridiagonal precision), dense M⁻¹ = Σ shared by all
    chains, 4096 chains, sampling at a fixed per-chain ϵ found by dual averaging.  Not the driver's line (that is
    configs[1]); run with --config 3 for the MFMA roofline of the dense path."""
    Dd, C, T, K = 1000, args.chains, args.transitions, args.steps
    rho = 0.5
    sig = np.logspace(-1, 1, Dd)
    Pc = np.zeros(Dd) + (1 + rho ** 2) / (1 - rho ** 2); Pc[0] = Pc[-1] = 1 / (1 - rho ** 2)
    diag = Pc / sig ** 2
    off = np.zeros(Dd); off[:Dd - 1] = -rho / (1 - rho ** 2) / (sig[:-1] * sig[1:])
    idx = np.arange(Dd)
    Sigma = np.outer(sig, sig) * rho ** np.abs(idx[:, None] - idx[None, :])
    ctx = pkg.DeviceContext(Dd, C, metric=pkg.abi.METRIC_DENSE, target=pkg.abi.TARGET_TRIDIAG_NORMAL,
                            target_params=np.concatenate([diag, off]), seed=args.seed,
                            stream=_work_stream(torch))
    ctx.set_metric_dense(Sigma)
    ctx.init(np.random.default_rng(5).normal(size=(C, Dd)) * sig)
    ctx.find_initial_stepsize()
    ctx.run_into(40, {}, da={})
    nprod = ctx.dense_products()                          # 1: u′ = ∇ℓq′·M⁻¹ only (default); 2: the reference's M⁻¹pₘ and M⁻¹p′
    peak = 78.6                                           # MI355X fp64 matrix peak, TFLOP/s

    def measure(T, K, W):
        out = {"draws": torch.empty((C, T, Dd), dtype=torch.float64, device="cuda"),
               "steps": torch.empty((C, T), dtype=torch.int64, device="cuda"),
               "depth": torch.empty((C, T), dtype=torch.int32, device="cuda"),
               "acceptance_rate": torch.empty((C, T), dtype=torch.float64, device="cuda")}
        for _ in range(W):
            ctx.run_into(T, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); lf = 0; rounds = 0; kms = []
        for _ in range(K):
            ctx.run_into(T, out)
            lf += ctx.last_run_leapfrogs(); rounds += ctx.last_run_rounds(); kms.append(ctx.last_run_kernel_ms())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # 2·D² flops per M⁻¹ product and chain; the kernels multiply padded rows (Dpad = 1024), kept as a second figure
        ach = rounds * nprod * 2.0 * C * Dd * Dd / (sum(kms) * 1e-3) / 1e12
        ach_pad = rounds * nprod * 2.0 * C * 1024 * 1024 / (sum(kms) * 1e-3) / 1e12
        q = out["draws"]
        res = {"value": lf / dt, "ms_per_step": 1e3 * dt / K, "steps": K, "transitions_per_step": T,
               "tree": {"mean_depth": float(out["depth"].double().mean()), "mean_leapfrogs_per_transition": float(out["steps"].double().mean()),
                        "mean_acceptance": float(out["acceptance_rate"].mean()),
                        "scaled_draw_var": float((q / torch.tensor(sig, device="cuda")).var())},
               "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                            "kernel": "gemm_rows_f64_kernel (v_mfma_f64_16x16x4_f64), share of whole round time",
                            "algorithmic_flops_per_leapfrog": nprod * 2.0 * Dd * Dd,
                            "padded": {"achieved": ach_pad, "frac": ach_pad / peak, "flops_per_leapfrog": nprod * 2.0 * 1024 * 1024},
                            "note": f"flops of the {nprod} M^-1 contraction(s) per leapfrog round (2·D² each, SURVEY.md §8d; `padded`: the 2·Dpad² "
                                    "the kernels execute; one-product recurrence: include/dhmc.h dhmc_set_dense_products) / total kernel time "
                                    "of the rounds (tree-logic kernels included); the GEMM launches alone reach ~50 TFLOP/s (profiles/)",
                            "rounds": rounds}}
        del out, q
        torch.cuda.empty_cache()
        return res
    m = measure(args.transitions, args.steps, args.warmup)
    line = {
        "metric": "leapfrog-steps/sec (all chains), 1000-dim correlated MVN, dense M^-1, @4096 chains",
        "value": m["value"], "unit": "leapfrog-steps/s", "n_gpus": 1, "steps": m["steps"], "warmup": args.warmup,
        "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "1000-dim correlated MVN (rho=0.5, sigma log-spaced 0.1..10), dense M^-1 = Sigma shared, "
                               f"{C} chains (BASELINE.json configs[2])", "transitions_per_step": args.transitions, "chains_per_gpu": C,
                   "dense_products_per_leapfrog": nprod},
        "tree": m["tree"], "roofline": m["roofline"]}
    if getattr(args, "config_n", 0):                      # the config's own N (1000 draws) in one call, beside the short steps
        line["at_config_n"] = measure(args.config_n, 1, 0)
    return line


def bench_config45(args, pkg, torch):
    """One GPU's share of BASELINE.json configs[3] (Neal's funnel D=30, 32768 chains over 8 GPUs -> 4096 here) and
    configs[4] (logistic regression N=1e5, p=256, 8192 chains over 8 GPUs -> 1024 here).  Not the driver's line."""
    T, K = args.transitions, args.steps
    if args.config == 4:
        D, C = 30, 4096 if args.chains == CHAINS_PER_GPU else args.chains
        ctx = pkg.DeviceContext(D, C, target=pkg.abi.TARGET_FUNNEL, seed=args.seed, stream=_work_stream(torch))
        ctx.init(); ctx.find_initial_stepsize()
        for n, metric in ((75, False), (25, True), (50, True), (100, True), (200, True), (50, False)):
            if metric:
                ctx.metric_window_begin()
            ctx.run_into(n, {}, da={})
            if metric:
                ctx.update_metric_diag_window()
        name = f"Neal's funnel D=30, diagonal metric, {C} chains" + (" = one GPU's share of 32768" if C == 4096 else " (all of them on this GPU)" if C == 32768 else "") + " (BASELINE.json configs[3])"
        flops_per_leapfrog = None
    else:
        N, D, C = 100000, 256, 1024 if args.chains == CHAINS_PER_GPU else args.chains
        rng = np.random.default_rng(0)
        X = rng.normal(size=(N, D)) / 16
        y = (rng.random(N) < 1 / (1 + np.exp(-X @ rng.normal(size=D)))).astype(float)
        ctx = pkg.DeviceContext(D, C, target=pkg.abi.TARGET_LOGISTIC, target_params=pkg.LogisticRegression(X, y).params(),
                                seed=args.seed, stream=_work_stream(torch))
        ctx.init(); ctx.set_stepsize(0.02)
        # A shortened default warmup (mcmc.jl:415-425: step-size stage, doubling metric windows, final step-size stage; 200 transitions,
        # ≈ 9 s).  Rounds 1-4 ran 20 + 15 transitions here, which left the chains at step sizes 5 × too small and ragged (0.05 … 0.37
        # against 0.35 ± 0.05 adapted): 49 leapfrogs per transition instead of 15, the slowest chain at 2.7 × the mean's work, so that
        # 63 % of the rows of the round engine's products idled (tools/experiments/c5_warmup_probe.py, profiles/r05_config5_*).
        for n, metric in ((40, False), (40, True), (80, True), (40, False)):
            if metric:
                ctx.metric_window_begin()
            ctx.run_into(n, {}, da={})
            if metric:
                ctx.update_metric_diag_window()
        name = f"logistic regression N=1e5 p=256, diagonal metric, {C} chains" + (" = one GPU's share of 8192" if C == 1024 else " (all of them on this GPU)" if C == 8192 else "") + " (BASELINE.json configs[4])"
        flops_per_leapfrog = 4.0 * 100032 * 256      # two GEMM passes over X per gradient
    def measure(T, K, W):
        out = {"steps": torch.empty((C, T), dtype=torch.int64, device="cuda"), "depth": torch.empty((C, T), dtype=torch.int32, device="cuda"),
               "acceptance_rate": torch.empty((C, T), dtype=torch.float64, device="cuda")}
        for _ in range(W):
            ctx.run_into(T, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); lf = 0; kms = []
        for _ in range(K):
            ctx.run_into(T, out)
            lf += ctx.last_run_leapfrogs(); kms.append(ctx.last_run_kernel_ms())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        work = out["steps"].sum(dim=1).double()           # the last step's leapfrogs per chain: a call ends with its slowest chain
        if flops_per_leapfrog:
            ach = lf * flops_per_leapfrog / (sum(kms) * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": ach, "peak": 78.6, "unit": "TFLOP/s", "frac": ach / 78.6, "traffic": None,
                    "kernel": "logistic_eta_link_kernel (fp64 MFMA: Q'·Xᵀ over the active rows with the link in its epilogue) + gemm_rows_f64_kernel (R·X split over blocks of observations), share of whole round time",
                    "note": "4·N·p flops per leapfrog of the chains that took it / total kernel time of the rounds"}
        else:
            ach = lf * 48.0 * D / (sum(kms) * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None,
                    "kernel": "nuts_run_pipeline_kernel<FunnelT> / nuts_run_packed_kernel / nuts_run_kernel<FunnelT,1> (chosen per launch from the previous launch's work: dhmc_run)",
                    "note": "48·D algorithmic bytes per leapfrog; a 30-dim chain is latency-, not bandwidth-bound: the call ends with its slowest "
                            "chain, whose leapfrogs are sequential (slowest_chain_leapfrogs × the kernel's latency per leapfrog ≈ the call's time)"}
        return {"value": lf / dt, "ms_per_step": 1e3 * dt / K, "steps": K, "transitions_per_step": T,
                "tree": {"mean_depth": float(out["depth"].double().mean()), "mean_leapfrogs_per_transition": float(out["steps"].double().mean()),
                         "mean_acceptance": float(out["acceptance_rate"].mean()),
                         "slowest_chain_leapfrogs": float(work.max()), "mean_chain_leapfrogs": float(work.mean())},
                "roofline": roof}
    m = measure(args.transitions, args.steps, args.warmup)
    line = {
        "metric": "leapfrog-steps/sec (all chains)", "value": m["value"], "unit": "leapfrog-steps/s", "n_gpus": 1, "steps": m["steps"],
        "warmup": args.warmup, "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "config": {"workload": name, "transitions_per_step": args.transitions, "chains_per_gpu": C},
        "tree": m["tree"], "roofline": m["roofline"]}
    if getattr(args, "config_n", 0):                      # the config's own N (funnel 1000 draws, logistic 200) in one call
        line["at_config_n"] = measure(args.config_n, 1, 0)
    return line


def other_configs(args, pkg, torch):
    """Short measurements of BASELINE.json configs[2], [3] (one GPU's share) and [4] (one GPU's share) for the default line:
    the same code paths as --config 3 / 4 / 5, two timed steps of 20 transitions after one warm-up step each, and then ONE call of
    the config's own N (1000 draws for the dense normal and the funnel, 200 for the logistic regression) as
    `at_config_n` — a call ends with its slowest chain, and a 20-transition call of a model with a heavy-tailed tree size (the funnel,
    the logistic regression) is mostly that wait."""
    import copy
    res = {}
    # (c4_32768: ALL of configs[3]'s chains on this one GPU — what the packed kernel's queue of places is for)
    for cfg, key, fn, config_n, chains in ((3, "c3", bench_config3, 1000, CHAINS_PER_GPU), (4, "c4", bench_config45, 1000, CHAINS_PER_GPU),
                                           (4, "c4_32768", bench_config45, 1000, 32768), (5, "c5", bench_config45, 200, CHAINS_PER_GPU)):
        a = copy.copy(args)
        a.config, a.transitions, a.steps, a.warmup, a.chains, a.config_n = cfg, 20, 2, 1, chains, config_n
        t0 = time.perf_counter()
        try:
            line = fn(a, pkg, torch)
            res[key] = {"workload": line["config"]["workload"], "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"],
                        "steps": line["steps"], "transitions_per_step": 20, "tree": line["tree"], "roofline": line["roofline"],
                        "at_config_n": line.get("at_config_n"), "seconds_with_setup": None}
        except Exception as e:      # noqa: the headline line is what must come out
            res[key] = {"error": repr(e)}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        res[key]["seconds_with_setup"] = time.perf_counter() - t0
    return res


def spawn_ranks(n):
    """Re-run this command line under torch.distributed.run with n ranks on this node (rendezvous on 127.0.0.1, a free
    port); the ranks' stdout/stderr are inherited, the return code is the launcher's."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config: 2 (default, diagonal metric), 3 (dense metric), 4 / 5 (one GPU's share of the funnel / logistic configs)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--transitions", type=int, default=None, help="NUTS transitions per chain per step (default 1000 for config 2, 20 otherwise)")
    ap.add_argument("--chains", type=int, default=CHAINS_PER_GPU, help="chains per GPU")
    ap.add_argument("--config-n", type=int, default=0, help="--config 3 / 4 / 5: after the timed steps, one more call of this many transitions "
                                                            "(the config's own N), reported as at_config_n")
    ap.add_argument("--warmup-draws", action="store_true", help="A/B: the metric windows' draws stored and re-read (dhmc_update_metric_diag) "
                                                                 "instead of accumulated by the kernel (dhmc_metric_window_begin)")
    ap.add_argument("--short-warmup", action="store_true", help="65-transition adaptive setup instead of the reference's 900")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-transitions", type=int, default=150)
    ap.add_argument("--seed", type=int, default=0x23EF614D)
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short configs[2..4] measurements attached to the default line")
    ap.add_argument("--traffic", choices=["live", "profile", "none"], default="live",
                    help="roofline.traffic: measured now with rocprofv3 counter passes (default, single GPU), from profiles/*traffic.json, or left out")
    ap.add_argument("--traffic-probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--allow-shared-gpu", action="store_true",
                    help="let several ranks share one physical GPU (collectives over gloo): a readiness run of the multi-process path, not a measurement of N GPUs")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, one per GPU, exactly as the driver's
        # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` form does; rank 0 prints the one JSON line
        return spawn_ranks(args.gpus)
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = max(1, torch.cuda.device_count())
    local_rank %= ndev                                   # a launcher that hides all but one GPU per rank leaves one device
    torch.cuda.set_device(local_rank)
    dist = None
    pg = None
    coll_dev = "cuda"
    backend = None
    if world > 1 or "WORLD_SIZE" in os.environ:       # (a launcher's single rank takes the RCCL path too: it is the 8-GPU code)
        import torch.distributed as dist
        # rendezvous over gloo first: do two ranks sit on the same physical GPU (a readiness run of the multi-process path
        # on a 1-GPU box)?  RCCL refuses that ("Duplicate GPU detected"), so then the three small collectives stay on gloo
        # with host tensors; the per-rank HIP path, the sharding and the reductions are the ones of the 8-GPU run.
        dist.init_process_group("gloo")
        props = torch.cuda.get_device_properties(local_rank)
        ident = str(getattr(props, "uuid", "")) or str(getattr(props, "pci_bus_id", "")) or props.name
        # (the device index and the visibility masks are part of the identity: eight ranks on eight indices, or on one
        # visible device each, are eight GPUs whatever the uuid field holds)
        ident += f"|dev{local_rank}|" + os.environ.get("ROCR_VISIBLE_DEVICES", "") + "|" + os.environ.get("HIP_VISIBLE_DEVICES", "")
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        if len(set(idents)) < world:
            # fewer distinct devices than ranks: a line with n_gpus = N measured on fewer GPUs would be a lie about the hardware.
            # Only a readiness run of the multi-process path on a small box asks for it explicitly.
            if not args.allow_shared_gpu:
                if rank == 0:
                    print(f"bench.py: --gpus {world} but only {len(set(idents))} distinct device(s) among the ranks ({sorted(set(idents))}); "
                          "refusing to print an n_gpus line for GPUs that are not there (--allow-shared-gpu runs the ranks on shared "
                          "devices over gloo, for testing the multi-process path)", file=sys.stderr)
                dist.destroy_process_group()
                raise SystemExit(3)
            backend, coll_dev = "gloo", "cpu"
        else:
            backend = "nccl"
            pg = dist.new_group(backend="nccl")
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    if args.traffic_probe:
        return traffic_probe(pkg, torch, args.seed)
    if args.transitions is None:
        args.transitions = 1000 if args.config == 2 else 20
    if args.config == 3:
        return print(json.dumps(bench_config3(args, pkg, torch)))
    if args.config in (4, 5):
        return print(json.dumps(bench_config45(args, pkg, torch)))
    C, T, K, Wn = args.chains, args.transitions, args.steps, args.warmup
    ctx, warm = setup_context(pkg, torch, rank, C, args.seed, args.short_warmup, args.warmup_draws)
    out = {
        "draws": torch.empty((C, T, D), dtype=torch.float64, device="cuda"),
        "steps": torch.empty((C, T), dtype=torch.int64, device="cuda"),
        "depth": torch.empty((C, T), dtype=torch.int32, device="cuda"),
        "acceptance_rate": torch.empty((C, T), dtype=torch.float64, device="cuda"),
        "logdensities": torch.empty((C, T), dtype=torch.float64, device="cuda"),
    }
    keep_draws = []

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            if backend == "nccl":
                dist.barrier(group=pg, device_ids=[local_rank])
            else:
                dist.barrier(group=pg)
            torch.cuda.synchronize()

    for _ in range(Wn):
        _run(ctx, T, out)
    sync()
    kernel_ms, leapfrogs = [], 0
    t0 = time.perf_counter()
    for _ in range(K):
        _run(ctx, T, out)
        kernel_ms.append(ctx.last_run_kernel_ms())
        leapfrogs += ctx.last_run_leapfrogs()
    sync()
    dt = time.perf_counter() - t0

    # untimed: statistics and ESS of the last timed step's draws, gather over RCCL
    mean_depth = float(out["depth"].double().mean())
    mean_acc = float(out["acceptance_rate"].mean())
    mean_steps = float(out["steps"].double().mean())
    per_chain = out["steps"].sum(1).double()         # the launch ends with its slowest chain: one wave walks a chain's T transitions
    slowest = float(per_chain.max() / per_chain.mean())
    q = out["draws"]
    mom = (float(q[:, -20:].mean()), float(q[:, -20:].var()))
    ess_T = min(T, 1000)
    ess_dt = dt / K * ess_T / T
    ess = bulk_ess_min(pkg, torch, q[:, T - ess_T:].contiguous() if ess_T < T else q)

    t_max, total_leapfrogs, ess_rate = dt, leapfrogs, ess / ess_dt
    if dist is not None:
        tt = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=pg)
        ll = torch.tensor([leapfrogs, ess / ess_dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(ll, op=dist.ReduceOp.SUM, group=pg)
        t_max, total_leapfrogs, ess_rate = float(tt[0]), float(ll[0]), float(ll[1])
        ss = torch.tensor([slowest], device=coll_dev, dtype=torch.float64)       # the rank whose slowest chain is furthest above its mean
        dist.all_reduce(ss, op=dist.ReduceOp.MAX, group=pg)                       # sets the job's time (profiles/r04_straggler_chain.txt)
        slowest = float(ss[0])
        # the one collective of the path: gather the last draw of every chain over RCCL/xGMI
        last = out["draws"][:, -1, :].contiguous().to(coll_dev)
        gathered = torch.empty((world * C, D), dtype=torch.float64, device=coll_dev)
        if backend == "nccl":
            dist.all_gather_into_tensor(gathered, last, group=pg)
        else:
            dist.all_gather(list(gathered.view(world, C, D).unbind(0)), last, group=pg)
        torch.cuda.synchronize()
        assert bool(torch.equal(gathered[rank * C:(rank + 1) * C], last))

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        per_launch = leapfrogs / K
        achieved = per_launch * ALGO_BYTES_PER_LEAPFROG / (k_ms * 1e-3) / 1e9
        valu_ach = per_launch * ALGO_FLOPS_PER_LEAPFROG / (k_ms * 1e-3) / 1e12
        kernel_name = "nuts_run_kernel<StdNormalT,16,true>"
        tpl, tsrc = None, None
        if args.traffic == "live" and dist is None:
            tpl, tsrc = live_traffic_per_leapfrog(args.seed)
            if tpl is None:
                why = tsrc
                tpl, tsrc = measured_traffic_per_leapfrog()
                tsrc = f"{tsrc}; live measurement unavailable ({why})"
        elif args.traffic != "none":
            tpl, tsrc = measured_traffic_per_leapfrog()
        line = {
            "metric": "leapfrog-steps/sec (all chains) + ESS/sec, 1000-dim MVN @4096 chains",
            "value": total_leapfrogs / t_max,
            "unit": "leapfrog-steps/s",
            "n_gpus": world, "steps": K, "warmup": Wn,
            "ms_per_step": 1e3 * t_max / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"1000-dim standard MVN, per-chain diagonal mass matrix, {C} chains per MI355X "
                                   "(BASELINE.json configs[1])",
                       "dim": D, "chains_per_gpu": C, "transitions_per_step": T,
                       "phase": "sampling (fixed adapted eps and M^-1 per chain)", "max_depth": 10,
                       "parallelism": f"chains sharded x{world}, no data-path collective",
                       "collective_backend": backend, "shared_gpu": bool(args.allow_shared_gpu and backend == "gloo")},
            "ess_per_sec": ess_rate,
            "ess_note": f"min rank-normalised split-chain bulk ESS (Vehtari et al. 2021; dhmc_ess_bulk) over 16 coordinates, "
                        f"the last {ess_T} draws x all chains of the last timed step, over that share of the step's time",
            "tree": {"mean_depth": mean_depth, "mean_leapfrogs_per_transition": mean_steps,
                     "mean_acceptance": mean_acc, "draw_mean": mom[0], "draw_var": mom[1],
                     "slowest_chain_over_mean_leapfrogs": slowest,
                     "slowest_chain_note": "a launch ends with its slowest chain (one wave walks a chain's transitions; 4096 chains = four "
                                           "rounds of 1024 resident waves): the largest per-chain leapfrog count of the last step over "
                                           "the mean, max over ranks; 1.0 = no chain holds a round open, 1.48 cost 11 % in "
                                           "profiles/r04_straggler_chain.txt"},
            "warmup_phase": warm,
            "roofline": {"bound": "valu", "achieved": valu_ach, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": valu_ach / VALU_PEAK_TFLOPS,
                         "traffic": None if tpl is None else tpl * per_launch, "traffic_source": tsrc,
                         "kernel": kernel_name, "kernel_ms": k_ms,
                         "algorithmic_flops_per_leapfrog": ALGO_FLOPS_PER_LEAPFROG,
                         "leapfrogs_per_launch": per_launch,
                         "note": "≈30·D useful flops per leapfrog (6D integrator, 3D density, 4D kinetic terms, ≈15D amortised "
                                 "turn checks; an unfused multiply or add counts 1, the peak counts an fma as 2) x leapfrogs per "
                                 "launch / kernel time (HIP events on the launch stream) against the fp64 vector peak; "
                                 "traffic = real HBM bytes per launch from the PMC passes",
                         # measured on this part (profiles/r06_mfma_vgpr_form.txt §7): what ONE wave per SIMD — this kernel's layout — can issue
                         "one_wave_per_simd_ceiling": {"value": 33.3, "unit": "TFLOP/s", "frac_of_it": valu_ach / 33.3,
                                                       "note": "a lone wave with eight independent v_fma_f64 chains: one fma per ~9.4 clocks"},
                         # the streaming model SURVEY.md §8(d) starts from: the chain state never leaves the CU, so this exceeds 1
                         "hbm_model": {"algorithmic_bytes_per_leapfrog": ALGO_BYTES_PER_LEAPFROG, "achieved": achieved,
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}},
            "all_run_leapfrogs": ALL_RUN_LEAPFROGS[0],
        }
        if not args.no_cpu_baseline and world == 1:   # reported baseline: rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(args.cpu_transitions, os.cpu_count() or 1)
        if dist is None and not args.no_other_configs:
            del out, q, ctx
            torch.cuda.empty_cache()
            line["other_configs"] = other_configs(args, pkg, torch)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
