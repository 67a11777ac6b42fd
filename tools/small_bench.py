"""Throughput of the other BASELINE configs on one GPU (not the driver's bench line)."""
import sys, os, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from __graft_entry__ import load_package
import oracle_lib as ol
pkg = load_package()


def run(name, ctx, warm, n):
    ctx.init(); ctx.find_initial_stepsize()
    r = ctx.run(warm, da={}, fields=["draws"])
    ctx.update_metric_diag(r["draws"])
    ctx.run(warm // 2, da={}, fields=[])
    ctx.run(n, fields=[]); ms = ctx.last_run_kernel_ms(); lf = ctx.last_run_leapfrogs()
    print(json.dumps({"config": name, "leapfrog_steps_per_s": lf / ms * 1e3, "kernel_ms": ms, "leapfrogs": lf,
                      "mean_leapfrogs_per_transition": lf / (n * ctx.C)}))


which = sys.argv[1:] or ["c4", "c5small", "c1"]
if "c4" in which:
    run("config 4 share: funnel D=30, 4096 chains", pkg.DeviceContext(30, 4096, target=ol.TARGET_FUNNEL, seed=1), 100, 200)
if "c1" in which:
    run("config 1 shape on GPU: D=100 std normal, 4096 chains", pkg.DeviceContext(100, 4096, seed=1), 100, 200)
if "c5" in which:
    rng = np.random.default_rng(0); N, D = 100000, 256
    X = rng.normal(size=(N, D)) / 16; y = (rng.random(N) < 1 / (1 + np.exp(-X @ rng.normal(size=D)))).astype(float)
    ctx = pkg.DeviceContext(D, 1024, target=ol.TARGET_LOGISTIC, target_params=ol.target_params_blob(ol.TARGET_LOGISTIC, D, X=X, y=y), seed=1)
    ctx.init(); ctx.set_stepsize(0.02)
    r = ctx.run(20, da={}, fields=["draws"]); ctx.update_metric_diag(r["draws"]); ctx.run(15, da={}, fields=[])
    ctx.run(10, fields=[]); ms = ctx.last_run_kernel_ms(); lf = ctx.last_run_leapfrogs(); rd = ctx.last_run_rounds()
    fl = rd * 2 * 2.0 * 1024 * 256 * 100032
    print(json.dumps({"config": "config 5 share: logistic N=1e5 p=256, 1024 chains", "leapfrog_steps_per_s": lf / ms * 1e3, "kernel_ms": ms,
                      "leapfrogs": lf, "rounds": rd, "gemm_tflops_over_total_time": fl / ms / 1e9, "eps_median": float(np.median(ctx.stepsize()))}))
if "c5small" in which:
    rng = np.random.default_rng(0); N, D = 20000, 256
    X = rng.normal(size=(N, D)) / 16; y = (rng.random(N) < 1 / (1 + np.exp(-X @ rng.normal(size=D)))).astype(float)
    run("config 5 model, N=20000 p=256, 1024 chains", pkg.DeviceContext(D, 1024, target=ol.TARGET_LOGISTIC, target_params=ol.target_params_blob(ol.TARGET_LOGISTIC, D, X=X, y=y), seed=1), 30, 10)
