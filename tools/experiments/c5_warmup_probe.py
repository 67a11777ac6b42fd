"""Config 5 (logistic regression N = 1e5, p = 256, 1024 chains): how much of the round engine's time is the wait for chains whose
adapted step size leaves them with deeper trees — the bench's short warmup against longer ones.  usage: c5_warmup_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
pkg = load_package()
N, D, C = 100000, 256, 1024
rng = np.random.default_rng(0)
X = rng.normal(size=(N, D)) / 16
y = (rng.random(N) < 1 / (1 + np.exp(-X @ rng.normal(size=D)))).astype(float)
params = pkg.LogisticRegression(X, y).params()
for name, stages in (("bench r4: 20w + 15", [(20, True), (15, False)]),
                     ("40 + 40w + 80w + 40", [(40, False), (40, True), (80, True), (40, False)]),
                     ("75 + 25w + 50w + 100w + 50", [(75, False), (25, True), (50, True), (100, True), (50, False)])):
    ctx = pkg.DeviceContext(D, C, target=pkg.abi.TARGET_LOGISTIC, target_params=params, seed=0x23EF614D)
    ctx.init(); ctx.set_stepsize(0.02)
    t0 = time.perf_counter()
    for n, metric in stages:
        if metric:
            ctx.metric_window_begin()
        ctx.run_into(n, {}, da={})
        if metric:
            ctx.update_metric_diag_window()
    torch.cuda.synchronize(); tw = time.perf_counter() - t0
    T = 60
    out = {"steps": torch.empty((C, T), dtype=torch.int64, device="cuda"), "depth": torch.empty((C, T), dtype=torch.int32, device="cuda")}
    t0 = time.perf_counter(); ctx.run_into(T, out); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    w = out["steps"].sum(1).double(); eps = ctx.stepsize()
    print(f"{name:30s} warmup {tw:6.1f} s | T={T}: {ctx.last_run_leapfrogs() / dt:.4g} leapfrogs/s, leapfrogs per transition {float(out['steps'].double().mean()):.1f}, "
          f"slowest/mean chain {float(w.max() / w.mean()):.2f}, depth histogram {torch.bincount(out['depth'].flatten().long()).tolist()}, eps min/median/max "
          f"{eps.min():.4g}/{np.median(eps):.4g}/{eps.max():.4g}", flush=True)
    ctx.close()
