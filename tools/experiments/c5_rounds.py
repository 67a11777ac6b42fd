"""Config 5's share on one GPU: how many lockstep rounds a call takes against the mean work per chain (stragglers)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
import torch
pkg = load_package()
N, D, C = 100000, 256, int(os.environ.get("C5_C", 1024))
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(0)
X = rng.normal(size=(N, D)) / 16
y = (rng.random(N) < 1 / (1 + np.exp(-X @ rng.normal(size=D)))).astype(float)
ctx = pkg.DeviceContext(D, C, target=pkg.abi.TARGET_LOGISTIC, target_params=pkg.LogisticRegression(X, y).params(), seed=1234)
ctx.init(); ctx.set_stepsize(0.02)
d = torch.empty((C, 20, D), dtype=torch.float64, device="cuda")
ctx.run_into(20, {"draws": d}, da={}); ctx.update_metric_diag(d); ctx.run_into(15, {}, da={})
out = {"steps": torch.empty((C, T), dtype=torch.int64, device="cuda")}
ctx.run_into(T, out)
t0 = time.perf_counter(); ctx.run_into(T, out); torch.cuda.synchronize(); dt = time.perf_counter() - t0
s = out["steps"].cpu().numpy().sum(1)
lf = ctx.last_run_leapfrogs(); ms = ctx.last_run_kernel_ms()
print(f"T={T} leapfrogs {lf} wall {dt:.3f}s kernel_ms {ms:.1f} -> {lf/dt:.4g}/s")
print(f"per-chain leapfrogs: mean {s.mean():.1f} median {np.median(s):.0f} p90 {np.percentile(s,90):.0f} p99 {np.percentile(s,99):.0f} max {s.max()}")
print(f"rounds >= max = {s.max()}: ms per round {ms/s.max():.3f}; mean/max {s.mean()/s.max():.3f}")
act = np.array([(s > r).sum() for r in range(0, int(s.max()), max(1, int(s.max()) // 20))])
print("active chains along the call:", act.tolist())
eps = ctx.get_stepsize() if hasattr(ctx, "get_stepsize") else None
if eps is not None: print("eps quantiles", np.percentile(eps, [0, 10, 50, 90, 100]).round(4).tolist())
