"""Dense round engine on a workload with UNEQUAL trees (a correlated normal under a metric that is only roughly right, step
size per chain from a short adaptation): products over all rows every round against products over the running chains only
(DHMC_DENSE="row_lists=0" / "row_lists=1", read at context creation)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import torch
pkg = load_package()
import oracle_lib as ol
D, C, T = int(os.environ.get("RL_D", 1000)), int(os.environ.get("RL_C", 4096)), 10
rho = 0.9
sig = np.logspace(-1, 1, D)
Pc = np.zeros(D) + (1 + rho ** 2) / (1 - rho ** 2); Pc[0] = Pc[-1] = 1 / (1 - rho ** 2)
diag = Pc / sig ** 2
off = -rho / (1 - rho ** 2) / (sig[:-1] * sig[1:])
params = ol.target_params_blob(ol.TARGET_TRIDIAG_NORMAL, D, diag=diag, off=off)
for mode in ("0", "1"):
    os.environ["DHMC_DENSE"] = "row_lists=" + mode
    ctx = pkg.DeviceContext(D, C, metric=pkg.abi.METRIC_DENSE, target=pkg.abi.TARGET_TRIDIAG_NORMAL, target_params=params, seed=5)
    ctx.set_metric_dense(np.diag(sig ** 2))            # the right scales, none of the correlation
    ctx.init(np.random.default_rng(5).normal(size=(C, D)) * sig)
    ctx.find_initial_stepsize()
    ctx.run_into(30, {}, da={})
    out = {"steps": torch.empty((C, T), dtype=torch.int64, device="cuda"), "draws": torch.empty((C, T, D), dtype=torch.float64, device="cuda")}
    ctx.run_into(T, out)
    torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.run_into(T, out); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = out["steps"].cpu().numpy().sum(1)
    print(f"row lists {mode}: {ctx.last_run_leapfrogs() / dt:.4g} leapfrog-steps/s  rounds {ctx.last_run_rounds()}  per-chain leapfrogs mean {s.mean():.0f} max {s.max()}  draws checksum {float(out['draws'].sum()):.17g}")
