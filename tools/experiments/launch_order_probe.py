"""Does the hardware start workgroups in blockIdx order, and does RunParams::launch_order move a slow chain to the front?
4096 chains of the 1000-dim standard normal, 64 transitions per call, one chain with ϵ/8 (depth ≈ 7 instead of 4)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
D, C, T = 1000, 4096, 64
for slow in (0, 1000, 2500, 4095, None):
    ctx = pkg.DeviceContext(D, C, seed=5, stream=torch.cuda.current_stream().cuda_stream)
    ctx.init(); ctx.find_initial_stepsize()
    ctx.run_into(60, {}, da={})
    eps = np.full(C, float(np.median(ctx.stepsize())))
    if slow is not None:
        eps[slow] /= 8
    ctx.set_stepsize(eps)
    out = {"steps": torch.empty((C, T), dtype=torch.int64, device="cuda")}
    ms = []
    for rep in range(4):
        ctx.run_into(T, out); ms.append(round(ctx.last_run_kernel_ms(), 2))
    w = out["steps"].sum(1).double()
    print(json.dumps({"order": os.environ.get("DHMC_LAUNCH_ORDER", "1"), "slow_chain": slow, "kernel_ms": ms, "slowest_over_mean": round(float(w.max() / w.mean()), 2),
                      "slow_chain_leapfrogs": int(w.max()), "mean": float(w.mean())}))
    ctx.close()
