"""bench.py's headline setup (default seed, --warmup-draws: the state with one chain at 1.48 x the mean work), then 1000-transition
calls: which chain is the slowest in each call, how much work it did, the kernel time — with the launch order on (default) or off."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from __graft_entry__ import load_package
pkg = load_package()
ctx, warm = bench.setup_context(pkg, torch, 0, 4096, 1234 if len(sys.argv) < 2 else int(sys.argv[1]), False, True)
C, T = 4096, 1000
out = {"steps": torch.empty((C, T), dtype=torch.int64, device="cuda"), "depth": torch.empty((C, T), dtype=torch.int32, device="cuda")}
eps = ctx.stepsize()
for rep in range(5):
    ctx.run_into(T, out)
    w = out["steps"].sum(1)
    top = torch.topk(w, 3)
    k = int(top.indices[0])
    print(json.dumps({"order": os.environ.get("DHMC_LAUNCH_ORDER", "1"), "kernel_ms": round(ctx.last_run_kernel_ms(), 2), "top_chains": top.indices.tolist(), "top_work": top.values.tolist(),
                      "mean_work": float(w.double().mean()), "eps_top": float(eps[k]), "eps_median": float(np.median(eps)),
                      "depth_hist_top": torch.bincount(out["depth"][k].long(), minlength=8).tolist()}))
