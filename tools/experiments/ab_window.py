"""A/B of the headline kernel with and without the metric window's code compiled in (csrc/nuts_kernels.hpp window_accumulate;
-DDHMC_NO_WINDOW), no window open in either: what the feature costs the sampling phase.  Same setup in both (two-pass metric
update from stored draws).   DHMC_LIB_PATH=tools/experiments/_v/nowin/libdhmc_amd.so python tools/experiments/ab_window.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
D, C, T = 1000, 8192, 20
ctx = pkg.DeviceContext(D, C, seed=1234, stream=torch.cuda.current_stream().cuda_stream)
ctx.init(); ctx.find_initial_stepsize()
ctx.run_into(40, {}, da={})
d = torch.empty((C, 25, D), dtype=torch.float64, device="cuda")
ctx.run_into(25, {"draws": d}, da={}); ctx.update_metric_diag(d); del d
ctx.run_into(40, {}, da={})
out = {"draws": torch.empty((C, T, D), dtype=torch.float64, device="cuda"), "steps": torch.empty((C, T), dtype=torch.int64, device="cuda")}
res = []
for rep in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.run_into(T, out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res.append(int(out["steps"].sum()) / dt)
print(json.dumps({"lib": os.environ.get("DHMC_LIB_PATH", "default"), "leapfrogs_per_s": [round(r / 1e6, 2) for r in res], "best_e8": max(res) / 1e8}))
