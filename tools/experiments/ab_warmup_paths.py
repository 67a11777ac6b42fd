"""Same seed, the reference's default warmup two ways — metric windows accumulated by the kernel (dhmc_metric_window_begin) vs the
windows' draws stored and re-read (dhmc_update_metric_diag) — then 200 sampling transitions: adapted state and kernel time."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
D, C = 1000, 4096
stages = [(75, False), (25, True), (50, True), (100, True), (200, True), (400, True), (50, False)]
for path in ("window", "draws", "window", "draws"):
    ctx = pkg.DeviceContext(D, C, seed=1234, stream=torch.cuda.current_stream().cuda_stream)
    ctx.init(); ctx.find_initial_stepsize()
    for n, metric in stages:
        if metric and path == "window":
            ctx.metric_window_begin(); ctx.run_into(n, {}, da={}); ctx.update_metric_diag_window()
        elif metric:
            d = torch.empty((C, n, D), dtype=torch.float64, device="cuda")
            ctx.run_into(n, {"draws": d}, da={}); ctx.update_metric_diag(d); del d
        else:
            ctx.run_into(n, {}, da={})
    eps = ctx.stepsize(); minv = ctx.metric_diag()
    T = 200
    out = {"steps": torch.empty((C, T), dtype=torch.int64, device="cuda"), "depth": torch.empty((C, T), dtype=torch.int32, device="cuda"),
           "acceptance_rate": torch.empty((C, T), dtype=torch.float64, device="cuda"), "draws": torch.empty((C, T, D), dtype=torch.float64, device="cuda")}
    ms = []
    for rep in range(3):
        ctx.run_into(T, out); ms.append(ctx.last_run_kernel_ms())
    lf = int(out["steps"].sum())
    print(json.dumps({"path": path, "eps_mean": float(eps.mean()), "eps_sd": float(eps.std()), "eps_min": float(eps.min()), "eps_max": float(eps.max()),
                      "minv_mean": float(minv.mean()), "minv_sd": float(minv.std()), "minv_min": float(minv.min()), "minv_max": float(minv.max()),
                      "kernel_ms": [round(m, 2) for m in ms], "ns_per_leapfrog_per_wave_slot": round(1e6 * ms[-1] / lf * 1024, 1),
                      "depth_hist": torch.bincount(out["depth"].flatten().long(), minlength=7).tolist(), "acc": float(out["acceptance_rate"].mean())}))
    ctx.close(); del out
