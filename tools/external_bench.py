"""Throughput of DHMC_TARGET_EXTERNAL (a torch log density behind the round engine) next to the built-in functor."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
for D, C in ((100, 4096), (1000, 4096)):
    for name, target, fn in (("torch callback", pkg.abi.TARGET_EXTERNAL, lambda q: (-0.5 * (q * q).sum(1), -q)), ("built-in functor", pkg.abi.TARGET_STD_NORMAL, None)):
        ctx = pkg.DeviceContext(D, C, target=target, seed=1, stream=torch.cuda.current_stream().cuda_stream)
        if fn:
            ctx.set_logdensity_callback(fn)
        ctx.init(); ctx.find_initial_stepsize()
        d = torch.empty((C, 40, D), dtype=torch.float64, device="cuda")
        ctx.run_into(40, {"draws": d}, da={}); ctx.update_metric_diag(d); ctx.run_into(30, {}, da={})
        ctx.run_into(20, {})
        print(json.dumps({"D": D, "chains": C, "density": name, "leapfrog_steps_per_s": ctx.last_run_leapfrogs() / ctx.last_run_kernel_ms() * 1e3,
                          "rounds": ctx.last_run_rounds(), "ms": ctx.last_run_kernel_ms()}))
