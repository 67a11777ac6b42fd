"""Chains-in-flight experiment for the per-draw kernels at D = 1000: kernel time of 40 transitions at a fixed
step size against the number of chains, multi-wave kernel (DHMC_MW=1) and one-wave kernel (DHMC_MW=0).
A plateau of width k·256 chains says k workgroups per CU are resident."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
pkg = load_package()
D = int(os.environ.get("MWS_D", "1000"))
chains = [int(x) for x in (sys.argv[1:] or "64 256 512 768 1024 1280 1536 2048 3072 4096".split())]
for mw in ("1", "0"):
    os.environ["DHMC_MW"] = mw
    for C in chains:
        ctx = pkg.DeviceContext(D, C, seed=1)
        ctx.init(); ctx.set_stepsize(0.3)
        ctx.run(10, fields=[])
        ctx.run(40, fields=[]); ms = ctx.last_run_kernel_ms(); lf = ctx.last_run_leapfrogs()
        print(json.dumps({"mw": mw, "chains": C, "kernel_ms": round(ms, 3), "leapfrogs": lf, "steps_per_s": lf / ms * 1e3,
                          "us_per_leapfrog_per_chain": ms * 1e3 / (lf / C)}), flush=True)
        ctx.close()
