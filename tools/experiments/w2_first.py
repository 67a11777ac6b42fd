"""First GPU look at the two-waves-per-chain kernel (nuts_run_kernel NW = 2, DHMC_W2=1): bits against the one-wave
kernel, then kernel time against the number of chains at a fixed step size (D = 1000).  One library per process:
DHMC_LIB_PATH=tools/experiments/_v/<name>/libdhmc_amd.so python tools/experiments/w2_first.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
pkg = load_package()
name = os.path.basename(os.path.dirname(os.environ.get("DHMC_LIB_PATH", "default/x")))


def steps(ctx):
    out = {}
    ctx.init(); ctx.find_initial_stepsize()
    a = ctx.run(20, da={})
    ctx.update_metric_diag(a["draws"])
    out.update({"w_" + k: v for k, v in a.items()})
    out.update({"i_" + k: v for k, v in ctx.run(10).items()})
    ctx.set_stepsize(4.0)
    out.update({"d_" + k: v for k, v in ctx.run(6).items()})
    ctx.set_stepsize(1e-3)
    out.update({"m_" + k: v for k, v in ctx.run(3).items()})
    q, lq, g = ctx.position()
    out.update(q=q, lq=lq, g=g)
    return out


res = {}
for w2 in ("1", "0"):
    os.environ["DHMC_W2"] = w2
    ctx = pkg.DeviceContext(1000, 7, seed=21, max_depth=6)
    res[w2] = steps(ctx)
    ctx.close()
bad = [k for k in res["1"] if not np.array_equal(res["1"][k], res["0"][k], equal_nan=True)]
print(json.dumps({"variant": name, "bits_equal": not bad, "mismatch": bad[:5]}), flush=True)

for w2 in ("1", "0"):
    os.environ["DHMC_W2"] = w2
    for C in (256, 1024, 2048, 4096):
        ctx = pkg.DeviceContext(1000, C, seed=1)
        ctx.init(); ctx.set_stepsize(0.3)
        ctx.run(10, fields=[])
        best = None
        for rep in range(2):
            ctx.run(40, fields=[]); ms = ctx.last_run_kernel_ms(); lf = ctx.last_run_leapfrogs()
            best = ms if best is None else min(best, ms)
        print(json.dumps({"variant": name, "w2": w2, "chains": C, "kernel_ms": round(best, 3), "leapfrogs": lf,
                          "steps_per_s": "%.4g" % (lf / best * 1e3), "us_per_leapfrog_per_chain": round(best * 1e3 / (lf / C), 3)}), flush=True)
        ctx.close()
