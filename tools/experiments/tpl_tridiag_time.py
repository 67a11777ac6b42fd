"""Kernel time of the one-kernel engine for a target whose gradient is not coordinate-wise (tridiagonal-precision normal,
D = 1000, diagonal metric, 4096 chains), default layout against the trajectory-ends-in-LDS layout (library built with
-DDHMC_FORCE_TRAJ_LDS).  DHMC_LIB_PATH selects the library."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from __graft_entry__ import load_package
pkg = load_package()
D, C = 1000, 4096
params = ol.target_params_blob(ol.TARGET_TRIDIAG_NORMAL, D, diag=np.full(D, 2.5), off=np.full(D - 1, -0.6))
for tgt, kw, name in ((ol.TARGET_TRIDIAG_NORMAL, dict(target_params=params), "tridiag"), (ol.TARGET_STD_NORMAL, {}, "std")):
    ctx = pkg.DeviceContext(D, C, target=tgt, seed=1, **kw)
    ctx.init(); ctx.set_stepsize(0.25)
    ctx.run(10, fields=[])
    best = 1e9
    for _ in range(2):
        ctx.run(40, fields=[]); best = min(best, ctx.last_run_kernel_ms()); lf = ctx.last_run_leapfrogs()
    print(json.dumps({"lib": os.path.basename(os.path.dirname(os.environ.get("DHMC_LIB_PATH", "default/x"))), "target": name,
                      "kernel_ms": round(best, 3), "leapfrogs": lf, "steps_per_s": "%.4g" % (lf / best * 1e3)}), flush=True)
    ctx.close()
