import json, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
pkg = load_package()
for D, C in ((64, 4), (100, 4), (128, 4), (200, 4), (256, 4), (256, 64)):
    rho = 0.5
    idx = np.arange(D)
    Sigma = rho ** np.abs(idx[:, None] - idx[None, :])
    Pm = np.linalg.inv(Sigma)
    diag = np.diag(Pm).copy(); off = np.zeros(D); off[:D - 1] = np.diag(Pm, 1)
    for tgt in ("tridiag", "std"):
        row = {"D": D, "C": C, "target": tgt}
        for products in (1, 2):
            if tgt == "tridiag":
                ctx = pkg.DeviceContext(D, C, target=pkg.abi.TARGET_TRIDIAG_NORMAL, target_params=np.concatenate([diag, off]), metric=pkg.abi.METRIC_DENSE, seed=3)
                ctx.set_metric_dense(Sigma)
            else:
                ctx = pkg.DeviceContext(D, C, metric=pkg.abi.METRIC_DENSE, seed=3)
            ctx.set_dense_products(products)
            ctx.init(); ctx.set_stepsize(0.3 if tgt == "tridiag" else 0.5)
            ctx.run(5, fields=[])
            N = 100
            ctx.run(N, fields=[])
            row[f"p{products}"] = {"us_per_leapfrog_chain": ctx.last_run_kernel_ms() * 1e3 / (ctx.last_run_leapfrogs() / C), "lf_per_tr": ctx.last_run_leapfrogs() / C / N}
            ctx.close()
        print(json.dumps(row), flush=True)
