"""Few chains with a dense metric (the reference's typical use: 1–8 chains): the GEMM round engine (default: one M⁻¹ product per
leapfrog) against the wave-per-chain dense kernel (DHMC_DENSE="products=2").   python tools/small_dense_bench.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()

out = []
for D, C in ((10, 4), (100, 4), (100, 32), (500, 8), (1000, 4), (1000, 64)):
    rho = 0.5
    idx = np.arange(D)
    Sigma = rho ** np.abs(idx[:, None] - idx[None, :])
    Pm = np.linalg.inv(Sigma)
    diag = np.diag(Pm).copy(); off = np.zeros(D); off[:D - 1] = np.diag(Pm, 1)
    row = {"D": D, "C": C}
    for products in (1, 2):
        ctx = pkg.DeviceContext(D, C, target=pkg.abi.TARGET_TRIDIAG_NORMAL, target_params=np.concatenate([diag, off]), metric=pkg.abi.METRIC_DENSE, seed=3)
        ctx.set_dense_products(products)
        ctx.set_metric_dense(Sigma)
        ctx.init(); ctx.find_initial_stepsize()
        ctx.run(50, da={}, fields=[])
        N = 200
        t0 = time.perf_counter()
        ctx.run(N, fields=[])
        dt = time.perf_counter() - t0
        row[f"products{products}"] = {"leapfrogs_per_s": ctx.last_run_leapfrogs() / dt, "ms_per_transition": dt / N * 1e3, "rounds": ctx.last_run_rounds()}
        ctx.close()
    out.append(row)
    print(json.dumps(row), flush=True)
