"""Logistic regression with a handful of chains: the GEMM round engine against the wave-per-chain functor (DHMC_LOGISTIC_ROUNDS=1 / 0),
to place the engine threshold.   python tools/experiments/small_logistic_engines.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package()
for N, D, C in ((1000, 16, 4), (10000, 32, 4), (10000, 32, 32), (100000, 64, 4), (100000, 256, 4), (2000, 256, 8)):
    rng = np.random.default_rng(0)
    X = rng.normal(size=(N, D)) / np.sqrt(D)
    y = (rng.random(N) < 1 / (1 + np.exp(-X @ rng.normal(size=D)))).astype(float)
    params = np.concatenate([np.array([N], np.int64).view(np.float64), X.ravel(), y])
    row = {"N": N, "D": D, "C": C}
    for rounds in ("1", "0"):
        os.environ["DHMC_LOGISTIC_ROUNDS"] = rounds
        ctx = pkg.DeviceContext(D, C, target=pkg.abi.TARGET_LOGISTIC, target_params=params, seed=2)
        ctx.init(np.zeros((C, D))); ctx.set_stepsize(0.5 / np.sqrt(N / 100))
        ctx.run(2, fields=[])
        T = 10
        t0 = time.perf_counter(); ctx.run(T, fields=[]); dt = time.perf_counter() - t0
        row["rounds" if rounds == "1" else "functor"] = {"us_per_leapfrog_chain": dt * 1e6 / (ctx.last_run_leapfrogs() / C), "lf_per_tr": ctx.last_run_leapfrogs() / C / T}
        ctx.close()
    print(json.dumps(row), flush=True)
