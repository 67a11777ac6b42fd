import json, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
pkg = load_package()
import torch
STREAM = torch.cuda.Stream()
for D, C in ((500, 8), (1000, 4), (1000, 64), (300, 200)):
    idx = np.arange(D)
    Sigma = 0.5 ** np.abs(idx[:, None] - idx[None, :])
    Pm = np.linalg.inv(Sigma)
    diag = np.diag(Pm).copy(); off = np.zeros(D); off[:D - 1] = np.diag(Pm, 1)
    row = {"D": D, "C": C}
    for graph in ("1", "0", "1", "0"):
        os.environ["DHMC_GRAPH"] = graph
        ctx = pkg.DeviceContext(D, C, target=pkg.abi.TARGET_TRIDIAG_NORMAL, target_params=np.concatenate([diag, off]), metric=pkg.abi.METRIC_DENSE, seed=3, stream=STREAM.cuda_stream)
        ctx.set_metric_dense(Sigma); ctx.init(); ctx.set_stepsize(0.3)
        ctx.run(5, fields=[])
        N = 100
        t0 = time.perf_counter(); ctx.run(N, fields=[]); dt = time.perf_counter() - t0
        row.setdefault("graph" + graph, []).append({"us_per_round": round(dt * 1e6 / max(1, ctx.last_run_rounds()), 1)})
        ctx.close()
    print(json.dumps(row), flush=True)
