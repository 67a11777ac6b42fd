"""Where the wave-per-chain dense kernel stops beating the GEMM rounds as the chain count grows (DHMC_DENSE="rounds=0" / "rounds=1")."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package()
for D, C in ((64, 256), (64, 4096), (128, 256), (128, 1024), (128, 4096), (256, 128), (256, 512), (256, 4096)):
    idx = np.arange(D)
    Sigma = 0.5 ** np.abs(idx[:, None] - idx[None, :])
    Pm = np.linalg.inv(Sigma)
    diag = np.diag(Pm).copy(); off = np.zeros(D); off[:D - 1] = np.diag(Pm, 1)
    row = {"D": D, "C": C}
    for rounds in ("0", "1"):
        os.environ["DHMC_DENSE"] = "rounds=" + rounds
        ctx = pkg.DeviceContext(D, C, target=pkg.abi.TARGET_TRIDIAG_NORMAL, target_params=np.concatenate([diag, off]), metric=pkg.abi.METRIC_DENSE, seed=3)
        ctx.set_metric_dense(Sigma); ctx.init(); ctx.set_stepsize(0.3)
        N = 60
        ctx.run(N, fields=[])                                                     # (first use of an engine's kernels loads their code)
        t0 = time.perf_counter(); ctx.run(N, fields=[]); dt = time.perf_counter() - t0
        row["rounds" if rounds == "1" else "wave"] = round(ctx.last_run_leapfrogs() / dt / 1e6, 2)
        ctx.close()
    print(json.dumps(row), flush=True)
