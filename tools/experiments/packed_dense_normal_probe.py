"""Round 6: the dense-precision normal (DHMC_TARGET_DENSE_NORMAL, diagonal metric) through the wave-per-chain kernel and the packed
kernel (a D x D matvec per gradient inside the chain's lane group): C chains, T transitions after a short adaptation.
usage: packed_dense_normal_probe.py D C T"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import oracle_lib as ol
pkg = load_package()
D, C, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(D)
A = rng.normal(size=(D, D)) * 0.3
Pm = A @ A.T + np.diag(rng.uniform(1.0, 2.0, size=D))
blob = ol.target_params_blob(ol.TARGET_DENSE_NORMAL, D, mu=rng.normal(size=D), P=Pm)
variants = [("wave", dict(DHMC_PACKED="0", DHMC_PIPELINE="0")), ("packed cpl4", dict(DHMC_PACKED="1", DHMC_PK="cpl=4"))]
if D <= 32:
    variants.append(("packed cpl2", dict(DHMC_PACKED="1", DHMC_PK="cpl=2")))
for name, env in variants:
    for k in ("DHMC_PACKED", "DHMC_PIPELINE", "DHMC_PK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = pkg.DeviceContext(D, C, target=ol.TARGET_DENSE_NORMAL, target_params=blob, seed=1)
    ctx.init(); ctx.find_initial_stepsize(); ctx.run(60, da={}, fields=[])
    ctx.run(T, fields=[])
    ms, lf = ctx.last_run_kernel_ms(), ctx.last_run_leapfrogs()
    print(f"D {D} chains {C} transitions {T} {name:12s} kernel_ms {ms:9.3f} leapfrogs {lf} -> {lf / ms * 1e3:.4g} /s")
    ctx.close()
