"""PCIe-inclusive rate of the drop-in path: dhmc_run with HOST result buffers (what the Julia shim's `run!` and
DeviceContext.run() hand over), BASELINE config 2 (D = 1000, 4096 chains, adapted ϵ / metric).  Never bench.py's `value`
(that keeps outputs in HBM); DESIGN.md §6 quotes these numbers.

    python tools/pcie_rate.py [transitions per call] [dense]      (default 100; "dense": BASELINE config 3's dense metric through the
                                                                   GEMM round engine, whose long host-output calls dhmc_run splits)

Three destinations: pageable numpy arrays (the runtime stages the copies), page-locked arrays (pinned_empty /
dhmc_host_alloc: asynchronous 2-D copies under the next chunk's kernel), and device buffers for comparison.  Draws are
8 000 B per transition and chain = 533 B per leapfrog at 15 leapfrogs per transition, so a PCIe 5 x16 link (≈ 55 GB/s
achievable of 64) caps the host-output rate at ≈ 1.0e8 leapfrog-steps/s whatever the kernel does."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
from dynamichmc_jl_amd.context import pinned_empty   # noqa: E402
import torch

D, C = 1000, 4096
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
DENSE = len(sys.argv) > 2 and sys.argv[2] == "dense"
if DENSE:
    import oracle_lib_free_config3 as c3                       # (tools/: the config-3 target without the test tree)
    ctx = c3.context(pkg, D, C)
else:
    ctx = pkg.DeviceContext(D, C, seed=1); ctx.init(); ctx.find_initial_stepsize()
    r = ctx.run(60, da={}, fields=["draws"]); ctx.update_metric_diag(r["draws"][:, 30:]); ctx.run(40, da={}, fields=[])
fields = (("draws", np.float64), ("steps", np.int64), ("depth", np.int32), ("acceptance_rate", np.float64), ("logdensities", np.float64))
shape = lambda k: (C, T, D) if k == "draws" else (C, T)


def timed(arrs, reps=3):
    ctx.run_into(T, arrs)
    t0 = time.perf_counter(); lf = 0; kms = 0.0
    for _ in range(reps):
        ctx.run_into(T, arrs); lf += ctx.last_run_leapfrogs(); kms += ctx.last_run_kernel_ms()
    dt = time.perf_counter() - t0
    nbytes = sum(a.numel() * a.element_size() if hasattr(a, "numel") else a.nbytes for a in arrs.values())
    return {"steps_per_s": lf / dt, "ms_per_call": dt / reps * 1e3, "kernel_ms_per_call": kms / reps, "GB_per_s": nbytes * reps / dt / 1e9}


out = {"transitions_per_call": T, "chains": C, "dim": D, "metric": "dense" if DENSE else "diag", "host_chunk_env": os.environ.get("DHMC_HOST_CHUNK")}
out["pageable"] = timed({k: np.zeros(shape(k), dt) for k, dt in fields})
out["page_locked"] = timed({k: pinned_empty(shape(k), dt) for k, dt in fields})
tdt = {np.float64: torch.float64, np.int64: torch.int64, np.int32: torch.int32}
out["device"] = timed({k: torch.empty(shape(k), dtype=tdt[dt], device="cuda") for k, dt in fields})
print(json.dumps(out))
