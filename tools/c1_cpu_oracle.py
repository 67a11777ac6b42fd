"""BASELINE.json configs[0] on the CPU oracle (the restatement of the reference; Julia is not available):
D = 100 standard MVN, 4 chains, default warmup (75/25/50/100/200/400/50) + 1000 draws, one chain per thread."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as ol
o = ol.Oracle(100, 4, seed=0x23EF614D, threads=4)
t0 = time.perf_counter()
o.init(); o.find_initial_stepsize()
wsteps = 0
for n, metric in ((75, False), (25, True), (50, True), (100, True), (200, True), (400, True), (50, False)):
    r = o.run(n, da={}, fields=["draws", "steps"]); wsteps += int(r["steps"].sum())
    if metric:
        o.update_metric_diag(r["draws"])
t1 = time.perf_counter()
r = o.run(1000, fields=["draws", "steps", "depth", "acceptance_rate"])
t2 = time.perf_counter()
print(json.dumps({"config": "1: D=100 standard MVN, 4 chains, CPU oracle (restatement of the reference), 4 threads",
                  "sampling_leapfrog_steps_per_s": int(r["steps"].sum()) / (t2 - t1), "warmup_leapfrog_steps_per_s": wsteps / (t1 - t0),
                  "sampling_s": t2 - t1, "warmup_s": t1 - t0, "mean_depth": float(r["depth"].mean()), "mean_acceptance": float(r["acceptance_rate"].mean()),
                  "draw_var": float(r["draws"].var()), "host_cores": os.cpu_count()}))
