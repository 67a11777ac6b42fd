"""Summarise the rocprofv3 output of tools/profile.sh: per-kernel stats and PMC counters of
the dominant kernel (averages per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))


for f in find("*kernel_stats.csv"):
    print("== kernel stats", os.path.relpath(f, out))
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:8]:
        name = r.get("Name", "")[:70]
        print(f"  {name:70s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} avg_ns={r.get('AverageNs')} pct={r.get('Percentage')}")

for f in find("*kernel_trace.csv"):
    print("== per-dispatch durations of nuts_run_kernel (ms), in launch order:", os.path.relpath(f, out))
    with open(f) as fh:
        rows = [r for r in csv.DictReader(fh) if "nuts_run_kernel" in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    print("  ", " ".join(f"{x:.3f}" for x in d))
    print("   (bench.py --steps 3 --warmup 1 --transitions T: 7 adaptive setup launches (75/25/50/100/200/400/50 transitions),"
          " then 1 warmup + 3 timed launches of T transitions — compare the last three with roofline.kernel_ms in"
          " bench_trace.json)")

for f in find("*counter_collection.csv"):
    print("== counters", os.path.relpath(f, out))
    agg = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r.get("Kernel_Name", "")
            agg[k][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
    for k, cs in agg.items():
        if "nuts_run_kernel" not in k:
            continue
        print("  kernel:", k[:80])
        for cn, vals in sorted(cs.items()):
            print(f"    {cn:24s} n={len(vals):3d} mean={sum(vals)/len(vals):.6g} last={vals[-1]:.6g}")

# HBM bytes per leapfrog over ALL nuts_run_kernel dispatches of the run (needs the bench JSON of the trace pass)
import json
tot = {}
for f in find("*counter_collection.csv"):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "nuts_run_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") in ("FETCH_SIZE", "WRITE_SIZE"):
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
try:
    with open(os.path.join(out, "bench_trace.json")) as fh:
        bench = json.loads(fh.read().strip().splitlines()[-1])
    n = bench["all_run_leapfrogs"]
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        fetch_b = tot["FETCH_SIZE"] * 1024 * 2      # KB; x2: gfx950 FETCH_SIZE reports half of wide streaming reads
        write_b = tot["WRITE_SIZE"] * 1024
        res = {"hbm_bytes_per_leapfrog": (fetch_b + write_b) / n, "fetch_bytes_per_leapfrog_x2": fetch_b / n,
               "write_bytes_per_leapfrog": write_b / n, "leapfrogs": n,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py, summed over all "
                         "nuts_run_kernel dispatches, FETCH_SIZE doubled per MI355X_MICROARCH.md"}
        print("== traffic", json.dumps(res))
        with open(os.path.join(out, "traffic.json"), "w") as fh:
            json.dump(res, fh, indent=1)
except Exception as e:  # noqa
    print("traffic summary unavailable:", e)
