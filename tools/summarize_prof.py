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
