#!/bin/bash
# kernel trace of the dense-metric round engine (config 3): do the two half-batches' kernels overlap?
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/c3trace; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/c3t
timeout 500 rocprofv3 --output-format csv --kernel-trace -d /tmp/c3t -o t -- python $REPO/bench.py --config 3 --steps 1 --warmup 0 --transitions 5 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.txt
f=$(find /tmp/c3t -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$OUT" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last 4000 dispatches (the timed rounds)
rows = rows[-4000:]
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2] + "/timeline.txt", "w") as o:
    for r in rows[-1300:-700]:
        o.write("%10.1f %10.1f q=%s %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                        r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
span = int(rows[-1]["End_Timestamp"]) - t0
busy = collections.Counter(); 
ev = []
for r in rows:
    g = "gemm" if "gemm" in r["Kernel_Name"] else "other"
    busy[g] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    ev.append((int(r["Start_Timestamp"]), 1, g)); ev.append((int(r["End_Timestamp"]), -1, g))
ev.sort()
act = collections.Counter(); last = ev[0][0]; both = anyk = 0
for t, d, g in ev:
    dt = t - last
    if act["gemm"] > 0 and act["other"] > 0: both += dt
    if act["gemm"] > 0 or act["other"] > 0: anyk += dt
    act[g] += d; last = t
print("span_us %.0f  gemm_busy_us %.0f  other_busy_us %.0f  any_kernel_us %.0f  gemm_and_other_concurrent_us %.0f" % (span / 1e3, busy["gemm"] / 1e3, busy["other"] / 1e3, anyk / 1e3, both / 1e3))
names = collections.Counter(); dur = collections.Counter()
for r in rows:
    n = r["Kernel_Name"][:50]; names[n] += 1; dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, c in names.most_common(12): print("%6d x %8.1f us avg  %s" % (c, dur[n] / c / 1e3, n))
PY
tail -2 $OUT/err.txt
