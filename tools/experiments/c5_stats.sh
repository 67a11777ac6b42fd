#!/bin/bash
# per-kernel time of config 5's round engine (rocprofv3 --kernel-trace --stats of tools/experiments/c5_rounds.py)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/c5stats; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/c5s
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/c5s -o t -- python $REPO/tools/experiments/c5_rounds.py ${1:-10} > $OUT/run.txt 2> $OUT/err.txt
f=$(find /tmp/c5s -name '*kernel_stats.csv' | head -1); cp $f $OUT/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-72s calls=%6s total_ms=%9.2f avg_us=%8.1f pct=%5s" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -4 $OUT/run.txt
