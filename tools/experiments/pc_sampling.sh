#!/bin/bash
# PC-sampling profile of the headline kernel (rocprofv3 beta feature): where the wave cycles go, per instruction.
# usage (GPU box): bash tools/experiments/pc_sampling.sh [stochastic|host_trap]
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pcs
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
METHOD=${1:-stochastic}
if [ "$METHOD" = stochastic ]; then UNIT=cycles; INT=${2:-4194304}; else UNIT=time; INT=${2:-2000}; fi
rocprofv3 -L > $OUT/avail.txt 2>&1
timeout 400 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD \
    --pc-sampling-interval $INT --kernel-trace -d /tmp/pcs_raw --output-format csv -- \
    python $REPO/bench.py --steps 1 --warmup 0 --transitions 200 --no-cpu-baseline > $OUT/run_$METHOD.log 2>&1
echo "rc=$?" >> $OUT/run_$METHOD.log
find /tmp/pcs_raw -type f | xargs ls -la > $OUT/files_$METHOD.txt 2>&1
python3 - "$METHOD" <<'PY'
import sys, glob, csv, collections, os, gzip, shutil
method = sys.argv[1]
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "pcs")
for f in glob.glob("/tmp/pcs_raw/**/*", recursive=True):
    if not os.path.isfile(f): continue
    base = os.path.basename(f)
    if "pc_sampling" in base and base.endswith(".csv"):
        with open(f) as fh:
            head = [next(fh, "") for _ in range(40)]
        open(os.path.join(out, "head_%s_%s.txt" % (method, base)), "w").writelines(head)
        rd = csv.DictReader(open(f))
        cols = rd.fieldnames
        hist = collections.Counter()
        n = 0
        for row in rd:
            n += 1
            key = tuple(row.get(c, "") for c in cols if any(t in c.lower() for t in
                        ("code_object_id", "offset", "stall", "inst_type", "issued", "dual", "arb", "no_inst", "reason", "instruction", "comment")))
            hist[key] += 1
        keycols = [c for c in cols if any(t in c.lower() for t in
                        ("code_object_id", "offset", "stall", "inst_type", "issued", "dual", "arb", "no_inst", "reason", "instruction", "comment"))]
        with open(os.path.join(out, "hist_%s_%s" % (method, base)), "w") as o:
            o.write(",".join(keycols + ["samples"]) + "\n")
            for k, v in hist.most_common():
                o.write(",".join('"%s"' % x if "," in x else x for x in k) + ",%d\n" % v)
        print(base, "rows", n, "distinct", len(hist))
    elif base.endswith(".csv") and os.path.getsize(f) < 2_000_000:
        shutil.copy(f, os.path.join(out, method + "_" + base))
PY
ls -la $OUT
