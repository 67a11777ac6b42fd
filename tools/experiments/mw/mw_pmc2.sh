#!/bin/bash
# instruction-cache and issue counters of the per-draw kernels at a given chain count
CH=${1:-768}
OUT=/tmp/pmc2; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVE_CYCLES -d $OUT/a -o a -- python $REPO/tools/experiments/mw_scaling.py $CH > /dev/null 2> $OUT/a.err
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT/b -o b -- python $REPO/tools/experiments/mw_scaling.py $CH > /dev/null 2> $OUT/b.err
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAIT_ANY SQ_LEVEL_WAVES SQC_DCACHE_MISSES -d $OUT/c -o c -- python $REPO/tools/experiments/mw_scaling.py $CH > /dev/null 2> $OUT/c.err
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)):
    agg = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "nuts_run" not in k: continue
        agg["mw" if "_mw_" in k else "onewave"][r["Counter_Name"]] += float(r["Counter_Value"])
    for key in agg:
        print(key, {c: f"{v:.4g}" for c, v in sorted(agg[key].items())})
PY
