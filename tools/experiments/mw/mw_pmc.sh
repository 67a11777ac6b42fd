#!/bin/bash
# PMC passes over the D = 1000 per-draw kernel at a fixed step size (tools/experiments/mw_scaling.py with one chain count).
# usage: tools/experiments/mw_pmc.sh <tag> [chains]
TAG=${1:-mw}; CH=${2:-4096}
OUT=/tmp/pmc_$TAG; KEEP=$PWD/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT $KEEP
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/tools/experiments/mw_scaling.py $CH > $OUT/trace.out 2> $OUT/trace.err
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/pmc1 -o pmc1 -- python $REPO/tools/experiments/mw_scaling.py $CH > /dev/null 2> $OUT/pmc1.err
rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o pmc2 -- python $REPO/tools/experiments/mw_scaling.py $CH > /dev/null 2> $OUT/pmc2.err
rocprofv3 --output-format csv --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC -d $OUT/pmc3 -o pmc3 -- python $REPO/tools/experiments/mw_scaling.py $CH > /dev/null 2> $OUT/pmc3.err
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc4 -o pmc4 -- python $REPO/tools/experiments/mw_scaling.py $CH > /dev/null 2> $OUT/pmc4.err
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc5 -o pmc5 -- python $REPO/tools/experiments/mw_scaling.py $CH > /dev/null 2> $OUT/pmc5.err
cd $REPO
cp $OUT/trace.out $KEEP/
python - "$OUT" > $KEEP/summary.txt 2>&1 <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True)):
    print("== kernel stats")
    for r in list(csv.DictReader(open(f)))[:6]:
        print("  ", r.get("Name", "")[:80], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage"))
for f in sorted(glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "nuts_run" in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    print("== per-dispatch ms:", " ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.3f}" for r in rows))
    for r in rows[:2]:
        print("   ", {k: r[k] for k in r if k in ("Kernel_Name", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size", "Grid_Size")})
for f in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)):
    agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "nuts_run" not in k: continue
        key = ("mw" if "_mw_" in k else "onewave")
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key][r["Counter_Name"]] += 1
    print("== counters (sum over dispatches of the kernel; dispatch count)", os.path.basename(os.path.dirname(f)))
    for key in agg:
        for c in sorted(agg[key]):
            print(f"   {key:8s} {c:28s} {agg[key][c]:.6g}  ({cnt[key][c]} dispatches)")
PY
cat $KEEP/summary.txt; tail -2 $OUT/*.err | head -40
