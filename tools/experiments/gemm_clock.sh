#!/bin/bash
# effective shader clock while the fp64-MFMA GEMMs run: GRBM_GUI_ACTIVE cycles / kernel duration (two rocprofv3 passes)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/gc1 /tmp/gc2
rocprofv3 --output-format csv --kernel-trace -d /tmp/gc1 -o t -- $REPO/tools/experiments/bin/gemm_tile_sweep > /dev/null 2>&1
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/gc2 -o p -- $REPO/tools/experiments/bin/gemm_tile_sweep > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
tr = list(csv.DictReader(open(glob.glob('/tmp/gc1/**/*kernel_trace.csv', recursive=True)[0])))
pm = list(csv.DictReader(open(glob.glob('/tmp/gc2/**/*counter_collection.csv', recursive=True)[0])))
dur = collections.defaultdict(list)
for r in tr: dur[r['Kernel_Name'][:60]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
cyc = collections.defaultdict(list)
for r in pm:
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': cyc[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k in dur:
    if k in cyc and len(cyc[k]) == len(dur[k]):
        # per dispatch clock, show min/median/max over dispatches longer than 50 us
        f = [c / d for c, d in zip(cyc[k], dur[k]) if d > 50000]
        if f: print('%-60s dispatches %3d  GHz min %.2f  median %.2f  max %.2f' % (k, len(f), min(f), sorted(f)[len(f)//2], max(f)))
PY
