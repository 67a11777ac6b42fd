#!/bin/bash
# round 4: (1) bisect the two funnel D = 1000 fuzz mismatches over builds without the inline-asm Horner chains / without the asm
# Philox products; (2) GRBM_GUI_ACTIVE (GPU clock cycles) beside kernel durations for the round-3 library and the working tree:
# the effective shader clock under each.
O=gpurun_out/r4f; mkdir -p $O
for v in noasm nophiloxasm; do
  echo "== fuzz with variant $v"; DHMC_LIB_PATH=$PWD/tools/experiments/_v/$v/libdhmc_amd.so FUZZ_VERBOSE=2 timeout -s KILL 120 python tools/fuzz_parity.py 20 12345 2> $O/fuzz_$v.err | tail -4
done
export TMPDIR=/tmp; REPO=$PWD
pmc() {  # name dir
  rm -rf /tmp/pm_$1; ( cd $2 && rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d /tmp/pm_$1 -o p -- python bench.py --steps 3 --warmup 1 --transitions 100 --no-cpu-baseline > /dev/null 2> /tmp/pm_$1.err )
  python - <<PY
import csv, glob
cnt = {}
for f in glob.glob('/tmp/pm_$1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'nuts_run_kernel' in r['Kernel_Name']:
            cnt.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
dur = {}
for f in glob.glob('/tmp/pm_$1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'nuts_run_kernel' in r['Kernel_Name']:
            dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
for d in sorted(cnt, key=int)[-3:]:
    c = cnt[d]; ms = dur.get(d, float('nan'))
    print('$1 dispatch', d, 'ms %.3f' % ms, 'GUI_ACTIVE/ms -> MHz %.0f' % (c.get('GRBM_GUI_ACTIVE', 0) / ms / 1e3), ' '.join('%s=%.4g' % kv for kv in sorted(c.items())))
PY
  tail -2 /tmp/pm_$1.err
}
pmc v1 tools/experiments/_ab/v1
pmc v2 .
