#!/bin/bash
# round 6: the numbers and profiles of the final build — GPU suite, the default bench line (other_configs incl. c4_32768, live traffic,
# CPU baseline), rocprofv3 trace + PMC passes of the headline kernel, kernel stats of configs 3 and 5, the fuzz sweep.  Every step
# under a hard timeout.
O=$PWD/gpurun_out/r6bfinal; mkdir -p $O
timeout -s KILL 1200 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -30 > $O/pytest.log; tail -4 $O/pytest.log
( time timeout -s KILL 400 python bench.py 2> $O/bench.err | tail -1 > $O/bench_default.json ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt
python - <<'PY'
import json
d = json.load(open('gpurun_out/r6bfinal/bench_default.json'))
print('headline %.4g' % d['value'], 'frac %.3f' % d['roofline']['frac'], d['roofline'].get('traffic_source'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
for k, v in d['other_configs'].items():
    n = v.get('at_config_n') or {}
    print(k, 'T20 %.4g' % v.get('value', float('nan')), '| at N=%s: %.4g' % (n.get('transitions_per_step'), n.get('value', float('nan'))), 'ms %.0f' % n.get('ms_per_step', float('nan')), 'sec %.1f' % v['seconds_with_setup'], v.get('error', ''))
PY
timeout -s KILL 200 python tools/fuzz_parity.py 60 20261002 2>/dev/null | tail -2 > $O/fuzz.txt; cat $O/fuzz.txt
timeout -s KILL 900 bash tools/profile.sh r06 > $O/profile.log 2>&1; tail -40 $O/profile.log
export TMPDIR=/tmp; REPO=$PWD; cd /tmp
for c in 3 5; do
  rm -rf /tmp/pk$c; timeout -s KILL 200 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/pk$c -o t -- python $REPO/bench.py --config $c --steps 3 --warmup 1 > $O/bench_c${c}_under_rocprof.json 2> /tmp/pk$c.err
  f=$(find /tmp/pk$c -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/c${c}_kernel_stats.csv
done
cd $REPO
for c in 3 5; do python - $c <<'PY'
import json,sys
c=sys.argv[1]
try:
    d=json.load(open('gpurun_out/r6bfinal/bench_c%s_under_rocprof.json'%c)); print('c%s under rocprof: %.4g frac %.3f'%(c,d['value'],d['roofline']['frac']))
except Exception as e: print('c%s: %s'%(c,e))
PY
done
