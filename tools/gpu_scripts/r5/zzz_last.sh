#!/bin/bash
# round 5: the GPU suite and the default bench line once more on the last build of the round
O=$PWD/gpurun_out/r5last; mkdir -p $O
timeout -s KILL 700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -30 > $O/pytest.log; tail -4 $O/pytest.log
( time timeout -s KILL 400 python bench.py 2> $O/bench.err | tail -1 > $O/bench_default.json ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt
python - <<'PY'
import json
d = json.load(open('gpurun_out/r5last/bench_default.json'))
print('headline %.4g' % d['value'], 'frac %.3f' % d['roofline']['frac'], d['roofline'].get('traffic_source'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
for k, v in d['other_configs'].items():
    n = v.get('at_config_n') or {}
    print(k, 'T20 %.4g' % v.get('value', float('nan')), '| at N=%s: %.4g' % (n.get('transitions_per_step'), n.get('value', float('nan'))), 'ms %.0f' % n.get('ms_per_step', float('nan')), 'sec %.1f' % v['seconds_with_setup'], v.get('error', ''))
PY
