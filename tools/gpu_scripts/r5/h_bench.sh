#!/bin/bash
# round 5: the default bench line (other_configs now also at each config's own N), timing of the whole invocation
O=gpurun_out/r5h; mkdir -p $O
( time python bench.py 2> $O/bench.err | tail -1 > $O/bench_default.json ) 2> $O/time.txt
cat $O/time.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5h/bench_default.json'))
print('headline %.4g' % d['value'], 'frac %.3f' % d['roofline']['frac'], d['tree'].get('slowest_chain_over_mean_leapfrogs'))
for k,v in d['other_configs'].items():
    n=v.get('at_config_n') or {}
    print(k, 'T20 %.4g' % v['value'], 'frac %.3f' % v['roofline']['frac'], '| at N=%s: %.4g' % (n.get('transitions_per_step'), n.get('value', float('nan'))), 'frac %.3f' % n.get('roofline',{}).get('frac', float('nan')), 'ms %.0f' % n.get('ms_per_step', float('nan')), n.get('tree',{}).get('slowest_chain_leapfrogs'), n.get('tree',{}).get('mean_chain_leapfrogs'), 'sec', v['seconds_with_setup'])
PY
