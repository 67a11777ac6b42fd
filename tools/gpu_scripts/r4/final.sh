#!/bin/bash
# round 4: the numbers and profiles of the final build — GPU suite, bench lines (default with other_configs + live traffic +
# CPU baseline; configs 3 / 4 / 5; funnel with all 32768 chains), rocprofv3 trace + PMC passes of the headline kernel, kernel
# stats of configs 3 and 5, a 90-second fuzz sweep.
O=$PWD/gpurun_out/r4final; mkdir -p $O
timeout -s KILL 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.log; tail -3 $O/pytest.log
timeout -s KILL 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_default.json
for c in 3 4 5; do timeout -s KILL 200 python bench.py --config $c --steps 5 --warmup 1 2>/dev/null | tail -1 > $O/bench_c$c.json; done
timeout -s KILL 200 python bench.py --config 4 --chains 32768 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c4_32768.json
timeout -s KILL 300 python bench.py --config 4 --transitions 1000 --steps 1 --warmup 0 2>/dev/null | tail -1 > $O/bench_c4_1000transitions.json
for f in bench_default bench_c3 bench_c4 bench_c5 bench_c4_32768 bench_c4_1000transitions; do python -c "
import json; d = json.load(open('$O/$f.json')); print('$f %.4g' % d['value'], 'frac %.4f' % d['roofline']['frac'], (d.get('warmup_phase') or {}).get('value'), list((d.get('other_configs') or {}).keys()), d['roofline'].get('traffic_source'))"; done
timeout -s KILL 200 python tools/fuzz_parity.py 60 20260928 2>/dev/null | tail -2 > $O/fuzz.txt; cat $O/fuzz.txt
timeout -s KILL 500 bash tools/profile.sh r04 > $O/profile.log 2>&1; tail -60 $O/profile.log
export TMPDIR=/tmp; REPO=$PWD; cd /tmp
for c in 3 5; do
  rm -rf /tmp/pk$c; rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/pk$c -o t -- python $REPO/bench.py --config $c --steps 3 --warmup 1 > $O/bench_c${c}_under_rocprof.json 2> /tmp/pk$c.err
  f=$(find /tmp/pk$c -name '*kernel_stats.csv' | head -1); cp $f $O/c${c}_kernel_stats.csv
done
