#!/bin/bash
# round 4: metric windows (warmup without the draws) — GPU suite, the A/B of the headline kernel with / without the window's code,
# the bench line (warmup_phase now through windows)
O=$PWD/gpurun_out/r4q; mkdir -p $O
timeout -s KILL 200 python -m pytest tests/test_gpu_parity.py -q -k "metric_window or adaptive_stages" 2>&1 | tail -8 > $O/pytest_window.log; cat $O/pytest_window.log
for i in 1 2; do
  timeout -s KILL 150 python tools/experiments/ab_window.py 2>/dev/null | tail -1 >> $O/ab.txt
  DHMC_LIB_PATH=$PWD/tools/experiments/_v/nowin/libdhmc_amd.so timeout -s KILL 150 python tools/experiments/ab_window.py 2>/dev/null | tail -1 >> $O/ab.txt
done
cat $O/ab.txt
timeout -s KILL 400 python bench.py --steps 10 --warmup 3 --no-other-configs --traffic none 2>/dev/null | tail -1 > $O/bench_headline.json
python -c "
import json; d = json.load(open('$O/bench_headline.json')); print('headline %.4g' % d['value'], 'frac %.4f' % d['roofline']['frac'], d.get('warmup_phase'))"
timeout -s KILL 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/pytest.log; tail -4 $O/pytest.log
