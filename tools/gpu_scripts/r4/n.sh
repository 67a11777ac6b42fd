#!/bin/bash
O=$PWD/gpurun_out/r4n; mkdir -p $O
timeout -s KILL 300 python -m pytest tests/test_gpu_dense.py -m gpu -q -x 2>&1 | tail -40 | cut -c1-220
run() { ( cd $2 && timeout -s KILL 200 python bench.py --steps 5 --warmup 2 --transitions 200 --no-cpu-baseline --no-other-configs --traffic none 2>/dev/null | tail -1 ) > $O/bench_$1.json
  python -c "
import json; d = json.load(open('$O/bench_$1.json')); print('$1 %.4g' % d['value'], 'ms/step %.2f' % d['ms_per_step'])"; }
run v2 .
run v2b .
