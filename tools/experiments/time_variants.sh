#!/bin/bash
# GPU box: quick bench line (config 2, 3 timed steps of 100 transitions) for every library under tools/experiments/_v/
ROOT=$(cd $(dirname $0)/../.. && pwd)
for so in $ROOT/tools/experiments/_v/*/libdhmc_amd.so; do
    n=$(basename $(dirname $so))
    DHMC_LIB_PATH=$so timeout 300 python $ROOT/bench.py --steps 3 --warmup 1 --transitions 100 --no-cpu-baseline "$@" 2>&1 | tail -1 | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', '%.4g' % d['value'], 'frac %.4f' % d['roofline']['frac'])" || echo "$n FAILED"
done
