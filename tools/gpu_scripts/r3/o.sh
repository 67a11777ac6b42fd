#!/bin/bash
export PYTHONPATH=tests
O=gpurun_out/r3o; mkdir -p $O; rm -f $O/log.txt
bash tools/experiments/time_variants.sh 2>/dev/null | tee -a $O/log.txt
python bench.py --steps 3 --warmup 1 --transitions 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', '%.4g' % d['value'])" | tee -a $O/log.txt
