#!/bin/bash
# config 3 (and others), three runs each of the current build:  g_c3_quick.sh <out dir> "<configs>"
O=gpurun_out/${1:-r6bg}; mkdir -p $O
for c in ${2:-3}; do
 for i in 1 2 3; do
  r=$(timeout -s KILL 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g leapfrog-steps/s, %.1f ms/step' % (d['value'], d['ms_per_step']))")
  echo "c$c: $r" | tee -a $O/bench.txt
 done
done
