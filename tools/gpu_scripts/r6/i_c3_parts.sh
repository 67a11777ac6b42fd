#!/bin/bash
# round 6: config 3, one batch on one stream against two half-batches (the two kinds of kernel stretch each other when they overlap: is the overlap worth
# the second set of small launches?)
O=gpurun_out/r6i; mkdir -p $O
for v in "parts2_free DHMC_DENSE=alternate=0" "parts1 DHMC_DENSE=parts=1" "parts2_alternate DHMC_NOTHING=1" "parts1_again DHMC_DENSE=parts=1" "parts4 DHMC_DENSE=parts=4,alternate=0"; do
  set -- $v
  r=$(env $2 timeout 600 python bench.py --config 3 --steps 3 --warmup 1 --transitions 100 --no-cpu-baseline 2>$O/err_$1.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3: %.4g leapfrog-steps/s, %.1f ms/step, frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
  echo "$1: $r" | tee -a $O/c3.txt
done
DHMC_DENSE=parts=1 bash tools/experiments/c3_trace.sh > $O/trace_parts1.txt 2>&1; cp gpurun_out/c3trace/timeline.txt $O/timeline_parts1.txt; head -3 $O/trace_parts1.txt
