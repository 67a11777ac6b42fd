#!/bin/bash
# round 6, second session: config 3 with the batch in 1 / 2 (default) / 4 parts on the build with the rotated product kernel
O=gpurun_out/r6be; mkdir -p $O
for p in 1 2 4 2 1; do
  r=$(DHMC_DENSE="parts=$p" timeout -s KILL 300 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g leapfrog-steps/s, %.1f ms/step' % (d['value'], d['ms_per_step']))")
  echo "parts=$p: $r" | tee -a $O/parts.txt
done
