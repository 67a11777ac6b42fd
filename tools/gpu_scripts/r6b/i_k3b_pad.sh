#!/bin/bash
# experiment: K3b's residency limited by an LDS pad, beside the product kernel with two k-tiles of loads in flight
O=gpurun_out/r6bi2; mkdir -p $O
for pad in 0 24000 44000 0 24000; do
  r=$(DHMC_K3B_PAD_EXPERIMENT=$pad timeout -s KILL 300 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g leapfrog-steps/s, %.1f ms/step' % (d['value'], d['ms_per_step']))")
  echo "pad=$pad: $r" | tee -a $O/pad.txt
done
