#!/bin/bash
# round 6, second session: config 5 at 960 / 1024 / 1088 chains — does the fused kernel's grid (98 x ceil(chains / 64) workgroups on 512 slots) quantise?
O=gpurun_out/r6br; mkdir -p $O
for ch in 1024 960 896 1088 1024 960; do
  r=$(timeout -s KILL 300 python bench.py --config 5 --chains $ch --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g leapfrog-steps/s, %.1f ms/step' % (d['value'], d['ms_per_step']))")
  echo "chains=$ch: $r" | tee -a $O/chains.txt
done
