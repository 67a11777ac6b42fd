#!/bin/bash
# round 5: packed engine with the tail of the launch order on the wave-per-chain kernel; config 4 at N = 1000
O=gpurun_out/r5f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_packed.py -x -q 2>&1 | tail -8 > $O/packed.log; cat $O/packed.log
for ch in 4096 32768; do
  for v in "tail1 DHMC_PK_TAIL=1" "tail0 DHMC_PK_TAIL=0" "wave DHMC_PACKED=0"; do
    set -- $v
    env $2 DHMC_DEBUG_ORDER=1 timeout 600 python bench.py --config 4 --chains $ch --transitions 1000 --steps 1 --warmup 0 2> $O/c4_${ch}_T1000_$1.err | tail -1 > $O/c4_${ch}_T1000_$1.json
    grep "launch order" $O/c4_${ch}_T1000_$1.err | tail -2
  done
done
DHMC_PK_TAIL=1 timeout 300 python bench.py --config 4 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/c4_4096_T20_tail1.json
for f in $O/c4_*.json; do python -c "
import json,sys; d = json.load(open('$f')); print('$f', '%.4g' % d['value'], 'ms/step %.1f' % d['ms_per_step'])"; done
