#!/bin/bash
# round 5: hybrid segments (deep chains through the pipeline kernel beside the packed launch of the rest) — parity, then all 32768
# funnel chains on the GPU
O=gpurun_out/r5t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k hybrid 2>&1 | tail -8 > $O/hybrid_test.log; cat $O/hybrid_test.log
for ch in 32768 16384; do
  for v in "hybrid8 DHMC_HYBRID=1" "hybrid16 DHMC_HYBRID_SEGMENTS=16" "hybrid4 DHMC_HYBRID_SEGMENTS=4" "off DHMC_HYBRID=0"; do
    set -- $v
    env $2 DHMC_DEBUG_ORDER=1 timeout 900 python bench.py --config 4 --chains $ch --transitions 1000 --steps 1 --warmup 0 2> $O/c4_${ch}_$1.err | tail -1 > $O/c4_${ch}_$1.json
    python -c "
import json; d = json.load(open('$O/c4_${ch}_$1.json')); print('$ch $1: %.4g' % d['value'], 'ms %.0f' % d['ms_per_step'])"
    grep "launch order" $O/c4_${ch}_$1.err | tail -3
  done
done
