#!/bin/bash
# round 5: calls in rounds (packed bulk with a leapfrog budget, given-up and deep chains through the pipeline kernel) — parity, then
# all 32768 / 16384 funnel chains on the GPU
O=gpurun_out/r5v; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_packed.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -12 > $O/tests.log; cat $O/tests.log
for ch in 32768 16384; do
  for v in "rounds8 DHMC_HYBRID=1" "rounds16 DHMC_HYBRID_SEGMENTS=16" "rounds32 DHMC_HYBRID_SEGMENTS=32" "rounds16_cap1 DHMC_HYBRID_SEGMENTS=16,DHMC_HYBRID_DEEP_CAP=1" \
           "rounds16_b2 DHMC_HYBRID_SEGMENTS=16,DHMC_HYBRID_BUDGET=2" "rounds16_b8 DHMC_HYBRID_SEGMENTS=16,DHMC_HYBRID_BUDGET=8" "packed_queue DHMC_PACKED=1"; do
    set -- $v
    env ${2//,/ } DHMC_HYBRID_MIN_CHAINS=8192 DHMC_DEBUG_ORDER=1 timeout 600 python bench.py --config 4 --chains $ch --transitions 1000 --steps 1 --warmup 0 2> $O/c4_${ch}_$1.err | tail -1 > $O/c4_${ch}_$1.json
    python -c "
import json; d = json.load(open('$O/c4_${ch}_$1.json')); print('$ch $1: %.4g' % d['value'], 'ms %.0f' % d['ms_per_step'])"
    grep "round" $O/c4_${ch}_$1.err | tail -40 | cut -c1-150 > $O/c4_${ch}_$1.rounds; tail -4 $O/c4_${ch}_$1.rounds
  done
done
