#!/bin/bash
# round 5: the packed kernel's queue of places — parity (queue forced by a small wave limit; hybrid segments), then all 32768 /
# 16384 funnel chains on the GPU through every engine choice
O=gpurun_out/r5u; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_packed.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -8 > $O/tests.log; cat $O/tests.log
for ch in 32768 16384; do
  for v in "wave DHMC_PACKED=0,DHMC_PIPELINE=0" "packed_queue DHMC_PACKED=1" "packed_noqueue DHMC_PACKED=1,DHMC_PK_QUEUE=0" \
           "hybrid8 DHMC_HYBRID=1" "hybrid16 DHMC_HYBRID_SEGMENTS=16" "hybrid8_w768 DHMC_PK_MAX_WAVES=768" "hybrid8_noqueue DHMC_PK_QUEUE=0" "policy_nohybrid DHMC_HYBRID=0"; do
    set -- $v
    env ${2//,/ } DHMC_DEBUG_ORDER=1 timeout 600 python bench.py --config 4 --chains $ch --transitions 1000 --steps 1 --warmup 0 2> $O/c4_${ch}_$1.err | tail -1 > $O/c4_${ch}_$1.json
    python -c "
import json; d = json.load(open('$O/c4_${ch}_$1.json')); print('$ch $1: %.4g' % d['value'], 'ms %.0f' % d['ms_per_step'])"
    grep "launch order" $O/c4_${ch}_$1.err | tail -2
  done
done
