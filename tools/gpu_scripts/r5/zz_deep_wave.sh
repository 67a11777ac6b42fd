#!/bin/bash
# round 5: calls in rounds with the wave-per-chain kernel as the deep chains' engine (no CU masks: its waves sit beside the packed ones)
O=gpurun_out/r5dw; mkdir -p $O
timeout -s KILL 300 python -m pytest tests/test_gpu_pipeline.py -x -q -k "hybrid" 2>&1 | tail -5 > $O/tests.log; cat $O/tests.log
run() {  # chains name env
  env ${3//,/ } DHMC_HYBRID=1 DHMC_HYBRID_DEEP=wave DHMC_HYBRID_DEEP_CUS=0 DHMC_HYBRID_MIN_CHAINS=8192 DHMC_DEBUG_ORDER=1 timeout -s KILL 120 python bench.py --config 4 --chains $1 --transitions 1000 --steps 1 --warmup 0 2> $O/c4_$1_$2.err | tail -1 > $O/c4_$1_$2.json
  python -c "
import json; d = json.load(open('$O/c4_$1_$2.json')); print('$1 $2: %.4g' % d['value'], 'ms %.0f' % d['ms_per_step'])"
  grep "round" $O/c4_$1_$2.err | tail -40 | cut -c1-200 > $O/c4_$1_$2.rounds; tail -${4:-2} $O/c4_$1_$2.rounds
}
run 32768 cap4_p3_b8 DHMC_HYBRID_DEEP_CAP=4,DHMC_HYBRID_PROMOTE=3,DHMC_HYBRID_BUDGET=8 9
run 32768 cap8_p2_b8 DHMC_HYBRID_DEEP_CAP=8,DHMC_HYBRID_PROMOTE=2,DHMC_HYBRID_BUDGET=8
run 32768 cap8_p2_b6 DHMC_HYBRID_DEEP_CAP=8,DHMC_HYBRID_PROMOTE=2,DHMC_HYBRID_BUDGET=6
run 32768 cap16_p1.5_b6 DHMC_HYBRID_DEEP_CAP=16,DHMC_HYBRID_PROMOTE=1.5,DHMC_HYBRID_BUDGET=6
run 32768 cap8_p2_b8_r16 DHMC_HYBRID_DEEP_CAP=8,DHMC_HYBRID_PROMOTE=2,DHMC_HYBRID_BUDGET=8,DHMC_HYBRID_SEGMENTS=16
run 32768 cap8_p2_b8_w768 DHMC_HYBRID_DEEP_CAP=8,DHMC_HYBRID_PROMOTE=2,DHMC_HYBRID_BUDGET=8,DHMC_PK_MAX_WAVES=768
run 16384 cap8_p2_b8 DHMC_HYBRID_DEEP_CAP=8,DHMC_HYBRID_PROMOTE=2,DHMC_HYBRID_BUDGET=8
