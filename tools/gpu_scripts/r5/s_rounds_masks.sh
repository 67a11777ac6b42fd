#!/bin/bash
# round 5: calls in rounds with the two kernels on disjoint CU masks; per-round timings of the two launches
O=gpurun_out/r5w; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_packed.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -12 > $O/tests.log; cat $O/tests.log
run() {  # chains name env
  env ${3//,/ } DHMC_HYBRID_MIN_CHAINS=8192 DHMC_DEBUG_ORDER=1 timeout 600 python bench.py --config 4 --chains $1 --transitions 1000 --steps 1 --warmup 0 2> $O/c4_$1_$2.err | tail -1 > $O/c4_$1_$2.json
  python -c "
import json; d = json.load(open('$O/c4_$1_$2.json')); print('$1 $2: %.4g' % d['value'], 'ms %.0f' % d['ms_per_step'])"
  grep "round" $O/c4_$1_$2.err | tail -40 | cut -c1-200 > $O/c4_$1_$2.rounds; tail -${4:-3} $O/c4_$1_$2.rounds
}
run 32768 r8_b8 DHMC_HYBRID_BUDGET=8 12
run 32768 r8_auto DHMC_HYBRID=1
run 32768 r8_b12 DHMC_HYBRID_BUDGET=12
run 32768 r8_b16_cus32 DHMC_HYBRID_BUDGET=16,DHMC_HYBRID_DEEP_CUS=32
run 32768 r8_b8_cus128 DHMC_HYBRID_BUDGET=8,DHMC_HYBRID_DEEP_CUS=128
run 32768 r8_b8_nomask DHMC_HYBRID_BUDGET=8,DHMC_HYBRID_DEEP_CUS=0
run 32768 r16_b8 DHMC_HYBRID_BUDGET=8,DHMC_HYBRID_SEGMENTS=16
run 32768 r4_b8 DHMC_HYBRID_BUDGET=8,DHMC_HYBRID_SEGMENTS=4
run 32768 r16_b12_cap1 DHMC_HYBRID_BUDGET=12,DHMC_HYBRID_SEGMENTS=16,DHMC_HYBRID_DEEP_CAP=1
run 16384 r8_b8 DHMC_HYBRID_BUDGET=8
run 16384 r8_b12_cap1 DHMC_HYBRID_BUDGET=12,DHMC_HYBRID_DEEP_CAP=1
