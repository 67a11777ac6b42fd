#!/bin/bash
# round 5: calls in rounds with the deepest chains promoted to the pipeline kernel by their work rate of the round before
O=gpurun_out/r5z; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_packed.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -12 > $O/tests.log; cat $O/tests.log
run() {  # chains name env
  env ${3//,/ } DHMC_HYBRID_MIN_CHAINS=8192 DHMC_DEBUG_ORDER=1 timeout 600 python bench.py --config 4 --chains $1 --transitions 1000 --steps 1 --warmup 0 2> $O/c4_$1_$2.err | tail -1 > $O/c4_$1_$2.json
  python -c "
import json; d = json.load(open('$O/c4_$1_$2.json')); print('$1 $2: %.4g' % d['value'], 'ms %.0f' % d['ms_per_step'])"
  grep "round" $O/c4_$1_$2.err | tail -40 | cut -c1-200 > $O/c4_$1_$2.rounds; tail -${4:-2} $O/c4_$1_$2.rounds
}
run 32768 packed_auto DHMC_PACKED=1 0
run 32768 r8_cus32_b16 DHMC_HYBRID_DEEP_CUS=32,DHMC_HYBRID_BUDGET=16,DHMC_HYBRID_DEEP_CAP=1 9
run 32768 r8_cus32_b8 DHMC_HYBRID_DEEP_CUS=32,DHMC_HYBRID_BUDGET=8,DHMC_HYBRID_DEEP_CAP=1
run 32768 r8_cus32_b16_cap2 DHMC_HYBRID_DEEP_CUS=32,DHMC_HYBRID_BUDGET=16,DHMC_HYBRID_DEEP_CAP=2
run 32768 r8_cus16_b16_cap2 DHMC_HYBRID_DEEP_CUS=16,DHMC_HYBRID_BUDGET=16,DHMC_HYBRID_DEEP_CAP=2
run 32768 r8_cus64_b16 DHMC_HYBRID_DEEP_CUS=64,DHMC_HYBRID_BUDGET=16,DHMC_HYBRID_DEEP_CAP=1
run 32768 r16_cus32_b16 DHMC_HYBRID_DEEP_CUS=32,DHMC_HYBRID_BUDGET=16,DHMC_HYBRID_DEEP_CAP=1,DHMC_HYBRID_SEGMENTS=16
run 32768 r5_cus32_b16 DHMC_HYBRID_DEEP_CUS=32,DHMC_HYBRID_BUDGET=16,DHMC_HYBRID_DEEP_CAP=1,DHMC_HYBRID_SEGMENTS=5
run 32768 r8_cus32_b16_p8 DHMC_HYBRID_DEEP_CUS=32,DHMC_HYBRID_BUDGET=16,DHMC_HYBRID_DEEP_CAP=1,DHMC_HYBRID_PROMOTE=8
run 16384 packed_auto DHMC_PACKED=1 0
run 16384 r8_cus32_b16 DHMC_HYBRID_DEEP_CUS=32,DHMC_HYBRID_BUDGET=16,DHMC_HYBRID_DEEP_CAP=1
run 16384 r8_cus64_b16 DHMC_HYBRID_DEEP_CUS=64,DHMC_HYBRID_BUDGET=16,DHMC_HYBRID_DEEP_CAP=1
run 8192 r8_cus64_b16 DHMC_HYBRID_DEEP_CUS=64,DHMC_HYBRID_BUDGET=16,DHMC_HYBRID_DEEP_CAP=1,DHMC_HYBRID_MIN_CHAINS=4096
