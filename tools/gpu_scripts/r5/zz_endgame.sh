#!/bin/bash
# round 5: the end game of a tail-bound packed launch (hand-over of the last chains to the pipeline kernel) — parity, then 32768 / 16384 chains
O=gpurun_out/r5eg; mkdir -p $O
timeout -s KILL 300 python -m pytest tests/test_gpu_packed.py -x -q -k "end_game or queue" 2>&1 | tail -6 > $O/tests.log; cat $O/tests.log
run() {  # chains name env
  env ${3//,/ } DHMC_DEBUG_ORDER=1 timeout -s KILL 120 python bench.py --config 4 --chains $1 --transitions 1000 --steps 1 --warmup 0 2> $O/c4_$1_$2.err | tail -1 > $O/c4_$1_$2.json
  python -c "
import json; d = json.load(open('$O/c4_$1_$2.json')); print('$1 $2: %.4g' % d['value'], 'ms %.0f' % d['ms_per_step'])"
  grep "end game\|engine:" $O/c4_$1_$2.err | tail -2
}
run 32768 off DHMC_PK_HANDOVER=0
run 32768 default DHMC_NOTHING=1
run 32768 h640 DHMC_PK_HANDOVER=640
run 32768 h2560 DHMC_PK_HANDOVER=2560
run 32768 h5120 DHMC_PK_HANDOVER=5120
run 16384 default DHMC_MANY_CHAINS=8192
run 16384 h2560 DHMC_MANY_CHAINS=8192,DHMC_PK_HANDOVER=2560
