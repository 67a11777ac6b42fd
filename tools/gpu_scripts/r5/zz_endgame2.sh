#!/bin/bash
# round 5: from how many chains the packed launch with its end game beats the pipeline kernel alone (config 4, one call of N = 1000)
O=gpurun_out/r5eg; mkdir -p $O
run() {  # chains name env
  env ${3//,/ } DHMC_DEBUG_ORDER=1 timeout -s KILL 120 python bench.py --config 4 --chains $1 --transitions 1000 --steps 1 --warmup 0 2> $O/c4_$1_$2.err | tail -1 > $O/c4_$1_$2.json
  python -c "
import json; d = json.load(open('$O/c4_$1_$2.json')); print('$1 $2: %.4g' % d['value'], 'ms %.0f' % d['ms_per_step'])"
  grep "end game\|engine:" $O/c4_$1_$2.err | tail -2 | tr '\n' ' '; echo
}
run 4096 pipeline DHMC_NOTHING=1
run 4096 endgame DHMC_MANY_CHAINS=2000
run 4096 endgame_h2560 DHMC_MANY_CHAINS=2000,DHMC_PK_HANDOVER=2560
run 8192 pipeline DHMC_NOTHING=1
run 8192 endgame DHMC_MANY_CHAINS=2000
run 8192 endgame_h2560 DHMC_MANY_CHAINS=2000,DHMC_PK_HANDOVER=2560
run 8192 endgame_h4000 DHMC_MANY_CHAINS=2000,DHMC_PK_HANDOVER=4000
run 16384 endgame_h640 DHMC_MANY_CHAINS=2000,DHMC_PK_HANDOVER=640
