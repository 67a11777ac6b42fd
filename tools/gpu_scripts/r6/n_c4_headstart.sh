#!/bin/bash
# round 6 experiment: the head of the launch order starts in the pipeline kernel beside the packed launch (DHMC_PK=headstart=<chains>)
O=gpurun_out/r6n; mkdir -p $O
DHMC_PK=headstart=300 timeout -s KILL 600 python -m pytest tests/test_gpu_packed.py -m gpu -x -q -k "end_game_as_shipped" 2>&1 | tail -3 | tee $O/parity.txt
for v in "h0 DHMC_NOTHING=1" "h256 DHMC_PK=headstart=256" "h512 DHMC_PK=headstart=512" "h1024 DHMC_PK=headstart=1024" "h1280 DHMC_PK=headstart=1280" "h2560 DHMC_PK=headstart=2560" "h0_again DHMC_NOTHING=1"; do
  set -- $v
  r=$(env $2 DHMC_DEBUG_ORDER=1 timeout -s KILL 300 python bench.py --config 4 --chains 32768 --steps 2 --warmup 1 --no-cpu-baseline --config-n 1000 2>$O/err_$1.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g (steps of 20)' % d['value'], '%.4g at N=1000, %.0f ms' % (d['at_config_n']['value'], d['at_config_n']['ms_per_step']))")
  echo "c4_32768 $1: $r" | tee -a $O/c4.txt
  grep "end game" $O/err_$1.txt | tail -1
done
