#!/bin/bash
# round 6: the end game's threshold with two packed waves per SIMD (c4_32768, N = 1000): hand over earlier?
O=gpurun_out/r6m; mkdir -p $O
for v in "h1280 DHMC_NOTHING=1" "h1920 DHMC_PK=handover=1920" "h2560 DHMC_PK=handover=2560" "h3840 DHMC_PK=handover=3840" "h5120 DHMC_PK=handover=5120" "h896 DHMC_PK=handover=896"; do
  set -- $v
  r=$(env $2 DHMC_DEBUG_ORDER=1 timeout -s KILL 300 python bench.py --config 4 --chains 32768 --steps 2 --warmup 1 --no-cpu-baseline --config-n 1000 2>$O/err_$1.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g (steps of 20)' % d['value'], '%.4g at N=1000, %.0f ms' % (d['at_config_n']['value'], d['at_config_n']['ms_per_step']))")
  echo "c4_32768 $1: $r" | tee -a $O/c4.txt
  grep "end game" $O/err_$1.txt | tail -2
done
