#!/bin/bash
# round 5: chains resident per CU in the pipeline kernel (more LDS per block = fewer), config 4's 4096-chain share, one call of N = 1000
for pad in 0 12288 25600 53248 100000; do
  r=$(env DHMC_PIPE_LDS_PAD=$pad timeout -s KILL 100 python bench.py --config 4 --transitions 1000 --steps 1 --warmup 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g' % d['value'], 'ms %.0f' % d['ms_per_step'], d['tree'].get('slowest_chain_leapfrogs'))")
  echo "pad $pad: $r"
done
