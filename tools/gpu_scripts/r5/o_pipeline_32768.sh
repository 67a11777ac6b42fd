#!/bin/bash
# round 5: which engine for ALL 32768 funnel chains on one GPU (config 4 unsharded), and for 16384 / 8192
O=gpurun_out/r5s; mkdir -p $O
for ch in 32768 8192; do
  for v in "pipeline DHMC_PIPELINE=1" "wave DHMC_PIPELINE=0,DHMC_PACKED=0" "auto X=1"; do
    set -- $v
    env $(echo $2 | tr ',' ' ') timeout 900 python bench.py --config 4 --chains $ch --transitions 1000 --steps 1 --warmup 0 2>/dev/null | tail -1 > $O/c4_${ch}_$1.json
    python -c "
import json; d = json.load(open('$O/c4_${ch}_$1.json')); print('$ch $1: %.4g' % d['value'], 'ms %.0f' % d['ms_per_step'], d['tree'].get('slowest_chain_leapfrogs'))"
  done
done
