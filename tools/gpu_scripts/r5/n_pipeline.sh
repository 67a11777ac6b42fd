#!/bin/bash
# round 5: the pair kernel (integrator ‖ tree builder) — parity, then config 4 timing: wave vs pair
O=gpurun_out/r5r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_packed.py -x -q 2>&1 | tail -12 > $O/pair.log; cat $O/pair.log
for v in "pipeline DHMC_PIPELINE=1" "wave DHMC_PIPELINE=0"; do
  set -- $v
  env $2 DHMC_PACKED=0 PH_STUCK=1 timeout 120 python tools/experiments/packed_probe.py 8 10 2>&1 | grep chains | sed "s/^/stuck $1: /"
  env $2 DHMC_PACKED=0 timeout 600 python bench.py --config 4 --transitions 1000 --steps 1 --warmup 0 2>/dev/null | tail -1 > $O/c4_T1000_$1.json
  env $2 DHMC_PACKED=0 timeout 300 python bench.py --config 4 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/c4_T20_$1.json
done
timeout 600 python bench.py --config 4 --steps 3 --warmup 1 --config-n 1000 2>/dev/null | tail -1 > $O/c4_auto.json
for f in $O/c4_*.json; do python -c "
import json,sys; d = json.load(open('$f')); n = d.get('at_config_n') or {}; print('$f', '%.4g' % d['value'], 'ms/step %.1f' % d['ms_per_step'], '| N=1000: %.4g' % n.get('value', float('nan')))"; done
