#!/bin/bash
# round 5, first GPU call: the packed small-D engine — parity (its own suite, the funnel config test, the goldens), then config 4
# through both engines at 20 and 1000 transitions per step and with all 32768 chains on the GPU
O=gpurun_out/r5a; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_packed.py -x -q 2>&1 | tail -25 > $O/packed.log; cat $O/packed.log
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_golden.py tests/test_gpu_parity.py -q -m gpu -k "funnel or golden or config4 or window or launch_order" 2>&1 | tail -15 > $O/funnel.log; cat $O/funnel.log
for pk in 1 0; do
  DHMC_PACKED=$pk timeout 300 python bench.py --config 4 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/c4_T20_pk$pk.json
  DHMC_PACKED=$pk timeout 600 python bench.py --config 4 --transitions 1000 --steps 1 --warmup 0 2>/dev/null | tail -1 > $O/c4_T1000_pk$pk.json
done
DHMC_PACKED=1 timeout 600 python bench.py --config 4 --chains 32768 --transitions 1000 --steps 1 --warmup 0 2>/dev/null | tail -1 > $O/c4_32768_T1000_pk1.json
for a in 1 2 8; do DHMC_PK_ALIGN=$a timeout 600 python bench.py --config 4 --transitions 1000 --steps 1 --warmup 0 2>/dev/null | tail -1 > $O/c4_T1000_align$a.json; done
for f in $O/c4_*.json; do python -c "
import json,sys; d = json.load(open('$f')); print('$f', '%.4g' % d['value'], 'ms/step %.1f' % d['ms_per_step'], d['tree'])"; done
