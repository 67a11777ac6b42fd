#!/bin/bash
# round 5: the headline kernel's two-waves-per-SIMD layout (DHMC_L1_LDS=0: level-1 rows and trajectory ends out of LDS, 256 VGPRs) against the library's
for v in "library DHMC_NOTHING=1" "two_waves_per_simd DHMC_L1_LDS=0"; do
  set -- $v
  r=$(env $2 timeout -s KILL 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --traffic none --short-warmup 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g leapfrog-steps/s' % d['value'], 'ms %.1f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'], d['roofline']['kernel'])")
  echo "$1: $r"
done
