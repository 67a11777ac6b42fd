#!/bin/bash
# round 4: fused logaddexp pair; parity of the per-draw kernels (suite subset + fuzz), A/B bench, phase timing, funnel share
O=$PWD/gpurun_out/r4j; mkdir -p $O
timeout -s KILL 400 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_detmath.py tests/test_gpu_dense.py -m gpu -q -x 2>&1 | tail -5
timeout -s KILL 100 python tools/fuzz_parity.py 30 4242 2>/dev/null | tail -3
run() { ( cd $2 && timeout -s KILL 200 python bench.py --steps 5 --warmup 2 --transitions 200 --no-cpu-baseline --no-other-configs --traffic none 2>/dev/null | tail -1 ) > $O/bench_$1.json
  python -c "
import json; d = json.load(open('$O/bench_$1.json')); print('$1 %.4g' % d['value'], 'ms/step %.2f' % d['ms_per_step'])"; }
( cd tools/experiments/_ab/v1 && timeout -s KILL 200 python bench.py --steps 5 --warmup 2 --transitions 200 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/bench_v1.json; python -c "
import json; d = json.load(open('$O/bench_v1.json')); print('v1 %.4g' % d['value'])"
run v2 .
echo "== phase v2"; timeout -s KILL 120 bash tools/experiments/phase_timing.sh 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/phase_v2.txt
timeout -s KILL 150 python bench.py --config 4 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c4.json; python -c "
import json; d = json.load(open('$O/bench_c4.json')); print('config 4 %.4g' % d['value'])"
