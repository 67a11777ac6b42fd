#!/bin/bash
# round 4: pipelined draw store, int64 lane array for step counts, leaf·leaf turn check fused into the odd leaf's reduction;
# plus the cross-rank metric pooling tests
O=$PWD/gpurun_out/r4m; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_engines.py tests/test_gpu_dense.py tests/test_gpu_statistics.py -m gpu -q -x 2>&1 | tail -4
timeout -s KILL 100 python tools/fuzz_parity.py 40 2718 2>/dev/null | tail -3
run() { ( cd $2 && timeout -s KILL 200 python bench.py --steps 5 --warmup 2 --transitions 200 --no-cpu-baseline --no-other-configs --traffic none 2>/dev/null | tail -1 ) > $O/bench_$1.json
  python -c "
import json; d = json.load(open('$O/bench_$1.json')); print('$1 %.4g' % d['value'], 'ms/step %.2f' % d['ms_per_step'])"; }
run v2 .
run v2b .
echo "== phase v2"; timeout -s KILL 120 bash tools/experiments/phase_timing.sh 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/phase_v2.txt
timeout -s KILL 150 python bench.py --config 4 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c4.json; python -c "
import json; d = json.load(open('$O/bench_c4.json')); print('config 4 %.4g' % d['value'])"
