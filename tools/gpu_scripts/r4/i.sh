#!/bin/bash
# round 4: asm-free device math policies (coefficients pinned to SGPRs by empty asm statements, every instruction the
# compiler's): the GPU suite, a 60-second fuzz sweep, A/B bench against the round-3 library, phase timing.
O=$PWD/gpurun_out/r4i; mkdir -p $O
timeout -s KILL 600 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest.log; tail -4 $O/pytest.log
FUZZ_VERBOSE=2 timeout -s KILL 200 python tools/fuzz_parity.py 60 777 2> $O/fuzz.err | tail -3
run() { ( cd $2 && timeout -s KILL 200 python bench.py --steps 5 --warmup 2 --transitions 200 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/bench_$1.json
  python -c "
import json; d = json.load(open('$O/bench_$1.json')); print('$1 %.4g' % d['value'], 'ms/step %.2f' % d['ms_per_step'])"; }
run v1 tools/experiments/_ab/v1
run v2 .
echo "== phase v2"; timeout -s KILL 120 bash tools/experiments/phase_timing.sh 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/phase_v2.txt
for c in 4 3 5; do timeout -s KILL 150 python bench.py --config $c --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c$c.json; python -c "
import json; d = json.load(open('$O/bench_c$c.json')); print('config $c %.4g' % d['value'], 'frac %.3f' % d['roofline']['frac'])"; done
