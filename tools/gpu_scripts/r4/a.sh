#!/bin/bash
# round 4, first GPU call: ABI v2 scalar math — device == CPU function by function, the whole GPU suite, the bench lines, PMC
O=gpurun_out/r4a; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_detmath.py -x -q 2>&1 | tail -15 > $O/detmath.log; cat $O/detmath.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest.log; tail -5 $O/pytest.log
python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_default.json
python bench.py --config 4 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c4.json
python bench.py --config 3 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c3.json
python bench.py --config 5 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c5.json
for f in bench_default bench_c4 bench_c3 bench_c5; do python -c "
import json; d = json.load(open('$O/$f.json')); print('$f %.4g' % d['value'], 'frac %.3f' % d['roofline']['frac'], d.get('warmup_phase'))"; done
bash tools/profile.sh r04a > $O/profile.log 2>&1; tail -40 $O/profile.log
