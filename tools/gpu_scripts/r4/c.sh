#!/bin/bash
# round 4: per-case detmath self-test first (stops at the first case that does not come back); only then the GPU suite, the
# bench lines and the PMC profile.
bash tools/gpu_scripts/r4/b.sh || exit 1
O=gpurun_out/r4c; mkdir -p $O
timeout -s KILL 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $O/pytest.log; tail -5 $O/pytest.log
timeout -s KILL 120 python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_default.json
timeout -s KILL 120 python bench.py --config 4 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c4.json
timeout -s KILL 120 python bench.py --config 3 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c3.json
timeout -s KILL 120 python bench.py --config 5 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c5.json
for f in bench_default bench_c4 bench_c3 bench_c5; do python -c "
import json; d = json.load(open('$O/$f.json')); print('$f %.4g' % d['value'], 'frac %.3f' % d['roofline']['frac'], d.get('warmup_phase'))"; done
timeout -s KILL 420 bash tools/profile.sh r04a > $O/profile.log 2>&1; tail -45 $O/profile.log
