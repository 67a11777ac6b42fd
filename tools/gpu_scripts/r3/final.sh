#!/bin/bash
# end of round 3: the bench lines of every config on the final build
O=gpurun_out/r3final; mkdir -p $O
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_default.json
python bench.py --config 3 --steps 5 --warmup 1 2>/dev/null | tail -1 > $O/bench_c3.json
python bench.py --config 4 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c4.json
python bench.py --config 5 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c5.json
python bench.py --gpus 2 --chains 2048 --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_two_ranks.json
for f in bench_default bench_c3 bench_c4 bench_c5 bench_two_ranks; do python -c "
import json; d = json.load(open('$O/$f.json')); print('$f %.4g' % d['value'], 'frac %.3f' % d['roofline']['frac'], 'n_gpus', d['n_gpus'])"; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
