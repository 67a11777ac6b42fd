#!/bin/bash
# round 3 profiles: config 2 (kernel trace + PMC passes), config 3 (kernel stats), the bench lines of all configs
O=gpurun_out/r3p; mkdir -p $O
bash tools/profile.sh r03 > $O/profile_log.txt 2>&1
REPO=$PWD
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/c3s
timeout 500 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/c3s -o t -- python $REPO/bench.py --config 3 --steps 3 --warmup 1 > $REPO/$O/c3_under_rocprof.json 2> /dev/null
cp $(find /tmp/c3s -name '*kernel_stats.csv' | head -1) $REPO/$O/c3_kernel_stats.csv
cd $REPO
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_default.json
python bench.py --config 3 --steps 5 --warmup 1 2>/dev/null | tail -1 > $O/bench_c3.json
python bench.py --config 4 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c4.json
python bench.py --config 5 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c5.json
for f in bench_default bench_c3 bench_c4 bench_c5; do python -c "
import json; d = json.load(open('$O/$f.json')); print('$f %.4g' % d['value'], 'frac %.3f' % d['roofline']['frac'])"; done
tail -30 gpurun_out/prof_r03/summary.txt
