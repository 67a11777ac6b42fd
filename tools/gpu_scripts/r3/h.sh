#!/bin/bash
set -u
O=gpurun_out/r3h; mkdir -p $O
export PYTHONPATH=tests
timeout -s KILL 900 python -m pytest tests/test_golden.py tests/test_gpu_parity.py tests/test_gpu_engines.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee $O/log.txt
tail -8 $O/pytest.txt | tee -a $O/log.txt
timeout -s KILL 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
python -c "
import json; d = json.load(open('$O/bench.json')); print('config 2: %.4g steps/s' % d['value'], 'frac %.4f' % d['roofline']['frac'], 'warm %.4g' % d['warmup_phase']['value'], d['tree'])" | tee -a $O/log.txt
DHMC_MOMENTUM_CHUNK=500 timeout -s KILL 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_chunk500.json
python -c "
import json; d = json.load(open('$O/bench_chunk500.json')); print('one chunk of 500: %.4g steps/s' % d['value'], 'frac %.4f' % d['roofline']['frac'])" | tee -a $O/log.txt
DHMC_MOMENTUM_CHUNK=16 timeout -s KILL 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_chunk16.json
python -c "
import json; d = json.load(open('$O/bench_chunk16.json')); print('chunks of 16: %.4g steps/s' % d['value'], 'frac %.4f' % d['roofline']['frac'])" | tee -a $O/log.txt
