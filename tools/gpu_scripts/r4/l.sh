#!/bin/bash
# round 4: short-chain reductions (skip all-zero butterfly rows): parity, fuzz, funnel share, funnel phase timing
O=$PWD/gpurun_out/r4l; mkdir -p $O
timeout -s KILL 400 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_engines.py tests/test_gpu_probes.py -m gpu -q -x 2>&1 | tail -4
timeout -s KILL 100 python tools/fuzz_parity.py 40 991 2>/dev/null | tail -3
timeout -s KILL 150 python bench.py --config 4 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c4.json; python -c "
import json; d = json.load(open('$O/bench_c4.json')); print('config 4 %.4g' % d['value'])"
echo "== phase funnel (1 chain per SIMD: 1024 chains)"; FAM=FunnelT PH_D=30 PH_TARGET=funnel timeout -s KILL 120 bash tools/experiments/phase_timing.sh 1024 50 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/phase_funnel.txt
