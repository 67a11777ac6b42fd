#!/bin/bash
# round 2, GPU call A: multi-wave kernel smoke (bounded), then the GPU suite, then bench both engines.
set -u
mkdir -p gpurun_out/r2a
export PYTHONPATH=tests
echo "== smoke (mw kernel vs oracle, 120 s cap)" | tee gpurun_out/r2a/log.txt
timeout -s KILL 120 python - >> gpurun_out/r2a/log.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, 'tests')
import numpy as np, oracle_lib as ol
from __graft_entry__ import load_package
pkg = load_package()
for D in (512, 1000):
    dev = pkg.DeviceContext(D, 5, seed=3); ora = ol.Oracle(D, 5, seed=3, threads=8)
    dev.init(); ora.init(); dev.find_initial_stepsize(); ora.find_initial_stepsize()
    t = time.time(); a = dev.run(8, da={}); print('D', D, 'mw run ok', time.time() - t, flush=True)
    b = ora.run(8, da={})
    for k in a:
        ok = np.array_equal(a[k], b[k])
        print('  ', k, 'OK' if ok else 'MISMATCH', flush=True)
        if not ok:
            bad = np.argwhere(a[k] != b[k]); print('    first', bad[:3].tolist(), a[k][tuple(bad[0])], b[k][tuple(bad[0])])
PY
echo "smoke rc=$?" | tee -a gpurun_out/r2a/log.txt
echo "== engines + golden + parity" | tee -a gpurun_out/r2a/log.txt
timeout -s KILL 900 python -m pytest tests/test_gpu_engines.py tests/test_golden.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2a/pytest_core.txt 2>&1
echo "core rc=$?" | tee -a gpurun_out/r2a/log.txt
tail -5 gpurun_out/r2a/pytest_core.txt | tee -a gpurun_out/r2a/log.txt
echo "== bench mw / one-wave" | tee -a gpurun_out/r2a/log.txt
timeout -s KILL 300 python bench.py --no-cpu-baseline > gpurun_out/r2a/bench_mw.json 2> gpurun_out/r2a/bench_mw.err
echo "bench mw rc=$?" | tee -a gpurun_out/r2a/log.txt
DHMC_MW=0 timeout -s KILL 300 python bench.py --no-cpu-baseline > gpurun_out/r2a/bench_onewave.json 2> gpurun_out/r2a/bench_onewave.err
echo "bench onewave rc=$?" | tee -a gpurun_out/r2a/log.txt
cat gpurun_out/r2a/bench_mw.json gpurun_out/r2a/bench_onewave.json | cut -c1-600 | tee -a gpurun_out/r2a/log.txt
echo "== rest of the GPU suite" | tee -a gpurun_out/r2a/log.txt
timeout -s KILL 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_engines.py --deselect tests/test_golden.py --deselect tests/test_gpu_parity.py > gpurun_out/r2a/pytest_rest.txt 2>&1
echo "rest rc=$?" | tee -a gpurun_out/r2a/log.txt
tail -15 gpurun_out/r2a/pytest_rest.txt | tee -a gpurun_out/r2a/log.txt
