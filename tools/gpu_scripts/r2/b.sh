#!/bin/bash
# round 2, GPU call B: mw kernel v3 — bounded smoke, engine + parity tests, scaling and PMC experiments
set -u
mkdir -p gpurun_out/r2c
export PYTHONPATH=tests
timeout -s KILL 120 python - > gpurun_out/r2c/smoke.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, 'tests')
import numpy as np, oracle_lib as ol
from __graft_entry__ import load_package
pkg = load_package()
for D in (512, 1000):
    dev = pkg.DeviceContext(D, 5, seed=3); ora = ol.Oracle(D, 5, seed=3, threads=8)
    dev.init(); ora.init(); dev.find_initial_stepsize(); ora.find_initial_stepsize()
    a = dev.run(8, da={}); print('D', D, 'mw run ok', flush=True)
    b = ora.run(8, da={})
    for k in a:
        ok = np.array_equal(a[k], b[k])
        print('  ', k, 'OK' if ok else 'MISMATCH', flush=True)
        if not ok:
            bad = np.argwhere(a[k] != b[k]); print('    first', bad[:3].tolist(), a[k][tuple(bad[0])], b[k][tuple(bad[0])])
PY
echo "smoke rc=$?"; tail -25 gpurun_out/r2c/smoke.txt
timeout -s KILL 900 python -m pytest tests/test_gpu_engines.py tests/test_golden.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2c/pytest_core.txt 2>&1
echo "core rc=$?"; tail -5 gpurun_out/r2c/pytest_core.txt
timeout -s KILL 300 python tools/experiments/mw_scaling.py 256 512 768 1024 1536 2304 3072 4096 > gpurun_out/r2c/scaling.txt 2>&1
cat gpurun_out/r2c/scaling.txt
bash tools/experiments/mw_pmc.sh r2c 3072 > gpurun_out/r2c/pmc.txt 2>&1
grep -A200 "kernel stats" gpurun_out/r2c/pmc.txt | grep -v onewave | head -60
