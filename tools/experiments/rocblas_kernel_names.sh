#!/bin/bash
# which Tensile kernels (macro tile, MFMA instruction) the vendor fp64 GEMM runs on our shapes — names from a kernel trace
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/rk
rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/rk -o t -- python $REPO/tools/experiments/rocblas_dgemm_ceiling.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for r in csv.DictReader(open(glob.glob('/tmp/rk/**/*kernel_stats.csv', recursive=True)[0])):
    print(r['Calls'], r['AverageNs'], r['Name'][:400])
PY
