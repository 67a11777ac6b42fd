#!/bin/bash
# round 4: (1) which use of the device math policies the funnel D = 1000 fuzz mismatches follow (funnel family rebuilt with one
# use at a time on the compiler's own code); (2) A/B bench and phase timing of the round-3 library against the working tree
# (momentum refresh now gathers its table rows in batches of four pairs).
O=$PWD/gpurun_out/r4h; mkdir -p $O
for v in f_cur f_rexpgen f_vecgen f_unigen; do
  echo "== fuzz with funnel variant $v"; DHMC_LIB_PATH=$PWD/tools/experiments/_v/$v/libdhmc_amd.so FUZZ_VERBOSE=2 timeout -s KILL 120 python tools/fuzz_parity.py 20 12345 2> $O/fuzz_$v.err | tail -3
done
run() { ( cd $2 && timeout -s KILL 200 python bench.py --steps 5 --warmup 2 --transitions 200 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/bench_$1.json
  python -c "
import json; d = json.load(open('$O/bench_$1.json')); print('$1 %.4g' % d['value'], 'ms/step %.2f' % d['ms_per_step'])"; }
run v1 tools/experiments/_ab/v1
run v2 .
echo "== phase v2"; timeout -s KILL 120 bash tools/experiments/phase_timing.sh 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/phase_v2.txt
