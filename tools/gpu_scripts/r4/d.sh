#!/bin/bash
# round 4: same-box A/B of the round-3 library (ABI v1 math) against the working tree (ABI v2), shader clock / power sampled
# beside each run; then the GPU suite on the working tree.
O=gpurun_out/r4d; mkdir -p $O
smi() { ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Average Graphics Package Power\|Current Socket" | tr -s ' ' | tr '\n' ' '; echo; sleep 1; done ) > $1 & echo $!; }
run() {  # name dir
  P=$(smi $O/smi_$1.txt)
  ( cd $2 && timeout -s KILL 200 python bench.py --steps 5 --warmup 2 --transitions 200 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/bench_$1.json
  kill $P
  python -c "
import json; d = json.load(open('$O/bench_$1.json')); print('$1 %.4g' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'kernel_ms %.2f' % d['roofline'].get('kernel_ms', 0))"
  sort $O/smi_$1.txt | uniq -c | sort -rn | head -3
}
run v1_a tools/experiments/_ab/v1
run v2_a .
run v1_b tools/experiments/_ab/v1
run v2_b .
timeout -s KILL 600 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/pytest.log; tail -5 $O/pytest.log
