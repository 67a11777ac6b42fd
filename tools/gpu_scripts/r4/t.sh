#!/bin/bash
# round 4: the straggler reading of the headline's two values (3.25e8 after --warmup-draws, 3.60e8 after metric windows)
O=$PWD/gpurun_out/r4t; mkdir -p $O
run() {
  timeout -s KILL 300 python bench.py --steps 6 --warmup 2 --no-other-configs --traffic none --no-cpu-baseline $1 2>/dev/null | tail -1 > $O/b.json
  python -c "
import json; d = json.load(open('$O/b.json')); print('[$1] headline %.4g' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'], d['tree'])" | tee -a $O/ab.txt
}
run ""
run "--warmup-draws"
run "--seed 7"
run "--seed 7 --warmup-draws"
run "--seed 8"
run "--seed 9"
