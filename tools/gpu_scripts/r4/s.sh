#!/bin/bash
# round 4: where do 11 % of the sampling kernel go after a warmup that held 12.5 GB of window draws (bench.py --warmup-draws)?
O=$PWD/gpurun_out/r4s; mkdir -p $O
run() {
  DHMC_BENCH_X=$1 timeout -s KILL 300 python bench.py --steps 6 --warmup 2 --no-other-configs --traffic none --no-cpu-baseline $2 2>/dev/null | tail -1 > $O/b.json
  python -c "
import json; d = json.load(open('$O/b.json')); print('[X=$1 $2] headline %.4g' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'], 'warmup_phase %.4g' % d['warmup_phase']['value'])" | tee -a $O/ab.txt
}
run "" ""
run "" "--warmup-draws"
run "emptycache" "--warmup-draws"
run "dummy" ""
run "dummy,emptycache" ""
run "dummyhold" ""
