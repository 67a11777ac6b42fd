#!/bin/bash
# round 4: launch order by the previous launch's work (longest chain first) — invariance tests, then the headline with the seeds
# that have a straggler chain (default seed after --warmup-draws: 1.48 x the mean; seed 7: 1.17 x) and one that has none
O=$PWD/gpurun_out/r4u; mkdir -p $O
timeout -s KILL 300 python -m pytest tests/test_gpu_engines.py -q -k "launch_order" 2>&1 | tail -5
run() {
  DHMC_LAUNCH_ORDER=$1 timeout -s KILL 300 python bench.py --steps 6 --warmup 2 --no-other-configs --traffic none --no-cpu-baseline $2 2>/dev/null | tail -1 > $O/b.json
  python -c "
import json; d = json.load(open('$O/b.json')); print('[order=$1 $2] headline %.4g' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'], 'slowest/mean %.3f' % d['tree']['slowest_chain_over_mean_leapfrogs'], 'warmup_phase %.4g' % d['warmup_phase']['value'])" | tee -a $O/ab.txt
}
run 1 "--warmup-draws"
run 0 "--warmup-draws"
run 1 "--seed 7"
run 0 "--seed 7"
run 1 ""
run 0 ""
for c in 4; do for o in 1 0; do
DHMC_LAUNCH_ORDER=$o timeout -s KILL 200 python bench.py --config $c --steps 5 --warmup 1 2>/dev/null | tail -1 > $O/c.json
python -c "
import json; d = json.load(open('$O/c.json')); print('[config $c order=$o] %.4g' % d['value'])" | tee -a $O/ab.txt
DHMC_LAUNCH_ORDER=$o timeout -s KILL 200 python bench.py --config $c --chains 32768 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/c.json
python -c "
import json; d = json.load(open('$O/c.json')); print('[config $c 32768 chains order=$o] %.4g' % d['value'])" | tee -a $O/ab.txt
done; done
