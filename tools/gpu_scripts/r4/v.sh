#!/bin/bash
# round 4: the per-draw kernel's launches scheduled in segments (long chains first and through, the others round-robin per XCD)
O=$PWD/gpurun_out/r4v; mkdir -p $O
timeout -s KILL 150 python -m pytest tests/test_gpu_engines.py -q -x -k "launch_order" 2>&1 | tail -8
timeout -s KILL 100 python tools/experiments/straggler_probe.py 602890573 2>&1 | grep '"order"' | tail -3
DHMC_SCHED=0 timeout -s KILL 100 python tools/experiments/straggler_probe.py 602890573 2>&1 | grep '"order"' | tail -2
timeout -s KILL 100 python tools/experiments/launch_order_probe.py 2>&1 | grep '"order"'
run() {
  DHMC_SCHED=$1 timeout -s KILL 300 python bench.py --steps 6 --warmup 2 --no-other-configs --traffic none --no-cpu-baseline $2 2>/dev/null | tail -1 > $O/b.json
  python -c "
import json; d = json.load(open('$O/b.json')); print('[sched=$1 $2] headline %.4g' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'], 'slowest/mean %.3f' % d['tree']['slowest_chain_over_mean_leapfrogs'], 'warmup_phase %.4g' % d['warmup_phase']['value'])" | tee -a $O/ab.txt
}
run 1 "--warmup-draws"
run 0 "--warmup-draws"
run 1 "--seed 7"
run 1 ""
run 0 ""
