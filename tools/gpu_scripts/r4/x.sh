#!/bin/bash
# round 4: the per-draw kernel's time under the compiler's scheduling strategies (variants of the StdNormalT family object, same box)
O=$PWD/gpurun_out/r4x; mkdir -p $O
run() {
  DHMC_LIB_PATH=$1 DHMC_SCHED=0 timeout -s KILL 120 python bench.py --steps 4 --warmup 1 --no-other-configs --traffic none --no-cpu-baseline 2>/dev/null | tail -1 > $O/b.json
  python -c "
import json; d = json.load(open('$O/b.json')); print('[$2] headline %.4g' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'], 'slowest/mean %.3f' % d['tree']['slowest_chain_over_mean_leapfrogs'])" | tee -a $O/ab.txt
}
V=$PWD/tools/experiments/_v
run "$V/head/libdhmc_amd.so" head
run "" tree
for n in "$@"; do run "$V/$n/libdhmc_amd.so" $n; done
