#!/bin/bash
# round 4: same box — the library before the segment scheduler (commit ee0b16f: tools/experiments/_v/head) against the working tree
O=$PWD/gpurun_out/r4w; mkdir -p $O
run() {
  DHMC_LIB_PATH=$1 DHMC_SCHED=$2 timeout -s KILL 300 python bench.py --steps 6 --warmup 2 --no-other-configs --traffic none --no-cpu-baseline $3 2>/dev/null | tail -1 > $O/b.json
  python -c "
import json; d = json.load(open('$O/b.json')); print('[lib=$1 sched=$2 $3] headline %.4g' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'], 'slowest/mean %.3f' % d['tree']['slowest_chain_over_mean_leapfrogs'], 'warmup_phase %.4g' % d['warmup_phase']['value'])" | tee -a $O/ab.txt
}
H=$PWD/tools/experiments/_v/head/libdhmc_amd.so
run "$H" 1 ""
run "" 1 ""
run "$H" 1 "--warmup-draws"
run "" 1 "--warmup-draws"
run "" 0 "--warmup-draws"
run "$H" 1 ""
run "" 1 ""
