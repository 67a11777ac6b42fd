#!/bin/bash
# round 6: (1) config 5 with the fused eta + link kernel, second version (64-k stages, the link in phases); (2) headline kernel with the cold
# paths' lane index made opaque (no loop-invariant registers for once-per-transition code); (3) packed kernel at two waves per SIMD
O=gpurun_out/r6c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_engines.py tests/test_gpu_tolerance.py tests/test_gpu_external.py tests/test_gpu_detmath.py -m gpu -q -x -k "logistic or config5 or detmath" 2>&1 | tail -5 | tee $O/pytest_c5.log
timeout 600 python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline 2>$O/err_c5.txt | tail -1 > $O/bench_c5.json; python -c "
import json; d=json.load(open('$O/bench_c5.json')); print('c5: %.4g leapfrog-steps/s, %.1f ms/step, frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/c5prof -o c5 -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/err_c5prof.txt
cd $GRAFT_REPO_ROOT; f=$(find /tmp/c5prof -name '*kernel_stats.csv' | head -1); cp $f $O/c5_kernel_stats.csv; head -6 $O/c5_kernel_stats.csv | cut -c1-150
for v in "cold DHMC_NOTHING=1" "base6 DHMC_LIB_PATH=tools/experiments/_v/base6/libdhmc_amd.so" "cold_again DHMC_NOTHING=1" "base6_again DHMC_LIB_PATH=tools/experiments/_v/base6/libdhmc_amd.so"; do
  set -- $v
  r=$(env $2 timeout -s KILL 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs --traffic none 2>$O/err_$1.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g leapfrog-steps/s' % d['value'], 'ms %.2f' % d['ms_per_step'], 'frac %.4f' % d['roofline']['frac'])")
  echo "$1: $r" | tee -a $O/variants.txt
done
for v in "default DHMC_NOTHING=1" "occ2 DHMC_PK_CPL=2,DHMC_PK_MAX_WAVES=2048,DHMC_PK_LDS_LEVELS=3" "cpl2_occ1 DHMC_PK_CPL=2,DHMC_PK_MAX_WAVES=1024" "occ2_l2 DHMC_PK_CPL=2,DHMC_PK_MAX_WAVES=2048,DHMC_PK_LDS_LEVELS=2"; do
  set -- $v
  r=$(env $(echo $2 | tr ',' ' ') DHMC_DEBUG_ORDER=1 timeout -s KILL 300 python bench.py --config 4 --chains 32768 --steps 3 --warmup 1 --no-cpu-baseline --config-n 1000 2>$O/err_c4_$1.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g (steps of 20)' % d['value'], '%.4g at N=1000, %.0f ms' % (d['at_config_n']['value'], d['at_config_n']['ms_per_step']))")
  echo "c4_32768 $1: $r" | tee -a $O/occ2.txt
  grep -c "end game" $O/err_c4_$1.txt
done
