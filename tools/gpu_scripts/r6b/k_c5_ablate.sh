#!/bin/bash
# round 6, second session: the fused eta + link kernel with its link executed TWICE (timing build, same results): what the link costs in place
O=$PWD/gpurun_out/r6bk; mkdir -p $O
export TMPDIR=/tmp; REPO=$PWD; cd /tmp
for v in full lk_twice; do
  if [ $v = full ]; then unset DHMC_LIB_PATH; else export DHMC_LIB_PATH=$REPO/tools/experiments/_v/$v/libdhmc_amd.so; fi
  rm -rf /tmp/pk_$v; timeout -s KILL 150 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/pk_$v -o t -- python $REPO/bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$v.json 2> /tmp/pk_$v.err
  f=$(find /tmp/pk_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v" | tee -a $O/ablate.txt
  [ -n "$f" ] && python3 -c "import csv,sys; [print('%-60s calls %s avg %.1f us max %.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MaxNs'])/1e3)) for r in list(csv.DictReader(open(sys.argv[1])))[:3]]" $f | tee -a $O/ablate.txt
done
