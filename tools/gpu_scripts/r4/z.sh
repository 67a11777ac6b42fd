#!/bin/bash
# round 4: the fast (committed) and the slow (segment-scheduler patch, scheduler off) builds of the per-draw kernel under the same counters
O=$PWD/gpurun_out/r4z; mkdir -p $O
export TMPDIR=/tmp; REPO=$PWD; cd /tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --traffic none --transitions 100"
for lib in head sched; do
  export DHMC_LIB_PATH=$REPO/tools/experiments/_v/$lib/libdhmc_amd.so DHMC_SCHED=0
  i=0
  for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM" \
             "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_IFETCH SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    i=$((i+1)); rm -rf /tmp/z_$lib_$i
    timeout -s KILL 120 rocprofv3 --output-format csv --pmc $pmc -d /tmp/z_${lib}_$i -o p -- python $REPO/bench.py $ARGS > /dev/null 2> /tmp/z_${lib}_$i.err || tail -2 /tmp/z_${lib}_$i.err
    f=$(find /tmp/z_${lib}_$i -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python - "$f" "$lib" >> $O/counters.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if 'nuts_run_kernel' in r.get('Kernel_Name', ''):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(sys.argv[2], k, 'n=%d' % len(v), 'last=%.6g' % v[-1])
PY
  done
done
cat $O/counters.txt
