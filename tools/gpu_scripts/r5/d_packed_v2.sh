#!/bin/bash
# round 5: packed engine v2 (2 or 4 coordinates per lane, cooperative Exp(1) refill, typed LDS / HBM paths, fused logaddexp pair)
O=gpurun_out/r5d; mkdir -p $O
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_packed.py -x -q 2>&1 | tail -8 > $O/packed.log; cat $O/packed.log
( export DHMC_LIB_PATH=$PWD/tools/experiments/_phase/libdhmc_amd_FunnelT.so
  for cpl in 2 4; do echo "== stuck chains, one wave, cpl $cpl"; DHMC_PK_CPL=$cpl PH_STUCK=1 timeout 300 python tools/experiments/packed_phase_timing.py 8 10; done ) > $O/phases.txt 2>&1
cat $O/phases.txt
for cpl in 2 4; do
  DHMC_PK_CPL=$cpl timeout 600 python bench.py --config 4 --transitions 1000 --steps 1 --warmup 0 2>/dev/null | tail -1 > $O/c4_T1000_cpl$cpl.json
  DHMC_PK_CPL=$cpl timeout 300 python bench.py --config 4 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/c4_T20_cpl$cpl.json
  DHMC_PK_CPL=$cpl timeout 600 python bench.py --config 4 --chains 32768 --transitions 1000 --steps 1 --warmup 0 2>/dev/null | tail -1 > $O/c4_32768_T1000_cpl$cpl.json
done
DHMC_PACKED=0 timeout 600 python bench.py --config 4 --chains 32768 --transitions 1000 --steps 1 --warmup 0 2>/dev/null | tail -1 > $O/c4_32768_T1000_wave.json
for f in $O/c4_*.json; do python -c "
import json,sys; d = json.load(open('$f')); print('$f', '%.4g' % d['value'], 'ms/step %.1f' % d['ms_per_step'])"; done
export TMPDIR=/tmp; cd /tmp
for cpl in 2 4; do
    OUT=/tmp/pmc_$cpl; rm -rf $OUT; mkdir -p $OUT
    DHMC_PK_CPL=$cpl PH_STUCK=1 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/a -o a -- python $REPO/tools/experiments/packed_probe.py 8 10 > $OUT/a.out 2> $OUT/a.err
    DHMC_PK_CPL=$cpl PH_STUCK=1 rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 -d $OUT/b -o b -- python $REPO/tools/experiments/packed_probe.py 8 10 > $OUT/b.out 2> $OUT/b.err
    echo "== stuck, 8 chains x 10, cpl $cpl"; tail -1 $OUT/a.out
    for f in $(find $OUT -name "*counter_collection.csv"); do python $REPO/tools/summarize_pmc.py $f nuts_run; done
done > $REPO/$O/counters.txt 2>&1
cd $REPO; cat $O/counters.txt
