#!/bin/bash
# round 5: the packed kernel's clocks by region (static marks) and its instruction counters, stuck chains (one wave, and 4096 chains)
O=gpurun_out/r5c; mkdir -p $O
REPO=$PWD
( export DHMC_LIB_PATH=$PWD/tools/experiments/_phase/libdhmc_amd_FunnelT.so
  echo "== stuck chains, one wave (8 chains x 10 transitions of 1023 leapfrogs)"; PH_STUCK=1 timeout 300 python tools/experiments/packed_phase_timing.py 8 10
  echo "== stuck chains, 4096 chains x 4"; PH_STUCK=1 timeout 300 python tools/experiments/packed_phase_timing.py 4096 4
  echo "== adapted chains, one wave"; timeout 300 python tools/experiments/packed_phase_timing.py 8 200 ) > $O/phases.txt 2>&1
cat $O/phases.txt
export TMPDIR=/tmp; cd /tmp
for sc in "8 10" "4096 4"; do
  tag=$(echo $sc | tr ' ' _)
  for pk in 1 0; do
    OUT=/tmp/pmc_${tag}_$pk; rm -rf $OUT; mkdir -p $OUT
    PH_STUCK=1 DHMC_PACKED=$pk rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/a -o a -- python $REPO/tools/experiments/packed_probe.py $sc > $OUT/a.out 2> $OUT/a.err
    PH_STUCK=1 DHMC_PACKED=$pk rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $OUT/b -o b -- python $REPO/tools/experiments/packed_probe.py $sc > $OUT/b.out 2> $OUT/b.err
    PH_STUCK=1 DHMC_PACKED=$pk rocprofv3 --output-format csv --pmc SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 -d $OUT/c -o c -- python $REPO/tools/experiments/packed_probe.py $sc > $OUT/c.out 2> $OUT/c.err
    echo "== scenario $sc packed=$pk"; cat $OUT/a.out | tail -1
    for f in $(find $OUT -name "*counter_collection.csv"); do python $REPO/tools/summarize_pmc.py $f nuts_run; done
  done
done > $REPO/$O/counters.txt 2>&1
cd $REPO; cat $O/counters.txt
