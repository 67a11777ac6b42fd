#!/bin/bash
# one PMC pass: outstanding-op level counters (average latency = LEVEL / INSTS)
OUT=/tmp/prof_lat; rm -rf $OUT; mkdir -p $OUT gpurun_out
export TMPDIR=/tmp; REPO=$PWD; cd /tmp
rocprofv3 --output-format csv --pmc SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES -d $OUT/a -o a -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/a.err
rocprofv3 --output-format csv --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_ANY SQ_BUSY_CYCLES -d $OUT/b -o b -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/b.err
cd $REPO; python tools/summarize_prof.py $OUT | grep -v "^  void\|kernel stats" 
tail -2 $OUT/a.err
