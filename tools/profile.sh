#!/bin/bash
# rocprofv3 passes for the bench kernel; run on the GPU box via gpurun.  Usage: tools/profile.sh <tag> [bench args]
# Pass 1: --kernel-trace --stats (per-kernel time).  Passes 2..4: PMC counters, each in its own run
# (never combined with other trace domains; FETCH_SIZE and WRITE_SIZE cannot share a pass).
TAG=${1:-r01}; shift
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --traffic none --transitions 100 $@"
KEEP=$PWD/gpurun_out/prof_$TAG
OUT=/tmp/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $KEEP
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout -s KILL 200 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/pmc1 -o pmc1 -- python $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc1.err
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o pmc2 -- python $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc2.err
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU -d $OUT/pmc5 -o pmc5 -- python $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc5.err
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE -d $OUT/pmc6 -o pmc6 -- python $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc6.err
timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc3.err
timeout -s KILL 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- python $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc4.err
cd $REPO
find $OUT -name "*.csv" > $KEEP/files.txt; python tools/summarize_prof.py $OUT > $KEEP/summary.txt 2>&1
cp $OUT/bench_trace.json $OUT/traffic.json $KEEP/ 2>/dev/null
for f in $(find $OUT -name '*kernel_stats.csv' | head -2) $(find $OUT -name '*agent_info.csv' | head -2); do cp $f $KEEP/; done
for e in $OUT/*.err; do echo "--- $e"; tail -3 $e; done
cat $KEEP/summary.txt
