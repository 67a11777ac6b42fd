#!/bin/bash
# round 6: counters of the fused eta + link kernel (config 5): how busy are the matrix cores, where does the wave wait
O=gpurun_out/r6d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--config 5 --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --output-format csv --kernel-include-regex "logistic_eta_link" --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -d /tmp/p1 -o p1 -- python $GRAFT_REPO_ROOT/bench.py $A > /dev/null 2> $GRAFT_REPO_ROOT/$O/err1.txt
timeout 600 rocprofv3 --output-format csv --kernel-include-regex "logistic_eta_link" --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64 -d /tmp/p2 -o p2 -- python $GRAFT_REPO_ROOT/bench.py $A > /dev/null 2> $GRAFT_REPO_ROOT/$O/err2.txt
cd $GRAFT_REPO_ROOT
python tools/summarize_prof.py /tmp 2>/dev/null | grep -A12 "logistic_eta_link" | head -60 | tee $O/summary.txt
for f in $(find /tmp/p1 /tmp/p2 -name "*counter_collection.csv"); do echo $f; python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(list)
for r in rows:
    acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    v2=sorted(v)
    print('  %-32s n=%d median=%.4g max=%.4g'%(k,len(v),v2[len(v2)//2],v2[-1]))
PY
done | tee -a $O/summary.txt
