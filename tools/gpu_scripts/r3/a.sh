#!/bin/bash
# round 3, GPU call A: the two-waves-per-chain kernel variants (tools/experiments/_v/*): bits against the one-wave kernel,
# kernel time against the number of chains, then the bench line of each.
set -u
O=gpurun_out/r3a; mkdir -p $O
export PYTHONPATH=tests
for so in tools/experiments/_v/*/libdhmc_amd.so; do
    n=$(basename $(dirname $so))
    echo "== $n" | tee -a $O/log.txt
    DHMC_LIB_PATH=$PWD/$so timeout -s KILL 240 python tools/experiments/w2_first.py >> $O/log.txt 2>&1
    echo "rc=$?" >> $O/log.txt
done
grep -v "amdgpu.ids" $O/log.txt | tail -80
echo "== bench lines (W2 on)"
for so in tools/experiments/_v/*/libdhmc_amd.so; do
    n=$(basename $(dirname $so))
    DHMC_W2=1 DHMC_LIB_PATH=$PWD/$so timeout -s KILL 300 python bench.py --steps 3 --warmup 1 --transitions 100 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$n.json
    python - <<PY
import json
try:
    d = json.load(open("$O/bench_$n.json")); print("$n", "W2 %.4g" % d["value"], "frac %.4f" % d["roofline"]["frac"], "warm %.4g" % d["warmup_phase"]["value"])
except Exception as e:
    print("$n FAILED", e)
PY
done
DHMC_W2=0 timeout -s KILL 300 python bench.py --steps 3 --warmup 1 --transitions 100 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_base.json
python -c "
import json; d = json.load(open('$O/bench_base.json')); print('base one-wave %.4g' % d['value'], 'frac %.4f' % d['roofline']['frac'])"
