#!/bin/bash
# round 6: where do the role-split kernel's waves wait?  (clock counters around the barriers, printed by block 1000 of the first launches)
O=gpurun_out/r6u; mkdir -p $O
DHMC_LOGISTIC_ROLES=1 timeout -s KILL 400 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > $O/out.txt 2> $O/err.txt
grep "roles\]" $O/out.txt $O/err.txt | head -12
tail -1 $O/out.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 roles: %.4g' % d['value'])"
