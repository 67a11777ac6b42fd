#!/bin/bash
# round 6, second session: strict alternation of the two half-batches again — now with the product kernel that keeps its rate beside an HBM-bound
# neighbour (two k-tiles in flight) and with K3b's residency capped by an LDS pad so that the product's workgroups find room
O=gpurun_out/r6bp; mkdir -p $O
for cfg in "alternate=0" "alternate=1" "alternate=1,k3b_pad=20000" "alternate=1,k3b_pad=30000" "alternate=1,k3b_pad=44000" "alternate=0,k3b_pad=30000" "alternate=0" "alternate=1,k3b_pad=30000"; do
  r=$(DHMC_DENSE="$cfg" timeout -s KILL 300 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g leapfrog-steps/s, %.1f ms/step' % (d['value'], d['ms_per_step']))")
  echo "$cfg: $r" | tee -a $O/alt.txt
done
