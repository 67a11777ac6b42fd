#!/bin/bash
# round 4: is the headline's jump (3.24e8 -> 3.62e8 with metric windows) the kernel, the box, or the 25 GB of window buffers?
O=$PWD/gpurun_out/r4r; mkdir -p $O
for v in "" "--warmup-draws" "" "--warmup-draws"; do
  timeout -s KILL 300 python bench.py --steps 10 --warmup 3 --no-other-configs --traffic none $v 2>/dev/null | tail -1 > $O/b.json
  python -c "
import json; d = json.load(open('$O/b.json')); print('[$v] headline %.4g' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'], 'leapfrogs/launch', d['roofline']['leapfrogs_per_launch'], 'warmup_phase %.4g' % d['warmup_phase']['value'])" | tee -a $O/ab.txt
done
