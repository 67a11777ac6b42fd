#!/bin/bash
# round 4: (1) the two funnel D = 1000 fuzz cases that differ from the oracle, with the first differing entry printed;
# (2) shader clock and socket power sampled at 10 Hz from sysfs beside a 6-second bench run of the round-3 library and of the
# working tree (the v2 build issues 17 % fewer vector instructions and 8 % fewer wave cycles, yet is not faster).
O=gpurun_out/r4e; mkdir -p $O
FUZZ_VERBOSE=2 timeout -s KILL 120 python tools/fuzz_parity.py 20 12345 2> $O/fuzz.err | tail -5; grep -v "^case" $O/fuzz.err | head -40
ls /sys/class/drm/ | head; CARD=$(ls -d /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | head -1); echo "sclk file: $CARD"
HW=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -1); echo "hwmon: $HW"; ls $HW 2>/dev/null | tr '\n' ' '
smi() { ( while true; do echo "$(grep '\*' $CARD 2>/dev/null | tr -d '\n') $(cat $HW/power1_average 2>/dev/null || cat $HW/power1_input 2>/dev/null) $(cat $HW/freq1_input 2>/dev/null)"; sleep 0.1; done ) > $1 & echo $!; }
run() {
  P=$(smi $O/clk_$1.txt)
  ( cd $2 && timeout -s KILL 200 python bench.py --steps 30 --warmup 2 --transitions 1000 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/bench_$1.json
  kill $P
  python -c "
import json; d = json.load(open('$O/bench_$1.json')); print('$1 %.4g' % d['value'], 'ms/step %.2f' % d['ms_per_step'])"
  sort $O/clk_$1.txt | uniq -c | sort -rn | head -6
}
run v1 tools/experiments/_ab/v1
run v2 .
