#!/bin/bash
# round 5: which counters of this rocprofv3 say where L2 misses go (Infinity Cache / HBM / fabric)
O=gpurun_out/r5c2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$O/avail.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -i -E "mall|hbm|dram|umc|_df_|fabric|infinity|EA0?_(RD|WR)|TCC_.*(MISS|HIT|REQ|EA)" $O/avail.txt | grep -i "name\|^\s*[A-Z_0-9]*\s" | sort -u | head -150 > $O/avail_mem.txt
wc -l $O/avail.txt $O/avail_mem.txt; head -80 $O/avail_mem.txt
