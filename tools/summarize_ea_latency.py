"""Per kernel name (and grid size): requests of the L2 to the fabric, their mean latency in L2 clocks, and the DRAM-destined share,
from a rocprofv3 --pmc counter_collection.csv with TCC_EA0_{RD,WR}REQ[_LEVEL]_sum (and optionally TCC_EA0_RDREQ_DRAM_sum)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for path in sys.argv[1:]:
    seen = set()
    for row in csv.DictReader(open(path)):
        key = (row["Kernel_Name"][:70], row.get("Grid_Size", ""))
        acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
        did = (key, row.get("Dispatch_Id"))
        if did not in seen:
            seen.add(did); cnt[key] += 1
for key, d in sorted(acc.items(), key=lambda kv: -kv[1].get("TCC_EA0_RDREQ_sum", 0.0))[:12]:
    rd, rl = d.get("TCC_EA0_RDREQ_sum", 0.0), d.get("TCC_EA0_RDREQ_LEVEL_sum", 0.0)
    wr, wl = d.get("TCC_EA0_WRREQ_sum", 0.0), d.get("TCC_EA0_WRREQ_LEVEL_sum", 0.0)
    dram = d.get("TCC_EA0_RDREQ_DRAM_sum")
    print(f"{key[0]:70s} grid {key[1]:>10s} dispatches {cnt[key]:5d}  RDREQ {rd:.4g} mean latency {rl / rd if rd else 0:8.1f} clk"
          f"{'' if dram is None else f' (DRAM-destined {dram / rd if rd else 0:.3f})'}   WRREQ {wr:.4g} mean latency {wl / wr if wr else 0:8.1f} clk")
