"""Per kernel: requests of the L2 to the fabric and their mean latency in L2 clocks (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ, same for WRREQ),
and the DRAM-destined share of the reads, from rocprofv3 --pmc counter_collection.csv files (one pass per file; ratios are taken
inside a file).  Dispatches of __amd_rocclr_copyBuffer are split into the first three (the calibration's 4 GiB copies: HBM) and the
rest (its 48 MiB copies: Infinity Cache)."""
import csv, sys, collections
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    ndisp = collections.defaultdict(set)
    copies = {}
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"][:60]
        if "copyBuffer" in name:
            did = int(row["Dispatch_Id"])
            copies.setdefault(did, None)
    order = {d: i for i, d in enumerate(sorted(copies))}
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"][:60]
        if "copyBuffer" in name:
            name += " [4 GiB: HBM]" if order[int(row["Dispatch_Id"])] < 3 else " [48 MiB: Infinity Cache]"
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
        ndisp[name].add(row["Dispatch_Id"])
    counters = sorted({c for d in acc.values() for c in d})
    print(f"-- {counters}")
    big = sorted(acc.items(), key=lambda kv: -max(kv[1].values()))[:6]
    for name, d in big:
        out = f"   {name:86s} dispatches {len(ndisp[name]):4d}"
        for req, lvl, what in (("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_LEVEL_sum", "reads"), ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_LEVEL_sum", "writes")):
            if req in d and lvl in d and d[req] > 0:
                out += f"  {what} {d[req]:.4g}, mean latency {d[lvl] / d[req]:7.1f} clk"
        if "TCC_EA0_RDREQ_DRAM_sum" in d and d.get("TCC_EA0_RDREQ_sum", 0) > 0:
            out += f"  DRAM-destined reads {d['TCC_EA0_RDREQ_DRAM_sum'] / d['TCC_EA0_RDREQ_sum']:.3f}"
        print(out)
