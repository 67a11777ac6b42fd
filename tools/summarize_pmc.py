"""Mean of each PMC counter per kernel name from a rocprofv3 --pmc counter_collection.csv."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} n={len(v):6d} mean={sum(v)/len(v):.6g}")
