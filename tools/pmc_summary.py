#!/usr/bin/env python3
"""Per-launch mean of PMC counters for one kernel from rocprofv3 --pmc ... --output-format csv runs:
pmc_summary.py <kernel-name-substring> <dir> [<dir> ...]  (each dir = one PMC pass)."""
import csv, glob, sys, collections
pat = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"]:
                acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for name, per in acc.items():
            vals = list(per.values())
            print(f"{name},{sum(vals)/len(vals):.6g},launches={len(vals)}")
