#!/usr/bin/env python3
"""Per-launch mean of PMC counters for one kernel from rocprofv3 --pmc ... --output-format csv runs:

    pmc_summary.py <kernel-name-substring> <dir> [<dir> ...]        (each dir = one PMC pass)

Normalisation (round 4; the round-3 file averaged SIX dispatches of which one was the 64-board calibration launch of the weight set,
so its per-launch values were 5/6 of the real ones): a counter's value for one dispatch is the SUM over the rows rocprofv3 writes for
that dispatch (one row per XCD / shader engine instance); only dispatches of the LARGEST grid seen for the kernel are kept (the launch
size under measurement -- warm-up / calibration launches of the same kernel have smaller grids or fewer workgroups); the value printed
is the MEAN over those dispatches.  Output rows:  counter, mean per launch, launches=<kept>/<seen>, grid=<workgroup count x size>."""
import csv, glob, sys, collections
pat = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        grid = {}
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"]:
                acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
                grid[row["Dispatch_Id"]] = int(row.get("Grid_Size", 0) or 0)
        gmax = max(grid.values()) if grid else 0
        for name, per in acc.items():
            vals = [v for k, v in per.items() if grid[k] == gmax]
            print(f"{name},{sum(vals)/len(vals):.6g},launches={len(vals)}/{len(per)},grid_threads={gmax}")
