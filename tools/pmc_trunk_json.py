#!/usr/bin/env python3
"""profiles/r04_pmc_trunk.json from a round's PMC summary (tools/profile_round4.sh) and the bench command's kernel statistics: the
matrix pipes' busy share of the default trunk, the number bench.py quotes as roofline.mfma_busy_pmc (read from the file, never typed in).

    pmc_trunk_json.py <pmc_summary.csv> <bench_kernel_stats.csv> <out.json>

busy share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1,024 SIMDs), both per launch from their own --pmc passes; for the
480-board launch also against the UN-instrumented launch time of the bench profile (kernel-trace only) x the 2.4 GHz clock."""
import csv, json, sys
pmc, stats, out = sys.argv[1:4]
val = {}
for row in csv.reader(open(pmc)):
    if len(row) >= 3 and not row[0].startswith("#"):
        val[(row[0], row[1])] = float(row[2])
res = {"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), separate rocprofv3 --pmc passes", "per_launch": {}}
for tag, boards in (("trunk_B65536", 65536), ("trunk_B480", 480)):
    busy, gui = val.get((tag, "SQ_VALU_MFMA_BUSY_CYCLES")), val.get((tag, "GRBM_GUI_ACTIVE"))
    if busy and gui:
        res["per_launch"][str(boards)] = {"mfma_busy": busy / (gui / 8 * 1024), "mfma_insts_per_board": val.get((tag, "SQ_INSTS_MFMA"), 0) / boards,
                                          "valu_insts_per_board": val.get((tag, "SQ_INSTS_VALU"), 0) / boards,
                                          "lds_bank_conflict_share": (val.get((tag, "SQ_LDS_BANK_CONFLICT"), 0) / val[(tag, "SQ_LDS_IDX_ACTIVE")]) if val.get((tag, "SQ_LDS_IDX_ACTIVE")) else None}
try:
    for row in csv.DictReader(open(stats)):
        if "gcn_trunk_boards_mm_kernel" in row["Name"]:
            ns = float(row["AverageNs"])
            busy = val.get(("trunk_B480", "SQ_VALU_MFMA_BUSY_CYCLES"))
            res["bench_trunk_avg_launch_ns"] = ns
            if busy:
                res["per_launch"].setdefault("480", {})["mfma_busy_vs_uninstrumented_launch"] = busy / (ns * 2.4 * 1024)
            break
except FileNotFoundError:
    pass
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
