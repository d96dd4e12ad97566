#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSVs (one directory per pass) for the rollout kernel.
    python tools/pmc_summary.py <dir> [kernel-name substring, default l2a_rollout] [directory prefix, default pmc_]"""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "l2a_rollout"
PREFIX = sys.argv[3] if len(sys.argv) > 3 else "pmc_"
agg = defaultdict(list)
for path in glob.glob(os.path.join(out, PREFIX + "*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if KERNEL not in row.get("Kernel_Name", ""):
                continue
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("counter, launches, mean per launch")
for k in sorted(agg):
    v = agg[k]
    print("%-28s %4d  %.6g" % (k, len(v), sum(v) / len(v)))
g = {k: sum(v) / len(v) for k, v in agg.items()}
if "SQ_WAVE_CYCLES" in g:
    wc = g["SQ_WAVE_CYCLES"]
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in g:
            print("%s / SQ_WAVE_CYCLES = %.3f" % (k, g[k] / wc))
if "TCC_HIT_sum" in g and "TCC_MISS_sum" in g:
    print("L2 hit rate = %.4f" % (g["TCC_HIT_sum"] / (g["TCC_HIT_sum"] + g["TCC_MISS_sum"])))
if "FETCH_SIZE" in g:
    print("FETCH_SIZE per launch = %.1f KB (x2 correction for wide coalesced reads on gfx950 -> %.1f KB)"
          % (g["FETCH_SIZE"], 2 * g["FETCH_SIZE"]))
if "WRITE_SIZE" in g:
    print("WRITE_SIZE per launch = %.1f KB" % g["WRITE_SIZE"])
if "SQ_VALU_MFMA_BUSY_CYCLES" in g and "GRBM_GUI_ACTIVE" in g:
    # (GRBM_GUI_ACTIVE comes back summed over the 8 XCDs: / 8 = the launch's clocks; 1024 SIMDs)
    print("matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = %.3f"
          % (g["SQ_VALU_MFMA_BUSY_CYCLES"] / (g["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)))
if "SQ_LDS_BANK_CONFLICT" in g and "SQ_LDS_IDX_ACTIVE" in g and g["SQ_LDS_IDX_ACTIVE"] > 0:
    print("SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.4f" % (g["SQ_LDS_BANK_CONFLICT"] / g["SQ_LDS_IDX_ACTIVE"]))
