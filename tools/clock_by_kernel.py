#!/usr/bin/env python3
"""Per-kernel sustained clock from a rocprofv3 --pmc GRBM_GUI_ACTIVE run: GUI_ACTIVE / 8 XCDs / duration."""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "force_" not in r["Kernel_Name"]:
            continue
        dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
        acc[(r["Kernel_Name"], r["Grid_Size"])].append((float(r["Counter_Value"]) / 8 / dur / 1e9, dur))
for k, v in acc.items():
    clk = sum(x for x, _ in v) / len(v); dur = sum(d for _, d in v) / len(v)
    n = 1 << 20
    print("%-60s grid=%-9s launches=%d  dur=%8.3f ms  clock=%.3f GHz  -> %.1f cycles per wave-pair" % (k[0][:60], k[1], len(v), dur * 1e3, clk, 1024 * 64 * clk * 1e9 * dur / (float(n) * n)))
