#!/usr/bin/env python3
"""Per-kernel shader clock from a rocprofv3 counter pass: GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs) / 8 over the dispatch's own
End - Start timestamps.  An independent check of the in-kernel clock stamps (profiles/r06_clock.txt).
    python tools/pmc_clock.py gpurun_out/final/pmc_all/GRBM_GUI_ACTIVE_counter_collection.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])      # ns
    name = r["Kernel_Name"].split("(")[0][-40:]
    if r["Kernel_Name"].startswith("void dfm::k_edge_msg<1, 1, 0>"):
        name += " full" if dur > 1.8e6 else " lig-only"
    agg[name].append((float(r["Counter_Value"]), dur))
print("kernel                                      launches   busy cycles / 8 XCDs   duration us      MHz")
for k, v in sorted(agg.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    c = sum(x for x, _ in v) / len(v) / 8
    d = sum(y for _, y in v) / len(v)
    print(f"{k:45s} {len(v):5d} {c:18.0f} {d / 1e3:14.1f} {c / d * 1e3:9.0f}" + ("   (too short for this estimate)" if d < 5e4 else ""))
