"""Summarise rocprofv3 counter_collection CSVs: mean per dispatch of every counter for the selected kernel."""
import csv, glob, sys, collections
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
acc = collections.defaultdict(list)
for f in sorted(glob.glob(d + "/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        acc[(row.get("Kernel_Name", "")[:40], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:42s} {c:34s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
