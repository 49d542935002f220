"""Top rows of a rocprofv3 kernel_stats.csv: calls, avg / min / max in us, share.   python tools/kstats.py <csv | dir> [rows] [name filter]"""
import csv, glob, os, sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 10
pat = sys.argv[3] if len(sys.argv) > 3 else ""
n = 0
for r in csv.DictReader(open(path)):
    if pat and pat not in r["Name"]:
        continue
    print(f'{r["Name"][:56]:56s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:8.1f} min {float(r["MinNs"]) / 1e3:8.1f} max {float(r["MaxNs"]) / 1e3:8.1f} us {float(r["Percentage"]):6.2f} %')
    n += 1
    if n >= rows:
        break
