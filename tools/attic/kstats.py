"""Print rocprofv3 kernel statistics (CSV) for kernels whose name matches any of the given substrings.

python tools/kstats.py <kernel_stats.csv> [substring ...]
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:]
for r in rows:
    if not pats or any(p in r["Name"] for p in pats):
        print("%9.1f us avg  %6d calls  %8.2f ms total  %s" % (float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                                               r["Name"][:70]))
