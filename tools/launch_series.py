"""Per-launch durations of one kernel from a rocprofv3 kernel_trace.csv: the time series, not just the average.
    python tools/launch_series.py <kernel_trace.csv> <kernel name substring> [bucket]
Prints count, min / median / mean / max, the first launches one by one, then the mean of every `bucket` consecutive launches
(default 100) with the idle gap before each -- a clock or power effect shows as a drift of the bucket means, a probe
artefact as outliers among the first few."""
import csv
import statistics
import sys

path, needle = sys.argv[1], sys.argv[2]
bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        if needle in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
d = [(e - s) / 1e3 for s, e in rows]
if not d:
    raise SystemExit("no launch of %r" % needle)
print("%s: %d launches, us: min %.1f median %.1f mean %.1f max %.1f stdev %.1f" % (needle, len(d), min(d), statistics.median(d), statistics.mean(d), max(d),
                                                                               statistics.pstdev(d)))
print("first 12:", " ".join("%.0f" % v for v in d[:12]))
t0 = rows[0][0]
for i in range(0, len(d), bucket):
    part = d[i:i + bucket]
    gap = (rows[i][0] - rows[i - 1][1]) / 1e3 if i else 0.0
    print("launches %5d..%5d  t=%8.1f ms  mean %8.1f us  min %8.1f  max %8.1f  (gap before: %.0f us)" % (i, i + len(part) - 1, (rows[i][0] - t0) / 1e6, statistics.mean(part),
                                                                                             min(part), max(part), gap))
