"""rocprofv3 *_kernel_trace.csv -> how busy the device was: union of the kernels' intervals over the span between the first
start and the last end, mean number of kernels in flight, time with 0 / 1 / 2+ kernels, and the totals per kernel name.
    python tools/trace_busy.py <kernel_trace.csv> [skip_fraction]   (skip_fraction: ignore that share of the span at its start: warm-up)"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, e, r["Kernel_Name"], r.get("Queue_Id", "?")))
t0 = min(s for s, _, _, _ in ev)
t1 = max(e for _, e, _, _ in ev)
lo = t0 + int((t1 - t0) * skip)
ev = [(max(s, lo), e, n, q) for s, e, n, q in ev if e > lo]
pts = []
for s, e, _, _ in ev:
    pts.append((s, 1))
    pts.append((e, -1))
pts.sort()
depth = 0
last = lo
hist = defaultdict(int)
for t, d in pts:
    hist[min(depth, 3)] += t - last
    last = t
    depth += d
span = t1 - lo
print("span %.1f ms; kernels in flight: none %.1f %%, one %.1f %%, two %.1f %%, three or more %.1f %%; %d queues" %
      (span / 1e6, 100.0 * hist[0] / span, 100.0 * hist[1] / span, 100.0 * hist[2] / span, 100.0 * hist[3] / span, len(set(q for _, _, _, q in ev))))
tot = defaultdict(lambda: [0, 0])
for s, e, n, _ in ev:
    n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", "").replace("imp::", ""))[:40]
    tot[n][0] += e - s
    tot[n][1] += 1
for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:12]:
    print("   %-40s calls %6d total %8.2f ms = %5.1f %% of the span, avg %8.1f us" % (n, c, d / 1e6, 100.0 * d / span, d / c / 1e3))
