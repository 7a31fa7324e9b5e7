"""rocprofv3 *_kernel_trace.csv -> per queue: how much of the span its kernels cover, the gaps between consecutive kernels
of the queue (the host / launch latency between dependent launches), and a sample cycle written out kernel by kernel.
    python tools/trace_lanes.py <kernel_trace.csv> [skip_fraction]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5


def short(n):
    return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", "").replace("imp::", ""))[:28]


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows)
t0, t1 = ev[0][0], max(e for _, e, _, _ in ev)
lo = t0 + int((t1 - t0) * skip)
ev = [x for x in ev if x[0] >= lo]
span = t1 - lo
by_q = defaultdict(list)
for s, e, n, q in ev:
    by_q[q].append((s, e, n))
for q, ks in sorted(by_q.items()):
    busy = sum(e - s for s, e, _ in ks)
    gaps = sorted(ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1))
    pos = [g for g in gaps if g > 0]
    print("queue %s: %d kernels, busy %.1f %% of the span; gap to the next kernel of the queue: median %.1f us, p90 %.1f us, overlapped (negative) %d"
          % (q, len(ks), 100.0 * busy / span, (pos[len(pos) // 2] if pos else 0) / 1e3, (pos[int(len(pos) * 0.9)] if pos else 0) / 1e3, len(gaps) - len(pos)))
# one stretch of one queue, kernel by kernel
q, ks = max(by_q.items(), key=lambda kv: len(kv[1]))
mid = len(ks) // 2
print("queue %s, 40 consecutive kernels from its middle (start us, duration us, gap before us):" % q)
base = ks[mid][0]
for i in range(mid, min(mid + 40, len(ks))):
    s, e, n = ks[i]
    print("   %9.1f %8.1f %8.1f  %s" % ((s - base) / 1e3, (e - s) / 1e3, (s - ks[i - 1][1]) / 1e3, n))
