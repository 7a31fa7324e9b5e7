"""stdin: the stderr of a run with IMPGPU_JPEG_TRACE=1 -> mean of each host phase per decode call, and the spread of the waits"""
import re, sys
rows = []
for line in sys.stdin:
    m = re.search(r"jpeg x(\d+) \((\d+) live\): headers (\d+) \S+ (\d+) jobs (\d+) enqueue (\d+) wait (\d+) us", line)
    if m:
        rows.append([int(x) for x in m.groups()])
if not rows:
    sys.exit("no trace lines")
n = len(rows)
names = ("files", "live", "headers", "unstuff", "jobs", "enqueue", "wait")
print("%d calls; mean per call: " % n + ", ".join("%s %.0f" % (names[i], sum(r[i] for r in rows) / n) for i in range(len(names))))
w = sorted(r[6] for r in rows)
print("wait us: min %d  median %d  p90 %d  max %d" % (w[0], w[n // 2], w[int(n * 0.9)], w[-1]))
