"""rocprofv3 kernel_stats.csv -> short table: name (trimmed), calls, total ms, avg us, %"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("imp::", "")
    name = re.sub(r"\(.*", "", name)
    print("%-44s calls %6s total %9.2f ms avg %9.1f us %6s%%" % (name[:44], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
