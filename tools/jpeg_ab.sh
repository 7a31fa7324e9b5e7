#!/bin/bash
# A/B of library builds on ONE box: for every libimpgpu*.so given (default: all in the package directory) the average
# duration of the JPEG kernels over the 64-file batch probe (rocprofv3 --kernel-trace --stats).  Run through gpurun.
R=${GRAFT_REPO_ROOT:-/root/repo}
LIBS=${@:-$(ls $R/ngx_http_imgproc_amd/libimpgpu*.so)}
for L in $LIBS; do
  IMPGPU_LIB=$L $R/tools/jpeg_prof_r04.sh ab > /dev/null 2>&1
  echo "== $(basename $L) ${AB_NOTE}"
  python3 - "$R/gpurun_out/ab_jpeg_batch64_kernel_stats.csv" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    if "jpeg" in n:
        import re
        m = re.search(r"(k_jpeg_\w+(?:<[^>]*>)?)", n)
        short = m.group(1) if m else n[:28]
        print("   %-28s calls %3s avg %8.1f us  min %8.1f" % (short, row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3))
PY
  grep -E "^batch" $R/gpurun_out/ab_jpeg_batch64.log | tail -1
done
