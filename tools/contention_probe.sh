#!/bin/bash
# Do concurrent request chains slow each other's KERNELS, or only the gaps between them?  One 640x480 file over and over
# (every launch the same work), T broker lanes with T workers (one request per lane at a time): per-kernel average durations.
#   tools/contention_probe.sh "1 2 4 8"
R=${GRAFT_REPO_ROOT:-/root/repo}
export POOL=$R/gpurun_out/one_file_pool.bin
python3 - <<PY
import sys, io
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
import worker_scaling as w
from PIL import Image
from ngx_http_imgproc_amd.workloads import photo_like
b = io.BytesIO(); Image.fromarray(photo_like(480, 640, seed=5)).save(b, "JPEG", quality=90, subsampling="4:2:0")
w.write_pool("$POOL", [b.getvalue()] * 8)
print("file: %d bytes" % len(b.getvalue()))
PY
for T in ${1:-1 4 8}; do
  echo "=== $T lanes, $T workers"
  SECS=1 timeout -k 10 120 bash $R/tools/r05_broker_prof.sh $T $T 2>&1 | grep -v "^queue [0-9]*:\|Killed" | head -16 || exit 1
done
rm -f $POOL
