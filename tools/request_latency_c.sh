#!/bin/bash
# One request at a time from C, in process (tests/c/latency_harness.c), four sizes, two repetitions; run through gpurun.
#   tools/request_latency_c.sh > profiles/rNN_request_latency_c.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; make -C tests/c > /dev/null
D=$R/gpurun_out/lat_c; mkdir -p $D
python3 - "$D" <<'PY'
import io, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from PIL import Image
from ngx_http_imgproc_amd.workloads import photo_like
for w, h in ((640, 480), (1280, 720), (1920, 1080), (3840, 2160)):
    b = io.BytesIO(); Image.fromarray(photo_like(h, w, 3)).save(b, "JPEG", quality=90, subsampling="4:2:0")
    open(os.path.join(sys.argv[1], "%dx%d.jpg" % (w, h)), "wb").write(b.getvalue())
PY
echo "# tests/c/latency_harness.c <file> 300: one request at a time from C, in process: JPEG in -> resize=224,0 -> JPEG out (quality 86);"
echo "# photo_like content, quality-90 4:2:0 files; medians and p95 in microseconds; one gpurun call (one box), two repetitions"
for run in 1 2; do
  echo "# run $run"
  for s in 640x480 1280x720 1920x1080 3840x2160; do echo -n "$s "; timeout -k 10 120 tests/c/_build/latency_harness $D/$s.jpg 300 2>/dev/null | tail -1; done
done
rm -rf $D
