#!/bin/bash
# The JPEG request stream driven by C threads (tests/c/stream_harness.c) over thread counts; run through gpurun.
#   JPEG_BATCH=files per call  JPEG_OUT=quality of JPEG answers (0 = raw thumbnails)  JPEG_AHEAD=1: begin the next batch before finishing the current one
#   tools/jpeg_stream_native.sh <requests> <threads>...
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-8192}; shift
POOL=$R/gpurun_out/jpeg_pool.bin
[ -f $POOL ] || python3 - <<PY
import struct, sys
sys.path.insert(0, "$R")
import bench
files = bench.jpeg_pool(64)
with open("$POOL", "wb") as f:
    f.write(struct.pack("<I", len(files)))
    for _, _, b in files:
        f.write(struct.pack("<I", len(b)))
        f.write(b)
PY
for T in "$@"; do
  timeout -k 10 300 $R/tests/c/_build/stream_harness $POOL $N $T ${JPEG_BATCH:-64} ${JPEG_OUT:-0} $((T * ${JPEG_BATCH:-64} * 8 > 2048 ? T * ${JPEG_BATCH:-64} * 8 : 2048)) ${JPEG_AHEAD:-0} || exit 1
done
