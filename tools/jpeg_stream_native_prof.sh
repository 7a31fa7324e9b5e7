#!/bin/bash
# device occupancy of the native JPEG request stream (tests/c/stream_harness.c) under rocprofv3 (run through gpurun, after
# tools/jpeg_stream_native.sh has written the pool):  JPEG_BATCH= JPEG_OUT= N=  tools/jpeg_stream_native_prof.sh <threads>
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-8}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_nstream_$T
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_nstream_$T -- $R/tests/c/_build/stream_harness $R/gpurun_out/jpeg_pool.bin ${N:-16384} $T ${JPEG_BATCH:-64} ${JPEG_OUT:-0} 2048 > $R/gpurun_out/prof_nstream_$T.log 2>&1
echo "threads $T: $(grep -o '"requests_per_s": [0-9.]*' $R/gpurun_out/prof_nstream_$T.log) under the profiler"
python3 $R/tools/trace_busy.py $(ls $R/gpurun_out/prof_nstream_$T/*/*kernel_trace.csv | head -1)
rm -f $R/gpurun_out/prof_nstream_$T/*/*kernel_trace.csv
