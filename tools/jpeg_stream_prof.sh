#!/bin/bash
# kernel durations and device occupancy of the JPEG request stream under rocprofv3 at a given thread count (run through gpurun)
#   N=requests JPEG_BATCH=files per call JPEG_OUT=quality  tools/jpeg_stream_prof.sh <threads>
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-32}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_jstream_$T
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_jstream_$T -- python3 $R/bench.py --stream ${N:-1024} --threads $T --jpeg device --jpeg-batch ${JPEG_BATCH:-1} --jpeg-out ${JPEG_OUT:-0} > $R/gpurun_out/prof_jstream_$T.log 2>&1
echo "threads $T, $(grep -o '"value": [0-9.]*' $R/gpurun_out/prof_jstream_$T.log) requests/s under the profiler"
python3 $R/tools/trace_busy.py $(ls $R/gpurun_out/prof_jstream_$T/*/*kernel_trace.csv | head -1)
rm -f $R/gpurun_out/prof_jstream_$T/*/*kernel_trace.csv
