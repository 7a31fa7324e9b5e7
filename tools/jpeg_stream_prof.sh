#!/bin/bash
# kernel durations of the JPEG request stream under rocprofv3 at a given thread count (run through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-32}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_jstream_$T
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_jstream_$T -- python3 $R/bench.py --stream ${N:-1024} --threads $T --jpeg device --jpeg-batch ${JPEG_BATCH:-1} > $R/gpurun_out/prof_jstream_$T.log 2>&1
f=$(ls $R/gpurun_out/prof_jstream_$T/*/*kernel_stats.csv | head -1)
echo "threads $T"; python3 $R/tools/kstats_fmt.py $f
grep -o '"value": [0-9.]*' $R/gpurun_out/prof_jstream_$T.log
rm -f $R/gpurun_out/prof_jstream_$T/*/*kernel_trace.csv
