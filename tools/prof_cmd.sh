#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python tool (run through gpurun):  tools/prof_cmd.sh <tag> <script> [args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
f=$(ls $R/gpurun_out/prof_$TAG/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/${TAG}_kernel_stats.csv
python3 $R/tools/kstats_fmt.py $f 12
tail -2 $R/gpurun_out/prof_$TAG.log | grep -v rocprofv3
rm -f $R/gpurun_out/prof_$TAG/*/*kernel_trace.csv
