#!/bin/bash
# rocprofv3 kernel trace of the JPEG decode probe (run through gpurun): per-kernel durations of one decode.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_jpeg
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_jpeg -- python3 $R/tools/jpeg_probe.py 20 > $R/gpurun_out/prof_jpeg.log 2>&1
f=$(ls $R/gpurun_out/prof_jpeg/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/jpeg_kernel_stats.csv
head -12 $R/gpurun_out/jpeg_kernel_stats.csv | cut -c1-200
