#!/bin/bash
# rocprofv3 kernel trace of the batched JPEG decode probe (run through gpurun): per-kernel durations of a 64-file launch
# (tools/jpeg_batch_probe.py 64) and of the lone-file probe.  $1 = tag for the output names.  IMPGPU_JPEG_WHOLE=1: the
# batch as ONE launch per kernel (the decode call otherwise cuts it in two to prepare the second half during the first).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_jpeg_b
IMPGPU_JPEG_WHOLE=1 IMPGPU_JPEG_TRACE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_jpeg_b -- python3 $R/tools/jpeg_batch_probe.py 64 > $R/gpurun_out/${TAG}_jpeg_batch64.log 2>&1
f=$(ls $R/gpurun_out/prof_jpeg_b/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/${TAG}_jpeg_batch64_kernel_stats.csv
head -12 $R/gpurun_out/${TAG}_jpeg_batch64_kernel_stats.csv | cut -c1-160
grep -E "^jpeg x|^batch" $R/gpurun_out/${TAG}_jpeg_batch64.log | tail -4
