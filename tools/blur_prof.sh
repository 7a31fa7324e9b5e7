#!/bin/bash
# which kernels a filter-blur call runs and how long they take: rocprofv3 --kernel-trace --stats over tools/blur_probe.py
#   tools/blur_prof.sh <sigma> <channels>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_blur
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_blur -- python3 $R/tools/blur_probe.py ${1:-8} ${2:-4} 32 > $R/gpurun_out/prof_blur.log 2>&1
tail -1 $R/gpurun_out/prof_blur.log
python3 $R/tools/kstats_fmt.py $(ls $R/gpurun_out/prof_blur/*/*kernel_stats.csv | head -1) 4
