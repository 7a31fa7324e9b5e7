#!/bin/bash
# rocprofv3 evidence for round 3 (run through gpurun): the headline under the profiler, and the batched JPEG stream
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench3 $R/gpurun_out/prof_jpeg3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench3 -- python3 $R/bench.py --no-cpu > $R/gpurun_out/r03_bench_under_rocprof.json 2> $R/gpurun_out/prof_bench3.log
cp $(ls $R/gpurun_out/prof_bench3/*/*kernel_stats.csv | head -1) $R/gpurun_out/r03_bench_kernel_stats.csv
rm -f $R/gpurun_out/prof_bench3/*/*kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_jpeg3 -- python3 $R/bench.py --stream 8192 --threads 4 --jpeg device --jpeg-batch 64 > $R/gpurun_out/r03_jpeg_stream_under_rocprof.json 2> $R/gpurun_out/prof_jpeg3.log
cp $(ls $R/gpurun_out/prof_jpeg3/*/*kernel_stats.csv | head -1) $R/gpurun_out/r03_jpeg_kernel_stats.csv
rm -f $R/gpurun_out/prof_jpeg3/*/*kernel_trace.csv
python3 $R/tools/kstats_fmt.py $R/gpurun_out/r03_bench_kernel_stats.csv 3
tail -1 $R/gpurun_out/r03_bench_under_rocprof.json | cut -c1-200
python3 $R/tools/kstats_fmt.py $R/gpurun_out/r03_jpeg_kernel_stats.csv 8
tail -1 $R/gpurun_out/r03_jpeg_stream_under_rocprof.json | cut -c1-260
