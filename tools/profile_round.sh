#!/bin/bash
# The round's rocprofv3 evidence, on the GPU box (run through gpurun; summaries are then made by tools/summarize_prof.py):
#   gpurun_out/prof_trace  : --kernel-trace --stats over tools/pmc_probe.py      (kernel durations)
#   gpurun_out/prof_fetch  : --pmc FETCH_SIZE   (own pass)                        (HBM reads)
#   gpurun_out/prof_write  : --pmc WRITE_SIZE   (own pass)                        (HBM writes)
#   gpurun_out/prof_bench  : --kernel-trace --stats over bench.py itself          (the benchmarked launch's average duration)
# Counters are collected in their own runs (no --sys-trace / marker domains next to --pmc); the program itself follows `--`.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_trace $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write $R/gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $R/tools/pmc_probe.py > $R/gpurun_out/prof_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/tools/pmc_probe.py > $R/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/tools/pmc_probe.py > $R/gpurun_out/prof_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --no-cpu > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/prof_bench.log
tail -1 $R/gpurun_out/bench_under_rocprof.json | cut -c1-200
