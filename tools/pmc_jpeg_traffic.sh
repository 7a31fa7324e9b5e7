#!/bin/bash
# HBM traffic of the JPEG kernels: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (with --kernel-trace only,
# the program itself after `--`) over tools/jpeg_pmc_probe.py -- three times a batch of 64 mixed-size files decoded, resized,
# encoded.  Run through gpurun; tools/summarize_jpeg_traffic.py <round> then writes profiles/<round>_jpeg_pmc.json and
# profiles/traffic_jpeg.json.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_jfetch $R/gpurun_out/prof_jwrite
export IMPGPU_JPEG_WHOLE=1          # one launch per kernel and batch
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_jfetch -- python3 $R/tools/jpeg_pmc_probe.py > $R/gpurun_out/prof_jfetch.log 2>&1 || { tail -5 $R/gpurun_out/prof_jfetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_jwrite -- python3 $R/tools/jpeg_pmc_probe.py > $R/gpurun_out/prof_jwrite.log 2>&1 || { tail -5 $R/gpurun_out/prof_jwrite.log; exit 1; }
rm -f $R/gpurun_out/prof_jfetch/*/*kernel_trace.csv $R/gpurun_out/prof_jwrite/*/*kernel_trace.csv
echo "counters are in gpurun_out/prof_jfetch, prof_jwrite: run tools/summarize_jpeg_traffic.py ${1:-r04} in the repository (gpurun merges gpurun_out/ only)"
