#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_multiproc.py -x -q -p no:cacheprovider 2>&1 | tail -3 && \
timeout -k 10 300 python tools/jpeg_random_sweep.py 2>&1 | tail -1 && \
timeout -k 10 200 python tools/jpeg_stage_probe.py 2>&1 | grep -v amdgpu.ids && \
timeout -k 10 200 python tools/_t.py 1920 1080 2> gpurun_out/trace1080.txt; grep "^wg" gpurun_out/trace1080.txt | awk '{print $9}' | sort -n | tail -3; grep "^wg" gpurun_out/trace1080.txt | sed -n 55,66p | cut -c14-120
