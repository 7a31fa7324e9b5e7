#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_jpeg_enc.py tests/test_gpu_jpeg.py -x -q -p no:cacheprovider 2>&1 | tail -3 && \
timeout -k 10 200 python tools/jpeg_enc_probe.py 2>&1 | grep -v amdgpu.ids | tail -12 && \
timeout -k 10 200 python tools/request_latency.py 2>&1 | grep "ONE wait"
