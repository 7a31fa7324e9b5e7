#!/bin/bash
# brightness: parity tests, then a kernel trace of tools/brightness_probe.py -> gpurun_out/bright_kernel_stats.csv
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_filters.py tests/test_golden.py tests/test_gpu_fuzz.py -x -q -k "brightness or golden" > gpurun_out/bright_tests.log 2>&1 || { tail -30 gpurun_out/bright_tests.log; exit 1; }
tail -2 gpurun_out/bright_tests.log
python tools/brightness_probe.py
bash tools/prof_cmd.sh bright $R/tools/brightness_probe.py
