cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_fuzz.py tests/test_gpu_stream.py tests/test_c_harness.py tests/test_gpu_broker.py -x -q -m gpu > $O/r05_jpeg_tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/r05_jpeg_tests.log
python tools/jpeg_probe.py 2>&1 | grep "no DRI"
python tools/request_latency.py 2>&1 | tail -4
python tools/jpeg_stage_probe.py 2>&1 | tail -4
