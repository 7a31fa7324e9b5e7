cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_jpeg_enc.py tests/test_gpu_jpeg.py tests/test_gpu_fuzz.py tests/test_gpu_broker.py -x -q -m gpu > $O/r05_enc_tests.log 2>&1; echo "enc tests rc=$?"; tail -8 $O/r05_enc_tests.log
python tools/request_latency.py 2>&1 | tail -4
python tools/jpeg_enc_probe.py 2>&1 | tail -12
