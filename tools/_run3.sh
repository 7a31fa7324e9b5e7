cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_fuzz.py tests/test_gpu_broker.py -x -q -m gpu > $O/r05_jpeg_tests.log 2>&1; echo "jpeg tests rc=$?"; tail -3 $O/r05_jpeg_tests.log
bash tools/jpeg_prof_r04.sh r05a > $O/r05a_jpeg_prof.txt 2>&1; tail -16 $O/r05a_jpeg_prof.txt
python tools/jpeg_probe.py > $O/r05a_jpeg_probe.txt 2>&1; tail -12 $O/r05a_jpeg_probe.txt
python tools/request_latency.py > $O/r05a_request_latency.txt 2>&1; tail -5 $O/r05a_request_latency.txt
