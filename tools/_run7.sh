cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/r05_jpeg_tests.log 2>&1; echo "jpeg tests rc=$?"; tail -15 $O/r05_jpeg_tests.log
python tools/jpeg_probe.py > $O/r05c_jpeg_probe.txt 2>&1; grep "no DRI\|DRI=row" $O/r05c_jpeg_probe.txt
IMPGPU_JPEG_FUSED=0 python tools/jpeg_probe.py 2>&1 | grep "no DRI" 
python tools/request_latency.py 2>&1 | tail -4
