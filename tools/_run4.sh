cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/r05_jpeg_tests.log 2>&1; echo "jpeg tests rc=$?"; tail -3 $O/r05_jpeg_tests.log
bash tools/jpeg_prof_r04.sh r05b > $O/r05b_jpeg_prof.txt 2>&1; head -7 $O/r05b_jpeg_prof.txt | cut -c1-150
python tools/jpeg_probe.py > $O/r05b_jpeg_probe.txt 2>&1; grep "no DRI" $O/r05b_jpeg_probe.txt
python3 tools/worker_scaling.py broker 1 16 32 --threads 4 --seconds 2 2>&1 | cut -c1-330
