cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_filters.py -x -q -m gpu -k blur > $O/r05_blur_tests.log 2>&1; echo "blur tests rc=$?"; tail -3 $O/r05_blur_tests.log
for sg in 2 4 7.5 8 16; do for cn in 4 3; do echo "== blur sigma $sg channels $cn"; bash tools/blur_prof.sh $sg $cn; done; done > $O/r05_blur_kernels.txt 2>&1
grep "==\|k_blur\|us/frame" $O/r05_blur_kernels.txt | cut -c1-170
BLUR_SIGMA=8 bash tools/pmc_blur.sh > $O/r05_blur_sq_counters_s8.txt 2>&1; tail -60 $O/r05_blur_sq_counters_s8.txt
