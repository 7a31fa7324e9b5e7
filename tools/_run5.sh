cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/r05_gpu_suite_2.log 2>&1; echo "suite rc=$?"; tail -6 $O/r05_gpu_suite_2.log
