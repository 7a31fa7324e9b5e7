cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_suite_1.log 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/r05_gpu_suite_1.log
{
for Q in 8; do for T in 4 8; do
  echo "## broker, $T threads, GPU_MAX_HW_QUEUES=$Q"
  timeout -k 10 120 python3 tools/worker_scaling.py broker 16 32 --threads $T --hw-queues $Q --seconds 2 || exit 1
done; done
} > gpurun_out/r05_broker_hwq.txt 2>&1
cat gpurun_out/r05_broker_hwq.txt | cut -c1-260
