#!/bin/bash
# Round 5, review item 1: the reference's process model measured properly -- C workers, one request at a time, every size
# warm, 3 s per point -- in process (a device context per worker; the box allows 6 processes on the card) and through the
# broker (1 context, any number of workers).  Run through gpurun; writes gpurun_out/r05_worker_scaling.txt.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_worker_scaling.txt
cd $R
{
  echo "# $(date -u +%FT%TZ) worker processes on one MI355X; pool: bench.jpeg_pool(64) (256 px - 4K, quality-90 4:2:0), resize=224,0, JPEG answers q86"
  echo "# host: $(nproc) CPUs usable"
  echo "## direct (libimpgpu.so in every worker)"
  python3 tools/worker_scaling.py direct 1 2 4 6 --seconds ${SECONDS_PER_POINT:-3} || exit 1
  for T in ${BROKER_THREADS:-1 2 4}; do
    echo "## broker, $T threads, gather 0"
    python3 tools/worker_scaling.py broker 1 2 4 8 16 32 --threads $T --seconds ${SECONDS_PER_POINT:-3} || exit 1
  done
  echo "## broker, 4 threads, the workers copy their files as they are (IMPGPU_BROKER_PREPARE=0: the lanes unstuff)"
  IMPGPU_BROKER_PREPARE=0 python3 tools/worker_scaling.py broker 1 8 16 32 --threads 4 --seconds ${SECONDS_PER_POINT:-3} || exit 1
  echo "## broker, 4 threads, --pipeline 1 (a lane unpacks the next batch behind the answers of the one before)"
  python3 tools/worker_scaling.py broker 8 16 32 --threads 4 --pipeline 1 --seconds ${SECONDS_PER_POINT:-3} || exit 1
  echo "## broker, 4 threads, --split-kb 400 (a launch takes files up to 400 KB, or above)"
  python3 tools/worker_scaling.py broker 16 32 --threads 4 --split-kb 400 --seconds ${SECONDS_PER_POINT:-3} || exit 1
  echo "## broker, 2 threads, gather 50 us"
  python3 tools/worker_scaling.py broker 8 16 32 --threads 2 --gather-us 50 --seconds ${SECONDS_PER_POINT:-3} || exit 1
} 2>&1 | tee $O
