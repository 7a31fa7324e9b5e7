#!/bin/bash
# round-3 bench modes on one box, one JSON line each (run through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
for m in "$@"; do
  timeout -k 10 300 python $R/bench.py --mode $m --steps 50 --no-cpu 2>/dev/null | tail -1
done
