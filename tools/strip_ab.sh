#!/bin/bash
# k_resize_strip's four bench modes, one session on one box: bench.py lines -> gpurun_out/strip_modes.jsonl
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
out=${1:-gpurun_out/strip_modes.jsonl}
: > $out
for m in linear_up lanczos_up upscale_x lanczos_15; do
  timeout -k 10 200 python bench.py --mode $m --steps 30 --warmup 20 --no-cpu >> $out 2>> gpurun_out/strip_modes.err || exit 1
done
python - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d["config"]["workload"][:60], d["ms_per_step"], d["roofline"]["frac"])
PY
