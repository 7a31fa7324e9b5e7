#!/bin/bash
# every bench.py mode on one box, one JSON line each (profiles/rNN_bench_modes.jsonl)
out=${1:-gpurun_out/bench_modes.jsonl}
shift
modes=${@:-cubic area chain chain224 lanczos gamma gotham upscale area2x}
: > "$out"
for m in $modes; do
  python bench.py --mode $m --no-cpu --steps 50 2>/dev/null | tail -1 >> "$out" || exit 1
done
python - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    r = d["roofline"]
    print("%-9s %10.0f img/s  %8.4f ms/step  alg %7.1f GB/s  frac %.3f" % (d["metric"].split()[-1], d["value"], d["ms_per_step"], r["achieved"], r["frac"]))
PY
