#!/bin/bash
# cfg5 thread scaling on one GPU: bench.py --stream at 1..16 host threads (profiles/rNN_stream_scaling.txt)
out=${1:-gpurun_out/stream_scaling.txt}
: > "$out"
for t in 1 2 4 8 12 16; do
  python bench.py --stream 4096 --threads $t >> "$out" 2>&1 || exit 1
done
cat "$out"
