#!/bin/bash
# is k_resize_area_rows<4,10>'s 1.39 <-> 2.22 ms a clock / power effect or a probe artefact?
# Per-launch durations (kernel trace) of `bench.py --mode <mode>` over a few thousand launches, clocks before and after.
R=${GRAFT_REPO_ROOT:-/root/repo}
MODE=${1:-area}; STEPS=${2:-2000}; NEEDLE=${3:-k_resize_area_rows}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/swing_$MODE
rm -rf $OUT; mkdir -p $OUT
(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6) > $OUT/clocks_before.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --mode $MODE --steps $STEPS --warmup 20 --no-cpu > $OUT/bench.json 2> $OUT/bench.err
(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6) > $OUT/clocks_after.txt
f=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python3 $R/tools/launch_series.py $f "$NEEDLE" 200 > $OUT/series.txt
cat $OUT/clocks_before.txt; head -40 $OUT/series.txt; cat $OUT/clocks_after.txt; cut -c1-160 $OUT/bench.json
rm -rf $OUT/trace
