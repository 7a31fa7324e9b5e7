#!/bin/bash
# SQ counter pass over one bench.py mode: tools/pmc_mode.sh <mode> <outdir> [counter sets...]
# (counters in their own runs: rocprofv3 --pmc with --kernel-trace only; the program itself after `--`)
mode=$1; out=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
mkdir -p $R/$out
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$out/p$i -- python3 $R/bench.py --mode $mode --steps 3 --warmup 1 --no-cpu --prewarm-sec 0 --batch ${PMC_BATCH:-64} > $R/$out/p$i.log 2>&1 || { tail -5 $R/$out/p$i.log; exit 1; }
done
python3 $R/tools/pmc_table.py $R/$out
