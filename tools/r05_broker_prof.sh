#!/bin/bash
# Kernel timeline of the broker under load (rocprofv3 --kernel-trace; the broker is the program after `--`, the workers are
# ordinary processes that never touch the GPU), and of ONE in-process worker alone.  Run through gpurun.
#   tools/r05_broker_prof.sh <broker threads> <workers> [gather_us]
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-4}; W=${2:-16}; G=${3:-0}
NAME=/impgpu-prof-$$
OUT=$R/gpurun_out/prof_broker_${T}_${W}
POOL=${POOL:-$R/gpurun_out/jpeg_pool.bin}
[ -f $POOL ] || python3 -c "import sys; sys.path.insert(0,'$R/tools'); import worker_scaling as w; w.make_pool('$POOL')"
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
READY=$OUT/ready
rocprofv3 --kernel-trace --output-format csv -d $OUT -- $R/ngx_http_imgproc_amd/impgpu_broker --name $NAME --threads $T --gather-us $G --slots 64 --ready-file $READY > $OUT/broker.log 2>&1 &
PROF=$!
for i in $(seq 1 600); do [ -f $READY ] && break; sleep 0.1; done
[ -f $READY ] || { echo "broker did not start"; cat $OUT/broker.log; kill $PROF; exit 1; }
D=$(mktemp -d)
PIDS=""
for i in $(seq 0 $((W-1))); do
  $R/tests/c/_build/worker_harness $POOL ${SECS:-2} $i $D broker:$NAME > $D/out.$i 2>&1 &
  PIDS="$PIDS $!"
done
for i in $(seq 1 600); do [ $(ls $D/ready.* 2>/dev/null | wc -l) -ge $W ] && break; sleep 0.1; done
touch $D/go
wait $PIDS
cat $D/out.* | python3 -c "
import sys, json
rs=[json.loads(l) for l in sys.stdin if l.startswith('{')]
n=sum(r['requests'] for r in rs); s=max(r['seconds'] for r in rs)
print('broker %s threads, %d workers under the profiler: %.0f requests/s, mean batch %.2f, p50 %.0f us' % ('$T', len(rs), n/s, sum(r['mean_batch']*r['requests'] for r in rs)/n, sum(r['p50_us']*r['requests'] for r in rs)/n))"
BP=$(cat $READY)
kill -TERM $BP
# (rocprofv3 writes its files from its own SIGTERM handler; a broker that then runs on into impgpu_env_destroy with the tool
#  already finalized may not return: give it a moment, then end it)
for i in $(seq 1 50); do kill -0 $BP 2>/dev/null || break; sleep 0.1; done
kill -KILL $BP 2>/dev/null
wait $PROF
rm -f /dev/shm$NAME
rm -rf $D
CSV=$(ls $OUT/*/*kernel_trace.csv | head -1)
python3 $R/tools/trace_busy.py $CSV 0.5
python3 $R/tools/trace_lanes.py $CSV
rm -f $OUT/*/*kernel_trace.csv
