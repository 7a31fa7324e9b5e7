#!/bin/bash
# Where a broker lane's decode time goes on the HOST: the decoder's own stopwatch (IMPGPU_JPEG_TRACE) per launch, averaged.
#   tools/broker_host_phases.sh <broker threads> <workers>
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-4}; W=${2:-16}
NAME=/impgpu-ph-$$
POOL=${POOL:-$R/gpurun_out/jpeg_pool.bin}
[ -f $POOL ] || python3 -c "import sys; sys.path.insert(0,'$R/tools'); import worker_scaling as w; w.make_pool('$POOL')"
D=$(mktemp -d)
IMPGPU_JPEG_TRACE=1 $R/ngx_http_imgproc_amd/impgpu_broker --name $NAME --threads $T --slots 64 --ready-file $D/bready > $D/broker.log 2>&1 &
for i in $(seq 1 600); do [ -f $D/bready ] && break; sleep 0.1; done
[ -f $D/bready ] || { echo "broker did not start"; cat $D/broker.log; exit 1; }
BP=$(cat $D/bready)
PIDS=""
for i in $(seq 0 $((W-1))); do
  $R/tests/c/_build/worker_harness $POOL ${SECS:-2} $i $D broker:$NAME > $D/out.$i 2>&1 &
  PIDS="$PIDS $!"
done
for i in $(seq 1 600); do [ $(ls $D/ready.* 2>/dev/null | wc -l) -ge $W ] && break; sleep 0.1; done
touch $D/go
wait $PIDS
kill -TERM $BP
for i in $(seq 1 50); do kill -0 $BP 2>/dev/null || break; sleep 0.1; done
kill -KILL $BP 2>/dev/null
rm -f /dev/shm$NAME
cat $D/out.* | python3 -c "
import sys, json
rs=[json.loads(l) for l in sys.stdin if l.startswith('{')]
n=sum(r['requests'] for r in rs); s=max(r['seconds'] for r in rs)
print('broker $T threads, %d workers, decoder stopwatch on: %.0f requests/s, mean batch %.2f' % (len(rs), n/s, sum(r['mean_batch']*r['requests'] for r in rs)/n))"
python3 - $D/broker.log <<'PY'
import re, sys
rows = []
for l in open(sys.argv[1], errors="replace"):
    m = re.match(r"jpeg x(\d+) \((\d+) live\): headers (\d+) unstuff (\d+) jobs (\d+) enqueue (\d+) wait (\d+) us", l)
    if m: rows.append([int(x) for x in m.groups()])
rows = rows[len(rows) // 4:]
n = len(rows)
print("%d launches, files per launch %.2f; us per launch: headers %.1f, unstuffing copy %.1f, tables + job table %.1f, enqueue %.1f, wait for the verdicts %.1f"
      % ((n,) + tuple(sum(r[k] for r in rows) / n for k in (0, 2, 3, 4, 5, 6))))
for k in (1, 2, 4, 8):
    sel = [r for r in rows if r[0] == k]
    if sel: print("   launches of %d: %5d; headers %.1f unstuff %.1f jobs %.1f enqueue %.1f wait %.1f" % ((k, len(sel)) + tuple(sum(r[j] for r in sel) / len(sel) for j in (2, 3, 4, 5, 6))))
PY
grep "impgpu_broker:" $D/broker.log | tail -3
rm -rf $D
