#!/bin/bash
# bench.py --stream --jpeg over thread counts (run through gpurun): one line per (threads, decoder)
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-1024}; shift
for T in "$@"; do
  timeout -k 10 400 python $R/bench.py --stream $N --threads $T --jpeg ${JPEG_MODE:-device} 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('threads', d['config']['threads_per_gpu'], d['decoder'], d['value'], 'req/s', d['compressed_MB_per_sec'], 'MB/s compressed', d['bits_per_pixel'], 'bpp', flush=True)
    except Exception: pass
"
done
