#!/bin/bash
# bench.py --stream --jpeg over thread counts (run through gpurun): one line per (threads, decoder)
#   JPEG_MODE=device|hosthuff|host|all  JPEG_BATCH=files per decode call  JPEG_OUT=quality of JPEG answers (0 = raw thumbnails)
#   tools/jpeg_stream.sh <requests> <threads>...
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-1024}; shift
for T in "$@"; do
  timeout -k 10 400 python $R/bench.py --stream $N --threads $T --jpeg ${JPEG_MODE:-device} --jpeg-batch ${JPEG_BATCH:-1} --jpeg-out ${JPEG_OUT:-0} 2>/dev/null | python $R/tools/jpeg_stream_fmt.py
done
