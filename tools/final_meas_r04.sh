#!/bin/bash
# Round 4's measurement sessions (each through ONE gpurun call, so everything inside a session is one box):
#   tools/final_meas_r04.sh a   bench modes (all, incl. the strip modes), mixed resident, operator survey, default bench line,
#                               bench.py under rocprofv3 --kernel-trace --stats
#   tools/final_meas_r04.sh b   rocprofv3 passes over tools/pmc_probe.py: kernel trace, FETCH_SIZE, WRITE_SIZE (tools/summarize_prof.py r04 afterwards, locally)
#   tools/final_meas_r04.sh c   the JPEG request path: native stream (raw and JPEG answers, 4 / 8 / 16 threads), kernel stats of the
#                               64-file launch, lone-request latency, traffic counters, N worker processes, the PNG experiment
#   tools/final_meas_r04.sh d   SQ counters of the JPEG kernels and of the blur kernels
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
case "$1" in
a)
  bash tools/bench_modes.sh $O/r04_bench_modes.jsonl cubic area chain chain224 lanczos gamma gotham upscale area2x upscale_x linear_up lanczos_up lanczos_15 > $O/r04_bench_modes.txt 2>&1
  cat $O/r04_bench_modes.txt
  python bench.py --mixed 4096 --steps 10 > $O/r04_mixed_bgra.json
  python bench.py --mixed 4096 --steps 10 --channels 3 > $O/r04_mixed_bgr.json
  python tools/perf_survey.py > $O/r04_operator_survey.txt 2>&1
  python bench.py > $O/r04_bench.json 2>/dev/null
  tail -1 $O/r04_bench.json | cut -c1-400
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/$O/prof_bench && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -- python3 $R/bench.py --no-cpu > $R/$O/r04_bench_under_rocprof.json 2> $R/$O/prof_bench.log )
  cp $(ls $O/prof_bench/*/*kernel_stats.csv | head -1) $O/r04_bench_kernel_stats.csv
  head -3 $O/r04_bench_kernel_stats.csv | cut -c1-200
  ;;
b)
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/$O/prof_trace $R/$O/prof_fetch $R/$O/prof_write
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace -- python3 $R/tools/pmc_probe.py > $R/$O/prof_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_fetch -- python3 $R/tools/pmc_probe.py > $R/$O/prof_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_write -- python3 $R/tools/pmc_probe.py > $R/$O/prof_write.log 2>&1
  tail -1 $R/$O/prof_write.log
  cd $R
  for sg in 2 8 16; do for cn in 4 3; do echo "== blur sigma $sg channels $cn"; bash tools/blur_prof.sh $sg $cn; done; done > $O/r04_blur_kernels.txt 2>&1
  cat $O/r04_blur_kernels.txt | tail -40
  ;;
c)
  make -C tests/c > /dev/null
  for q in 0 86; do
    echo "# tests/c/stream_harness: 65536 requests, 64 files per call, answers: $([ $q = 0 ] && echo raw thumbnails || echo JPEG quality $q); one box, one session"
    JPEG_BATCH=64 JPEG_OUT=$q bash tools/jpeg_stream_native.sh 65536 1 2 4 8 16
  done > $O/r04_jpeg_stream_native.txt 2>&1
  cat $O/r04_jpeg_stream_native.txt
  python bench.py --stream 16384 --jpeg device --native --threads 8 --jpeg-batch 64 > $O/r04_jpeg_stream_line.json 2>/dev/null || true
  python bench.py --stream 16384 --jpeg device --native --threads 8 --jpeg-batch 64 --jpeg-out 86 >> $O/r04_jpeg_stream_line.json 2>/dev/null || true
  bash tools/jpeg_prof_r04.sh r04 > $O/r04_jpeg_prof.txt 2>&1; tail -20 $O/r04_jpeg_prof.txt
  python tools/jpeg_probe.py > $O/r04_jpeg_probe.txt 2>&1; tail -12 $O/r04_jpeg_probe.txt
  python tools/request_latency.py > $O/r04_request_latency.txt 2>&1; tail -6 $O/r04_request_latency.txt
  JPEG_BATCH=64 JPEG_OUT=0 N=32768 bash tools/jpeg_stream_native_prof.sh 8 > $O/r04_jpeg_stream_busy.txt 2>&1; cat $O/r04_jpeg_stream_busy.txt
  bash tools/pmc_jpeg_traffic.sh r04 > $O/r04_pmc_jpeg_traffic.log 2>&1; tail -2 $O/r04_pmc_jpeg_traffic.log
  timeout -k 10 300 python -m pytest tests/test_gpu_multiproc.py -q -m gpu -s > $O/r04_multiproc.txt 2>&1; tail -5 $O/r04_multiproc.txt
  python tools/png_probe.py --out $O/r04_png_probe.json > $O/r04_png_probe.txt 2>&1; tail -4 $O/r04_png_probe.txt
  ;;
d)
  OUT=r04_jpeg_sq_counters.txt bash tools/pmc_jpeg.sh
  ;;
*) echo "usage: $0 a|b|c|d"; exit 2;;
esac
