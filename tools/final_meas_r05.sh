#!/bin/bash
# Round 5's measurement sessions (each through ONE gpurun call, so everything inside a session is one box):
#   tools/final_meas_r05.sh a   bench modes, mixed resident (cfg5), default bench line, bench.py under rocprofv3 --kernel-trace --stats,
#                               FETCH_SIZE / WRITE_SIZE passes over tools/pmc_probe.py (summarize_prof.py r05 afterwards, locally)
#   tools/final_meas_r05.sh b   the JPEG request path: native stream with and without begin/finish (ahead 0 / 1), kernel stats of the
#                               64-file launch, lone-file probe, lone-request latency, the stream's bench lines
#   tools/final_meas_r05.sh c   worker processes: in-process (1-6) and through the broker (1-32 workers; 2, 3, 4, 6 lanes), broker under the profiler
#   tools/final_meas_r05.sh d   SQ counters: cfg4's and cfg5's kernels, the JPEG kernels, the blur kernels; blur kernel times
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
case "$1" in
a)
  bash tools/bench_modes.sh $O/r05_bench_modes.jsonl cubic area chain chain224 lanczos gamma gotham upscale area2x upscale_x linear_up lanczos_up lanczos_15 > $O/r05_bench_modes.txt 2>&1
  cat $O/r05_bench_modes.txt
  python bench.py --mixed 4096 --steps 10 > $O/r05_mixed_bgra.json
  python bench.py --mixed 4096 --steps 10 --channels 3 > $O/r05_mixed_bgr.json
  python bench.py > $O/r05_bench.json 2>/dev/null
  tail -1 $O/r05_bench.json | cut -c1-400
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/$O/prof_bench && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -- python3 $R/bench.py --no-cpu > $R/$O/r05_bench_under_rocprof.json 2> $R/$O/prof_bench.log )
  cp $(ls $O/prof_bench/*/*kernel_stats.csv | head -1) $O/r05_bench_kernel_stats.csv
  head -3 $O/r05_bench_kernel_stats.csv | cut -c1-200
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/$O/prof_trace $R/$O/prof_fetch $R/$O/prof_write
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace -- python3 $R/tools/pmc_probe.py > $R/$O/prof_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_fetch -- python3 $R/tools/pmc_probe.py > $R/$O/prof_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_write -- python3 $R/tools/pmc_probe.py > $R/$O/prof_write.log 2>&1
  tail -1 $R/$O/prof_write.log
  du -sh $R/$O | tail -1
  ;;
b)
  make -C tests/c > /dev/null
  for ahead in 0 1; do for q in 0 86; do
    echo "# tests/c/stream_harness: 65536 requests, 64 files per call, ahead = $ahead, answers: $([ $q = 0 ] && echo raw thumbnails || echo JPEG quality $q); one box, one session"
    JPEG_AHEAD=$ahead JPEG_BATCH=64 JPEG_OUT=$q bash tools/jpeg_stream_native.sh 65536 1 2 4 8 16
  done; done > $O/r05_jpeg_stream_native.txt 2>&1
  cat $O/r05_jpeg_stream_native.txt
  python bench.py --stream 16384 --jpeg device --native --threads 8 --jpeg-batch 64 > $O/r05_jpeg_stream_line.json 2>/dev/null || true
  python bench.py --stream 16384 --jpeg device --native --threads 8 --jpeg-batch 64 --jpeg-out 86 >> $O/r05_jpeg_stream_line.json 2>/dev/null || true
  bash tools/jpeg_prof_r04.sh r05 > $O/r05_jpeg_prof.txt 2>&1; tail -20 $O/r05_jpeg_prof.txt
  python tools/jpeg_probe.py > $O/r05_jpeg_probe.txt 2>&1; tail -12 $O/r05_jpeg_probe.txt
  python tools/request_latency.py > $O/r05_request_latency.txt 2>&1; tail -6 $O/r05_request_latency.txt
  python tools/jpeg_tiny_probe.py > $O/r05_jpeg_small_files.txt 2>&1; tail -12 $O/r05_jpeg_small_files.txt
  python tools/jpeg_stage_probe.py 2>&1 | grep -v amdgpu.ids > $O/r05_jpeg_stage_probe.txt; { echo "# IMPGPU_JPEG_SPLIT=0: one lane of k_jpeg_write per chunk"; IMPGPU_JPEG_SPLIT=0 python tools/jpeg_stage_probe.py 2>&1 | grep -v amdgpu.ids; } >> $O/r05_jpeg_stage_probe.txt; cat $O/r05_jpeg_stage_probe.txt
  python tools/jpeg_enc_probe.py 2>&1 | grep -v amdgpu.ids > $O/r05_jpeg_enc_probe.txt; cat $O/r05_jpeg_enc_probe.txt
  rm -f $O/jpeg_pool.bin; rm -rf $O/prof_jpeg_b      # (gpurun copies at most 64 MB back)
  ;;
c)
  make -C tests/c > /dev/null
  BROKER_THREADS="2 3 4 6" SECONDS_PER_POINT=3 bash tools/r05_workers.sh
  bash tools/r05_broker_prof.sh 4 16 > $O/r05_broker_prof_4_16.txt 2>&1; cat $O/r05_broker_prof_4_16.txt | head -30
  timeout -k 10 300 bash tools/contention_probe.sh "1 2 4 5 6 8" > $O/r05_lane_contention.txt 2>&1; tail -5 $O/r05_lane_contention.txt
  { timeout -k 10 120 bash tools/broker_host_phases.sh 4 16; IMPGPU_BROKER_PREPARE=0 timeout -k 10 120 bash tools/broker_host_phases.sh 4 16; } > $O/r05_broker_host_phases.txt 2>&1; cat $O/r05_broker_host_phases.txt
  rm -rf $O/prof_broker_*
  timeout -k 10 300 python -m pytest tests/test_gpu_multiproc.py tests/test_gpu_broker.py -q -m gpu -s > $O/r05_multiproc.txt 2>&1; tail -8 $O/r05_multiproc.txt
  python bench.py --workers 16 --seconds 3 > $O/r05_bench_workers.json 2>/dev/null; python bench.py --workers 32 --seconds 3 >> $O/r05_bench_workers.json 2>/dev/null; cat $O/r05_bench_workers.json | cut -c1-300
  rm -f $O/jpeg_pool.bin; rm -rf $O/prof_broker_4_16      # (gpurun copies at most 64 MB back; the trace's summary is in r05_broker_prof_4_16.txt)
  ;;
d)
  PMC_BATCH=32 bash tools/pmc_mode.sh lanczos $O/pmc_r05_lanczos "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" > $O/r05_sq_cfg4.txt 2>&1 || true
  tail -30 $O/r05_sq_cfg4.txt
  rm -rf $O/pmc_r05_lanczos
  bash tools/r05_pmc_mixed.sh > $O/r05_sq_cfg5.txt 2>&1 || true
  tail -40 $O/r05_sq_cfg5.txt
  rm -rf $O/pmc_r05_mixed_4 $O/pmc_r05_mixed_3
  OUT=r05_jpeg_sq_counters.txt bash tools/pmc_jpeg.sh || true
  rm -rf $O/pmc_jpeg $O/jpeg_pool.bin
  for sg in 2 4 8 16; do for cn in 4 3; do echo "== blur sigma $sg channels $cn"; bash tools/blur_prof.sh $sg $cn; done; done > $O/r05_blur_kernels.txt 2>&1
  BLUR_SIGMA=8 bash tools/pmc_blur.sh > $O/r05_blur_sq_counters.txt 2>&1 || true
  rm -rf $O/pmc_blur $O/prof_blur
  du -sh $O | tail -1
  ;;
*) echo "usage: $0 a|b|c|d"; exit 2;;
esac
