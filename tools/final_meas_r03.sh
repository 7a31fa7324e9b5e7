#!/bin/bash
# Round 3's measurement session, one box, one gpurun call; the files land in gpurun_out/ and are copied to profiles/r03_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
set -x
# 1. every bench.py mode (kernels resident in HBM)
bash tools/bench_modes.sh $O/r03_bench_modes.jsonl cubic area chain chain224 lanczos gamma gotham upscale area2x upscale_x lanczos_up linear_up lanczos_15 > $O/r03_bench_modes.txt 2>&1
# 2. configs[4], frames resident: one descriptor launch
python bench.py --mixed 4096 --steps 10 > $O/r03_mixed_bgra.json 2>/dev/null
python bench.py --mixed 4096 --steps 10 --channels 3 > $O/r03_mixed_bgr.json 2>/dev/null
# 3. configs[4] as a request stream: raw pixels over the link (BGRA, BGR) ...
( for t in 4 8 16; do python bench.py --stream 4096 --threads $t 2>/dev/null; done; for t in 4 8; do python bench.py --stream 4096 --threads $t --channels 3 2>/dev/null; done ) > $O/r03_stream_raw.jsonl
# 4. ... and as JPEG files: decoded on the device (one file per call, then batched), entropy stage on the host, whole decode on the host
( JPEG_MODE=device JPEG_BATCH=1 GPU_MAX_HW_QUEUES=16 tools/jpeg_stream.sh 2048 8 16
  JPEG_MODE=device JPEG_BATCH=16 tools/jpeg_stream.sh 4096 1 4 8
  JPEG_MODE=device JPEG_BATCH=64 tools/jpeg_stream.sh 8192 1 2 4 8 16
  JPEG_MODE=hosthuff JPEG_BATCH=16 tools/jpeg_stream.sh 2048 4 8 16
  JPEG_MODE=host JPEG_BATCH=1 tools/jpeg_stream.sh 512 1 4 8 16 ) > $O/r03_jpeg_stream.txt 2>&1
python bench.py --stream 8192 --threads 4 --jpeg device --jpeg-batch 64 > $O/r03_jpeg_stream_line.json 2>/dev/null
# 4b. ... with JPEG answers as well (cvEncodeImage at bridge.c:704 on the device), against the reference's structure (host codecs at both ends)
( JPEG_MODE=device JPEG_BATCH=64 JPEG_OUT=86 tools/jpeg_stream.sh 8192 1 4 8
  JPEG_MODE=host JPEG_BATCH=1 JPEG_OUT=86 tools/jpeg_stream.sh 512 8 16 ) > $O/r03_jpeg_out_stream.txt 2>&1
python tools/jpeg_enc_probe.py > $O/r03_jpeg_enc_probe.txt 2>&1
python tools/album_probe.py > $O/r03_album_probe.txt 2>&1
python tools/jpeg_wg_trace.py 2> $O/r03_jpeg_wg_trace.txt
# 5. one decode at a time: latency per file, next to Pillow on one core
python tools/jpeg_probe.py 40 > $O/r03_jpeg_probe.txt 2>&1
python tools/jpeg_batch_probe.py 256 > $O/r03_jpeg_batch_probe.txt 2>&1
# 6. operators one by one on a 1080p frame
python tools/perf_survey.py > $O/r03_operator_survey.txt 2>&1
# 7. the default line last (what the driver runs)
python bench.py > $O/r03_bench.json 2>/dev/null
tail -1 $O/r03_bench.json | cut -c1-300
