#!/bin/bash
# INTER_AREA (general path) across shrink factors, uniform batches of ~2 GB: tools/area_scales.sh [channels]
R=$(dirname $(dirname $(readlink -f $0)))
C=${1:-4}
for g in "3840 2160 224 126 64" "2560 1440 224 126 128" "1920 1080 224 126 256" "1920 1080 500 281 256" "1280 720 224 126 512" \
         "1000 750 224 168 512" "640 480 224 168 1024" "400 300 224 168 2048" "256 256 224 224 4096" "1920 1080 1500 844 256" \
         "1920 1080 640 360 256" "1920 1080 480 270 256" "1920 1080 384 216 256" "1920 1080 240 135 256"; do
  set -- $g
  python $R/tools/resize_probe.py $1 $2 $3 $4 $C 3 $5
done
