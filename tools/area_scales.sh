#!/bin/bash
# INTER_AREA (general path) across shrink factors, uniform batches of ~2 GB: tools/area_scales.sh [env assignments...]
R=$(dirname $(dirname $(readlink -f $0)))
for g in "3840 2160 224 126 64" "2560 1440 224 126 128" "1920 1080 224 126 256" "1920 1080 500 281 256" "1280 720 224 126 512" \
         "1000 750 224 168 512" "640 480 224 168 1024" "400 300 224 168 2048" "256 256 224 224 4096" "1920 1080 1500 844 256"; do
  set -- $g
  python $R/tools/resize_probe.py $1 $2 $3 $4 4 3 $5
done
