#!/bin/bash
# The parity suites under every A/B switch of DESIGN.md section 4 (run through gpurun): a switch that changes a pixel shows here.
# The switches exist only in a library built with -DIMPGPU_AB_SWITCHES (imp_internal.h); build it BEFORE sending the tree:
#     IMPGPU_EXTRA_FLAGS=-DIMPGPU_AB_SWITCHES IMPGPU_LIB=$PWD/ngx_http_imgproc_amd/libimpgpu_ab.so python ngx_http_imgproc_amd/build.py
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
out=${1:-gpurun_out/r04_switch_matrix.txt}
export IMPGPU_LIB=$R/ngx_http_imgproc_amd/libimpgpu_ab.so
[ -f $IMPGPU_LIB ] || { echo "no $IMPGPU_LIB: build the A/B library first (see the head of this script)"; exit 1; }
echo "# tests/test_gpu_resize.py + test_gpu_chain.py + test_gpu_stream.py + test_gpu_jpeg.py + test_gpu_jpeg_enc.py + test_gpu_filters.py + test_gpu_fuzz.py (-m gpu) under each A/B switch of DESIGN.md section 4, one MI355X" > $out
for sw in "" IMPGPU_DMA_DEPTH=0 IMPGPU_DMA_DEPTH=2 IMPGPU_DMA_DEPTH=4 IMPGPU_DMA_WPB=1 IMPGPU_DMA_WPB=2 IMPGPU_NO_ROLL=1 IMPGPU_NO_DMA3=1 \
          IMPGPU_CHAIN_STREAM=0 IMPGPU_CHAIN_TILE=128 IMPGPU_CHAIN_ORDER=1 IMPGPU_AREA_BH=8 IMPGPU_NO_C4=1 IMPGPU_BOXL=1 IMPGPU_NO_ROWS4=1 \
          IMPGPU_MIX_NOSORT=1 IMPGPU_NO_UP=1 IMPGPU_UP_ROWS=32 IMPGPU_UP_NO_PERIOD=1 IMPGPU_JPEG_HUFF=host IMPGPU_SYNC=spin IMPGPU_POOL_CAP_MB=64 \
          IMPGPU_NUMA_BIND=1 IMPGPU_BLUR_FOUR=1 IMPGPU_BLUR_NO1=1 IMPGPU_JPEG_ENC_ONE_WG=1 IMPGPU_JPEG_CHUNK_WORDS=8 IMPGPU_JPEG_CHUNK_WORDS=16 IMPGPU_JPEG_CHUNK_WORDS=32 IMPGPU_BLUR_MFMA2=1 IMPGPU_STRIP_DYNAMIC=1 IMPGPU_STRIP_NARROW=1 IMPGPU_BLUR_FUSE_KB=128 IMPGPU_JPEG_WHOLE=1 IMPGPU_SYSTEM_HIP=1; do
  echo "== ${sw:-(defaults)}" >> $out
  env $sw timeout -k 10 300 python -m pytest tests/test_gpu_resize.py tests/test_gpu_chain.py tests/test_gpu_stream.py tests/test_gpu_jpeg.py tests/test_gpu_jpeg_enc.py tests/test_gpu_filters.py tests/test_gpu_fuzz.py tests/test_gpu_png.py -q -m gpu 2>&1 | tail -1 >> $out
done
cat $out
