cd $GRAFT_REPO_ROOT
AB=$(ls ngx_http_imgproc_amd/libimpgpu_*.so | grep -v client | head -1)
for W in 4 8 16; do
  echo "== chunk words $W"
  IMPGPU_LIB=$PWD/$AB IMPGPU_JPEG_CHUNK_WORDS=$W python tools/jpeg_probe.py 2>&1 | grep "4:2:0 q90 no DRI"
done
