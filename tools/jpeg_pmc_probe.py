"""The workload tools/pmc_jpeg.sh counts: three times a batch of 64 mixed-size JPEG files (the stream's pool) decoded on the
device, resized and encoded again."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ngx_http_imgproc_amd as gpu

gpu.env_start(0)
files = bench.jpeg_pool(64)
for _ in range(3):
    res = gpu.batch_decode_jpeg([b for _, _, b in files])
    ims = [im for code, im in res if code == 0]
    for im in ims:
        assert im.resize("224,0") == 0
    out = gpu.batch_encode_jpeg(ims, 86)
    assert all(c == 0 for c, _ in out)
    for im in ims:
        im.release()
