"""The clocks of k_jpeg_select's workgroups for ONE lone file (IMPGPU_JPEG_TRACE=2 on the fifth decode): on stderr a line per
workgroup -- start | candidates loaded | twins | maps | scan + look-back | done, rounds of picking, chunks chased -- and per workgroup of
k_jpeg_write its start and end.  profiles/r05_jpeg_select_chain.txt is cut from it.
    python tools/jpeg_chain_trace.py 1920 1080 2> trace.txt"""
import io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
import ngx_http_imgproc_amd as imp
from ngx_http_imgproc_amd.workloads import photo_like
os.environ["IMPGPU_JPEG_HUFF"] = "device"
imp.env_start(0)
w, h = int(sys.argv[1]), int(sys.argv[2])
b = io.BytesIO(); Image.fromarray(photo_like(h, w, 3)).save(b, "JPEG", quality=90, subsampling="4:2:0"); blob = b.getvalue()
for i in range(5):
    if i == 4: os.environ["IMPGPU_JPEG_TRACE"] = "2"
    rc, im = imp.Image.decode_jpeg(blob); assert rc == 0; im.release()
