"""One decode per file with IMPGPU_JPEG_TRACE=2: prints the workgroups' clocks at their phase boundaries (the library writes
them to stderr): start | rounds0 | wait1 rounds1 | wait2 rounds2 | count scan carry write, microseconds since the first start."""
import io, os, sys
os.environ["IMPGPU_JPEG_TRACE"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ngx_http_imgproc_amd as gpu
from ngx_http_imgproc_amd.workloads import photo_like
from PIL import Image

gpu.env_start(0)
for (w, h, sub) in ((640, 480, 2), (640, 480, 0), (1920, 1080, 2), (3840, 2160, 2)):
    b = io.BytesIO()
    Image.fromarray(photo_like(h, w, seed=1)).save(b, format="JPEG", quality=90, subsampling=sub)
    blob = b.getvalue()
    for rep in range(3):
        if rep == 2:
            sys.stderr.write("== %dx%d subsampling %d, %d bytes\n" % (w, h, sub, len(blob)))
            sys.stderr.flush()
        os.environ["IMPGPU_JPEG_TRACE"] = "2" if rep == 2 else ""
        if rep < 2:
            os.environ.pop("IMPGPU_JPEG_TRACE")
        rc, im = gpu.Image.decode_jpeg(blob)
        assert rc == 0
        im.release()
