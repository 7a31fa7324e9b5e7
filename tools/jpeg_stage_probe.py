"""Where a lone decode's time goes (impgpu_jpeg_profile / impgpu_jpeg_stage_times): host clock per step, device events per stage.
    python tools/jpeg_stage_probe.py [w h] ..."""
import ctypes as C
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
import ngx_http_imgproc_amd as imp
from ngx_http_imgproc_amd.workloads import photo_like

os.environ["IMPGPU_JPEG_HUFF"] = "device"
imp.env_start(0)
imp.lib.impgpu_jpeg_profile(1)
sizes = [(640, 480), (1280, 720), (1920, 1080), (3840, 2160)]
names = ["headers", "unstuff", "tables+jobs", "enqueue", "wait", "upload", "walks", "mend", "select", "write", "dcfix", "verdict+pixels"]
for w, h in sizes:
    b = io.BytesIO()
    Image.fromarray(photo_like(h, w, 3)).save(b, "JPEG", quality=90, subsampling="4:2:0")
    blob = b.getvalue()
    acc = np.zeros(16)
    n = 40
    for i in range(n + 5):
        rc, im = imp.Image.decode_jpeg(blob)
        assert rc == 0
        im.release()
        t = (C.c_double * 16)()
        imp.lib.impgpu_jpeg_stage_times(t, 16)
        if i >= 5:
            acc += np.array(list(t))
    acc /= n
    print("%dx%d %d B: " % (w, h, len(blob)) + ", ".join("%s %.0f" % (names[i], acc[i]) for i in range(12)) + " us (host steps sum %.0f, device stages sum %.0f)" % (acc[:5].sum(), acc[5:12].sum()))
imp.env_destroy()
