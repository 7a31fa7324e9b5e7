"""Decode timing of impgpu_image_decode_jpeg on one MI355X: per-call latency from one thread, device vs host entropy stage,
next to Pillow (libjpeg-turbo) on one host core.  python tools/jpeg_probe.py [reps]"""
import io
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (first: see tests/conftest.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from PIL import Image  # noqa: E402

import ngx_http_imgproc_amd as imp  # noqa: E402
from ngx_http_imgproc_amd.workloads import photo_like  # noqa: E402


def encode(arr, **kw):
    b = io.BytesIO()
    Image.fromarray(arr).save(b, "JPEG", **kw)
    return b.getvalue()


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    imp.env_start(0)
    for (h, w) in ((1080, 1920), (2160, 3840), (480, 640)):
        for sub, q, extra, label in (("4:2:0", 90, {}, "no DRI"), ("4:2:0", 90, dict(restart_marker_rows=1), "DRI=row"), ("4:4:4", 90, {}, "no DRI")):
            arr = photo_like(h, w, seed=1)
            blob = encode(arr[:, :, ::-1], quality=q, subsampling=sub, **extra)
            bpp = len(blob) * 8.0 / (h * w)
            t0 = time.perf_counter()
            for _ in range(max(3, reps // 10)):
                np.asarray(Image.open(io.BytesIO(blob)))
            pil_ms = (time.perf_counter() - t0) / max(3, reps // 10) * 1e3
            line = "%dx%d %s q%d %-8s %7d B (%.2f bpp)  pillow 1 core %6.2f ms" % (w, h, sub, q, label, len(blob), bpp, pil_ms)
            for mode in ("device", "host"):
                os.environ["IMPGPU_JPEG_HUFF"] = mode
                for _ in range(3):
                    rc, im = imp.Image.decode_jpeg(blob)
                    assert rc == 0
                    im.release()
                imp.sync()
                t0 = time.perf_counter()
                for _ in range(reps):
                    rc, im = imp.Image.decode_jpeg(blob)
                    im.release()
                imp.sync()
                ms = (time.perf_counter() - t0) / reps * 1e3
                line += " | %s %6.3f ms = %7.1f MB/s" % (mode, ms, len(blob) / ms / 1e3)
            print(line, flush=True)
    imp.env_destroy()


if __name__ == "__main__":
    main()
