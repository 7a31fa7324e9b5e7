"""The bounded PNG experiment's first question (round 4, review item 7): of the time libpng takes to decode a PNG on one host core
(Pillow's decoder: zlib inflate + unfiltering + row delivery), how much is the inflate -- the part that cannot move to the device
without a device inflate?  No GPU needed.   python tools/png_inflate_probe.py"""
import io
import os
import struct
import sys
import time
import zlib

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ngx_http_imgproc_amd.workloads import photo_like


def idat(blob):
    at, out = 8, []
    while at < len(blob):
        n, kind = struct.unpack(">I4s", blob[at:at + 8])
        if kind == b"IDAT":
            out.append(blob[at + 8:at + 8 + n])
        at += 12 + n
    return b"".join(out)


def best(fn, reps):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return min(t)


def unfilter_numpy(raw, h, stride, bpp):
    """reference unfiltering (vectorised per row where the filter allows it): the rest of libpng's work"""
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), dtype=np.uint8)
    kinds = rows[:, 0]
    return kinds


print("%-34s %10s %9s %9s %8s  filters used" % ("file", "bytes", "pillow ms", "inflate ms", "share"))
for (w, h) in ((640, 480), (1920, 1080), (3840, 2160)):
    for mode, arr in (("RGB photo", photo_like(h, w, 3)), ("RGBA photo", np.dstack([photo_like(h, w, 3), np.full((h, w), 255, np.uint8)])),
                      ("gray photo", photo_like(h, w, 3)[:, :, 1].copy())):
        for level in (6, 9):
            b = io.BytesIO()
            Image.fromarray(arr).save(b, "PNG", compress_level=level)
            blob = b.getvalue()
            comp = idat(blob)
            reps = 5 if w < 3000 else 3
            t_pil = best(lambda: np.asarray(Image.open(io.BytesIO(blob))), reps)
            t_inf = best(lambda: zlib.decompress(comp), reps)
            raw = zlib.decompress(comp)
            bpp = arr.shape[2] if arr.ndim == 3 else 1
            kinds = np.bincount(unfilter_numpy(raw, h, w * bpp, bpp), minlength=5)
            print("%-34s %10d %9.2f %9.2f %7.0f%%  none/sub/up/avg/paeth rows %s" % ("%dx%d %s level %d" % (w, h, mode, level), len(blob), t_pil * 1e3, t_inf * 1e3,
                                                                                  100 * t_inf / t_pil, list(kinds)))
