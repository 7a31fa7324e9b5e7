"""Round 4's bounded PNG experiment, measured: one host core decoding a PNG with Pillow (libpng: inflate + unfilter + row
delivery) beside impgpu_image_decode_png (host inflate, filters undone by k_png_unfilter) on the same files.

    python tools/png_probe.py [--out gpurun_out/png_probe.json]
Per file: Pillow ms, zlib.decompress ms (the inflate alone), the product call's host ms split by impgpu_png_stage_times,
the wall time until the pixels are in HBM (call + impgpu_sync), and the inflate's share of that wall time."""
import argparse
import io
import json
import os
import struct
import sys
import time
import zlib

import numpy as np
import torch  # noqa: F401  (first: see tests/conftest.py)
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ngx_http_imgproc_amd as imp
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
    return min(t) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--reps", type=int, default=7)
    a = ap.parse_args()
    imp.env_start(0)
    rows = []
    for (w, h) in ((640, 480), (1920, 1080), (3840, 2160)):
        rgb = photo_like(h, w, 3)
        for mode, arr in (("RGB", rgb), ("RGBA", np.dstack([rgb, np.full((h, w), 255, np.uint8)])), ("gray", np.ascontiguousarray(rgb[:, :, 1]))):
            b = io.BytesIO()
            Image.fromarray(arr).save(b, "PNG", compress_level=6)
            blob = b.getvalue()
            comp = idat(blob)
            want = np.asarray(Image.open(io.BytesIO(blob)))
            want = want[:, :, None] if want.ndim == 2 else want[:, :, [2, 1, 0] + ([3] if want.shape[2] == 4 else [])]
            rc, im = imp.Image.decode_png(blob)
            assert rc == 0
            got = im.numpy()
            assert np.array_equal(got if got.ndim == 3 else got[:, :, None], want)
            stages = []

            def product():
                rc, im = imp.Image.decode_png(blob)
                stages.append(imp.png_stage_times())
                imp.sync()
                return im

            t_gpu = best(product, a.reps)
            st = np.array(stages).min(axis=0)
            t_pil = best(lambda: np.asarray(Image.open(io.BytesIO(blob))), a.reps)
            t_inf = best(lambda: zlib.decompress(comp), a.reps)
            rows.append({"file": "%dx%d %s level 6" % (w, h, mode), "bytes": len(blob), "pillow_ms": round(t_pil, 3), "zlib_inflate_ms": round(t_inf, 3),
                         "impgpu_ms": round(t_gpu, 3), "impgpu_host_inflate_ms": round(st[1] / 1e3, 3), "impgpu_host_other_ms": round((st[0] + st[2]) / 1e3, 3),
                         "inflate_share_of_impgpu": round(st[1] / 1e3 / t_gpu, 3), "inflate_share_of_pillow": round(t_inf / t_pil, 3),
                         "speedup_vs_pillow": round(t_pil / t_gpu, 2)})
            print(json.dumps(rows[-1]), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump({"device": torch.cuda.get_device_name(0) if torch.cuda.is_available() else "", "rows": rows}, f, indent=1)
    imp.env_destroy()


if __name__ == "__main__":
    main()
