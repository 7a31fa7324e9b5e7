"""How often does the phase-parallel entropy scheme (k_jpeg_sync; here its host model, jpeg_emulate_entropy) have to walk a
chunk again?  A "miss" = none of the chunk's walks, started `overlap` bits in front of it from every block phase of the MCU,
had fallen into step with the true decoder by the chunk's first bit.  No GPU needed.

    python tools/jpeg_sync_probe.py            # table over content / quality / sampling / chunk size / overlap
"""
import ctypes as C
import io
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ngx_http_imgproc_amd._lib import lib
from ngx_http_imgproc_amd.workloads import photo_like


def misses(blob, words, overlap):
    os.environ["IMPGPU_JPEG_CHUNK_WORDS"] = str(words)
    os.environ["IMPGPU_JPEG_OVERLAP"] = str(overlap)
    out = np.zeros(16_000_000, dtype=np.int16)
    info = (C.c_int * 12)()
    rc = lib.impgpu_jpeg_coefficients(blob, len(blob), 1, out.ctypes.data, out.size, info)
    assert rc == 0 and info[1] == 0, (rc, list(info))
    st = (C.c_int * 8)()
    lib.impgpu_jpeg_sync_stats(st)
    return list(st)


def main():
    rng = np.random.Generator(np.random.PCG64(5))
    cases = []
    for (w, h) in ((640, 480), (1920, 1080)):
        photo = photo_like(h, w, 3)
        noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        flat = np.full((h, w, 3), 128, np.uint8)
        flat[::37, :, :] = 90
        for name, arr in (("photo", photo), ("noise", noise), ("flat", flat)):
            for q in (50, 90, 98):
                for sub in ("4:2:0", "4:4:4", "4:2:2"):
                    if name != "photo" and (sub == "4:2:2" or q == 50):
                        continue
                    b = io.BytesIO()
                    Image.fromarray(arr).save(b, "JPEG", quality=q, subsampling=sub)
                    cases.append(("%s %dx%d q%d %s" % (name, w, h, q, sub), b.getvalue(), w * h))
    print("%-34s %9s %7s | true-path misses / repair walks / chased chunks, %% of chunks, at chunk bits / overlap bits" % ("file", "bytes", "bit/px"))
    combos = [(8, 128), (8, 256), (16, 128), (16, 256), (16, 512), (32, 256), (32, 512), (32, 1024)]
    print(" " * 54 + "".join("%17s" % ("%d/%d" % (wd * 32, ov)) for wd, ov in combos))
    for name, blob, px in cases:
        row = ""
        for wd, ov in combos:
            st = misses(blob, wd, ov)
            nch = max(1, st[0])
            row += "%17s" % ("%.1f/%.1f/%.2f" % (100.0 * st[1] / nch, 100.0 * st[2] / nch, 100.0 * st[3] / nch))
        print("%-34s %9d %7.2f |%s" % (name, len(blob), 8.0 * len(blob) / px, row))


if __name__ == "__main__":
    main()
