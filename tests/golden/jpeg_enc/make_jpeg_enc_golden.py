#!/usr/bin/env python3
"""Golden vectors for the JPEG ENCODE pin: B,G,R(,A) / gray frames and the files Pillow's encoder (libjpeg-turbo, third-party
C) writes for them with libjpeg's defaults -- what cvEncodeImage(".jpg", frame, {CV_IMWRITE_JPEG_QUALITY, q}) hands to libjpeg
at bridge.c:704.  Run once where Pillow is installed; the outputs are committed (enc_cases.npz + manifest.json)."""
import hashlib
import io
import json
import os

import numpy as np
import PIL
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))


def frame(h, w, c, kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 3 + yy * (k + 1) * 2 + 20 * k) % 256 for k in range(c)], -1)
    return np.clip(base + rng.integers(-6, 7, base.shape), 0, 255).astype(np.uint8)


def pillow_file(arr, q):
    b = io.BytesIO()
    if arr.shape[2] == 1:
        Image.fromarray(arr[:, :, 0], "L").save(b, format="JPEG", quality=q)
    else:
        Image.fromarray(np.ascontiguousarray(arr[:, :, [2, 1, 0]])).save(b, format="JPEG", quality=q, subsampling=2)
    return b.getvalue()


CASES = [  # (h, w, c, quality, kind): every padding / dummy-block rule, the quality scale's ends and both branches
    (16, 16, 3, 75, "noise"), (8, 8, 3, 90, "noise"), (1, 1, 3, 50, "noise"), (17, 23, 3, 95, "smooth"), (24, 40, 3, 10, "noise"),
    (100, 75, 3, 100, "smooth"), (126, 224, 3, 90, "smooth"), (126, 224, 4, 90, "noise"), (31, 47, 1, 75, "smooth"),
    (8, 8, 1, 0, "noise"), (33, 9, 4, 1, "noise"), (240, 320, 3, 49, "smooth"), (15, 17, 1, 100, "noise"), (64, 64, 3, 51, "noise"),
]


def main():
    data, manifest = {}, []
    for i, (h, w, c, q, kind) in enumerate(CASES):
        a = frame(h, w, c, kind, 7000 + i)
        f = pillow_file(a, q)
        data["in_%02d" % i] = a
        data["file_%02d" % i] = np.frombuffer(f, dtype=np.uint8)
        manifest.append({"case": i, "height": h, "width": w, "channels": c, "quality": q, "kind": kind, "bytes": len(f),
                         "sha256": hashlib.sha256(f).hexdigest()})
    np.savez_compressed(os.path.join(HERE, "enc_cases.npz"), **data)
    with open(os.path.join(HERE, "manifest.json"), "w") as fh:
        json.dump({"pillow": PIL.__version__, "libjpeg_turbo": features.version("jpg"), "cases": manifest}, fh, indent=1)
    print("wrote", len(CASES), "cases")


if __name__ == "__main__":
    main()
