"""Generates tests/golden/jpeg/: JPEG files + the pixels PILLOW's decoder (libjpeg-turbo, third-party C) returns for them.

    python tests/golden/jpeg/make_jpeg_golden.py

The expected pixels come from Pillow, NOT from oracle/orc_jpeg.c: these fixtures are what pins that oracle (and,
through it, the device decoder) to libjpeg's arithmetic.  The reference holds no JPEG fixtures of its own
(readme.md:6); its decode is cvDecodeImage at bridge.c:545-552, i.e. libjpeg with default parameters + an R/B swap,
which is exactly `np.asarray(Image.open(...))[:, :, ::-1]`.
Files are encoded by Pillow as well (quality / subsampling / restart interval as listed in manifest.json); the
4:4:0 cases are 4:2:2 files whose SOF header was patched (luma sampling 2x1 -> 1x2, width and height swapped):
the same entropy-coded MCUs, read as 8x16 MCUs.
"""
import hashlib
import io
import json
import os
import sys

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from conftest import noise_image, smooth_image  # noqa: E402


def encode(arr, **kw):
    b = io.BytesIO()
    Image.fromarray(arr).save(b, "JPEG", **kw)
    return b.getvalue()


def pillow_bgr(blob):
    a = np.asarray(Image.open(io.BytesIO(blob)))
    return a[:, :, None].copy() if a.ndim == 2 else a[:, :, ::-1].copy()


def to_440(blob):
    """4:2:2 file of a WxH image (W multiple of 16, H multiple of 8) -> 4:4:0 file of a (W/2)x(2H) image."""
    b = bytearray(blob)
    i = b.index(b"\xff\xc0")
    h = (b[i + 5] << 8) | b[i + 6]
    w = (b[i + 7] << 8) | b[i + 8]
    assert b[i + 9] == 3 and b[i + 11] == 0x21 and w % 16 == 0 and h % 8 == 0
    nw, nh = w // 2, h * 2
    b[i + 5], b[i + 6], b[i + 7], b[i + 8] = nh >> 8, nh & 255, nw >> 8, nw & 255
    b[i + 11] = 0x12
    return bytes(b)


CASES = [
    # name, h, w, kind, channels, save kwargs
    ("c444_q90_48x40", 40, 48, "smooth", 3, dict(quality=90, subsampling="4:4:4")),
    ("c422_q85_49x37", 37, 49, "smooth", 3, dict(quality=85, subsampling="4:2:2")),
    ("c420_q90_67x45", 45, 67, "smooth", 3, dict(quality=90, subsampling="4:2:0")),
    ("c420_q100_noise_33x31", 31, 33, "noise", 3, dict(quality=100, subsampling="4:2:0")),
    ("c420_q30_noise_64x64", 64, 64, "noise", 3, dict(quality=30, subsampling="4:2:0")),
    ("c420_q90_dri4_95x51", 51, 95, "smooth", 3, dict(quality=90, subsampling="4:2:0", restart_marker_blocks=4)),
    ("c420_q75_drirow_80x64", 64, 80, "noise", 3, dict(quality=75, subsampling="4:2:0", restart_marker_rows=1)),
    ("c422_q90_dri1_50x20", 20, 50, "smooth", 3, dict(quality=90, subsampling="4:2:2", restart_marker_blocks=1)),
    ("c444_q95_dri7_41x29", 29, 41, "noise", 3, dict(quality=95, subsampling="4:4:4", restart_marker_blocks=7)),
    ("gray_q90_57x43", 43, 57, "smooth", 1, dict(quality=90)),
    ("gray_q60_dri3_40x24", 24, 40, "noise", 1, dict(quality=60, restart_marker_blocks=3)),
    ("c420_q90_1x1", 1, 1, "smooth", 3, dict(quality=90, subsampling="4:2:0")),
    ("c420_q90_3x2", 2, 3, "noise", 3, dict(quality=90, subsampling="4:2:0")),      # downsampled_width <= 2: replicating upsampler
    ("c422_q90_4x9", 9, 4, "noise", 3, dict(quality=90, subsampling="4:2:2")),
    ("c420_q90_5x17", 17, 5, "noise", 3, dict(quality=90, subsampling="4:2:0")),
    ("c420_q92_opt_120x90", 90, 120, "smooth", 3, dict(quality=92, subsampling="4:2:0", optimize=True)),   # per-file Huffman tables
    ("c420_q50_640x480", 480, 640, "smooth", 3, dict(quality=50, subsampling="4:2:0")),                     # BASELINE configs[0]'s frame size
]


def main():
    manifest = {"generator": "tests/golden/jpeg/make_jpeg_golden.py", "pillow": Image.__version__ if hasattr(Image, "__version__") else None,
                "libjpeg_turbo": features.version("libjpeg_turbo"), "expected_from": "Pillow decode (libjpeg-turbo), channels reversed to B,G,R",
                "cases": []}
    expected = {}
    for name, h, w, kind, ch, kw in CASES:
        arr = smooth_image(h, w, 3, seed=len(name)) if kind == "smooth" else noise_image(h, w, 3, len(name))
        if ch == 1:
            arr = arr[:, :, 1]
        blob = encode(arr, **kw)
        with open(os.path.join(HERE, name + ".jpg"), "wb") as f:
            f.write(blob)
        want = pillow_bgr(blob)
        if want.size <= 40000:          # larger frames are pinned by their digest alone (keeps the fixture set small)
            expected[name] = want
        manifest["cases"].append({"name": name, "height": h, "width": w, "content": kind, "save": kw, "bytes": len(blob),
                                  "shape": list(want.shape), "sha256_bgr": hashlib.sha256(want.tobytes()).hexdigest()})
    for name, h, w, kind, kw in [("c440_q90_patched_24x64", 32, 48, "smooth", dict(quality=90, subsampling="4:2:2")),
                                 ("c440_q80_dri2_patched_40x48", 24, 80, "noise", dict(quality=80, subsampling="4:2:2", restart_marker_blocks=2))]:
        arr = smooth_image(h, w, 3, seed=3) if kind == "smooth" else noise_image(h, w, 3, 3)
        blob = to_440(encode(arr, **kw))
        with open(os.path.join(HERE, name + ".jpg"), "wb") as f:
            f.write(blob)
        want = pillow_bgr(blob)
        expected[name] = want
        manifest["cases"].append({"name": name, "height": 2 * h, "width": w // 2, "content": kind + " (scrambled by the header patch)",
                                  "save": kw, "patched": "4:2:2 -> 4:4:0", "bytes": len(blob),
                                  "shape": list(want.shape), "sha256_bgr": hashlib.sha256(want.tobytes()).hexdigest()})
    np.savez_compressed(os.path.join(HERE, "expected_bgr.npz"), **expected)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print(len(manifest["cases"]), "files,", sum(c["bytes"] for c in manifest["cases"]), "bytes of JPEG")


if __name__ == "__main__":
    main()
