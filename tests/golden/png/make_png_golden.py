"""Writes tests/golden/png/: small PNG files plus the pixels Pillow (libpng + zlib inside Pillow 12.2.0) decodes from them,
in the channel order the reference's cvDecodeImage(.., -1) delivers (gray, B,G,R or B,G,R,A) -- the golden vectors that pin
oracle/orc_png.c and, through it and directly, impgpu_image_decode_png.

libpng's own encoder picks the row filters by a heuristic, so files written by Pillow do not reach every predictor on demand.
The files named f*_ are therefore written by the small encoder below (PNG specification 9.2: it APPLIES a chosen filter per
row; the DECODING, the thing under test, is Pillow's); the files named pil_* are Pillow's own.  Negative cases carry the code
the decoders must return instead of pixels.

    python tests/golden/png/make_png_golden.py          (deterministic: rewrites identical files)
"""
import io
import json
import os
import struct
import zlib

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
UNSUPPORTED, DECODE_FAILED = 1, 3


import sys

sys.path.insert(0, os.path.join(HERE, "..", ".."))
from png_writer import chunk, filter_rows, write_png  # noqa: E402


def content(rng, h, w, c):
    """smooth ramps plus a little noise and a few hard edges: every predictor gets both small and wrapping residuals"""
    yy, xx = np.mgrid[0:h, 0:w]
    planes = []
    for k in range(c):
        p = (xx * (3 + k) + yy * (5 - k) + 40 * k) % 256
        p = p + rng.integers(-6, 7, size=(h, w))
        p[(xx // 9 + yy // 7) % 5 == 0] = rng.integers(0, 256)
        planes.append(np.clip(p, 0, 255))
    return np.stack(planes, axis=2).astype(np.uint8)


def to_reference_order(arr):
    if arr.ndim == 2:
        return arr[:, :, None]
    return arr[:, :, [2, 1, 0] + ([3] if arr.shape[2] == 4 else [])]


def main():
    rng = np.random.default_rng(20261004)
    files, expected, manifest = {}, {}, {}

    def add(name, blob, code=0, note=""):
        files[name] = blob
        manifest[name] = {"code": code, "note": note}
        if code == 0:
            got = np.asarray(Image.open(io.BytesIO(blob)))
            expected[name] = np.ascontiguousarray(to_reference_order(got))
            manifest[name]["shape"] = list(expected[name].shape)

    cyc = lambda h, start=0: [(start + y) % 5 for y in range(h)]
    add("f_gray_67x45.png", write_png(content(rng, 45, 67, 1)[:, :, 0], cyc(45), 0), note="every filter in turn, one band")
    add("f_rgb_130x70.png", write_png(content(rng, 70, 130, 3), cyc(70, 1), 2), note="two bands of 64 rows")
    add("f_rgba_33x140.png", write_png(content(rng, 140, 33, 4), cyc(140, 2), 6), note="three bands, odd width")
    add("f_rgb_41x600.png", write_png(content(rng, 600, 41, 3), [int(v) for v in rng.integers(0, 5, 600)], 2), note="ten bands: the edge slots are reused")
    add("f_rgb_paeth_64x64.png", write_png(content(rng, 64, 64, 3), [4] * 64, 2), note="Paeth only, exactly one band")
    add("f_rgba_avg_65x65.png", write_png(content(rng, 65, 65, 4), [3] * 65, 6), note="Average only, one row into the second band")
    add("f_gray_sub_up_256x9.png", write_png(content(rng, 9, 256, 1)[:, :, 0], [1, 2] * 4 + [1], 0), note="Sub / Up")
    add("f_rgb_noise_31x29.png", write_png(rng.integers(0, 256, size=(29, 31, 3), dtype=np.uint8), cyc(29, 3), 2, level=1), note="noise: every residual wraps")
    for (w, h) in ((1, 1), (3, 2), (4, 3), (5, 3)):
        add("f_rgb_%dx%d.png" % (w, h), write_png(content(rng, h, w, 3), cyc(h, 4), 2), note="tiny")
        add("f_rgba_%dx%d.png" % (w, h), write_png(content(rng, h, w, 4), cyc(h, 3), 6), note="tiny")
        add("f_gray_%dx%d.png" % (w, h), write_png(content(rng, h, w, 1)[:, :, 0], cyc(h, 2), 0), note="tiny")
    add("f_rgb_3idat_50x40.png", write_png(content(rng, 40, 50, 3), cyc(40), 2, pieces=3), note="the zlib stream in three IDAT chunks")
    trns = chunk(b"tRNS", struct.pack(">HHH", 10, 20, 30))
    gama = chunk(b"gAMA", struct.pack(">I", 45455))
    add("f_rgb_trns_gama_20x20.png", write_png(content(rng, 20, 20, 3), cyc(20), 2, extra=(gama, trns)), note="ancillary chunks: tRNS is not expanded, gAMA not applied")
    # libpng's own encoder (adaptive filters)
    for name, mode, (w, h) in (("pil_rgb_200x120.png", "RGB", (200, 120)), ("pil_rgba_90x77.png", "RGBA", (90, 77)), ("pil_gray_150x101.png", "L", (150, 101))):
        c = {"RGB": 3, "RGBA": 4, "L": 1}[mode]
        a = content(rng, h, w, c)
        b = io.BytesIO()
        Image.fromarray(a[:, :, 0] if c == 1 else a, mode).save(b, "PNG", optimize=True)
        add(name, b.getvalue(), note="written by Pillow / libpng")

    # ---- what the device path does not take: UNSUPPORTED (the host decoder's), and damaged files: DECODE_FAILED
    a = content(rng, 12, 12, 3)
    good = write_png(a, cyc(12), 2)
    add("n_16bit.png", write_png(np.concatenate([a, a], axis=2), cyc(12), 2, depth=16), UNSUPPORTED, "16 bits per sample")
    b = io.BytesIO(); Image.fromarray(a).convert("P").save(b, "PNG"); add("n_palette.png", b.getvalue(), UNSUPPORTED, "colour type 3")
    b = io.BytesIO(); Image.fromarray(a).convert("LA").save(b, "PNG"); add("n_gray_alpha.png", b.getvalue(), UNSUPPORTED, "colour type 4")
    add("n_interlaced.png", write_png(a, cyc(12), 2, interlace=1), UNSUPPORTED, "Adam7 (the scanlines here are not even interlaced: refused by the header alone)")
    add("n_not_png.png", b"\xff\xd8\xff\xe0" + bytes(40), UNSUPPORTED, "not a PNG")
    bad = bytearray(good); bad[60] ^= 0x01
    add("d_idat_crc.png", bytes(bad), DECODE_FAILED, "one bit of the IDAT payload flipped: CRC")
    add("d_truncated.png", good[:len(good) - 30], DECODE_FAILED, "file cut inside IDAT")
    raw = bytearray(filter_rows(a, cyc(12))); raw[37 * 5] = 7            # row 5's filter byte (1 + 12 * 3 = 37 bytes per row)
    z = zlib.compress(bytes(raw), 9)
    add("d_filter7.png", b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 12, 12, 8, 2, 0, 0, 0)) + chunk(b"IDAT", z) + chunk(b"IEND", b""), DECODE_FAILED, "filter type 7")
    short = zlib.compress(filter_rows(a[:10], cyc(10)), 9)
    add("d_short_stream.png", b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 12, 12, 8, 2, 0, 0, 0)) + chunk(b"IDAT", short) + chunk(b"IEND", b""), DECODE_FAILED, "the stream ends two rows early")
    hdr = bytearray(good); hdr[16:20] = struct.pack(">I", 13)               # width changed, IHDR CRC now wrong
    add("d_ihdr_crc.png", bytes(hdr), DECODE_FAILED, "IHDR CRC")
    add("d_no_iend.png", good[:-12], DECODE_FAILED, "IEND missing")
    badcrc = bytearray(chunk(b"tEXt", b"Comment\x00x"))
    badcrc[-1] ^= 0x55
    add("d_ancillary_crc.png", write_png(a, cyc(12), 2, extra=(bytes(badcrc),)), DECODE_FAILED,
        "a damaged ancillary chunk: libpng's default would warn and skip it, Pillow refuses the file; left to the host decoder")

    for name, blob in files.items():
        with open(os.path.join(HERE, name), "wb") as f:
            f.write(blob)
    np.savez_compressed(os.path.join(HERE, "expected_pixels.npz"), **expected)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump({"pillow": Image.__version__ if hasattr(Image, "__version__") else "", "files": manifest}, f, indent=1, sort_keys=True)
    print(len(files), "files,", sum(len(b) for b in files.values()), "bytes;", len(expected), "with pixels")


if __name__ == "__main__":
    main()
