"""A small PNG ENCODER for the tests (PNG specification 9.2: it applies a chosen filter type per row, deflates with zlib and
cuts the stream into IDAT chunks).  libpng's own encoder picks the row filters by a heuristic, so files written by Pillow do
not reach every predictor on demand; the DECODING -- the thing under test -- is never done with this file: expected pixels
come from Pillow (libpng).  Used by tests/golden/png/make_png_golden.py and tests/test_gpu_png.py."""
import struct
import zlib

import numpy as np


def chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def filter_rows(arr, kinds):
    """arr: H x W x C uint8 in file order (R,G,B,A); kinds[y] in 0..4 -> the filtered scanlines"""
    h, w, c = arr.shape
    flat = arr.reshape(h, w * c).astype(np.int32)
    out = bytearray()
    zero = np.zeros(w * c, dtype=np.int32)
    for y in range(h):
        cur, up = flat[y], (flat[y - 1] if y else zero)
        left = np.concatenate([np.zeros(c, np.int32), cur[:-c]])
        upleft = np.concatenate([np.zeros(c, np.int32), up[:-c]])
        k = kinds[y]
        if k == 0:
            pred = zero
        elif k == 1:
            pred = left
        elif k == 2:
            pred = up
        elif k == 3:
            pred = (left + up) // 2
        else:
            pred = np.array([paeth(int(a), int(b), int(cc)) for a, b, cc in zip(left, up, upleft)], dtype=np.int32)
        out.append(k)
        out += bytes(((cur - pred) & 255).astype(np.uint8))
    return bytes(out)


def write_png(arr, kinds, colour, depth=8, interlace=0, pieces=1, extra=(), level=9):
    h, w = arr.shape[:2]
    a = arr if arr.ndim == 3 else arr[:, :, None]
    ihdr = struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, interlace)
    z = zlib.compress(filter_rows(a, kinds), level)
    cuts = [len(z) * i // pieces for i in range(pieces + 1)]
    body = b"".join(chunk(b"IDAT", z[cuts[i]:cuts[i + 1]]) for i in range(pieces))
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + b"".join(extra) + body + chunk(b"IEND", b"")


