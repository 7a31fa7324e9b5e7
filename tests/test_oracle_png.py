"""oracle/orc_png.c (test infrastructure) pinned against third-party C: Pillow's PNG decoder (libpng + zlib) on the committed
files of tests/golden/png/ -- rows using every filter type, files libpng wrote, the refused and the damaged ones -- and on
files made here.  The oracle's inflate is its own (RFC 1951), so this also checks it against zlib."""
import io
import json
import os

import numpy as np
import pytest
from PIL import Image

import oracle_lib as O
from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "png")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))["files"]
EXPECTED = np.load(os.path.join(GOLD, "expected_pixels.npz"))


def blob_of(name):
    with open(os.path.join(GOLD, name), "rb") as f:
        return f.read()


def reference_order(arr):
    if arr.ndim == 2:
        return arr[:, :, None]
    if arr.shape[2] == 1:
        return arr
    return arr[:, :, [2, 1, 0] + ([3] if arr.shape[2] == 4 else [])]


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_oracle_on_golden_file(name):
    rc, got = O.png_decode(blob_of(name))
    assert rc == MANIFEST[name]["code"], MANIFEST[name]["note"]
    if rc == 0:
        got = got if got.ndim == 3 else got[:, :, None]
        assert got.shape == tuple(MANIFEST[name]["shape"])
        assert np.array_equal(got, EXPECTED[name])


def test_golden_vectors_are_pillows():
    """the committed pixels are what THIS Pillow decodes too (the fixtures were not edited by hand)"""
    for name, meta in MANIFEST.items():
        if meta["code"] == 0:
            assert np.array_equal(reference_order(np.asarray(Image.open(io.BytesIO(blob_of(name))))), EXPECTED[name]), name
        elif name.startswith("d_") and name not in ("d_no_iend.png", "d_short_stream.png"):
            # (those two: Pillow's own chunk reader pads a short stream and does not ask for IEND; libpng's png_read_image /
            # png_read_end -- what OpenCV calls -- fail on both, and a refused file goes to that decoder anyway)
            with pytest.raises(Exception):
                Image.open(io.BytesIO(blob_of(name))).load()


@pytest.mark.parametrize("mode,size,level", [("RGB", (97, 75), 6), ("RGBA", (64, 131), 1), ("L", (333, 40), 9), ("RGB", (258, 66), 0)])
def test_oracle_on_files_libpng_writes(mode, size, level):
    rng = np.random.default_rng(hash((mode, size)) & 0xffff)
    w, h = size
    c = {"RGB": 3, "RGBA": 4, "L": 1}[mode]
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([(xx * (2 + k) + yy * 3 + rng.integers(0, 9, size=(h, w))) % 256 for k in range(c)], axis=2).astype(np.uint8)
    b = io.BytesIO()
    Image.fromarray(a[:, :, 0] if c == 1 else a, mode).save(b, "PNG", compress_level=level)       # level 0: stored blocks
    rc, got = O.png_decode(b.getvalue())
    assert rc == 0
    got = got if got.ndim == 3 else got[:, :, None]
    assert np.array_equal(got, reference_order(a))
