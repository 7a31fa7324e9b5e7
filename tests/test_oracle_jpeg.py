"""The JPEG oracle (oracle/orc_jpeg.c) pinned against third-party C: Pillow's decoder = libjpeg-turbo.

bridge.c:545-552 decodes every JPEG with cvDecodeImage = libjpeg (default parameters) + an R/B swap, so
`np.asarray(Image.open(f))[:, :, ::-1]` IS the reference's decoded frame up to the libjpeg build.  This is the one
part of the oracle whose last bit is pinned by code that shares nothing with it.
"""
import hashlib
import io
import json
import os

import numpy as np
import pytest

import oracle_lib as orc
from conftest import noise_image, smooth_image

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))
EXPECTED = np.load(os.path.join(GOLD, "expected_bgr.npz"))


def golden_blob(name):
    with open(os.path.join(GOLD, name + ".jpg"), "rb") as f:
        return f.read()


@pytest.mark.parametrize("case", MANIFEST["cases"], ids=[c["name"] for c in MANIFEST["cases"]])
def test_oracle_matches_committed_pillow_pixels(case):
    rc, got = orc.jpeg_decode(golden_blob(case["name"]))
    assert rc == 0
    assert list(got.shape) == case["shape"]
    assert hashlib.sha256(got.tobytes()).hexdigest() == case["sha256_bgr"]
    if case["name"] in EXPECTED.files:
        assert np.array_equal(got, EXPECTED[case["name"]])


def _pil():
    return pytest.importorskip("PIL.Image")


def _encode(arr, **kw):
    b = io.BytesIO()
    _pil().fromarray(arr).save(b, "JPEG", **kw)
    return b.getvalue()


def _pillow_bgr(blob):
    a = np.asarray(_pil().open(io.BytesIO(blob)))
    return a[:, :, None] if a.ndim == 2 else a[:, :, ::-1]


SIZES = [(16, 16), (17, 23), (1, 1), (2, 3), (8, 8), (33, 65), (100, 75), (3, 300), (300, 3), (5, 4), (240, 321)]


@pytest.mark.parametrize("sub", ["4:4:4", "4:2:2", "4:2:0"])
@pytest.mark.parametrize("kind", ["smooth", "noise"])
def test_oracle_matches_live_pillow_decode(sub, kind):
    """Every size x quality x restart interval, decoded by both: bit for bit."""
    for h, w in SIZES:
        arr = smooth_image(h, w, 3) if kind == "smooth" else noise_image(h, w, 3, 1)
        for q in (30, 75, 90, 100):
            for rst in (0, 1, 5):
                kw = dict(quality=q, subsampling=sub)
                if rst:
                    kw["restart_marker_blocks"] = rst
                blob = _encode(arr, **kw)
                rc, got = orc.jpeg_decode(blob)
                assert rc == 0, (h, w, q, rst)
                assert np.array_equal(got, _pillow_bgr(blob)), (h, w, q, rst)


def test_oracle_gray_and_optimized_tables_match_pillow():
    for h, w in [(16, 16), (17, 23), (1, 1), (100, 75)]:
        g = smooth_image(h, w, 3)[:, :, 1]
        for kw in (dict(quality=50), dict(quality=95, optimize=True), dict(quality=80, restart_marker_rows=1)):
            blob = _encode(g, **kw)
            rc, got = orc.jpeg_decode(blob)
            assert rc == 0 and got.shape == (h, w, 1)
            assert np.array_equal(got, _pillow_bgr(blob))


def test_oracle_info_and_coefficients():
    blob = golden_blob("c420_q90_dri4_95x51")
    rc, info = orc.jpeg_info(blob)
    assert rc == 0
    assert info == dict(width=95, height=51, components=3, hs=2, vs=2, restart_interval=4, mcux=6, mcuy=4)
    rc, y = orc.jpeg_coefficients(blob, 0)
    assert rc == 0 and y.shape == (8, 12, 8, 8)
    rc, cb = orc.jpeg_coefficients(blob, 1)
    assert rc == 0 and cb.shape == (4, 6, 8, 8)
    # a smooth frame at quality 90: the DC term dominates, most AC terms are zero
    assert np.count_nonzero(y[:, :, 0, 0]) > 0.9 * y.shape[0] * y.shape[1]
    assert np.count_nonzero(y) < 0.25 * y.size


def test_oracle_refuses_what_it_does_not_cover():
    arr = smooth_image(40, 40, 3)
    prog = _encode(arr, quality=90, progressive=True)
    assert orc.jpeg_decode(prog)[0] == orc.UNSUPPORTED
    cmyk = io.BytesIO()
    _pil().fromarray(np.dstack([arr, arr[:, :, :1]])).convert("CMYK").save(cmyk, "JPEG")
    assert orc.jpeg_decode(cmyk.getvalue())[0] == orc.UNSUPPORTED
    assert orc.jpeg_decode(b"\x89PNG\r\n\x1a\n" + b"\0" * 64)[0] == orc.UNSUPPORTED


def test_oracle_rejects_truncated_and_damaged_files_without_crashing():
    blob = golden_blob("c420_q90_dri4_95x51")
    for cut in (3, 20, 200, len(blob) // 2, len(blob) - 40):
        assert orc.jpeg_decode(blob[:cut])[0] in (orc.UNSUPPORTED, orc.DECODE_FAILED)
    rng = np.random.Generator(np.random.PCG64(7))
    for name in ("c420_q90_dri4_95x51", "c444_q90_48x40", "gray_q90_57x43"):
        src = golden_blob(name)
        for _ in range(300):
            b = bytearray(src)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(2, len(b)))] = int(rng.integers(0, 256))
            rc, got = orc.jpeg_decode(bytes(b))     # any verdict is fine; it must come back
            assert rc in (0, orc.UNSUPPORTED, orc.DECODE_FAILED, 2)
