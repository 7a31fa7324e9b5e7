"""The JPEG ENCODE oracle (oracle/orc_jpeg_enc.c) pinned against third-party C: the files Pillow's libjpeg-turbo writes.
CPU only; the GPU encoder is compared with this oracle (and with the same golden files) in tests/test_gpu_jpeg_enc.py."""
import io
import json
import os

import numpy as np
import pytest

import oracle_lib as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_enc")
CASES = json.load(open(os.path.join(GOLD, "manifest.json")))["cases"]
DATA = np.load(os.path.join(GOLD, "enc_cases.npz"))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%dx%d-q%d" % (c["width"], c["height"], c["channels"], c["quality"]))
def test_oracle_writes_the_file_pillow_wrote(case):
    i = case["case"]
    rc, got = orc.jpeg_encode(DATA["in_%02d" % i], case["quality"])
    assert rc == 0
    assert got == DATA["file_%02d" % i].tobytes()


def test_oracle_refuses_what_libjpeg_refuses():
    a = np.zeros((4, 4, 3), np.uint8)
    assert orc.jpeg_encode(a[:, :, :2], 50)[0] != 0          # two channels
    rc, f = orc.jpeg_encode(a, 150)                            # OpenCV clamps the quality to 0..100
    assert rc == 0 and f == orc.jpeg_encode(a, 100)[1]
    assert orc.jpeg_encode(a, -5)[1] == orc.jpeg_encode(a, 0)[1]


def test_encode_then_decode_round_trip_is_close():
    """The two JPEG oracles against each other: decode(encode(x)) stays near x on a smooth frame at quality 95."""
    yy, xx = np.mgrid[0:48, 0:80]
    a = np.stack([(xx * 2 + yy) % 256, (xx + yy * 2) % 256, (xx * 3) % 256], -1).astype(np.uint8)
    a = (a // 8 * 4 + 60).astype(np.uint8)
    rc, f = orc.jpeg_encode(a, 95)
    rc2, back = orc.jpeg_decode(f)
    assert rc == 0 and rc2 == 0 and back.shape == a.shape
    assert np.abs(back.astype(int) - a.astype(int)).mean() < 6


def test_live_against_pillow_when_it_is_installed():
    PIL = pytest.importorskip("PIL")
    from PIL import Image

    rng = np.random.default_rng(11)
    for h, w, c, q in [(16, 16, 3, 75), (23, 57, 3, 88), (40, 24, 4, 30), (50, 50, 1, 60), (8, 24, 3, 100), (126, 224, 3, 90), (2, 300, 3, 5)]:
        a = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        b = io.BytesIO()
        if c == 1:
            Image.fromarray(a[:, :, 0], "L").save(b, format="JPEG", quality=q)
        else:
            Image.fromarray(np.ascontiguousarray(a[:, :, [2, 1, 0]])).save(b, format="JPEG", quality=q, subsampling=2)
        rc, got = orc.jpeg_encode(a, q)
        assert rc == 0 and got == b.getvalue(), (h, w, c, q)
