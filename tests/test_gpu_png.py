"""impgpu_image_decode_png (round 4's bounded experiment: host inflate, filters undone on the device) through the C ABI:
bit-exact against the committed Pillow pixels of tests/golden/png/, against the oracle on files made here, and against
Pillow at the sizes the experiment is quoted on."""
import io
import json
import os

import numpy as np
import pytest
from PIL import Image

import oracle_lib as O
from conftest import ROOT

pytestmark = pytest.mark.gpu

GOLD = os.path.join(ROOT, "tests", "golden", "png")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))["files"]
EXPECTED = np.load(os.path.join(GOLD, "expected_pixels.npz"))


def reference_order(arr):
    if arr.ndim == 2:
        return arr[:, :, None]
    if arr.shape[2] == 1:
        return arr
    return arr[:, :, [2, 1, 0] + ([3] if arr.shape[2] == 4 else [])]


def decode(imp, blob):
    rc, im = imp.Image.decode_png(blob)
    if rc:
        return rc, None
    a = im.numpy()
    return rc, (a if a.ndim == 3 else a[:, :, None])


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_golden_file(gpu, name):
    imp = gpu
    with open(os.path.join(GOLD, name), "rb") as f:
        blob = f.read()
    rc, got = decode(gpu, blob)
    assert rc == MANIFEST[name]["code"], MANIFEST[name]["note"]
    if rc == 0:
        assert np.array_equal(got, EXPECTED[name])
        orc, want = O.png_decode(blob)
        assert orc == 0 and np.array_equal(got, want if want.ndim == 3 else want[:, :, None])


def png_of(arr, **kw):
    b = io.BytesIO()
    Image.fromarray(arr[:, :, 0] if arr.shape[2] == 1 else arr).save(b, "PNG", **kw)
    return b.getvalue()


@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("size", [(640, 480), (1920, 1080), (1023, 769), (4096, 70), (61, 1500)])
def test_photo_sizes_match_pillow(gpu, c, size):
    imp = gpu
    from ngx_http_imgproc_amd.workloads import photo_like

    w, h = size
    rgb = photo_like(h, w, 3)
    a = rgb[:, :, 1:2] if c == 1 else rgb if c == 3 else np.dstack([rgb, (rgb[:, :, 0] // 2 + 100).astype(np.uint8)])
    blob = png_of(np.ascontiguousarray(a), compress_level=1)
    rc, got = decode(gpu, blob)
    assert rc == 0
    assert np.array_equal(got, reference_order(np.asarray(Image.open(io.BytesIO(blob)))))
    assert imp.png_info(blob) == (0, (w, h, c))


def test_outside_the_kernels_limits_is_left_to_the_host(gpu):
    imp = gpu
    a = np.zeros((4, 4097, 3), np.uint8)
    assert decode(gpu, png_of(a))[0] == O.UNSUPPORTED
    b = np.zeros((16385, 2, 1), np.uint8)
    assert decode(gpu, png_of(b))[0] == O.UNSUPPORTED


def test_many_files_in_flight(gpu):
    imp = gpu
    """the decode does not wait for the device: 40 files enqueued back to back, then checked (staging buffers alternate)"""
    rng = np.random.default_rng(5)
    blobs, handles = [], []
    for i in range(40):
        h, w, c = int(rng.integers(1, 300)), int(rng.integers(1, 300)), int(rng.choice([1, 3, 4]))
        blobs.append(png_of(rng.integers(0, 256, size=(h, w, c), dtype=np.uint8) // int(rng.integers(1, 40))))
    for blob in blobs:
        rc, im = imp.Image.decode_png(blob)
        assert rc == 0
        handles.append(im)
    for blob, im in zip(blobs, handles):
        got = im.numpy()
        assert np.array_equal(got if got.ndim == 3 else got[:, :, None], reference_order(np.asarray(Image.open(io.BytesIO(blob)))))


def test_decoded_png_enters_the_chain(gpu):
    imp = gpu
    """the decoded frame is an ordinary device frame: resize it and compare with the oracle's resize of Pillow's pixels"""
    from ngx_http_imgproc_amd.workloads import photo_like

    blob = png_of(photo_like(300, 400, 3))
    rc, im = imp.Image.decode_png(blob)
    assert rc == 0
    bgr = reference_order(np.asarray(Image.open(io.BytesIO(blob))))
    rc = im.resize("200,0", imp.Config())
    orc, want = O.resize(np.ascontiguousarray(bgr), "200,0")
    assert rc == 0 and orc == 0 and np.array_equal(im.numpy(), want)


def test_random_files_every_filter_any_geometry(gpu):
    """files written by tests/png_writer.py with a RANDOM filter type per row, random geometry (bands of 64 rows cut anywhere,
    groups of four pixels cut anywhere), colour type, deflate level (0 = stored blocks) and IDAT cuts; Pillow decodes them,
    the device must agree.  IMP_FUZZ_SCALE multiplies the number of files."""
    from png_writer import write_png

    imp = gpu
    n = 40 * int(os.environ.get("IMP_FUZZ_SCALE", "1"))
    rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little") if os.environ.get("IMP_FUZZ_RANDOM") else 20261005)
    for k in range(n):
        c = int(rng.choice([1, 3, 4]))
        w = int(rng.choice([int(rng.integers(1, 12)), int(rng.integers(1, 300)), int(rng.integers(1, 1500))]))
        h = int(rng.choice([int(rng.integers(1, 8)), int(rng.integers(1, 200)), int(rng.integers(60, 700))])) if w < 400 else int(rng.integers(1, 140))
        style = int(rng.integers(0, 3))
        if style == 0:
            arr = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
        elif style == 1:
            yy, xx = np.mgrid[0:h, 0:w]
            arr = np.stack([(xx * (k % 7 + 1) + yy * (ch + 2) + rng.integers(0, 5, size=(h, w))) % 256 for ch in range(c)], axis=2).astype(np.uint8)
        else:
            arr = np.full((h, w, c), int(rng.integers(0, 256)), dtype=np.uint8)
        kinds = [int(v) for v in (rng.integers(0, 5, size=h) if rng.integers(0, 3) else np.full(h, int(rng.integers(0, 5))))]
        blob = write_png(arr[:, :, 0] if c == 1 else arr, kinds, {1: 0, 3: 2, 4: 6}[c], pieces=int(rng.integers(1, 5)), level=int(rng.choice([0, 1, 6, 9])))
        want = reference_order(np.asarray(Image.open(io.BytesIO(blob))))
        assert np.array_equal(want, reference_order(arr if c > 1 else arr[:, :, 0])), "the test's own encoder wrote a file Pillow reads differently"
        rc, got = decode(imp, blob)
        assert rc == 0, (k, w, h, c)
        assert np.array_equal(got, want), (k, w, h, c, kinds[:8])
