"""GPU parity for the JPEG ENCODE end of the request (cvEncodeImage(".jpg") at bridge.c:704): the device writes the same
FILE, byte for byte, as the oracle -- which is pinned against Pillow's libjpeg-turbo -- and as the committed Pillow files."""
import json
import os

import numpy as np
import pytest

import oracle_lib as orc
from conftest import noise_image, smooth_image

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_enc")
CASES = json.load(open(os.path.join(GOLD, "manifest.json")))["cases"]
DATA = np.load(os.path.join(GOLD, "enc_cases.npz"))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%dx%d-q%d" % (c["width"], c["height"], c["channels"], c["quality"]))
def test_device_writes_the_file_pillow_wrote(gpu, case):
    i = case["case"]
    im = gpu.Image(DATA["in_%02d" % i])
    rc, got = im.encode_jpeg(case["quality"])
    assert rc == 0
    assert got == DATA["file_%02d" % i].tobytes()
    im.release()


SIZES = [(1, 1), (7, 9), (8, 8), (16, 16), (17, 33), (24, 40), (31, 257), (100, 75), (126, 224), (224, 126), (255, 255), (480, 640)]


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("c", [1, 3, 4])
def test_encode_matches_oracle(gpu, h, w, c):
    for q, arr in ((90, smooth_image(h, w, c) if c > 1 and h > 1 and w > 1 else noise_image(h, w, c, 5)), (75, noise_image(h, w, c, 1200 + h + w)), (100, noise_image(h, w, c, 7)),
                   (3, noise_image(h, w, c, 8))):
        rc_o, want = orc.jpeg_encode(arr, q)
        im = gpu.Image(arr)
        rc, got = im.encode_jpeg(q)
        assert rc == rc_o == 0
        assert got == want, (h, w, c, q, len(got), len(want))
        im.release()


def test_extreme_frames(gpu):
    """Saturated and alternating content: the longest codes, FF bytes in the entropy-coded segment (stuffing), runs of
    sixteen zeros (ZRL), quality 100 (divisor 8: the largest coefficients)."""
    h, w = 64, 96
    yy, xx = np.mgrid[0:h, 0:w]
    frames = [np.full((h, w, 3), 255, np.uint8), np.zeros((h, w, 3), np.uint8),
              np.where(((xx + yy) & 1)[..., None] == 0, 255, 0).astype(np.uint8).repeat(3, axis=2),
              np.where((xx & 8)[..., None] == 0, 255, 0).astype(np.uint8).repeat(3, axis=2),
              np.stack([(xx * 255 // (w - 1)), (yy * 255 // (h - 1)), ((xx ^ yy) * 4) % 256], -1).astype(np.uint8)]
    for arr in frames:
        for q in (100, 95, 50, 0):
            rc_o, want = orc.jpeg_encode(arr, q)
            im = gpu.Image(arr)
            rc, got = im.encode_jpeg(q)
            assert rc == rc_o == 0 and got == want, q
            im.release()


def test_full_hd_frame_many_strips(gpu):
    arr = smooth_image(1080, 1920, 3)
    arr = (arr.astype(int) + np.random.default_rng(3).integers(-20, 21, arr.shape)).clip(0, 255).astype(np.uint8)
    rc_o, want = orc.jpeg_encode(arr, 90)
    im = gpu.Image(arr)
    rc, got = im.encode_jpeg(90)
    assert rc == rc_o == 0 and got == want
    im.release()


def test_batch_and_refusals(gpu):
    rng = np.random.default_rng(5)
    frames = [noise_image(int(rng.integers(1, 200)), int(rng.integers(1, 300)), int(rng.choice([1, 3, 4])), 1300 + i) for i in range(40)]
    ims = [gpu.Image(f) for f in frames]
    res = gpu.batch_encode_jpeg(ims, 85)
    for f, (code, data) in zip(frames, res):
        assert code == 0 and data == orc.jpeg_encode(f, 85)[1]
    # a buffer that is too small: the code says so, the length says what it takes, nothing is written
    import ctypes as C

    small = (C.c_ubyte * 100)()
    n = C.c_size_t()
    rc = gpu.lib.impgpu_image_encode_jpeg(ims[0].h, 85, small, 100, C.byref(n))
    assert rc == gpu.IMP_ERROR_MALLOC_FAILED and n.value == len(res[0][1]) and bytes(small) == bytes(100)
    assert gpu.lib.impgpu_image_encode_jpeg(None, 85, small, 100, C.byref(n)) == gpu.IMP_ERROR_INVALID_ARGS
    for im in ims:
        im.release()


def test_request_end_to_end_jpeg_in_jpeg_out(gpu):
    """A whole request on the device: JPEG file -> decode -> resize -> JPEG file; the answer equals the reference's host
    pipeline restated by the oracle (decode, Resize(), encode)."""
    gold = os.path.join(os.path.dirname(GOLD), "jpeg")
    blob = open(os.path.join(gold, sorted(f for f in os.listdir(gold) if f.endswith(".jpg"))[0]), "rb").read()
    rc, im = gpu.Image.decode_jpeg(blob)
    assert rc == 0
    assert im.resize("48,0") == 0
    rc, got = im.encode_jpeg(80)
    rc_o, dec = orc.jpeg_decode(blob)
    rc_o2, small = orc.resize(dec, "48,0", 2000, 2000, 0)
    rc_o3, want = orc.jpeg_encode(small, 80)
    assert rc == rc_o == rc_o2 == rc_o3 == 0 and got == want
    im.release()


def test_encodes_from_several_threads(gpu):
    """Every thread has its own lane (stream, pool, staging); the shared pieces are the code tables (built once) and the device."""
    import threading

    frames = [noise_image(40 + 7 * i, 90 + 11 * i, [3, 1, 4][i % 3], 1400 + i) for i in range(8)]
    wants = [orc.jpeg_encode(f, 80 + i)[1] for i, f in enumerate(frames)]
    errors = []

    def worker(i):
        try:
            im = gpu.Image(frames[i])
            for _ in range(25):
                rc, got = im.encode_jpeg(80 + i)
                if rc != 0 or got != wants[i]:
                    errors.append((i, rc))
                    break
            im.release()
        except Exception as e:          # noqa: BLE001
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("h,w,c,q,kind", [(767, 1023, 3, 92, "smooth"), (900, 1200, 1, 75, "noise"), (1080, 1920, 4, 100, "noise"), (2160, 3840, 3, 86, "smooth"),
                                         (401, 3001, 3, 50, "noise"), (3001, 401, 3, 0, "smooth")])
def test_large_frames_many_workgroups(gpu, h, w, c, q, kind):
    """Frames of more than 2048 block slots: k_jpeg_enc_pack + k_jpeg_enc_stuff, a workgroup per 256 blocks / per 16 KB of the
    stream.  Sizes that leave partial segments, partial chunks and dummy blocks; noise at quality 100 (FF bytes everywhere)."""
    arr = noise_image(h, w, c, 1500 + h) if kind == "noise" else smooth_image(h, w, c)
    if kind == "smooth":
        arr = (arr.astype(int) + np.random.default_rng(h).integers(-9, 10, arr.shape)).clip(0, 255).astype(np.uint8)
    rc_o, want = orc.jpeg_encode(arr, q)
    im = gpu.Image(arr)
    rc, got = im.encode_jpeg(q)
    im.release()
    assert rc == rc_o == 0
    assert got == want, (h, w, c, q, len(got), len(want))


def test_batch_of_large_and_small_frames(gpu):
    frames = [noise_image(40, 50, 3, 1), smooth_image(1080, 1920, 3), noise_image(9, 9, 1, 2), smooth_image(700, 900, 4), noise_image(126, 224, 3, 3)]
    ims = [gpu.Image(f) for f in frames]
    for (code, data), f in zip(gpu.batch_encode_jpeg(ims, 88), frames):
        assert code == 0 and data == orc.jpeg_encode(f, 88)[1]
    for im in ims:
        im.release()


def test_encode_in_two_halves_with_a_decode_between_them(gpu):
    """impgpu_batch_encode_jpeg_begin / _finish (round 5): the files are the one-call form's although a prepared decode -- begun
    on the same stream, like a broker lane's next batch -- was enqueued behind the encode and runs while the files are fetched;
    a second encode begun before the first is finished is refused cleanly; _finish from another thread too."""
    import ctypes as C
    import threading

    from ngx_http_imgproc_amd._lib import CJpegPrepared
    from ngx_http_imgproc_amd.workloads import photo_like

    frames = [photo_like(168, 224, 31), photo_like(126, 224, 32), photo_like(1080, 1920, 33), noise_image(40, 56, 1, 3)]
    ims = [gpu.Image(f) for f in frames]
    want = gpu.batch_encode_jpeg(ims, 86)
    n = len(ims)
    hs = (C.c_void_p * n)(*[im.h.value for im in ims])
    enc = C.c_void_p()
    assert gpu.lib.impgpu_batch_encode_jpeg_begin(hs, n, 86, C.byref(enc)) == 0 and enc.value
    # the next batch's decode goes on the stream behind it
    blob = want[2][1]
    prepared = gpu.jpeg_unstuff(blob)
    assert prepared
    head = np.frombuffer(prepared[0], np.uint8)
    scan = np.concatenate([np.frombuffer(prepared[1], np.uint8), np.full(1024, 255, np.uint8)])
    f = (CJpegPrepared * 1)()
    f[0].head, f[0].head_size, f[0].scan, f[0].scan_size, f[0].registered = head.ctypes.data, head.size, scan.ctypes.data, len(prepared[1]), 0
    dec = C.c_void_p()
    assert gpu.lib.impgpu_batch_decode_jpeg_prepared_begin(f, 1, C.byref(dec)) == 0 and dec.value
    # a second encode may be begun (the thread's other staging buffer); with both holding answers a third one -- like anything
    # else that stages through pinned memory -- is refused, cleanly
    again, third = C.c_void_p(), C.c_void_p()
    assert gpu.lib.impgpu_batch_encode_jpeg_begin(hs, 2, 86, C.byref(again)) == 0 and again.value
    assert gpu.lib.impgpu_batch_encode_jpeg_begin(hs, n, 86, C.byref(third)) == gpu.IMP_ERROR_INVALID_ARGS and not third.value
    assert b"staging buffers" in gpu.lib.impgpu_last_error()
    caps = [gpu.lib.impgpu_jpeg_encode_bound(im.shape[1], im.shape[0], im.shape[2]) for im in ims]
    bufs = [np.empty(c, np.uint8) for c in caps]
    outs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    ccaps = (C.c_size_t * n)(*caps)
    lens = (C.c_size_t * n)()
    codes = (C.c_int * n)()
    other = []
    t = threading.Thread(target=lambda: other.append(gpu.lib.impgpu_batch_encode_jpeg_finish(C.byref(enc), outs, ccaps, lens, codes)))
    t.start(); t.join()
    assert other == [gpu.IMP_ERROR_INVALID_ARGS] and enc.value               # still the owner's
    assert gpu.lib.impgpu_batch_encode_jpeg_finish(C.byref(enc), outs, ccaps, lens, codes) == 0 and not enc.value
    for i in range(n):
        assert codes[i] == want[i][0] == 0
        assert bufs[i][:lens[i]].tobytes() == want[i][1]
    assert gpu.lib.impgpu_batch_encode_jpeg_finish(C.byref(again), outs, ccaps, lens, codes) == 0 and not again.value
    for i in range(2):
        assert codes[i] == 0 and bufs[i][:lens[i]].tobytes() == want[i][1]
    img = (C.c_void_p * 1)()
    code = (C.c_int * 1)()
    assert gpu.lib.impgpu_batch_decode_jpeg_finish(C.byref(dec), img, code) == 0 and code[0] == 0
    got = gpu.Image(handle=img[0])
    rc, ref = gpu.Image.decode_jpeg(blob)
    assert rc == 0 and np.array_equal(got.numpy(), ref.numpy())
    got.release(); ref.release()
    for im in ims:
        im.release()


@pytest.mark.parametrize("seg", ["1", "0"])
def test_thumbnails_in_segments_are_the_same_files(gpu, monkeypatch, seg):
    """k_jpeg_enc_huff_seg (round 5: a frame of 257 .. 2048 block slots cut into segments of up to 256 blocks, a workgroup each,
    bit offsets, the byte two segments share and the FF counts exchanged between them) against the oracle's files: block counts
    around every segment boundary, gray frames (one block per MCU), noise (long codes: FF bytes, also at the seams), smooth
    frames (a segment of a few hundred bits), high and low quality; forced on and forced off (the one-workgroup kernel), in one
    batch with frames of the other two kinds."""
    from ngx_http_imgproc_amd.workloads import photo_like

    monkeypatch.setenv("IMPGPU_JPEG_ENC_SEG", seg)
    shapes = [(168, 224, 3), (224, 224, 3), (126, 224, 3), (104, 104, 3), (112, 104, 3), (120, 104, 3),      # 924, 1176, 672, 294 (two segments), 294, 336 slots
              (160, 168, 1), (136, 136, 1), (256, 264, 1),                                                     # gray: 420, 289, 1056 blocks
              (299, 224, 4), (330, 224, 3)]                                                                     # 1596, 1764 slots: seven segments
    frames = []
    for k, (h, w, c) in enumerate(shapes):
        kind = k % 3
        f = photo_like(h, w, 40 + k) if kind == 0 else noise_image(h, w, 3, 60 + k) if kind == 1 else smooth_image(h, w, 3, seed=k)
        f = np.ascontiguousarray(f[:, :, :1]) if c == 1 else np.dstack([f, np.full((h, w, 1), 255, np.uint8)]) if c == 4 else f
        frames.append(f)
    for q in (86, 100, 20):
        ims = [gpu.Image(f) for f in frames]
        got = gpu.batch_encode_jpeg(ims, q)
        for (code, data), f in zip(got, frames):
            assert code == 0 and data == orc.jpeg_encode(f, q)[1], (f.shape, q)
        one = ims[1].encode_jpeg(q)
        assert one == (0, got[1][1])
        for im in ims:
            im.release()
    # with a tiny frame and a large one in the same call
    mixed = [noise_image(40, 50, 3, 1), frames[0], smooth_image(1080, 1920, 3), frames[7], frames[10]]
    ims = [gpu.Image(f) for f in mixed]
    for (code, data), f in zip(gpu.batch_encode_jpeg(ims, 90), mixed):
        assert code == 0 and data == orc.jpeg_encode(f, 90)[1]
    for im in ims:
        im.release()
