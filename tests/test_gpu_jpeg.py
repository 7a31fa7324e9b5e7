"""impgpu_image_decode_jpeg on the device against the Pillow-pinned oracle (bridge.c:545-552's cvDecodeImage).

Bit-exact: the whole decode is integer arithmetic (Huffman, dequantisation, ISLOW IDCT, fancy upsampling, YCbCr tables).
"""
import hashlib
import io
import json
import os

import numpy as np
import pytest

import oracle_lib as orc
from conftest import ROOT, noise_image, smooth_image

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))
EXPECTED = np.load(os.path.join(GOLD, "expected_bgr.npz"))
MODES = ["device", "host", "auto"]      # IMPGPU_JPEG_HUFF: where the entropy decoding runs; auto (unset) = by the size of the launch


def golden_blob(name):
    with open(os.path.join(GOLD, name + ".jpg"), "rb") as f:
        return f.read()


def encode(arr, **kw):
    Image = pytest.importorskip("PIL.Image")
    b = io.BytesIO()
    Image.fromarray(arr).save(b, "JPEG", **kw)
    return b.getvalue()


@pytest.fixture(autouse=True)
def _device_entropy_unless_a_test_says_otherwise(monkeypatch):
    """Unset, a lone small file's Huffman stage runs on the calling thread (decode_group's size rule): the files of this
    module are small, and it is the DEVICE's entropy stage they are here to test."""
    monkeypatch.setenv("IMPGPU_JPEG_HUFF", "device")


@pytest.fixture(params=MODES)
def huff(request, monkeypatch):
    if request.param == "auto":
        monkeypatch.delenv("IMPGPU_JPEG_HUFF", raising=False)
    else:
        monkeypatch.setenv("IMPGPU_JPEG_HUFF", request.param)
    return request.param


def decode(gpu, blob):
    rc, im = gpu.Image.decode_jpeg(blob)
    if rc:
        return rc, None
    out = im.numpy()
    im.release()
    return rc, out


@pytest.mark.parametrize("case", MANIFEST["cases"], ids=[c["name"] for c in MANIFEST["cases"]])
def test_golden_files_decode_to_pillows_pixels(gpu, huff, case):
    """The committed files against the pixels Pillow's libjpeg-turbo produced for them (not against our oracle)."""
    rc, got = decode(gpu, golden_blob(case["name"]))
    assert rc == 0
    assert list(got.shape) == case["shape"]
    if case["name"] in EXPECTED.files:
        assert np.array_equal(got, EXPECTED[case["name"]])
    assert hashlib.sha256(got.tobytes()).hexdigest() == case["sha256_bgr"]


@pytest.mark.parametrize("sub", ["4:4:4", "4:2:2", "4:2:0"])
def test_sizes_qualities_restart_intervals(gpu, huff, sub):
    for h, w in [(16, 16), (17, 23), (1, 1), (2, 3), (8, 8), (33, 65), (100, 75), (3, 300), (300, 3), (5, 4), (240, 321), (64, 256), (65, 257), (128, 511)]:
        for kind, q, rst in (("smooth", 90, 0), ("noise", 75, 5), ("noise", 100, 0), ("smooth", 30, 1)):
            arr = smooth_image(h, w, 3) if kind == "smooth" else noise_image(h, w, 3, h + w)
            kw = dict(quality=q, subsampling=sub)
            if rst:
                kw["restart_marker_blocks"] = rst
            blob = encode(arr, **kw)
            rc_o, want = orc.jpeg_decode(blob)
            rc, got = decode(gpu, blob)
            assert rc == 0 and rc_o == 0, (h, w, kind, q, rst, rc)
            assert np.array_equal(got, want), (h, w, kind, q, rst)


def test_gray_files(gpu, huff):
    for h, w in [(16, 16), (17, 23), (1, 1), (100, 75), (64, 260), (300, 300)]:
        g = smooth_image(h, w, 3)[:, :, 1]
        for kw in (dict(quality=50), dict(quality=95, optimize=True), dict(quality=80, restart_marker_rows=1)):
            blob = encode(g, **kw)
            rc, got = decode(gpu, blob)
            assert rc == 0 and got.shape == (h, w, 1)
            assert np.array_equal(got, orc.jpeg_decode(blob)[1])


@pytest.mark.parametrize("h,w,sub,kind,q,extra", [
    (1080, 1920, "4:2:0", "smooth", 90, {}),                          # BASELINE configs[1]'s frame
    (1080, 1920, "4:2:0", "noise", 90, {}),
    (1080, 1920, "4:4:4", "smooth", 95, {}),
    (1080, 1920, "4:2:2", "noise", 60, dict(restart_marker_rows=1)),
    (1080, 1920, "4:2:0", "smooth", 85, dict(restart_marker_blocks=1)),   # an interval per MCU: 8160 intervals
    (2160, 3840, "4:2:0", "smooth", 90, {}),                          # configs[3]'s frame
    (2160, 3840, "4:2:0", "noise", 50, dict(optimize=True)),
])
def test_full_size_frames(gpu, huff, h, w, sub, kind, q, extra):
    arr = smooth_image(h, w, 3) if kind == "smooth" else noise_image(h, w, 3, 5)
    blob = encode(arr, quality=q, subsampling=sub, **extra)
    rc, got = decode(gpu, blob)
    assert rc == 0
    rc_o, want = orc.jpeg_decode(blob)
    assert rc_o == 0
    assert np.array_equal(got, want)


def test_cfg1_jpeg_crop_on_the_device(gpu):
    """BASELINE configs[0]: a 640x480 JPEG, crop to 320x240 -- decoded and cropped on the device, against Pillow's pixels."""
    blob = golden_blob("c420_q50_640x480")
    rc, im = gpu.Image.decode_jpeg(blob)
    assert rc == 0
    cfg = gpu.Config()
    rc, step = gpu.run_ops(im, cfg, crop="320px,240px,0px,0px")
    assert rc == 0
    got = im.numpy()
    rc_o, full = orc.jpeg_decode(blob)
    assert got.shape == (240, 320, 3) and np.array_equal(got, full[:240, :320])
    im.release()


def test_decode_feeds_the_operator_chain(gpu):
    """decode -> resize (AREA thumbnail) -> rotate -> gamma, all on the device, equals the oracle's decode + chain."""
    arr = smooth_image(540, 960, 3, seed=4)
    blob = encode(arr, quality=88, subsampling="4:2:0")
    rc, im = gpu.Image.decode_jpeg(blob)
    assert rc == 0
    cfg = gpu.Config(allow_experiments=True)
    rc, step = gpu.run_ops(im, cfg, resize="224,0", filters=["rotate=90", "gamma=1.6"])
    assert rc == 0
    _, o = orc.jpeg_decode(blob)
    _, o = orc.resize(o, "224,0")
    _, o = orc.filter(o, "rotate=90")
    _, o = orc.filter(o, "gamma=1.6")
    assert np.array_equal(im.numpy(), o)
    im.release()
    # a gray file takes the gray->BGR promotion of bridge.c:613-618 inside run_ops
    blob = encode(arr[:, :, 0], quality=80)
    rc, im = gpu.Image.decode_jpeg(blob)
    assert rc == 0 and im.shape[2] == 1
    rc, step = gpu.run_ops(im, cfg, resize="100,0", filters=["contrast=1.2"])
    assert rc == 0
    _, o = orc.jpeg_decode(blob)
    _, o = orc.resize(o, "100,0")
    o = orc.gray2bgr(o)
    _, o = orc.filter(o, "contrast=1.2")
    assert np.array_equal(im.numpy(), o)
    im.release()


def test_refused_files(gpu, huff):
    arr = smooth_image(40, 40, 3)
    assert decode(gpu, encode(arr, quality=90, progressive=True))[0] == gpu.IMP_ERROR_UNSUPPORTED
    assert decode(gpu, b"\x89PNG\r\n\x1a\n" + b"\0" * 64)[0] == gpu.IMP_ERROR_UNSUPPORTED
    blob = golden_blob("c420_q90_dri4_95x51")
    for cut in (3, 20, 200, len(blob) // 2, len(blob) - 40):
        assert decode(gpu, blob[:cut])[0] in (gpu.IMP_ERROR_UNSUPPORTED, gpu.IMP_ERROR_DECODE_FAILED)
    # the env keeps working after a refusal
    rc, got = decode(gpu, blob)
    assert rc == 0 and np.array_equal(got, EXPECTED["c420_q90_dri4_95x51"])


def test_damaged_files_same_verdict_and_pixels_as_the_oracle(gpu):
    """Bytes flipped anywhere: the device accepts exactly the files the oracle accepts, with the same pixels."""
    rng = np.random.Generator(np.random.PCG64(23))
    accepted = 0
    for name in ("c420_q90_dri4_95x51", "c444_q90_48x40", "gray_q90_57x43", "c420_q30_noise_64x64", "c420_q92_opt_120x90"):
        src = golden_blob(name)
        for _ in range(120):
            b = bytearray(src)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(2, len(b)))] = int(rng.integers(0, 256))
            diff = [(i, src[i], b[i]) for i in range(len(src)) if src[i] != b[i]]
            b = bytes(b)
            rc_o, want = orc.jpeg_decode(b)
            rc, got = decode(gpu, b)
            if rc_o == 0:
                assert rc == 0, (name, rc, diff, gpu.lib.impgpu_last_error())
                assert np.array_equal(got, want), name
                accepted += 1
            elif rc_o == orc.DECODE_FAILED:
                assert rc in (gpu.IMP_ERROR_DECODE_FAILED, gpu.IMP_ERROR_UNSUPPORTED), (name, rc)
            else:
                assert rc != 0, (name, rc_o, rc)
    assert accepted > 40


def test_decodes_from_several_threads(gpu):
    """Each thread has its own lane (stream, pool, staging): concurrent decodes do not disturb one another."""
    import threading

    blobs = [encode(smooth_image(200 + 16 * i, 300 + 8 * i, 3, seed=i), quality=85, subsampling="4:2:0") for i in range(8)]
    wants = [orc.jpeg_decode(b)[1] for b in blobs]
    errors = []

    def work(k):
        try:
            for rep in range(6):
                i = (k + rep) % len(blobs)
                rc, got = decode(gpu, blobs[i])
                if rc or not np.array_equal(got, wants[i]):
                    errors.append((k, i, rc))
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_batch_decode_mixed_files(gpu, huff):
    """impgpu_batch_decode_jpeg: files of every sampling, size and restart layout in ONE call, refused files among them;
    each gets the code and the pixels the single-file call gives."""
    blobs, wants = [], []
    for name in [c["name"] for c in MANIFEST["cases"]]:
        blobs.append(golden_blob(name))
    rng = np.random.Generator(np.random.PCG64(5))
    for k in range(24):
        h, w = int(rng.integers(1, 500)), int(rng.integers(1, 700))
        arr = smooth_image(h, w, 3, seed=k) if k % 2 else noise_image(h, w, 3, k)
        kw = dict(quality=int(rng.integers(20, 100)), subsampling=["4:4:4", "4:2:2", "4:2:0"][k % 3])
        if k % 4 == 0:
            kw["restart_marker_blocks"] = int(rng.integers(1, 9))
        blobs.append(encode(arr, **kw))
    blobs.append(encode(smooth_image(40, 40, 3), quality=90, progressive=True))      # UNSUPPORTED
    blobs.append(golden_blob("c420_q90_dri4_95x51")[:700])                           # truncated
    blobs.append(b"\x89PNG\r\n\x1a\n" + b"\0" * 64)
    blobs.append(encode(smooth_image(1080, 1920, 3), quality=90, subsampling="4:2:0"))
    res = gpu.batch_decode_jpeg(blobs)
    assert len(res) == len(blobs)
    for b, (code, im) in zip(blobs, res):
        rc_o, want = orc.jpeg_decode(b)
        if rc_o == 0:
            assert code == 0
            assert np.array_equal(im.numpy(), want)
            im.release()
        else:
            assert code in (gpu.IMP_ERROR_UNSUPPORTED, gpu.IMP_ERROR_DECODE_FAILED) and im is None
    assert gpu.batch_decode_jpeg([]) == []


def test_batch_decode_more_files_than_one_launch_takes(gpu):
    blob = golden_blob("c420_q90_67x45")
    want = EXPECTED["c420_q90_67x45"]
    res = gpu.batch_decode_jpeg([blob] * 300)
    assert all(code == 0 for code, _ in res)
    for i in (0, 255, 256, 299):
        assert np.array_equal(res[i][1].numpy(), want)
    for _, im in res:
        im.release()


def test_batch_download_one_wait_for_an_album(gpu):
    """impgpu_batch_download (the encoders' hand-over for every frame of an album): frames of different geometry and
    channel count, caller rows with their own pitch."""
    import ctypes as C

    frames = [noise_image(37, 53, 4, 1), noise_image(20, 31, 3, 2), noise_image(5, 7, 1, 3), noise_image(64, 64, 4, 4)]
    ims = [gpu.Image(f) for f in frames]
    n = len(ims)
    outs = [np.full((f.shape[0], f.shape[1] * f.shape[2] + 5), 0xEE, dtype=np.uint8) for f in frames]
    handles = (C.c_void_p * n)(*[im.h for im in ims])
    datas = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    steps = (C.c_int * n)(*[o.shape[1] for o in outs])
    assert gpu.lib.impgpu_batch_download(handles, n, datas, steps) == 0
    for f, o in zip(frames, outs):
        assert np.array_equal(o[:, : f.shape[1] * f.shape[2]].reshape(f.shape), f)
        assert np.all(o[:, f.shape[1] * f.shape[2]:] == 0xEE)          # the caller's padding is not touched
    assert gpu.lib.impgpu_batch_download(handles, n, datas, (C.c_int * n)(1, 1, 1, 1)) == gpu.IMP_ERROR_INVALID_ARGS
    for im in ims:
        im.release()


def test_batch_download_into_pinned_memory_is_written_by_the_kernel(gpu):
    """destinations inside impgpu_host_alloc memory get their rows straight from the gather kernel (no staging pass): same
    bytes, the caller's pitch and padding respected, odd row lengths and unaligned pitches included"""
    import ctypes as C

    frames = [noise_image(37, 53, 4, 1), noise_image(20, 31, 3, 2), noise_image(5, 7, 1, 3), noise_image(224, 224, 3, 4), noise_image(9, 33, 3, 5)]
    pads = [8, 5, 3, 0, 1]                                   # pitch = row bytes + pad: 4-aligned and not
    ims = [gpu.Image(f) for f in frames]
    n = len(ims)
    sizes = [f.shape[0] * (f.shape[1] * f.shape[2] + p) for f, p in zip(frames, pads)]
    offs = np.concatenate([[0], np.cumsum([(sz + 64 + 3) & ~3 for sz in sizes])])[:-1] + np.array([0, 1, 0, 2, 0])     # some destinations start on odd bytes
    total = int(offs[-1] + sizes[-1] + 64)
    gpu.lib.impgpu_host_alloc.restype = C.c_void_p
    base = gpu.lib.impgpu_host_alloc(total)
    assert base
    buf = np.ctypeslib.as_array((C.c_ubyte * total).from_address(base))
    buf[:] = 0xEE
    handles = (C.c_void_p * n)(*[im.h for im in ims])
    datas = (C.c_void_p * n)(*[base + int(o) for o in offs])
    steps = (C.c_int * n)(*[f.shape[1] * f.shape[2] + p for f, p in zip(frames, pads)])
    assert gpu.lib.impgpu_batch_download(handles, n, datas, steps) == 0
    touched = np.zeros(total, dtype=bool)
    for f, p, o in zip(frames, pads, offs):
        rb = f.shape[1] * f.shape[2]
        view = buf[int(o): int(o) + f.shape[0] * (rb + p)].reshape(f.shape[0], rb + p)
        assert np.array_equal(view[:, :rb].reshape(f.shape), f)
        for y in range(f.shape[0]):
            touched[int(o) + y * (rb + p): int(o) + y * (rb + p) + rb] = True
    assert np.all(buf[~touched] == 0xEE)                     # nothing outside the rows was written
    gpu.lib.impgpu_host_free(C.c_void_p(base))
    for im in ims:
        im.release()


def test_batches_begun_and_finished_apart(gpu):
    """impgpu_batch_decode_jpeg_begin / _finish: four batches in flight on one thread, a fifth refused, finished in another
    order than begun, other work enqueued in between -- every frame is what the one-call form gives"""
    import ctypes as C

    gold = os.path.join(ROOT, "tests", "golden", "jpeg")
    names = sorted(n for n in os.listdir(gold) if n.endswith(".jpg"))
    blobs = [open(os.path.join(gold, n), "rb").read() for n in names]
    groups = [blobs[0:5], blobs[5:9], blobs[9:16], blobs[16:19] + [b"\xff\xd8\xff\xe0 not a jpeg"]]
    os.environ["IMPGPU_JPEG_HUFF"] = "device"
    try:
        want = []
        for g in groups:
            res = gpu.batch_decode_jpeg(g)
            want.append([(code, None if im is None else im.numpy()) for code, im in res])
            for _, im in res:
                if im is not None:
                    im.release()
        held, handles = [], []
        for g in groups:
            n = len(g)
            arr = (C.c_char_p * n)(*g)
            sizes = (C.c_size_t * n)(*[len(b) for b in g])
            h = C.c_void_p()
            assert gpu.lib.impgpu_batch_decode_jpeg_begin(arr, sizes, n, C.byref(h)) == 0 and h.value
            held.append((arr, sizes, n))
            handles.append(h)
        extra = C.c_void_p()
        arr, sizes, n = held[0]
        assert gpu.lib.impgpu_batch_decode_jpeg_begin(arr, sizes, n, C.byref(extra)) == gpu.IMP_ERROR_INVALID_ARGS and not extra.value
        busy = gpu.Image(noise_image(64, 64, 4, 9))                # other work on the thread's stream between the halves
        assert busy.cv_resize(32, 32, gpu.INTER_AREA) == 0
        for k in (2, 0, 3, 1):
            arr, sizes, n = held[k]
            imgs = (C.c_void_p * n)()
            codes = (C.c_int * n)()
            assert gpu.lib.impgpu_batch_decode_jpeg_finish(C.byref(handles[k]), imgs, codes) == 0 and not handles[k].value
            for i in range(n):
                code, pixels = want[k][i]
                assert codes[i] == code
                if code == 0:
                    im = gpu.Image(handle=imgs[i])
                    assert np.array_equal(im.numpy(), pixels)
                    im.release()
                else:
                    assert not imgs[i]
        # all slots are free again
        h = C.c_void_p()
        arr, sizes, n = held[1]
        assert gpu.lib.impgpu_batch_decode_jpeg_begin(arr, sizes, n, C.byref(h)) == 0
        imgs = (C.c_void_p * n)()
        codes = (C.c_int * n)()
        assert gpu.lib.impgpu_batch_decode_jpeg_finish(C.byref(h), imgs, codes) == 0
        for i in range(n):
            if imgs[i]:
                gpu.Image(handle=imgs[i]).release()
        busy.release()
    finally:
        os.environ.pop("IMPGPU_JPEG_HUFF", None)


def test_jpeg_request_batch_end_to_end(gpu):
    """What bench.py --stream --jpeg device --jpeg-batch N does per batch: decode all files with one call, resize all
    decoded frames with one descriptor launch (resize=224,0 -> the reference's per-frame Resize(), bridge.c:588-604),
    download -- every thumbnail equals the oracle's decode + Resize()."""
    import ctypes as C

    rng = np.random.Generator(np.random.PCG64(77))
    blobs = []
    for k in range(12):
        h, w = int(rng.integers(230, 900)), int(rng.integers(230, 1200))
        arr = smooth_image(h, w, 3, seed=k) if k % 3 else noise_image(h, w, 3, k)
        blobs.append(encode(arr, quality=int(rng.integers(60, 96)), subsampling=["4:2:0", "4:2:2", "4:4:4"][k % 3]))
    res = gpu.batch_decode_jpeg(blobs)
    cfg = gpu.Config()
    items, outs = [], []
    for code, im in res:
        assert code == 0
        h, w, c = im.shape
        rc, (ow, oh, interp) = gpu.resize_geometry(w, h, "224,0", cfg)
        assert rc == 0
        o = gpu.Image(np.zeros((oh, ow, 3), np.uint8))
        outs.append(o)
        items.append((im.device_ptr, w, h, im.step, o.device_ptr, ow, oh, o.step))
    assert gpu.batch_resize_mixed(items, 3) == 0
    for b, (code, im), o in zip(blobs, res, outs):
        _, full = orc.jpeg_decode(b)
        _, want = orc.resize(full, "224,0")
        assert np.array_equal(o.numpy(), want)
        im.release(); o.release()
    cfg.release()


def _box_files():
    """JPEG files this image ships (sample photographs of python packages: camera files, not files Pillow wrote for us)."""
    import glob
    pats = ["/usr/local/lib/python3.10/dist-packages/sklearn/datasets/images/*.jpg",
            "/usr/local/lib/python3.10/dist-packages/matplotlib/mpl-data/sample_data/*.jpg",
            "/opt/conda/lib/python3.9/site-packages/skimage/data/*.jpg",
            "/opt/conda/lib/python3.9/site-packages/anaconda_navigator/static/images/*.jpg",
            "/opt/conda/lib/python3.9/site-packages/anaconda_navigator/static/images/logos/*.jpg",
            "/usr/share/javascript/highlight.js/styles/*.jpg"]
    return sorted(p for pat in pats for p in glob.glob(pat))


def test_photographs_found_on_the_box_decode_to_pillows_pixels(gpu, huff):
    """Files nobody made for this test -- other encoders, optimised tables, EXIF and ICC segments, odd sizes: each is either
    refused with the reason impgpu_jpeg_classify names (the cvDecodeImage fallback takes it) or decodes to exactly the pixels
    libjpeg-turbo gives (Pillow, converted to the B,G,R order cvDecodeImage returns), and the refusal is counted."""
    Image = pytest.importorskip("PIL.Image")
    import ctypes as C
    files = _box_files()
    if not files:
        pytest.skip("no JPEG files on this box")
    taken = refused = 0
    before = (C.c_ulonglong * 13)()
    gpu.lib.impgpu_jpeg_counters(before, 13)
    for path in files:
        blob = open(path, "rb").read()
        kind = gpu.lib.impgpu_jpeg_classify(blob, len(blob))
        rc, got = decode(gpu, blob)
        if kind != 0:
            assert rc in (gpu.IMP_ERROR_UNSUPPORTED, gpu.IMP_ERROR_DECODE_FAILED), (path, kind, rc)
            refused += 1
            continue
        im = Image.open(io.BytesIO(blob))
        if im.mode not in ("RGB", "L"):
            continue
        want = np.asarray(im)
        want = want[:, :, ::-1] if want.ndim == 3 else want[:, :, None]
        assert rc == 0, (path, rc)
        assert np.array_equal(got, want), path
        taken += 1
    after = (C.c_ulonglong * 13)()
    gpu.lib.impgpu_jpeg_counters(after, 13)
    assert sum(after[5:13]) - sum(before[5:13]) == refused
    assert taken > 0


def test_slots_of_a_thread_are_shared_and_a_batch_belongs_to_its_thread(gpu):
    """A thread has four slots for decodes in flight (impgpu.h).  With three batches begun, a one-call batch of 32 files
    stays whole instead of failing for want of a second slot; with four begun, a further decode is refused (INVALID_ARGS),
    not corrupted; _finish from another thread is refused and the batch is still its owner's."""
    import ctypes as C
    import threading
    lib = gpu.lib
    blob = encode(smooth_image(64, 64, 3), quality=90)
    want = decode(gpu, blob)[1]
    n = 32
    blobs = (C.c_char_p * n)(*([blob] * n))
    sizes = (C.c_size_t * n)(*([len(blob)] * n))
    handles = []
    for _ in range(3):
        h = C.c_void_p()
        assert lib.impgpu_batch_decode_jpeg_begin(blobs, sizes, 2, C.byref(h)) == 0
        handles.append(h)
    images = (C.c_void_p * n)()
    codes = (C.c_int * n)()
    assert lib.impgpu_batch_decode_jpeg(blobs, sizes, n, images, codes) == 0          # three slots held: stays whole
    assert all(c == 0 for c in codes)
    for i in range(n):
        p = C.c_void_p(images[i])
        lib.impgpu_image_release(C.byref(p))
    h4 = C.c_void_p()
    assert lib.impgpu_batch_decode_jpeg_begin(blobs, sizes, 2, C.byref(h4)) == 0
    one = C.c_void_p()
    assert lib.impgpu_image_decode_jpeg(blob, len(blob), C.byref(one)) == gpu.IMP_ERROR_INVALID_ARGS   # all four held
    # another thread may not finish this thread's batch
    seen = []
    def other():
        im2 = (C.c_void_p * 2)()
        c2 = (C.c_int * 2)()
        seen.append(lib.impgpu_batch_decode_jpeg_finish(C.byref(h4), im2, c2))
    t = threading.Thread(target=other)
    t.start(); t.join()
    assert seen == [gpu.IMP_ERROR_INVALID_ARGS] and h4.value
    for h in handles + [h4]:
        im2 = (C.c_void_p * 2)()
        c2 = (C.c_int * 2)()
        assert lib.impgpu_batch_decode_jpeg_finish(C.byref(h), im2, c2) == 0 and list(c2) == [0, 0]
        for i in range(2):
            img = gpu.Image.__new__(gpu.Image)
            p = C.c_void_p(im2[i])
            lib.impgpu_image_release(C.byref(p))
    assert np.array_equal(decode(gpu, blob)[1], want)                                  # the slots are all free again


def _pillow_bgr(blob):
    Image = pytest.importorskip("PIL.Image")
    a = np.asarray(Image.open(io.BytesIO(blob)))
    return a if a.ndim == 2 else np.ascontiguousarray(a[:, :, ::-1])


@pytest.mark.parametrize("pinned", [False, True])
def test_files_unstuffed_by_the_caller_decode_to_the_same_pixels(gpu, huff, pinned):
    """impgpu_batch_decode_jpeg_prepared (round 5: the worker of a broker takes the scan out of its byte stuffing while it
    copies the file into shared memory): the frames of impgpu_batch_decode_jpeg -- and Pillow's -- whether the scans are
    staged or go to the device from page-locked memory, whole files and prepared ones in one launch."""
    from ngx_http_imgproc_amd.workloads import photo_like

    files = [encode(photo_like(480, 640, seed=3), quality=90),
             encode(photo_like(720, 1280, seed=4), quality=92, subsampling="4:4:4"),
             encode(photo_like(600, 800, seed=5)[:, :, 0], quality=95),
             encode(photo_like(1080, 1920, seed=6), quality=85, optimize=True),
             encode(photo_like(300, 400, seed=7), quality=90, restart_marker_blocks=5),       # goes as it is (restart interval)
             golden_blob("c420_q90_67x45")]                                                    # goes as it is (small)
    prepared = [gpu.jpeg_unstuff(f) or f for f in files]
    assert [isinstance(p, tuple) for p in prepared] == [True, True, True, True, False, False]
    plain = gpu.batch_decode_jpeg(files)
    got = gpu.batch_decode_jpeg_prepared(prepared, pinned=pinned)
    for f, (c0, a), (c1, b) in zip(files, plain, got):
        assert c0 == 0 and c1 == 0
        x, y = a.numpy(), b.numpy()
        a.release(); b.release()
        assert np.array_equal(x, y)
        want = _pillow_bgr(f)
        assert np.array_equal(y if y.shape[2] > 1 else y[:, :, 0], want)
    # one at a time too (the lone request of a lone worker)
    c, im = gpu.batch_decode_jpeg_prepared([prepared[0]], pinned=pinned)[0]
    assert c == 0 and np.array_equal(im.numpy(), _pillow_bgr(files[0]))
    im.release()


def test_prepared_files_the_device_defers_or_refuses(gpu, monkeypatch):
    """Dense blocks (quality-100 noise) keep their Huffman stage on the host even inside a device launch: a prepared file is
    put back into its stuffing for that.  A head that does not end at its scan, or announces restart intervals, is the
    caller's mistake; a damaged scan gets the verdict the file would have got."""
    monkeypatch.delenv("IMPGPU_JPEG_HUFF", raising=False)          # (forced to the device, nothing is deferred)
    rng = np.random.default_rng(11)
    noise = encode(rng.integers(0, 256, (480, 640, 3), dtype=np.uint8), quality=100, subsampling="4:4:4")
    ok = encode(smooth_image(480, 640, 3, seed=2), quality=90)
    pn, pk = gpu.jpeg_unstuff(noise), gpu.jpeg_unstuff(ok)
    assert pn and pk
    import ctypes as C

    def deferred():
        cnt = (C.c_ulonglong * 16)()
        assert gpu.lib.impgpu_jpeg_counters(cnt, 16) == 0
        return cnt[3]

    before = deferred()
    (c0, a), (c1, b) = gpu.batch_decode_jpeg_prepared([pn, pk], pinned=True)
    assert c0 == 0 and c1 == 0
    assert np.array_equal(a.numpy(), _pillow_bgr(noise)) and np.array_equal(b.numpy(), _pillow_bgr(ok))
    a.release(); b.release()
    assert deferred() > before                                     # the noise file did take the deferred path
    head, scan = pk
    (c, im), = gpu.batch_decode_jpeg_prepared([(head[:-3], scan)])
    assert c in (gpu.IMP_ERROR_INVALID_ARGS, gpu.IMP_ERROR_DECODE_FAILED) and im is None
    dri = encode(smooth_image(480, 640, 3, seed=2), quality=90, restart_marker_blocks=4)
    hd = dri[:dri.index(b"\xff\xda") + 14]
    (c, im), = gpu.batch_decode_jpeg_prepared([(hd, scan)])
    assert c == gpu.IMP_ERROR_INVALID_ARGS and im is None
    # damage: the scan cut in half / bytes flipped -> the verdict and (if any) the pixels of the file damaged the same way
    cut = scan[:len(scan) // 2]
    (c, im), = gpu.batch_decode_jpeg_prepared([(head, cut)], pinned=True)
    restuffed = head + cut.replace(b"\xff", b"\xff\x00") + b"\xff\xd9"
    c_ref, ref = decode(gpu, restuffed)
    assert c == c_ref
    if c == 0:
        assert np.array_equal(im.numpy(), ref)
        im.release()


def test_frames_taken_ahead_of_their_verdicts(gpu, monkeypatch):
    """impgpu_batch_decode_jpeg_pending (round 5): a request whose operators and answer are enqueued behind its decode and
    that waits once gives the file the two-wait form gives; a scan that turns out damaged comes back with the decode's code
    and no answer -- whatever had been made of its frame meanwhile; a file refused at its header (no frame to run ahead
    with) takes the two-wait form inside the same call."""
    from ngx_http_imgproc_amd.workloads import photo_like

    cfg = gpu.Config()
    blob = encode(photo_like(480, 640, seed=9), quality=90)
    rc, im = gpu.Image.decode_jpeg(blob)
    assert rc == 0
    assert gpu.run_ops(im, cfg, resize="224,0")[0] == 0
    rc, want = im.encode_jpeg(86)
    im.release()
    assert rc == 0
    for _ in range(3):
        assert gpu.jpeg_request_one_wait(blob, cfg, 86, resize="224,0") == (0, want)
    # filters too (the whole operator segment runs on a frame whose pixels are not there yet when it is enqueued)
    rc, im = gpu.Image.decode_jpeg(blob)
    assert gpu.run_ops(im, cfg, crop="4,3,c,c", filters=["gamma=1.3", "rotate=90"])[0] == 0
    rc, want2 = im.encode_jpeg(70)
    im.release()
    assert gpu.jpeg_request_one_wait(blob, cfg, 70, crop="4,3,c,c", filters=["gamma=1.3", "rotate=90"]) == (0, want2)
    # a scan damaged in its middle: the verdict arrives behind the answer that was made of the frame, and wins
    bad = bytearray(blob)
    at = len(bad) // 2
    bad[at:at + 64] = bytes(64)
    rc_ref, _ = decode(gpu, bytes(bad))
    rc, out = gpu.jpeg_request_one_wait(bytes(bad), cfg, 86, resize="224,0")
    assert rc == rc_ref and (rc != 0) == (out is None)
    # no frame ahead of the verdict for a progressive file: the code of the plain decode
    prog = encode(photo_like(480, 640, seed=9), quality=90, progressive=True)
    rc, out = gpu.jpeg_request_one_wait(prog, cfg, 86, resize="224,0")
    assert rc == gpu.IMP_ERROR_UNSUPPORTED and out is None
    # the thread's slots and buffers are its own again
    assert gpu.jpeg_request_one_wait(blob, cfg, 86, resize="224,0") == (0, want)


def test_dense_streams_forced_onto_the_device_are_exact_or_refused(gpu):
    """Files of dense blocks (noise at quality 90-100: 300-700 bits per block) keep their Huffman stage on the host by default
    (DENSE_BITS_PER_BLOCK); FORCED onto the device they are the chain's worst case -- hardly a walk falls into step inside its
    overlap, nearly every chunk is reached by an explicit state.  Round 5 found k_jpeg_select's look-back taking "every candidate
    leads to the same candidate" for "so does the true state" there; the guess is now checked against the predecessor's final
    word (JPEG_ST_CHAIN_GUESS).  So: Pillow's pixels, or a clean refusal (the caller's cvDecodeImage path) -- never other pixels;
    and the file that showed it is refused."""
    Image = pytest.importorskip("PIL.Image")
    seen_refusal = False
    for (h, w) in [(240, 321), (512, 512), (333, 777), (64, 2048), (600, 200)]:
        for q in (100, 95, 90):
            for sub in ("4:2:0", "4:2:2", "4:4:4"):
                blob = encode(noise_image(h, w, 3, h + w + q), quality=q, subsampling=sub)
                rc, got = decode(gpu, blob)
                if rc == 0:
                    want = np.asarray(Image.open(io.BytesIO(blob)))[:, :, ::-1]
                    assert np.array_equal(got, want), (h, w, q, sub)
                else:
                    assert rc == gpu.IMP_ERROR_DECODE_FAILED, (h, w, q, sub, rc)
                    seen_refusal = seen_refusal or (h, w, q, sub) == (512, 512, 95, "4:2:0")
    assert seen_refusal
    # by default (the entropy stage's place chosen per file) the same file is decoded, on the host
    os.environ.pop("IMPGPU_JPEG_HUFF", None)
    blob = encode(noise_image(512, 512, 3, 512 + 512 + 95), quality=95, subsampling="4:2:0")
    rc, got = decode(gpu, blob)
    assert rc == 0 and np.array_equal(got, np.asarray(Image.open(io.BytesIO(blob)))[:, :, ::-1])


def test_files_nobody_chose_every_way_of_decoding_them(gpu, monkeypatch):
    """A seeded sample of tools/jpeg_random_sweep.py (which runs 158 files): random sizes, content, quality, sampling, optimised
    tables, restart intervals, gray -- one at a time, as one batch, prepared by the caller (staged and pinned): Pillow's pixels."""
    Image = pytest.importorskip("PIL.Image")
    from ngx_http_imgproc_amd.workloads import photo_like

    monkeypatch.delenv("IMPGPU_JPEG_HUFF", raising=False)          # (the launch's size decides, as in production)
    rng = np.random.default_rng(7)
    files, wants = [], []
    while len(files) < 40:
        i = len(files)
        h, w = int(rng.integers(8, 900)), int(rng.integers(8, 1200))
        kind = int(rng.integers(0, 3))
        a = photo_like(h, w, i) if kind == 0 else smooth_image(h, w, 3, seed=i) if kind == 1 else noise_image(h, w, 3, i)
        kw = dict(quality=int(rng.choice([30, 60, 85, 92])), subsampling=str(rng.choice(["4:2:0", "4:2:2", "4:4:4"])))
        r = int(rng.integers(0, 4))
        if r == 1:
            kw["optimize"] = True
        if r == 2:
            kw["restart_marker_blocks"] = int(rng.integers(1, 40))
        try:
            blob = encode(a[:, :, 0], quality=kw["quality"]) if rng.integers(0, 8) == 0 else encode(a, **kw)
        except OSError:
            continue
        d = np.asarray(Image.open(io.BytesIO(blob)))
        files.append(blob)
        wants.append(d[:, :, None] if d.ndim == 2 else d[:, :, ::-1])
    prepared = [gpu.jpeg_unstuff(f) or f for f in files]
    assert 5 < sum(isinstance(p, tuple) for p in prepared) < 40
    for tag, res in (("lone", [gpu.batch_decode_jpeg([f])[0] for f in files]), ("batch", gpu.batch_decode_jpeg(files)),
                     ("prepared", gpu.batch_decode_jpeg_prepared(prepared)), ("pinned", gpu.batch_decode_jpeg_prepared(prepared, pinned=True))):
        for k, ((code, im), want) in enumerate(zip(res, wants)):
            assert code == 0, (tag, k, code)
            got = im.numpy()
            im.release()
            assert np.array_equal(got, want), (tag, k, want.shape)
