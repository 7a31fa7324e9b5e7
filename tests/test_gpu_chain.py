"""GPU parity for Crop and RunJob's operator segment (bridge.c:574-656) vs the oracle chain."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as orc
from conftest import noise_image, smooth_image

pytestmark = pytest.mark.gpu


def oracle_chain(arr, crop=None, gravity=None, resize=None, simple=0, filters=(), overlay=None, wm=None, flatten=0,
                 allow=1, max_w=2000, max_h=2000):
    cur = arr
    if crop is not None:
        rc, cur = orc.crop(cur, crop, gravity)
        if rc:
            return rc, 3, None
    if resize is not None:
        rc, cur = orc.resize(cur, resize, max_w, max_h, simple)
        if rc:
            return rc, 4, None
    if cur.shape[2] == 1:
        cur = orc.gray2bgr(cur)
    for f in filters:
        rc, cur = orc.filter(cur, f, allow)
        if rc:
            return rc, 5, None
    if overlay is not None:
        rc, cur = orc.watermark(cur, overlay, *wm)
        if rc:
            return rc, 6, None
    if flatten and cur.shape[2] == 4:
        cur = orc.blend_with_paper(cur)
    return 0, 7, cur


CROPS = ["320px,240px,0px,0px", "320px,240px", "1,1", "16,9,l,t", "4,3,r,b", "400px,200px,46px,0px", "1,2,c,c",
         "100px,100px,c,c", "640px,480px", "3,1,0px,20px"]


@pytest.mark.parametrize("args", CROPS)
@pytest.mark.parametrize("c", [1, 3, 4])
def test_crop_bit_exact(gpu, args, c):
    arr = noise_image(480, 640, c, 40)
    rc_o, want = orc.crop(arr, args)
    im = gpu.Image(arr)
    rc = im.crop(args)
    assert rc == rc_o == 0
    assert np.array_equal(im.numpy(), want)
    im.release()


def test_crop_with_gravity_and_errors(gpu):
    arr = noise_image(120, 160, 3, 41)
    for args, grav in [("1,1,c,c", "r,b"), ("1,1", "l,t"), ("100px,50px", "10px,20px"), ("1,1", "r"), ("1,1", "xx"),
                       ("0,0,320,240", None), ("500px,10px", None), ("10px,10", None), ("", None), ("1,1,q,t", None),
                       ("50px,50px,150px,0px", None), ("50px,50px,-5px,0px", None)]:
        rc_o, want = orc.crop(arr, args, grav)
        im = gpu.Image(arr)
        rc = im.crop(args, grav)
        assert rc == rc_o, (args, grav, rc, rc_o)
        if rc == 0:
            assert np.array_equal(im.numpy(), want)
        im.release()


CHAINS = [
    dict(crop="16,9", resize="160,90", filters=["rotate=90", "gamma=1.8"]),
    dict(resize="100,0", filters=["modulate=20,130,90", "colorize=203040,0.3", "contrast=1.2"]),
    dict(crop="1,1,c,c", gravity="r,b", resize="64,64", filters=["gotham=1", "flip=10", "kelvin=1"]),
    dict(filters=["blur=1.5", "lomo=1", "rotate=270", "vignette=0.5"]),
    dict(crop="200px,100px,7px,9px", filters=["rainbow=mid", "scanline=0.4,0.2,2,2"]),
    dict(crop="150px,111px,3px,5px"),
    dict(resize="300,300,up", filters=["gradmap=001122,ffeedd"]),
    dict(resize="120,90", simple=1),
    dict(crop="2,1", resize="0,40", filters=["rotate=180", "blur=0.8", "gamma=0.7"], flatten=1),
    # Resize + a leading rotation (+ the watermark when it is the only filter) leave as ONE launch for BGRA frames
    dict(resize="100,0", filters=["rotate=270"]),
    dict(resize="150,100", filters=["rotate=90"]),
    dict(crop="301px,177px,13px,9px", resize="99,0", filters=["rotate=180"]),
    dict(resize="160,120", filters=["rotate=90"]),              # exact 2x: the box kernels' arithmetic, step by step
]


@pytest.mark.parametrize("chain", CHAINS, ids=lambda d: "-".join(k for k in d))
@pytest.mark.parametrize("c", [1, 3, 4])
def test_run_ops_matches_oracle_chain(gpu, chain, c):
    arr = smooth_image(240, 320, c) if c > 1 else noise_image(240, 320, 1, 42)
    ov = noise_image(20, 48, 4, 43)
    wm = ("r", "b", 4, 4, 60)
    has_vig = any(f.startswith("vignette") for f in chain.get("filters", ()))
    rc_o, step_o, want = oracle_chain(arr, overlay=ov, wm=wm, **chain)
    cfg = gpu.Config(allow_experiments=True, max_filters=5)
    assert cfg.prepare_watermark(ov, *wm) == 0
    im = gpu.Image(arr)
    kw = dict(chain)
    kw["need_flatten"] = kw.pop("flatten", 0)
    rc, step = gpu.run_ops(im, cfg, **kw)
    assert rc == rc_o == 0, (rc, step)
    got = im.numpy()
    if has_vig:
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    else:
        assert np.array_equal(got, want)
    im.release(); cfg.release()


def test_run_ops_error_steps(gpu):
    arr = noise_image(60, 80, 3, 44)
    cfg = gpu.Config(max_w=100, max_h=100, max_filters=2)
    cases = [
        (dict(crop="0,0,320,240"), 50, 3),
        (dict(resize="0,0"), 50, 4),
        (dict(resize="500,500,up"), 54, 4),
        (dict(filters=["nosuch=1"]), 52, 5),
        (dict(filters=["gotham=1"]), 52, 5),            # experiments off
        (dict(filters=["gamma=1", "gamma=1", "gamma=1"]), 55, 0),
    ]
    for kw, code, step in cases:
        im = gpu.Image(arr)
        rc, st = gpu.run_ops(im, cfg, **kw)
        assert (rc, st) == (code, step), (kw, rc, st)
        im.release()


def test_cfg1_crop_plumbing(gpu):
    """BASELINE configs[0] in IMP grammar: 640x480 3-channel frame, crop=320px,240px,0px,0px."""
    arr = smooth_image(480, 640, 3)
    im = gpu.Image(arr)
    assert im.crop("0,0,320,240") == 50              # the literal BASELINE spelling is invalid IMP grammar
    assert im.crop("320px,240px,0px,0px") == 0
    got = im.numpy()
    assert got.shape == (240, 320, 3) and np.array_equal(got, arr[:240, :320])
    im.release()


def test_cfg3_chain_batch(gpu):
    """BASELINE configs[2]: resize(960x540, AREA 2x2) -> rotate 90 -> watermark, on a small batch at full size."""
    n = 3
    frames = [noise_image(1080, 1920, 4, 50 + i) for i in range(n)]
    ov = noise_image(64, 256, 4, 0xFF)
    ov[:, :, 3] = np.linspace(0, 255, 256).astype(np.uint8)[None, :]
    cfg = gpu.Config()
    assert cfg.prepare_watermark(ov, "r", "b", 16, 16, 60) == 0
    src = gpu.Image(np.concatenate(frames, axis=0))
    dst = gpu.Image(np.zeros((n * 960, 540, 4), np.uint8))
    gpu.batch_resize_rotate_watermark(src.device_ptr, 1080 * 1920 * 4, 1920, 1080, 1920 * 4, dst.device_ptr,
                                      960 * 540 * 4, 540 * 4, 960, 540, 90, cfg, 4, n)
    out = dst.numpy().reshape(n, 960, 540, 4)
    for i in range(n):
        rc, step, want = oracle_chain(frames[i], resize="960,540", filters=["rotate=90"], overlay=ov, wm=("r", "b", 16, 16, 60))
        assert rc == 0 and np.array_equal(out[i], want), i
    src.release(); dst.release(); cfg.release()


def test_cfg3_chain_batch_224_variant(gpu):
    """SURVEY 8(d) cfg3, second variant: resize=224,224 (general AREA) -> rotate 90 -> the 256x64 watermark, which is
    wider than the 224-pixel target and gets clipped by the ROI rule (bridge.c:257-271)."""
    n = 2
    frames = [noise_image(1080, 1920, 4, 55 + i) for i in range(n)]
    ov = noise_image(64, 256, 4, 0xFF)
    ov[:, :, 3] = np.linspace(0, 255, 256).astype(np.uint8)[None, :]
    cfg = gpu.Config()
    assert cfg.prepare_watermark(ov, "r", "b", 16, 16, 60) == 0
    src = gpu.Image(np.concatenate(frames, axis=0))
    dst = gpu.Image(np.zeros((n * 224, 224, 4), np.uint8))
    gpu.batch_resize_rotate_watermark(src.device_ptr, 1080 * 1920 * 4, 1920, 1080, 1920 * 4, dst.device_ptr,
                                      224 * 224 * 4, 224 * 4, 224, 224, 90, cfg, 4, n)
    out = dst.numpy().reshape(n, 224, 224, 4)
    for i in range(n):
        rc, step, want = oracle_chain(frames[i], resize="224,224", filters=["rotate=90"], overlay=ov, wm=("r", "b", 16, 16, 60))
        assert rc == 0 and np.array_equal(out[i], want), i
    src.release(); dst.release(); cfg.release()


@pytest.mark.parametrize("rotate", [0, 90, 180, 270])
@pytest.mark.parametrize("geom", [((96, 128), (64, 48)), ((96, 130), (64, 48)), ((70, 90), (45, 35)), ((66, 70), (35, 33)),
                                  ((300, 500), (150, 77)), ((96, 128), (32, 32))])
@pytest.mark.parametrize("c", [4, 3])
def test_batch_resize_rotate_watermark_all_turns(gpu, rotate, geom, c):
    """Batch chain API: the fused 2x2-box + quarter-turn kernel (exact halves), the row-streaming AREA kernel with the turn
    and the blend on its stores (any other shrink, BGRA and -- round 3 -- BGR, what every JPEG decodes to; several column
    strips, a partial last one) and the unfused fallback (integer factors other than 2) agree with the oracle."""
    (sh, sw), (rw, rh) = geom
    if c == 3 and (sw * 3) % 4:
        sw += 4 - sw % 4                                  # batch frames sit back to back: keep the BGR rows 4-byte aligned
    n = 3
    frames = [noise_image(sh, sw, c, 60 + i) for i in range(n)]
    ov = noise_image(10, 14, 4, 61)
    cfg = gpu.Config()
    assert cfg.prepare_watermark(ov, "c", "b", 1, 2, 70) == 0
    fw, fh = (rh, rw) if rotate in (90, 270) else (rw, rh)
    src = gpu.Image(np.concatenate(frames, axis=0))
    dst = gpu.Image(np.zeros((n * fh, fw, c), np.uint8))
    dstep = dst.step                                      # rows padded to 4 bytes like every frame
    gpu.batch_resize_rotate_watermark(src.device_ptr, sh * src.step, sw, sh, src.step, dst.device_ptr, fh * dstep, dstep,
                                      rw, rh, rotate, cfg, c, n)
    out = dst.numpy().reshape(n, fh, fw, c)
    filters = ["rotate=%d" % rotate] if rotate else []
    for i in range(n):
        rc, step, want = oracle_chain(frames[i], resize="%d,%d" % (rw, rh), filters=filters, overlay=ov, wm=("c", "b", 1, 2, 70))
        assert rc == 0 and np.array_equal(out[i], want), (rotate, geom, i)
    src.release(); dst.release(); cfg.release()


@pytest.mark.parametrize("rotate", [90, 270, 180])
@pytest.mark.parametrize("c", [4, 3])
def test_fused_area_rotate_with_full_bands(gpu, rotate, c):
    """Enough frames that the row-streaming kernel takes its 16-row bands (as in bench.py --mode chain224): the turned
    band leaves through the LDS tile in runs of 16 pixels, the last band of 50 rows is partial.  BGRA and BGR."""
    n, sh, sw, rw, rh = 1100, 120, 80, 37, 50
    rng = np.random.Generator(np.random.PCG64(0x1A4D7700 + rotate))
    frames = rng.integers(0, 256, size=(n, sh, sw, c), dtype=np.uint8)
    ov = noise_image(9, 20, 4, 62)
    cfg = gpu.Config()
    assert cfg.prepare_watermark(ov, "l", "t", 3, 1, 55) == 0
    fw, fh = (rh, rw) if rotate in (90, 270) else (rw, rh)
    src = gpu.Image(frames.reshape(n * sh, sw, c))
    dst = gpu.Image(np.zeros((n * fh, fw, c), np.uint8))
    gpu.batch_resize_rotate_watermark(src.device_ptr, sh * src.step, sw, sh, src.step, dst.device_ptr, fh * dst.step, dst.step,
                                      rw, rh, rotate, cfg, c, n)
    out = dst.numpy().reshape(n, fh, fw, c)
    for i in range(0, n, 7):
        rc, step, want = oracle_chain(frames[i], resize="%d,%d" % (rw, rh), filters=["rotate=%d" % rotate], overlay=ov,
                                      wm=("l", "t", 3, 1, 55))
        assert rc == 0 and np.array_equal(out[i], want), (rotate, i)
    src.release(); dst.release(); cfg.release()


def test_threads_have_independent_streams_and_pools(gpu):
    """Requests driven from several host threads (one HIP stream + buffer pool per thread) give the same bytes as serial ones."""
    import threading

    cfg = gpu.Config(allow_experiments=True)
    jobs = [(noise_image(120 + 7 * i, 160 + 5 * i, 4, 80 + i), dict(crop="4,3", resize="%d,0" % (40 + i), filters=["gotham=1", "rotate=90"]))
            for i in range(12)]
    want = []
    for arr, kw in jobs:
        rc, step, w = oracle_chain(arr, **kw)
        assert rc == 0
        want.append(w)
    got = [None] * len(jobs)
    errs = []

    def worker(tid):
        try:
            for rep in range(3):
                for i in range(tid, len(jobs), 4):
                    im = gpu.Image(jobs[i][0])
                    rc, step = gpu.run_ops(im, cfg, **jobs[i][1])
                    assert rc == 0
                    got[i] = im.numpy()
                    im.release()
        except Exception as e:      # surfaced below; a thread must not die silently
            errs.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for i in range(len(jobs)):
        assert np.array_equal(got[i], want[i]), i
    # threads alive at the same time really have distinct streams ...
    streams, gate = [], threading.Barrier(3)

    def grab():
        streams.append(gpu.lib.impgpu_env_stream())
        gate.wait()

    ts = [threading.Thread(target=grab) for _ in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert len(set(streams + [gpu.lib.impgpu_env_stream()])) == 4
    # ... and a thread that has ended hands its lane to the next new thread instead of leaking a stream per thread
    later = []
    for _ in range(6):
        t = threading.Thread(target=lambda: later.append(gpu.lib.impgpu_env_stream()))
        t.start()
        t.join()
    assert len(set(later)) == 1 and gpu.lib.impgpu_env_stream() not in later


@pytest.mark.parametrize("c", [3, 4])
@pytest.mark.parametrize("bpp", [24, 32])
def test_freeimage_side_repacks(gpu, c, bpp):
    """advancedio.c:65-101 (IplToFI32/24) on the device and advancedio.c:310-318 (LoadSingle's flip) on upload."""
    import ctypes as C

    arr = noise_image(37, 53, c, 90)
    im = gpu.Image(arr)
    pitch = (53 * (bpp // 8) + 3) & ~3
    out = np.zeros((37, pitch), np.uint8)
    assert gpu.lib.impgpu_image_download_fi(im.h, bpp, out.ctypes.data, pitch) == 0
    want = orc.ipl_to_fi(arr, bpp)
    assert np.array_equal(out[:, : 53 * (bpp // 8)], want[:, : 53 * (bpp // 8)])
    im.release()
    if c == 4 and bpp == 32:
        bits = np.ascontiguousarray(noise_image(37, 53, 4, 91))
        h = C.c_void_p()
        assert gpu.lib.impgpu_image_upload_fi32(bits.ctypes.data, 53, 37, 53 * 4, C.byref(h)) == 0
        up = gpu.Image(handle=h.value)
        assert np.array_equal(up.numpy(), orc.fi32_to_ipl(bits, 53, 37))
        assert np.array_equal(up.numpy(), bits[::-1])
        up.release()


# ---------------------------------------------------------------- albums: the frames of one animation behind one handle
@pytest.mark.parametrize("chain", CHAINS, ids=lambda d: "-".join(k for k in d))
@pytest.mark.parametrize("c", [1, 3, 4])
def test_album_run_ops_is_every_frame_of_the_loop(gpu, chain, c):
    """bridge.c:577-655 loops `for fid < album.Count` around each operator; an album handle runs each operator once for
    all frames.  Every frame must come out as the oracle chain makes it."""
    n = 5
    frames = [noise_image(240, 321, c, 900 + i) if i % 2 else smooth_image(240, 321, c) for i in range(n)]
    if c == 1:
        frames = [noise_image(240, 321, 1, 910 + i) for i in range(n)]
    ov = noise_image(20, 48, 4, 43)
    wm = ("r", "b", 4, 4, 60)
    has_vig = any(f.startswith("vignette") for f in chain.get("filters", ()))
    cfg = gpu.Config(allow_experiments=True, max_filters=5)
    assert cfg.prepare_watermark(ov, *wm) == 0
    al = gpu.Image.album(frames)
    assert al.count == n
    kw = dict(chain)
    kw["need_flatten"] = kw.pop("flatten", 0)
    rc, step = gpu.run_ops(al, cfg, **kw)
    assert rc == 0, (rc, step)
    assert al.count == n
    outs = al.frames()
    for i in range(n):
        rc_o, _, want = oracle_chain(frames[i], overlay=ov, wm=wm, **chain)
        assert rc_o == 0
        if has_vig:
            assert np.abs(outs[i].astype(int) - want.astype(int)).max() <= 1
        else:
            assert np.array_equal(outs[i], want), i
    al.release(); cfg.release()


def test_album_single_operators_and_round_trip(gpu):
    n = 7
    frames = [noise_image(33, 47, 4, 950 + i) for i in range(n)]
    al = gpu.Image.album(frames)
    assert al.count == n and al.shape == (33, 47, 4)
    back = al.frames()
    assert all(np.array_equal(a, b) for a, b in zip(back, frames))
    assert al.crop("20px,10px,3px,5px") == 0
    assert al.filter("gamma=1.4") == 0
    assert al.filter("rotate=90") == 0
    assert al.blend_with_paper() == 0
    outs = al.frames()
    for i in range(n):
        rc, cur = orc.crop(frames[i], "20px,10px,3px,5px")
        rc, cur = orc.filter(cur, "gamma=1.4", 1)
        rc, cur = orc.filter(cur, "rotate=90", 1)
        cur = orc.blend_with_paper(cur)
        assert np.array_equal(outs[i], cur), i
    # Info() reads frame 0 (bridge.c:283-300)
    assert al.calc_perceived_brightness() == np.float32(orc.brightness(outs[0]))
    al.release()


def test_album_of_one_is_an_image(gpu):
    arr = noise_image(40, 50, 3, 960)
    al = gpu.Image.album([arr])
    assert al.count == 1
    assert al.resize("25,0") == 0
    rc, want = orc.resize(arr, "25,0", 2000, 2000, 0)
    assert np.array_equal(al.frames()[0], want) and np.array_equal(al.numpy(), want)
    al.release()
