"""Randomised parity (hypothesis, derandomised): arbitrary geometries, channel counts and modes through every resize
kernel variant (gather, LDS-tiled, rolling 2x, vector AREA for BGR / BGRA, integer AREA, generic), random filter
chains through run_ops, random watermark placements.  Bit-exact against the oracle."""
import numpy as np
import pytest
from hypothesis import assume, given, settings, strategies as st, HealthCheck

import oracle_lib as orc
from conftest import noise_image, smooth_image

pytestmark = pytest.mark.gpu
import os
# IMP_FUZZ_RANDOM=1 draws fresh examples instead of the fixed derandomised set; IMP_FUZZ_SCALE=N multiplies their number
COMMON = dict(deadline=None, derandomize=not os.environ.get("IMP_FUZZ_RANDOM"),
              suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
SCALE = int(os.environ.get("IMP_FUZZ_SCALE", "1"))


@settings(max_examples=250 * SCALE, **COMMON)
@given(sw=st.integers(1, 300), sh=st.integers(1, 200), dw=st.integers(1, 300), dh=st.integers(1, 200),
       c=st.sampled_from([1, 3, 4]), interp=st.integers(0, 4), seed=st.integers(0, 1000))
def test_resize_any_geometry(gpu, sw, sh, dw, dh, c, interp, seed):
    if interp == orc.INTER_AREA and (dw > sw or dh > sh):
        dw, dh = min(dw, sw), min(dh, sh)
    arr = noise_image(sh, sw, c, seed)
    want = orc.cv_resize(arr, dw, dh, interp)
    im = gpu.Image(arr)
    assert im.cv_resize(dw, dh, interp) == 0
    got = im.numpy()
    im.release()
    assert np.array_equal(got, want), (sw, sh, dw, dh, c, interp)


@settings(max_examples=80 * SCALE, **COMMON)
@given(quarter_w=st.integers(1, 96), half_h=st.integers(1, 170), c=st.sampled_from([3, 4]), interp=st.sampled_from([2, 4]),
       mult=st.sampled_from([4, 8, 16]), seed=st.integers(0, 100))
def test_resize_exact_halves_ring_kernels(gpu, quarter_w, half_h, c, interp, mult, seed):
    """Widths that are multiples of 4 (BGRA) / 16 (BGR) take the LDS-DMA ring kernels, the rest the register strips."""
    sw = quarter_w * mult
    arr = noise_image(2 * half_h, sw, c, seed)
    want = orc.cv_resize(arr, sw // 2, half_h, interp)
    im = gpu.Image(arr)
    assert im.cv_resize(sw // 2, half_h, interp) == 0
    assert np.array_equal(im.numpy(), want), (sw, 2 * half_h, c, interp)
    im.release()


@settings(max_examples=60 * SCALE, **COMMON)
@given(half_w=st.integers(1, 200), half_h=st.integers(1, 150), interp=st.sampled_from([1, 2, 4]), seed=st.integers(0, 100))
def test_resize_exact_halves(gpu, half_w, half_h, interp, seed):
    arr = noise_image(2 * half_h, 2 * half_w, 4, seed)
    want = orc.cv_resize(arr, half_w, half_h, interp)
    im = gpu.Image(arr)
    assert im.cv_resize(half_w, half_h, interp) == 0
    assert np.array_equal(im.numpy(), want), (half_w, half_h, interp)
    im.release()


POINTWISE = ["modulate=%d,%d,%d", "colorize=%02x%02x%02x,0.%d", "gamma=%d.%d", "contrast=%d.%d", "gotham=1", "lomo=1", "kelvin=1",
             "rainbow=full", "rainbow=mid", "scanline=0.%d,0.%d,%d,%d", "flip=10", "flip=01", "rotate=90", "rotate=180",
             "rotate=270", "blur=%d.%d", "gradmap=%02x%02x%02x,%02x%02x%02x"]


@st.composite
def filter_chain(draw):
    n = draw(st.integers(1, 5))
    out = []
    for _ in range(n):
        t = draw(st.sampled_from(POINTWISE))
        k = t.count("%")
        if t.startswith("modulate"):
            vals = (draw(st.integers(0, 180)), draw(st.integers(-50, 300)), draw(st.integers(1, 300)))
        elif t.startswith("colorize"):
            vals = (draw(st.integers(0, 255)), draw(st.integers(0, 255)), draw(st.integers(0, 255)), draw(st.integers(0, 9)))
        elif t.startswith("gradmap"):
            vals = tuple(draw(st.integers(0, 255)) for _ in range(6))
        elif t.startswith("scanline"):
            vals = (draw(st.integers(0, 9)), draw(st.integers(0, 9)), draw(st.integers(1, 5)), draw(st.integers(1, 5)))
        elif t.startswith("blur"):
            vals = (draw(st.integers(0, 6)), draw(st.integers(1, 9)))
        elif k:
            vals = (draw(st.integers(0, 3)), draw(st.integers(1, 9)))
        else:
            vals = ()
        out.append(t % vals)
    return out


@settings(max_examples=80 * SCALE, **COMMON)
@given(w=st.integers(2, 120), h=st.integers(2, 90), c=st.sampled_from([3, 4]), filters=filter_chain(), seed=st.integers(0, 100),
       crop=st.sampled_from([None, "1,1", "4,3,r,b", "2,1,l,t"]), resize=st.sampled_from([None, "40", "0,33", "50,20", "200,200,up"]))
def test_run_ops_random_chain(gpu, w, h, c, filters, seed, crop, resize):
    from test_gpu_chain import oracle_chain

    arr = noise_image(h, w, c, seed)
    rc_o, step_o, want = oracle_chain(arr, crop=crop, resize=resize, filters=filters)
    cfg = gpu.Config(allow_experiments=True, max_filters=8)
    im = gpu.Image(arr)
    rc, step = gpu.run_ops(im, cfg, crop=crop, resize=resize, filters=filters)
    assert rc == rc_o, (rc, rc_o, step, filters)
    if rc == 0:
        assert np.array_equal(im.numpy(), want), (w, h, c, crop, resize, filters)
    im.release()


@settings(max_examples=80 * SCALE, **COMMON)
@given(bw=st.integers(1, 90), bh=st.integers(1, 70), ow=st.integers(1, 60), oh=st.integers(1, 50), dc=st.sampled_from([3, 4]),
       sc=st.sampled_from([3, 4]), gx=st.sampled_from("lcr"), gy=st.sampled_from("tcb"), ox=st.integers(-40, 40),
       oy=st.integers(-40, 40), op=st.integers(1, 100), seed=st.integers(0, 50))
def test_watermark_random_placement(gpu, bw, bh, ow, oh, dc, sc, gx, gy, ox, oy, op, seed):
    base = noise_image(bh, bw, dc, seed)
    ov = noise_image(oh, ow, sc, seed + 1)
    rc_o, want = orc.watermark(base, ov, gx, gy, ox, oy, op)
    cfg = gpu.Config()
    assert cfg.prepare_watermark(ov, gx, gy, ox, oy, op) == 0
    im = gpu.Image(base)
    rc = im.watermark(cfg)
    assert rc == rc_o
    if rc == 0:
        assert np.array_equal(im.numpy(), want)
    im.release(); cfg.release()


@settings(max_examples=60 * SCALE, **COMMON)
@given(sw=st.integers(8, 420), sh=st.integers(4, 260), fx=st.floats(1.0, 9.0), fy=st.floats(1.0, 9.0), whole=st.booleans(),
       c=st.sampled_from([3, 4]), count=st.sampled_from([1, 7, 40, 160, 700]), seed=st.integers(0, 1000))
def test_area_batches_any_geometry(gpu, sw, sh, fx, fy, whole, c, count, seed):
    """INTER_AREA through the batch entry point: the band height, the four-columns-per-lane kernel and the box kernels are
    picked from the batch size as well as the geometry, so the same geometries are tried at several frame counts (frames are
    repeats of three random ones; three are compared with the oracle)."""
    if whole:                                              # exact integer factors: resizeAreaFast_
        kx, ky = max(1, int(fx)), max(1, int(fy))
        dw, dh = max(1, sw // kx), max(1, sh // ky)
        sw, sh = dw * kx, dh * ky
    else:
        dw, dh = max(1, int(sw / fx)), max(1, int(sh / fy))
    if count * sw * sh * c > 120_000_000:
        count = max(1, 120_000_000 // (sw * sh * c))
    base = [noise_image(sh, sw, c, seed + k) for k in range(3)]
    sstep, dstep = (sw * c + 3) & ~3, (dw * c + 3) & ~3
    src = gpu.Image(np.concatenate([base[i % 3] for i in range(count)], axis=0))
    dst = gpu.Image(np.zeros((count * dh, dw, c), np.uint8))
    gpu.batch_cv_resize(src.device_ptr, sh * sstep, sw, sh, sstep, dst.device_ptr, dh * dstep, dw, dh, dstep, c, count, orc.INTER_AREA)
    out = dst.numpy().reshape(count, dh, dw, c)
    for i in sorted({0, count // 2, count - 1}):
        assert np.array_equal(out[i], orc.cv_resize(base[i % 3], dw, dh, orc.INTER_AREA)), (sw, sh, dw, dh, c, count, i)
    src.release(); dst.release()


@settings(max_examples=120 * SCALE, **COMMON)
@given(sw=st.integers(8, 700), sh=st.integers(2, 300), fx=st.floats(0.26, 2.0), fy=st.floats(0.26, 2.0), c=st.sampled_from([3, 4]),
       interp=st.sampled_from([1, 2, 4]), seed=st.integers(0, 100))
def test_resize_rolling_strips(gpu, sw, sh, fx, fy, c, interp, seed):
    """Round 3's k_resize_strip (it replaced the LDS-tiled kernel and the per-pixel gather BGR frames fell to): LINEAR,
    CUBIC and LANCZOS4 at scales up to 2 per axis in either direction, frames wider than one 256-column block and taller
    than one strip of rows, BGRA and BGR.  (Exact halves and CUBIC enlargements in y have their own kernels and tests.)"""
    dw, dh = max(1, int(round(sw / fx))), max(1, int(round(sh / fy)))
    arr = noise_image(sh, sw, c, seed)
    want = orc.cv_resize(arr, dw, dh, interp)
    im = gpu.Image(arr)
    assert im.cv_resize(dw, dh, interp) == 0
    got = im.numpy()
    im.release()
    assert np.array_equal(got, want), (sw, sh, dw, dh, c, interp)


def test_resize_one_axis_grows_the_other_shrinks(gpu):
    """What Resize() sends to CUBIC besides plain enlargements (bridge.c:190: width > col || height > row): a portrait
    frame made landscape and the reverse, through the request grammar, BGR (a JPEG) and BGRA."""
    for c in (3, 4):
        for (sh, sw), args in (((320, 180), "320,180,up"), ((180, 320), "180,320,up"), ((300, 100), "150,200,up")):
            arr = smooth_image(sh, sw, c, seed=c)
            rc_o, want = orc.resize(arr, args)
            im = gpu.Image(arr)
            assert im.resize(args) == rc_o == 0
            assert np.array_equal(im.numpy(), want), (c, sh, sw, args)
            im.release()


@settings(max_examples=120 * SCALE, **COMMON)
@given(w=st.integers(1, 200), h=st.integers(1, 150), c=st.sampled_from([1, 3, 4]), q=st.integers(0, 100), kind=st.integers(0, 3),
       seed=st.integers(0, 1000))
def test_jpeg_encode_any_frame(gpu, w, h, c, q, kind, seed):
    """cvEncodeImage(".jpg") on the device: the same file, byte for byte, as the oracle (itself pinned against libjpeg-turbo)
    for any geometry (every padding and dummy-block rule), channel count, quality and kind of content."""
    if kind == 0:
        arr = noise_image(h, w, c, seed)
    elif kind == 1:
        arr = smooth_image(h, w, c) if c > 1 and h > 1 and w > 1 else noise_image(h, w, c, seed)
    elif kind == 2:                                       # blocky: long zero runs, end-of-block everywhere
        arr = np.repeat(np.repeat(noise_image((h + 7) // 8, (w + 7) // 8, c, seed), 8, axis=0), 8, axis=1)[:h, :w]
    else:                                                 # extremes only: the longest codes, FF bytes in the segment
        arr = (noise_image(h, w, c, seed) > 127).astype(np.uint8) * 255
    rc_o, want = orc.jpeg_encode(arr, q)
    im = gpu.Image(arr)
    rc, got = im.encode_jpeg(q)
    im.release()
    assert rc == rc_o == 0 and got == want, (w, h, c, q, kind)


@settings(max_examples=40 * SCALE, **COMMON)
@given(n=st.integers(1, 9), w=st.integers(1, 120), h=st.integers(1, 90), c=st.sampled_from([1, 3, 4]), pick=st.integers(0, 5), seed=st.integers(0, 1000))
def test_album_is_the_per_frame_loop(gpu, n, w, h, c, pick, seed):
    """An album handle through an operator: every frame equals the single-image call on that frame."""
    frames = [noise_image(h, w, c, seed * 16 + i) for i in range(n)]
    ops = [("resize", "%d,0" % max(1, w // 2)), ("resize", "%d,%d,up" % (w + 3, h + 2)), ("filter", "gamma=1.7"), ("filter", "rotate=90"),
           ("filter", "blur=1.2"), ("crop", "1,1")]
    kind, arg = ops[pick]
    al = gpu.Image.album(frames)
    rc_a = getattr(al, kind)(arg)
    outs = al.frames() if rc_a == 0 else []
    for i, f in enumerate(frames):
        im = gpu.Image(f)
        rc = getattr(im, kind)(arg)
        assert rc == rc_a, (kind, arg, rc, rc_a)
        if rc == 0:
            assert np.array_equal(im.numpy(), outs[i]), (kind, arg, i)
        im.release()
    al.release()


def _pil_jpeg(arr, **kw):
    import io

    Image = pytest.importorskip("PIL.Image")
    b = io.BytesIO()
    (Image.fromarray(arr[:, :, 0], "L") if arr.shape[2] == 1 else Image.fromarray(arr)).save(b, "JPEG", **kw)
    return b.getvalue()


@settings(max_examples=120 * SCALE, **COMMON)
@given(w=st.integers(1, 700), h=st.integers(1, 500), sub=st.sampled_from(["4:4:4", "4:2:2", "4:2:0", "gray"]), q=st.integers(20, 100),
       rst=st.sampled_from([0, 0, 1, 3, 8, 40]), kind=st.integers(0, 2), opt=st.booleans(), flips=st.integers(0, 2), seed=st.integers(0, 10000))
def test_jpeg_decode_any_file(gpu, w, h, sub, q, rst, kind, opt, flips, seed):
    """Any baseline file (size, sampling, quality, restart interval, optimised tables, content), intact or with a few bytes
    flipped: the device gives the oracle's verdict and, when that is OK, the oracle's pixels (the oracle = libjpeg-turbo's)."""
    c = 1 if sub == "gray" else 3
    if kind == 0:
        arr = noise_image(h, w, c, seed)
    elif kind == 1:
        arr = smooth_image(h, w, c) if c > 1 and h > 1 and w > 1 else noise_image(h, w, c, seed)
    else:
        arr = np.repeat(np.repeat(noise_image((h + 15) // 16, (w + 15) // 16, c, seed), 16, axis=0), 16, axis=1)[:h, :w]
    kw = dict(quality=q, optimize=opt)
    if c == 3:
        kw["subsampling"] = sub
    if rst:
        kw["restart_marker_blocks"] = rst
    try:
        blob = bytearray(_pil_jpeg(arr, **kw))
    except OSError:                      # Pillow's own buffer estimate (noise at quality 100 with optimised tables): not a case
        assume(False)
    rng = np.random.Generator(np.random.PCG64(seed))
    for _ in range(flips):
        blob[int(rng.integers(2, len(blob)))] = int(rng.integers(0, 256))
    blob = bytes(blob)
    rc_o, want = orc.jpeg_decode(blob)
    os.environ["IMPGPU_JPEG_HUFF"] = "device" if seed % 3 else "host"     # (unset, a lone small file's entropy stage runs on the caller)
    try:
        rc, im = gpu.Image.decode_jpeg(blob)
    finally:
        del os.environ["IMPGPU_JPEG_HUFF"]
    if rc_o == 0:
        assert rc == 0, (w, h, sub, q, rst, kind, opt, flips, seed, gpu.lib.impgpu_last_error())
        got = im.numpy()
        im.release()
        assert np.array_equal(got, want), (w, h, sub, q, rst, kind, opt, flips, seed)
    else:
        assert rc != 0, (w, h, sub, q, rst, kind, opt, flips, seed, rc_o)
