"""GPU parity: filter-* operators, HSV conversions, blends, brightness, ASCII vs the CPU oracle (C ABI)."""
import numpy as np
import pytest

import oracle_lib as orc
from conftest import noise_image, smooth_image

pytestmark = pytest.mark.gpu


def run_filter(imp, arr, request, allow=1):
    im = imp.Image(arr)
    rc = im.filter(request, allow)
    out = im.numpy() if rc == 0 else None
    im.release()
    return rc, out


EXACT_FILTERS = [
    "flip=10", "flip=01", "flip=11", "flip=00", "rotate=90", "rotate=180", "rotate=270",
    "modulate=0,100,100", "modulate=30,120,80", "modulate=180,0,250", "modulate=90,-50,100",
    "colorize=ff8000", "colorize=102030,0.25", "colorize=ffffff,1", "colorize=000000,0",
    "gamma=2.2", "gamma=0.45", "gamma=1", "gamma=0", "gamma=-1.5",
    "contrast=1.5", "contrast=0.3", "contrast=10",
    "gradmap=000000,ffffff", "gradmap=ff0000,00ff00,0000ff", "gradmap=102030,405060,708090,a0b0c0",
    "gradmap=000000,111111,222222,333333,444444,555555,666666,777777",
    "gotham=1", "lomo=1", "kelvin=1", "rainbow=full", "rainbow=mid", "rainbow=pale",
    "scanline=0.5", "scanline=0.3,0.6,2,3", "scanline=1,1,1,1", "scanline=0,0,5,2",
]


@pytest.mark.parametrize("request_", EXACT_FILTERS)
@pytest.mark.parametrize("c", [3, 4])
def test_filter_bit_exact(gpu, request_, c):
    for arr in (noise_image(37, 53, c, 7), smooth_image(64, 48, c)):
        rc_o, want = orc.filter(arr, request_)
        rc, got = run_filter(gpu, arr, request_)
        assert rc == rc_o == 0, (rc, rc_o)
        assert got.shape == want.shape
        assert np.array_equal(got, want), "%s: %d px differ, max %d" % (
            request_, (got != want).any(axis=2).sum(), np.abs(got.astype(int) - want.astype(int)).max())


@pytest.mark.parametrize("request_", ["vignette=0.5", "vignette=0.8,0.6", "vignette=1.3,1.0"])
@pytest.mark.parametrize("c", [3, 4])
def test_vignette_within_one(gpu, request_, c):
    """Vignette goes through libm cos/pow in double on the CPU; device cos differs in the last ulp.
    Tolerance: +-1 per channel (north_star: stated per-channel tolerance for float paths), and it must be rare."""
    arr = smooth_image(120, 160, c)
    rc_o, want = orc.filter(arr, request_)
    rc, got = run_filter(gpu, arr, request_)
    assert rc == rc_o == 0
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1
    assert (d > 0).mean() < 1e-3


@pytest.mark.parametrize("sigma", ["0.5", "1", "2", "3.7", "8", "0.1", "0"])
@pytest.mark.parametrize("c", [1, 3, 4])
def test_blur_bit_exact(gpu, sigma, c):
    # 3-channel widths 67 / 50 / 45 leave 1 / 2 / 3 row elements to SymmColumnVec's scalar integer tail
    for arr in (noise_image(45, 67, c, 9), smooth_image(33, 50, c), noise_image(29, 45, c, 10)):
        want = orc.gaussian(arr, float(np.float32(sigma)))
        rc, got = run_filter(gpu, arr, "blur=" + sigma)
        assert rc == 0
        assert np.array_equal(got, want), "sigma %s: max diff %d" % (sigma, np.abs(got.astype(int) - want.astype(int)).max())


def test_blur_large_sigma_and_thin_images(gpu):
    arr = noise_image(64, 80, 4, 11)
    want = orc.gaussian(arr, 25.0)
    rc, got = run_filter(gpu, arr, "blur=25")
    assert rc == 0 and np.array_equal(got, want)
    for shape in [(1, 40), (40, 1), (2, 2)]:
        a = noise_image(shape[0], shape[1], 3, 12)
        rc, got = run_filter(gpu, a, "blur=2")
        assert rc == 0 and np.array_equal(got, orc.gaussian(a, 2.0)), shape


@pytest.mark.parametrize("c", [3, 4])
@pytest.mark.parametrize("sigma", ["3.4", "4", "5.5", "8", "9.9", "12", "20", "25", "31"])    # non-zero radii 17 .. 60 (ring 64 / 128), 31: two-pass
def test_blur_large_radius_strip_kernel_bit_exact(gpu, sigma, c):
    """k_blur_strip4: radius 17..60 (after the zero taps are trimmed) on BGRA and BGR -- column strips with the row sums in
    an LDS ring.  Frames narrower than a strip, shorter than the radius (every tap row clamps), taller than one block's
    strip, widths that are no multiple of 64 and, for BGR, rows whose last 1..3 elements take the integer tail."""
    for shape in [(70, 130), (300, 67), (9, 200), (131, 64), (64, 1), (2, 3), (40, 65), (33, 66)]:
        arr = noise_image(shape[0], shape[1], c, 14) if shape[0] % 2 else smooth_image(shape[0], shape[1], c)
        want = orc.gaussian(arr, float(np.float32(sigma)))
        rc, got = run_filter(gpu, arr, "blur=" + sigma)
        assert rc == 0
        assert np.array_equal(got, want), "sigma %s %r: max diff %d" % (sigma, shape, np.abs(got.astype(int) - want.astype(int)).max())


def test_blur_every_dispatch_across_sigmas(gpu):
    """A sweep fine enough to meet every kernel choice: fused (radius <= 16, taps summing to at most 257), strips with the
    float ring (64 rows; and any radius whose rounded taps sum ABOVE 257 -- sigma = 4 is one -- which took the two-pass
    fallback until round 3), strips with the u16 ring (128 rows), two-pass (radius > 60)."""
    arr3, arr4 = noise_image(150, 203, 3, 77), noise_image(150, 203, 4, 78)
    for i in range(1, 60):
        sigma = "%.2f" % (0.3 + 0.47 * i)
        for arr in (arr3, arr4):
            want = orc.gaussian(arr, float(np.float32(sigma)))
            rc, got = run_filter(gpu, arr, "blur=" + sigma)
            assert rc == 0 and np.array_equal(got, want), (sigma, arr.shape[2])


def test_blur_sigma8_full_hd_and_after_crop(gpu):
    arr = noise_image(1080, 1920, 4, 15)
    rc, got = run_filter(gpu, arr, "blur=8")
    assert rc == 0 and np.array_equal(got, orc.gaussian(arr, 8.0))
    # a cropped view as the source (unaligned window, pitch of the uncropped frame)
    im = gpu.Image(arr[:300, :400].copy())
    cfg = gpu.Config()
    rc, step = gpu.run_ops(im, cfg, crop="301px,177px,13px,9px", filters=["blur=6"])
    want = orc.gaussian(orc.crop(arr[:300, :400].copy(), "301px,177px,13px,9px")[1], 6.0)
    assert rc == 0 and np.array_equal(im.numpy(), want)
    im.release()


def test_filter_error_codes_match(gpu):
    arr = noise_image(16, 16, 4, 13)
    for req, allow in [("nosuch=1", 1), ("flip", 1), ("flip=2", 1), ("flip=101", 1), ("rotate=45", 1), ("modulate=1,2", 1),
                       ("modulate=181,1,1", 1), ("modulate=0,1,0", 1), ("colorize=fff", 1), ("colorize=ffffff,2", 1),
                       ("blur=-1", 1), ("contrast=0", 1), ("contrast=-2", 1), ("gradmap=ff", 1), ("gradmap=ffffff", 1),
                       ("rainbow=dark", 1), ("scanline=2", 1), ("scanline=0.5,3", 1), ("scanline=0.5,0.5,0", 1),
                       ("gotham=1", 0), ("vignette=1", 0), ("lomo=1", 0), ("=1", 1)]:
        rc_o, _ = orc.filter(arr, req, allow)
        rc, _ = run_filter(gpu, arr, req, allow)
        assert rc == rc_o != 0, (req, rc, rc_o)


def all_colours():
    """Every (b, g, r) once: 4096 x 4096 x 3."""
    v = np.arange(1 << 24, dtype=np.uint32)
    return np.stack([(v & 0xff), (v >> 8) & 0xff, (v >> 16) & 0xff], axis=1).astype(np.uint8).reshape(4096, 4096, 3)


def test_rgb2hsv_hsv2rgb_exhaustive(gpu):
    """helpers.c:70-176 over the whole 24-bit input space, both directions."""
    arr = all_colours()
    im = gpu.Image(arr)
    assert im.rgb2hsv() == 0
    assert np.array_equal(im.numpy(), orc.rgb2hsv(arr))
    im.release()
    im = gpu.Image(arr)            # any (h, s, v) byte triple, including the out-of-range hues ModulateHSV can store
    assert im.hsv2rgb() == 0
    assert np.array_equal(im.numpy(), orc.hsv2rgb(arr))
    im.release()


@pytest.mark.parametrize("dc,sc", [(4, 4), (4, 3), (3, 4), (3, 3)])
def test_watermark_bit_exact(gpu, dc, sc):
    base = noise_image(90, 120, dc, 14)
    ov = noise_image(24, 40, sc, 15)
    if sc == 4:
        ov[:, :, 3] = np.linspace(0, 255, 40).astype(np.uint8)[None, :]
    for gx, gy, ox, oy, op in [("r", "b", 16, 16, 60), ("l", "t", 0, 0, 100), ("c", "c", -5, 7, 35), ("r", "b", -10, -10, 80),
                               ("l", "t", -20, -10, 50), ("c", "b", 0, 80, 1), ("l", "t", 100, 80, 90)]:
        rc_o, want = orc.watermark(base, ov, gx, gy, ox, oy, op)
        cfg = gpu.Config()
        assert cfg.prepare_watermark(ov, gx, gy, ox, oy, op) == 0
        im = gpu.Image(base)
        rc = im.watermark(cfg)
        assert rc == rc_o, (gx, gy, ox, oy, rc, rc_o)
        if rc == 0:
            assert np.array_equal(im.numpy(), want), (gx, gy, ox, oy, op)
        im.release(); cfg.release()


def test_watermark_every_alpha_pair(gpu):
    """All 256 x 256 (source alpha, destination alpha) pairs at three opacities."""
    a = np.arange(256, dtype=np.uint8)
    base = noise_image(256, 256, 4, 16); base[:, :, 3] = a[:, None]
    ov = noise_image(256, 256, 4, 17); ov[:, :, 3] = a[None, :]
    for op in (100, 60, 7):
        rc_o, want = orc.watermark(base, ov, "l", "t", 0, 0, op)
        cfg = gpu.Config(); cfg.prepare_watermark(ov, "l", "t", 0, 0, op)
        im = gpu.Image(base)
        assert im.watermark(cfg) == rc_o == 0
        assert np.array_equal(im.numpy(), want), op
        im.release(); cfg.release()


def test_blend_with_paper_bit_exact(gpu):
    a = np.arange(256, dtype=np.uint8)
    arr = noise_image(256, 256, 4, 18); arr[:, :, 3] = a[:, None]; arr[:, :, 0] = a[None, :]
    im = gpu.Image(arr)
    assert im.blend_with_paper() == 0
    assert np.array_equal(im.numpy(), orc.blend_with_paper(arr))
    im.release()


@pytest.mark.parametrize("c", [1, 3, 4])
def test_brightness_exact(gpu, c):
    """The float accumulator's running rounding (filters.c:708-726) is reproduced, not approximated."""
    for arr in (noise_image(120, 200, c, 19), smooth_image(333, 77, c), np.full((300, 500, c), 100, np.uint8)):
        if c == 1:
            arr = arr[:, :, :1]
        im = gpu.Image(arr)
        got = im.calc_perceived_brightness()
        im.release()
        want = orc.brightness(arr)
        assert np.float32(got) == np.float32(want), (got, want)


def test_brightness_1080p_exact(gpu):
    arr = smooth_image(1080, 1920, 4)
    im = gpu.Image(arr)
    got = im.calc_perceived_brightness()
    im.release()
    assert np.float32(got) == np.float32(orc.brightness(arr))


@pytest.mark.parametrize("args", ["", "wide"])
def test_ascii_exact(gpu, args):
    arr = smooth_image(30, 72, 3)
    im = gpu.Image(arr)
    got = im.ascii(args)
    hsv_after = im.numpy()
    im.release()
    assert got == orc.ascii_art(arr, args)
    assert np.array_equal(hsv_after, orc.rgb2hsv(arr))     # the reference leaves the frame in HSV


def test_gray2bgr(gpu):
    arr = noise_image(31, 45, 1, 20)
    im = gpu.Image(arr)
    assert im.gray2bgr() == 0
    assert np.array_equal(im.numpy(), orc.gray2bgr(arr))
    im.release()


def test_brightness_adversarial_exact(gpu):
    """Inputs that stress the regime replay: exact ties (gray pixels: sqrt term is integral), a long zero prefix,
    all black, saturated white, 1-pixel frames, and a 4K frame whose sum passes 2^30."""
    rng = np.random.Generator(np.random.PCG64(99))
    gray = rng.integers(0, 256, size=(1080, 1920, 1), dtype=np.uint8).repeat(3, axis=2)
    zero_prefix = np.zeros((700, 900, 3), np.uint8); zero_prefix[:, 600:] = rng.integers(0, 256, size=(700, 300, 3), dtype=np.uint8)
    cases = [gray, gray[:, :, :1], zero_prefix, np.zeros((300, 300, 4), np.uint8), np.full((1080, 1920, 3), 255, np.uint8),
             np.full((1, 1, 3), 7, np.uint8), np.full((1, 5000, 1), 255, np.uint8), np.full((3000, 1, 1), 3, np.uint8),
             rng.integers(0, 256, size=(2160, 3840, 4), dtype=np.uint8),
             np.full((2160, 3840, 1), 254, np.uint8), (np.arange(1500 * 1500) % 2 * 255).astype(np.uint8).reshape(1500, 1500, 1),
             # the multi-kernel path (>= 2^19 pixels): a whole number of 16384-term chunks, the threshold itself and one
             # past it, a dark frame that never reaches the first round's binade, a frame that is dark then bright
             rng.integers(0, 256, size=(1024, 1024, 3), dtype=np.uint8), rng.integers(0, 256, size=(512, 1024, 1), dtype=np.uint8),
             rng.integers(0, 256, size=(1, 524289, 1), dtype=np.uint8), rng.integers(0, 3, size=(800, 800, 3), dtype=np.uint8),
             np.concatenate([rng.integers(0, 2, size=(900, 500, 3), dtype=np.uint8), rng.integers(200, 256, size=(900, 500, 3), dtype=np.uint8)], axis=1)]
    for arr in cases:
        im = gpu.Image(arr)
        got = im.calc_perceived_brightness()
        im.release()
        want = orc.brightness(arr)
        assert np.float32(got) == np.float32(want), (arr.shape, got, want)


def test_batch_filters_matches_per_frame(gpu):
    n, h, w = 5, 60, 80
    frames = [noise_image(h, w, 4, 500 + i) for i in range(n)]
    filters = ["modulate=20,130,90", "colorize=203040,0.3", "gamma=1.4", "rainbow=mid", "gradmap=001122,ffeedd"]
    batch = gpu.Image(np.concatenate(frames, axis=0))
    assert gpu.batch_filters(batch.device_ptr, h * w * 4, w, h, 4, w * 4, n, filters) == 0
    out = batch.numpy().reshape(n, h, w, 4)
    for i in range(n):
        cur = frames[i]
        for f in filters:
            rc, cur = orc.filter(cur, f)
            assert rc == 0
        assert np.array_equal(out[i], cur), i
    assert gpu.batch_filters(batch.device_ptr, h * w * 4, w, h, 4, w * 4, n, ["rotate=90"]) == 1       # not pointwise
    assert gpu.batch_filters(batch.device_ptr, h * w * 4, w, h, 4, w * 4, n, ["nosuch=1"]) == 52
    batch.release()


@pytest.mark.parametrize("c", [3, 4])
@pytest.mark.parametrize("filters", [["gamma=1.7"], ["gotham=1", "gamma=1.7"], ["scanline=0.6,0.4,3,2", "contrast=1.2"]],
                         ids=["table-only", "hsv+tables", "needs-coordinates"])
def test_batch_filters_vector_paths_big_batch(gpu, c, filters):
    """Contiguous frames take the vectorised pixel program: 16-byte groups (BGRA) or 12-byte groups (BGR, every JPEG);
    past 16 M pixels a thread carries four groups.  9 frames of 1080p cross that line; a 36 x 20 frame does not."""
    for n, h, w in ((9, 1080, 1920), (3, 20, 36)):
        rng = np.random.default_rng(700 + c)
        frames = rng.integers(0, 256, (n, h, w, c), dtype=np.uint8)
        batch = gpu.Image(frames.reshape(n * h, w, c))
        assert batch.step == w * c
        assert gpu.batch_filters(batch.device_ptr, h * w * c, w, h, c, w * c, n, filters) == 0
        out = batch.numpy().reshape(n, h, w, c)
        for i in (0, n // 2, n - 1):
            cur = frames[i]
            for f in filters:
                rc, cur = orc.filter(cur, f)
                assert rc == 0
            assert np.array_equal(out[i], cur), (n, i)
        batch.release()


@pytest.mark.parametrize("amount", [90, 270])
@pytest.mark.parametrize("shape", [(64, 96), (96, 64), (70, 130), (33, 32), (32, 33), (31, 31), (128, 36), (1, 40), (40, 1)])
def test_rotate_bgr_tiles(gpu, amount, shape):
    """3-channel quarter turns go through 32 x 32 LDS tiles: whole dword-aligned tiles (64 x 96 both ways), byte-path
    borders, source columns that are not dword aligned for 270 (70 x 130), and images smaller than a tile."""
    arr = noise_image(shape[0], shape[1], 3, 21)
    rc, got = run_filter(gpu, arr, "rotate=%d" % amount)
    assert rc == 0
    assert np.array_equal(got, np.rot90(arr, k=-1 if amount == 90 else 1))
    assert np.array_equal(got, orc.filter(arr, "rotate=%d" % amount)[1])


@pytest.mark.parametrize("sigma", ["1.7", "2.4", "3.3", "4", "6.5", "7.3", "7.4", "7.5", "7.7", "7.9", "10.5"])
def test_blur_taps_summing_past_257_on_the_matrix_unit(gpu, sigma):
    """The eleven sigmas between 0.5 and 25.0 (steps of 0.1) whose rounded taps sum to 258 .. 260: row sums of bright pixels
    need 17 bits.  Round 5 gives them a third byte plane (bit 16) and a third MFMA per chunk instead of the VALU kernels;
    bright frames are the ones that reach bit 16 -- all-255, noise in 200..255 -- next to ordinary noise, all three channel
    counts, frames larger than one 64 x 64 tile and with a partial last tile."""
    rng = np.random.Generator(np.random.PCG64(int(float(sigma) * 10)))
    for c in (1, 3, 4):
        frames = [np.full((70, 131, c), 255, np.uint8), rng.integers(200, 256, size=(133, 67, c), dtype=np.uint8),
                  noise_image(96, 203, c, 5), rng.integers(250, 256, size=(3, 300, c), dtype=np.uint8)]
        for arr in frames:
            want = orc.gaussian(arr, float(np.float32(sigma)))
            rc, got = run_filter(gpu, arr, "blur=" + sigma)
            assert rc == 0
            assert np.array_equal(got, want), "sigma %s c %d %r: max diff %d" % (sigma, c, arr.shape, np.abs(got.astype(int) - want.astype(int)).max())
