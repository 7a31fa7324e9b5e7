"""Pins on the CPU oracle that need no GPU.

The reference ships no tests or golden vectors, so the only reference-derived known answers are the
seven Crop results SURVEY.md 8(c) recorded from the reference's own Crop(); everything else here is a closed-form
property of the restated algorithm (identity, exact 2x2 box, constant images, NN index rule) or an independent
cross-check (torch CPU bicubic, same a = -0.75 kernel and sample centres, float arithmetic, +-1 LSB).
"""
import numpy as np
import pytest

import oracle_lib as orc
from conftest import noise_image, smooth_image

# (args, gravity) -> rc or (x, y, w, h) on a 640x480 frame: SURVEY.md 8(c)
SURVEY_CROP_KAT = [
    ("0,0,320,240", None, 50),
    ("320px,240px,0px,0px", None, (0, 0, 320, 240)),
    ("320px,240px", None, (160, 0, 320, 240)),
    ("1,1", None, (80, 0, 480, 480)),
    ("16,9,l,t", None, (0, 0, 640, 360)),
    ("1,1,c,c", "r,b", (160, 0, 480, 480)),
    ("400px,200px,46px,0px", None, (46, 0, 400, 200)),
]


@pytest.mark.parametrize("args,gravity,want", SURVEY_CROP_KAT)
def test_crop_known_answers_from_survey(args, gravity, want):
    rc, geom = orc.crop_geometry(640, 480, args, gravity)
    if isinstance(want, int):
        assert rc == want
    else:
        assert rc == 0 and geom == want


def test_resize_mode_rule_and_dims():
    # bridge.c:190: NN when simple, CUBIC when any axis grows, AREA otherwise; :167-173 aspect fill; :178-181 clamp
    assert orc.resize_geometry(1920, 1080, "224,224") == (0, (224, 224, orc.INTER_AREA))
    assert orc.resize_geometry(1920, 1080, "224") == (0, (224, 126, orc.INTER_AREA))
    assert orc.resize_geometry(1920, 1080, "0,224") == (0, (398, 224, orc.INTER_AREA))
    assert orc.resize_geometry(100, 100, "300,300") == (0, (100, 100, orc.INTER_AREA))
    assert orc.resize_geometry(100, 100, "300,300,up") == (0, (300, 300, orc.INTER_CUBIC))
    assert orc.resize_geometry(100, 100, "300,50,up") == (0, (300, 50, orc.INTER_CUBIC))
    assert orc.resize_geometry(100, 100, "50,50", simple=1) == (0, (50, 50, orc.INTER_NN))
    assert orc.resize_geometry(100, 100, "0,0")[0] == 50
    assert orc.resize_geometry(3000, 3000, "2500,100")[0] == 54
    # the reference compares width against H (bridge.c:184): a tall target passes, a wide one does not
    assert orc.resize_geometry(3000, 3000, "100,2500")[0] == 0


@pytest.mark.parametrize("c", [1, 3, 4])
def test_resize_closed_forms(c):
    a = noise_image(60, 80, c, 1)
    assert np.array_equal(orc.cv_resize(a, 80, 60, orc.INTER_AREA), a)                 # scale 1
    assert np.array_equal(orc.cv_resize(a, 80, 60, orc.INTER_NN), a)
    for interp in (orc.INTER_LINEAR, orc.INTER_CUBIC, orc.INTER_LANCZOS4):
        assert np.array_equal(orc.cv_resize(a, 80, 60, interp), a), interp             # taps collapse to the centre
    box = (a[0::2, 0::2].astype(int) + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2
    assert np.array_equal(orc.cv_resize(a, 40, 30, orc.INTER_AREA), box.astype(np.uint8))
    k = np.full((33, 47, c), 173, np.uint8)
    for interp in (orc.INTER_NN, orc.INTER_LINEAR, orc.INTER_AREA):
        assert (orc.cv_resize(k, 20, 11, interp) == 173).all()
    sy = np.minimum(np.floor(np.arange(11) * (1.0 / (11 / 33))).astype(int), 32)
    sx = np.minimum(np.floor(np.arange(20) * (1.0 / (20 / 47))).astype(int), 46)
    b = noise_image(33, 47, c, 2)
    assert np.array_equal(orc.cv_resize(b, 20, 11, orc.INTER_NN), b[sy][:, sx])


def test_cubic_matches_float_bicubic_within_one():
    import torch
    import torch.nn.functional as F

    for arr, size in ((smooth_image(270, 480, 4), (56, 56)), (noise_image(100, 140, 3, 3), (37, 51)),
                      (noise_image(40, 50, 3, 4), (90, 120))):
        got = orc.cv_resize(arr, size[1], size[0], orc.INTER_CUBIC).astype(np.float64)
        t = torch.from_numpy(arr).permute(2, 0, 1)[None].double()
        ref = F.interpolate(t, size=size, mode="bicubic", align_corners=False)[0].permute(1, 2, 0).numpy()
        d = np.abs(got - np.clip(np.rint(ref), 0, 255))
        # OpenCV 2.4.9's x-edge rule (sx < 0 -> src[0], sx >= w-1 -> src[w-1]; see test_cubic_upscale_edge_columns...)
        # pins the outermost columns of an enlargement; torch has no such rule, so those columns are checked there
        sx = np.floor((np.arange(size[1]) + 0.5) * (arr.shape[1] / size[1]) - 0.5)
        inner = (sx >= 0) & (sx < arr.shape[1] - 1)
        d = d[:, inner]
        assert d.max() <= 1        # 11-bit fixed-point weights vs float: never more than one LSB
        assert (d > 0).mean() < 0.08


def test_cubic_upscale_edge_columns_hand_derived():
    """OpenCV 2.4.9 cv::resize, xofs loop: `if( sx < 0 ) fx = 0, sx = 0;` and `if( sx >= ssize.width-1 ) fx = 0,
    sx = ssize.width-1;` for EVERY generic mode (3.x later exempted CUBIC / LANCZOS4).  The reference dispatches CUBIC only
    when enlarging (bridge.c:190), where dx = 0 always has sx = -1: the edge columns are copies of src[0] / src[w-1].
    Row [10 20 30 40] -> 8 wide, worked by hand with the 11-bit weights of interpolateCubic (A = -0.75):
      x = .25 -> (-216, 1800, 536, -72) / 2048,  x = .75 -> (-72, 536, 1800, -216) / 2048
      dx=1: taps 10,10,20,30 -> 24400 / 2048 = 11.91 -> 12      dx=2: 10,10,20,30 @.75 -> 34160 / 2048 = 16.68 -> 17
      dx=3: 10,20,30,40 @.25 -> 47040 / 2048 = 22.97 -> 23      dx=4: 10,20,30,40 @.75 -> 55360 / 2048 = 27.03 -> 27
      dx=5: 20,30,40,40 @.25 -> 68240 / 2048 = 33.32 -> 33      dx=6: 20,30,40,40 @.75 -> 78000 / 2048 = 38.09 -> 38
    and dx=0 -> src[0] = 10, dx=7 -> src[3] = 40 (blending clamped taps instead would give 9 and 41)."""
    w25, w75 = (-216, 1800, 536, -72), (-72, 536, 1800, -216)
    src = [10, 20, 30, 40]
    tap = lambda i: src[min(max(i, 0), 3)]
    want = [10]
    for dx in range(1, 7):
        f = (dx + 0.5) * 0.5 - 0.5
        s0 = int(np.floor(f))
        wts = w25 if f - s0 == 0.25 else w75
        want.append(int(np.rint(sum(tap(s0 - 1 + k) * wts[k] for k in range(4)) / 2048.0)))
    want.append(40)
    assert want == [10, 12, 17, 23, 27, 33, 38, 40]
    for c in (1, 3, 4):
        a = np.repeat(np.array(src, np.uint8)[None, :, None], c, axis=2)
        assert orc.cv_resize(a, 8, 1, orc.INTER_CUBIC)[0, :, 0].tolist() == want, c
        tall = np.repeat(a, 5, axis=0)                      # constant columns: the y pass must not change them
        assert (orc.cv_resize(tall, 8, 11, orc.INTER_CUBIC)[:, :, 0] == np.array(want)[None, :]).all()
    # LANCZOS4 and LINEAR follow the same x rule; the y axis has no such rule (rows blend clamped taps)
    col = np.array(src, np.uint8)[:, None, None]
    assert orc.cv_resize(np.array(src, np.uint8)[None, :, None], 8, 1, orc.INTER_LANCZOS4)[0, [0, 7], 0].tolist() == [10, 40]
    assert orc.cv_resize(col, 1, 8, orc.INTER_CUBIC)[[0, 7], 0, 0].tolist() == [9, 41]


def test_cubic_simd_and_scalar_vertical_paths_differ_by_at_most_one():
    arr = noise_image(200, 300, 4, 5)
    orc.lib.orc_set_cv_simd(1)
    a = orc.cv_resize(arr, 57, 41, orc.INTER_CUBIC)
    orc.lib.orc_set_cv_simd(0)
    b = orc.cv_resize(arr, 57, 41, orc.INTER_CUBIC)
    orc.lib.orc_set_cv_simd(1)
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1


def test_gaussian_properties():
    assert orc.lib.orc_gaussian_ksize(2.0) == 13 and orc.lib.orc_gaussian_ksize(25.0) == 151
    assert orc.lib.orc_gaussian_ksize(0.1) == 3 and orc.lib.orc_gaussian_ksize(0.01) == 1
    a = noise_image(40, 50, 3, 6)
    assert np.array_equal(orc.gaussian(a, 0.01), a)       # ksize 1 -> copy
    assert np.array_equal(orc.gaussian(a, 0.0), a)        # defined no-op (reference would assert)
    s = orc.gaussian(a, 3.0)
    assert s.std() < a.std() / 3 and abs(s.mean() - a.mean()) < 2


def test_hsv_known_values_and_ranges():
    px = np.array([[[0, 0, 255], [0, 255, 0], [255, 0, 0], [0, 0, 0], [255, 255, 255], [10, 200, 100]]], np.uint8)  # B,G,R
    hsv = orc.rgb2hsv(px)[0]
    assert hsv[0].tolist() == [0, 255, 255]          # red
    assert hsv[1].tolist() == [60, 255, 255]         # green
    assert hsv[2].tolist() == [120, 255, 255]        # blue
    assert hsv[3].tolist() == [0, 0, 0] and hsv[4].tolist() == [0, 0, 255]
    allpx = noise_image(64, 64, 3, 7)
    h = orc.rgb2hsv(allpx)
    assert h[:, :, 0].max() <= 179
    back = orc.hsv2rgb(h)
    assert np.abs(back.astype(int) - allpx.astype(int)).max() <= 12    # lossy by construction (integer hue), but close


def test_filter_codes_and_geometry_ops():
    a = noise_image(20, 30, 3, 8)
    assert orc.filter(a, "rotate=90")[1].shape == (30, 20, 3)
    r90 = orc.filter(a, "rotate=90")[1]
    assert np.array_equal(r90, np.rot90(a, k=-1))                       # clockwise
    assert np.array_equal(orc.filter(a, "rotate=270")[1], np.rot90(a, k=1))
    assert np.array_equal(orc.filter(a, "rotate=180")[1], a[::-1, ::-1])
    assert np.array_equal(orc.filter(a, "flip=10")[1], a[:, ::-1])      # first digit = horizontal
    assert np.array_equal(orc.filter(a, "flip=01")[1], a[::-1])
    assert orc.filter(a, "gotham=1", 0)[0] == 52 and orc.filter(a, "gotham=1", 1)[0] == 0
    assert orc.filter(a, "flip")[0] == 50 and orc.filter(a, "bogus=1")[0] == 52


def test_scanline_row_rule_closed_form():
    """rows freq <= (y mod (freq+width+1)) < freq+width are drawn."""
    a = np.full((40, 8, 3), 200, np.uint8)
    for freq, width in ((1, 1), (2, 3), (5, 2)):
        rc, out = orc.filter(a, "scanline=0,0,%d,%d" % (freq, width))     # drawn rows become V=0 -> black
        drawn = [y for y in range(40) if freq <= y % (freq + width + 1) < freq + width]
        assert rc == 0 and [y for y in range(40) if out[y, 0, 0] == 0] == drawn


def test_brightness_float_accumulator_is_not_the_true_mean():
    """Why the GPU replays the accumulation serially: on a large flat frame the float running sum stalls."""
    a = np.full((1080, 1920, 3), 100, np.uint8)
    b = orc.brightness(a)
    # past 2^27 the float sum can only move in steps of 16, so each +100.0 lands as +96: -2.7 % on this frame
    assert abs(b / (100 / 255) - 1) > 0.02
    assert int(round(b * 100)) == 38 and int(round(100 / 255 * 100)) == 39    # the JSON integer percent differs too


# ---- independent float cross-checks of the OpenCV-internal arithmetic the oracle restates from knowledge:
# none of these share code or tables with oracle/*.c; they pin sample centres, kernels, border rules and rounding to
# within the fixed-point error of OpenCV's 11-bit weights.
def _separable_resize(arr, dw, dh, taps):
    """Float reference: taps(frac) -> (first offset, weights); sample centre (d + 0.5) * scale - 0.5, replicate border;
    on the x axis OpenCV 2.4.9's edge rule: a centre left of pixel 0 or at / right of the last pixel snaps onto it."""
    def axis_matrix(ssize, dsize, is_x):
        scale = ssize / dsize
        m = np.zeros((dsize, ssize))
        for d in range(dsize):
            f = (d + 0.5) * scale - 0.5
            s = int(np.floor(f))
            if is_x and s < 0:
                f, s = 0.0, 0
            if is_x and s >= ssize - 1:
                f, s = float(ssize - 1), ssize - 1
            first, w = taps(f - s)
            for k, wk in enumerate(w):
                m[d, min(max(s + first + k, 0), ssize - 1)] += wk
        return m
    a = arr.astype(np.float64)
    my, mx = axis_matrix(arr.shape[0], dh, False), axis_matrix(arr.shape[1], dw, True)
    return np.einsum("ys,sxc->yxc", my, np.einsum("xs,ysc->yxc", mx, a))


def test_linear_matches_float_bilinear_within_one():
    import torch
    import torch.nn.functional as F

    for arr, size in ((smooth_image(120, 160, 4), (45, 77)), (noise_image(64, 48, 3, 11), (100, 90)), (noise_image(90, 70, 1, 12), (31, 29))):
        got = orc.cv_resize(arr, size[1], size[0], orc.INTER_LINEAR).astype(np.float64)
        t = torch.from_numpy(arr).permute(2, 0, 1)[None].double()
        ref = F.interpolate(t, size=size, mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(got - np.clip(np.rint(ref), 0, 255)).max() <= 1


def test_lanczos4_matches_windowed_sinc_within_two():
    def taps(x):                      # Lanczos a = 4 sampled at the 8 integer offsets -3..4, normalised (imgwarp.cpp)
        t = np.arange(-3, 5) - x
        w = np.sinc(t) * np.sinc(t / 4.0)
        return -3, w / w.sum()
    for arr, (dw, dh) in ((smooth_image(90, 120, 4), (50, 40)), (noise_image(60, 80, 3, 13), (75, 95)), (noise_image(128, 128, 4, 14), (64, 64))):
        got = orc.cv_resize(arr, dw, dh, orc.INTER_LANCZOS4).astype(np.float64)
        ref = np.clip(np.rint(_separable_resize(arr, dw, dh, taps)), 0, 255)
        d = np.abs(got - ref)
        assert d.max() <= 2 and (d > 1).mean() < 0.002          # 11-bit weights on 8 taps: rarely 2, never more


def test_area_matches_exact_overlap_average_within_one():
    def overlap_matrix(ssize, dsize):
        scale = ssize / dsize
        m = np.zeros((dsize, ssize))
        for d in range(dsize):
            lo, hi = d * scale, (d + 1) * scale
            for s in range(int(np.floor(lo)), min(int(np.ceil(hi)), ssize)):
                m[d, s] = max(0.0, min(hi, s + 1) - max(lo, s)) / scale
        return m
    for arr, (dw, dh) in ((noise_image(108, 192, 4, 15), (22, 23)), (smooth_image(100, 100, 3), (33, 77)), (noise_image(50, 70, 1, 16), (7, 49))):
        got = orc.cv_resize(arr, dw, dh, orc.INTER_AREA).astype(np.float64)
        my, mx = overlap_matrix(arr.shape[0], dh), overlap_matrix(arr.shape[1], dw)
        ref = np.einsum("ys,sxc->yxc", my, np.einsum("xs,ysc->yxc", mx, arr.astype(np.float64)))
        assert np.abs(got - np.clip(np.rint(ref), 0, 255)).max() <= 1


def test_gaussian_matches_float_convolution_within_one():
    from scipy.ndimage import correlate1d

    for sigma in (0.8, 2.0, 5.5):
        n = orc.lib.orc_gaussian_ksize(sigma)
        x = np.arange(n) - (n - 1) / 2
        k = np.exp(-x * x / (2 * sigma * sigma))
        k /= k.sum()                                            # getGaussianKernel, ksize = round(6 sigma + 1) | 1
        # the 8-bit path converts each coefficient with convertTo(CV_32S, 256) and does NOT renormalise, so the kernel
        # sums to 256 +- a few and the output carries that gain; the float reference uses the same rounded coefficients
        kq = np.rint(k.astype(np.float32) * 256.0) / 256.0
        a = noise_image(60, 70, 3, 17)
        ref = correlate1d(correlate1d(a.astype(np.float64), kq, axis=1, mode="nearest"), kq, axis=0, mode="nearest")
        got = orc.gaussian(a, sigma).astype(np.float64)
        assert np.abs(got - np.clip(np.rint(ref), 0, 255)).max() <= 1
        assert abs(kq.sum() - 1.0) < 0.05


# ---- independent pins for the hand-written filters (filters.c:524-729, helpers.c:70-176).  The product's host code
# (imp_args.cpp) and the oracle (oracle/orc_filters.c) were written from the same reading of the reference, so a shared
# misreading would pass every product-vs-oracle test.  Each check below restates the DOCUMENTED behaviour in float64
# numpy (docs/03 - Usage.md, the textbook colour maths) or as a closed form, shares no code or table with oracle/, and
# states its tolerance: the reference truncates where the textbook rounds, so +-1 is the usual bound.
def _textbook_hsv(bgr):
    """OpenCV-convention 8-bit HSV in float64: H in [0, 180), S and V in [0, 255]."""
    b, g, r = [bgr[..., i].astype(np.float64) for i in range(3)]
    v = np.maximum(np.maximum(r, g), b)
    mn = np.minimum(np.minimum(r, g), b)
    d = v - mn
    s = np.where(v > 0, 255.0 * d / np.maximum(v, 1), 0.0)
    dd = np.maximum(d, 1e-12)
    h = np.where(v == r, 30.0 * (g - b) / dd, np.where(v == g, 60.0 + 30.0 * (b - r) / dd, 120.0 + 30.0 * (r - g) / dd))
    h = np.where(d == 0, 0.0, h)
    h = np.where(h < 0, h + 180.0, h)
    return h, s, v


def _textbook_hsv_to_bgr(h, s, v):
    """Sector formula on float64 (h in half-degrees)."""
    hh = (h * 2.0) / 60.0
    i = np.floor(hh)
    f = hh - i
    s1 = s / 255.0
    p, q, t = v * (1 - s1), v * (1 - s1 * f), v * (1 - s1 * (1 - f))
    i = i.astype(int) % 6
    r = np.choose(i, [v, q, p, p, t, v])
    g = np.choose(i, [t, v, v, q, p, p])
    b = np.choose(i, [p, p, t, v, v, q])
    return np.stack([b, g, r], axis=-1)


def test_pin_a9_rgb2hsv_against_textbook_within_one():
    a = noise_image(64, 64, 3, 40)
    got = orc.rgb2hsv(a).astype(np.float64)
    h, s, v = _textbook_hsv(a)
    assert np.array_equal(got[..., 2], v)                            # value is the exact maximum
    # the reference truncates the integer quotients toward zero (helpers.c:89-97): within one unit below the real value
    assert np.all(np.abs(got[..., 1] - s) < 1.0 + 1e-9)
    dh = np.abs(got[..., 0] - h)
    dh = np.minimum(dh, 180.0 - dh)
    assert np.all(dh[s >= 1.0] < 1.0 + 1e-9)                         # (s < 1 truncates to 0, where the reference skips the hue)


def test_pin_a9_hsv2rgb_against_sector_formula_within_one():
    rng = np.random.Generator(np.random.PCG64(41))
    hsv = np.stack([rng.integers(0, 180, (48, 48)), rng.integers(0, 256, (48, 48)), rng.integers(0, 256, (48, 48))], axis=-1).astype(np.uint8)
    got = orc.hsv2rgb(hsv).astype(np.float64)
    want = _textbook_hsv_to_bgr(hsv[..., 0].astype(np.float64), hsv[..., 1].astype(np.float64), hsv[..., 2].astype(np.float64))
    # p, q, t are truncated to int (helpers.c:132-134) from float32 products: never above the real value, less than 1 below
    d = want - got
    assert d.min() > -1e-3 and d.max() < 1.0 + 1e-3


def test_pin_a10_modulate_closed_forms():
    a = noise_image(40, 50, 3, 42)
    # saturation 0: every pixel becomes its own maximum on all channels, exactly (HSV2RGB's s == 0 branch)
    rc, out = orc.filter(a, "modulate=0,0,100")
    assert rc == 0 and np.array_equal(out, np.repeat(a.max(axis=2, keepdims=True), 3, axis=2))
    # brightness 50 %: value halves (docs/03: "brightness ... percent"), so does every channel -- up to the 8-bit HSV trip:
    # the hue is an integer count of half-degrees, worth chroma / 30 per unit, plus the truncations of S, p, q, t
    rc, out = orc.filter(a, "modulate=0,100,50")
    chroma = (a.max(axis=2, keepdims=True).astype(float) - a.min(axis=2, keepdims=True)) * 0.5
    assert (np.abs(out.astype(float) - a.astype(float) * 0.5) <= chroma / 30.0 + 2.5).all()
    assert np.array_equal(out.max(axis=2), a.max(axis=2) // 2)                 # the value channel itself is exact
    # hue +90 half-degrees = 180 degrees: primaries turn into their complements exactly
    prim = np.array([[[0, 0, 255], [0, 255, 0], [255, 0, 0], [0, 255, 255]]], np.uint8)          # B,G,R: red, green, blue, yellow
    rc, out = orc.filter(prim, "modulate=90,100,100")
    assert out[0].tolist() == [[255, 255, 0], [255, 0, 255], [0, 255, 255], [255, 0, 0]]        # cyan, magenta, yellow, blue
    # documented argument ranges (filters.c:148-153)
    assert orc.filter(a, "modulate=181,100,100")[0] == 50 and orc.filter(a, "modulate=0,100,0")[0] == 50


def test_pin_a11_colorize_is_a_weighted_average_within_one():
    a = noise_image(30, 40, 4, 43)
    for color, op in (("ff8000", 0.3), ("102030", 0.5), ("00ff00", 0.85)):
        rc, out = orc.filter(a, "colorize=%s,%s" % (color, op))
        rgb = [int(color[i:i + 2], 16) for i in (0, 2, 4)]
        want = (1 - op) * a[..., :3].astype(np.float64) + np.array(rgb[::-1]) * op      # memory order is B,G,R
        d = want - out[..., :3].astype(np.float64)
        assert rc == 0 and d.min() > -0.01 and d.max() < 1.01                           # truncated, float32 arithmetic
        assert np.array_equal(out[..., 3], a[..., 3])                                   # alpha untouched (filters.c:613)
    assert np.array_equal(orc.filter(a, "colorize=123456,0")[1], a)
    solid = orc.filter(a, "colorize=123456,1")[1]
    assert (solid[..., 0] == 0x56).all() and (solid[..., 1] == 0x34).all() and (solid[..., 2] == 0x12).all()


def test_pin_a12_gamma_closed_form():
    a = noise_image(30, 40, 4, 44)
    assert np.array_equal(orc.filter(a, "gamma=1")[1], a)
    for g in (2.2, 0.45, 1.6):
        want = np.floor(255.0 * (a.astype(np.float64) / 255.0) ** (1.0 / g) + 1e-9)
        out = orc.filter(a, "gamma=%s" % g)[1].astype(np.float64)
        assert np.abs(out - want).max() <= 1          # 1/g is a float in the reference (filters.c:562); alpha included (:554)
        assert (out[a == 0] == 0).all() and (out[a == 255] == 255).all()


def test_pin_a13_contrast_closed_form():
    a = noise_image(30, 40, 4, 45)
    assert np.array_equal(orc.filter(a, "contrast=1")[1], a)
    out = orc.filter(a, "contrast=2")[1]
    assert np.array_equal(out[..., :3], np.minimum(a[..., :3].astype(int) * 2, 255).astype(np.uint8))     # exact in float32
    assert np.array_equal(out[..., 3], a[..., 3])                                                         # channels < 3 only (filters.c:599)
    half = orc.filter(a, "contrast=0.5")[1]
    assert np.array_equal(half[..., :3], a[..., :3] // 2)
    assert orc.filter(a, "contrast=0")[0] == 50 and orc.filter(a, "contrast=-1")[0] == 50


def test_pin_a14_gradmap_endpoints_and_midpoints():
    gray = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)              # (R+G+B)/3 = the level itself
    lvl = np.arange(256, dtype=np.float64)
    rc, out = orc.filter(gray, "gradmap=000000,ffffff")
    # 256 steps from the first colour towards (never reaching) the second: level i -> round(255 i / 256) (filters.c:583-586)
    assert rc == 0 and np.array_equal(out[0, :, 0], np.floor(lvl * 255 / 256 + 0.5).astype(np.uint8)) and out[0, 255, 0] == 254
    assert (out[..., 0] == out[..., 1]).all() and (out[..., 1] == out[..., 2]).all()
    rc, out = orc.filter(gray, "gradmap=ff0000,0000ff")                                     # colours are R,G,B; memory is B,G,R
    assert np.array_equal(out[0, :, 2], np.floor(255 - lvl * 255 / 256 + 0.5).astype(np.uint8))    # red falls ...
    assert np.array_equal(out[0, :, 0], np.floor(lvl * 255 / 256 + 0.5).astype(np.uint8))          # ... blue rises, green stays 0
    assert (out[0, :, 1] == 0).all()
    rc, out = orc.filter(gray, "gradmap=000000,ff0000,ffffff")                              # three colours: two segments of 128
    assert out[0, 0].tolist() == [0, 0, 0] and out[0, 128].tolist() == [0, 0, 255] and out[0, 64].tolist() == [0, 0, 128]
    assert out[0, 192].tolist() == [128, 128, 255]
    # a colourful pixel is indexed by its integer mean (filters.c:267-271)
    px = np.array([[[10, 20, 40]]], np.uint8)
    assert orc.filter(px, "gradmap=000000,ffffff")[1][0, 0].tolist() == [23, 23, 23]         # (10+20+40)/3 = 23 -> round(22.9)


def test_pin_a16_vignette_on_gray_follows_cos4():
    h, w = 61, 81
    a = np.full((h, w, 3), 200, np.uint8)
    intensity, radius = 0.8, 1.0
    rc, out = orc.filter(a, "vignette=%s,%s" % (intensity, radius))
    cx, cy = w // 2, h // 2
    yy, xx = np.mgrid[0:h, 0:w]
    d = np.sqrt((xx - cx) ** 2.0 + (yy - cy) ** 2.0)
    maxd = max(np.hypot(cx, cy), np.hypot(w - cx, cy), np.hypot(cx, h - cy), np.hypot(w - cx, h - cy))    # corners at (w, h): helpers.c:52-55
    want = 200.0 * np.cos(d / (radius * maxd) * intensity) ** 4
    # gray stays gray (S = 0) and V is truncated (filters.c:316): +-1 covers float32 vs float64
    assert rc == 0 and (out[..., 0] == out[..., 1]).all() and (out[..., 1] == out[..., 2]).all()
    diff = want - out[..., 0].astype(np.float64)
    assert diff.min() > -1.0 and diff.max() < 1.0 + 1e-6
    assert out[cy, cx, 0] == 200


def test_pin_a17_lomo_kelvin_gotham():
    a = noise_image(30, 40, 4, 46)
    lomo = orc.filter(a, "lomo=1")[1]
    want = np.clip(np.trunc(a[..., 1:3].astype(np.float64) * 1.5 - 50), 0, 255)             # G and R only (filters.c:340)
    assert np.array_equal(lomo[..., 1:3], want.astype(np.uint8)) and np.array_equal(lomo[..., [0, 3]], a[..., [0, 3]])
    # Kelvin and Gotham are fixed chains of the public filters (filters.c:325-354): the same bytes as spelling them out
    k = orc.filter(orc.filter(a, "modulate=120,50,100")[1], "colorize=ff9900,0.5")[1]
    assert np.array_equal(orc.filter(a, "kelvin=1")[1], k)
    g = orc.filter(orc.filter(orc.filter(a, "modulate=120,5,100")[1], "colorize=111b5d,0.15")[1], "gamma=0.3")[1]
    # last stage: brightness -0.07, contrast 1.5 -> trunc(1.5 v - 17.85) clamped, channels B,G,R (no public spelling)
    want = np.clip(np.trunc(1.5 * g[..., :3].astype(np.float64) + np.float32(-0.07) * np.float32(255)), 0, 255)
    got = orc.filter(a, "gotham=1")[1]
    assert np.abs(got[..., :3].astype(np.float64) - want).max() <= 1 and np.array_equal(got[..., 3], g[..., 3])


def test_pin_a18_rainbow_bands():
    # one pixel per documented band centre (filters.c:375-394), value 200: the band's hue at full saturation
    def pure(hue_deg, v=200):
        return _textbook_hsv_to_bgr(np.array([[hue_deg / 2.0]]), np.array([[255.0]]), np.array([[float(v)]]))[0, 0]
    cases = {0: 0, 20: 30, 50: 60, 100: 120, 170: 195, 230: 225, 300: 285, 350: 0}
    for hue_in, hue_out in cases.items():
        px = np.rint(pure(hue_in)).astype(np.uint8)[None, None, :]
        got = orc.filter(px, "rainbow=full")[1][0, 0].astype(np.float64)
        want = pure((hue_out // 2) * 2)                                                     # the stored hue is an integer half-degree count
        assert np.abs(got - want).max() <= 1.0, (hue_in, got, want)
    dark = np.array([[[5, 10, 15]]], np.uint8)
    assert orc.filter(dark, "rainbow=full")[1][0, 0].tolist() == [0, 0, 0]                  # V < 20 -> black
    white = np.array([[[255, 255, 255]]], np.uint8)
    assert orc.filter(white, "rainbow=mid")[1][0, 0].tolist() == [255, 255, 255]            # V > 254 -> unsaturated
    assert orc.filter(white, "rainbow=sepia")[0] == 50


def test_pin_a6_alpha_blend_is_porter_duff_over_with_subtracted_opacity():
    rng = np.random.Generator(np.random.PCG64(47))
    base = rng.integers(0, 256, (40, 60, 4), dtype=np.uint8)
    ov = rng.integers(0, 256, (16, 24, 4), dtype=np.uint8)
    for opacity in (100, 60, 25):
        rc, out = orc.watermark(base, ov, "l", "t", 3, 5, opacity)
        sa = np.maximum(ov[..., 3] / 255.0 - (1 - opacity / 100.0), 0.0)                    # opacity SUBTRACTS from alpha (filters.c:620,642)
        da = base[5:21, 3:27, 3] / 255.0
        ta = sa + da * (1 - sa)
        num = ov[..., :3] * sa[..., None] + base[5:21, 3:27, :3] * (da * (1 - sa))[..., None]
        want = np.where(ta[..., None] > 0, num / np.maximum(ta[..., None], 1e-30), 0.0)
        region = out[5:21, 3:27].astype(np.float64)
        d = want - region[..., :3]
        assert rc == 0 and d.min() > -1.01 and d.max() < 1.01                                # truncation + float32
        assert np.abs(ta * 255.0 - region[..., 3]).max() < 1.01
        untouched = out.copy()
        untouched[5:21, 3:27] = base[5:21, 3:27]
        assert np.array_equal(untouched, base)                                              # nothing outside the overlay rectangle
    # 3-channel destination: destination alpha is 1, no alpha written
    rc, out = orc.watermark(base[..., :3].copy(), ov, "r", "b", 0, 0, 100)
    sa = ov[..., 3] / 255.0
    want = ov[..., :3] * sa[..., None] + base[-16:, -24:, :3] * (1 - sa)[..., None]
    d = want - out[-16:, -24:].astype(np.float64)
    assert rc == 0 and d.min() > -1.01 and d.max() < 1.01


def test_pin_a8_flatten_on_white_paper():
    a = noise_image(30, 40, 4, 48)
    out = orc.blend_with_paper(a)
    al = a[..., 3:4].astype(np.float64) / 255.0
    want = 255.0 * (1 - al) + a[..., :3] * al                                               # white paper under the pixel
    d = want - out[..., :3].astype(np.float64)
    assert d.min() > -0.01 and d.max() < 1.01 and (out[..., 3] == 255).all()
    opaque = a.copy()
    opaque[..., 3] = 255
    assert np.array_equal(orc.blend_with_paper(opaque), opaque)
    clear = a.copy()
    clear[..., 3] = 0
    assert (orc.blend_with_paper(clear)[..., :3] == 255).all()


def test_pin_a19_brightness_is_the_weighted_rms_mean():
    for c, seed in ((3, 49), (4, 50)):
        a = noise_image(48, 64, c, seed)                                                    # small: the float accumulator is still exact to 1e-6
        b, g, r = [a[..., i].astype(np.float64) for i in range(3)]
        want = np.sqrt(0.241 * r * r + 0.691 * g * g + 0.068 * b * b).mean() / 255.0
        got = orc.brightness(a)
        assert abs(got - want) < 1e-4 and int(round(got * 100)) == int(round(want * 100))  # Info() reports the integer percent
    gray = noise_image(20, 30, 1, 51)
    assert abs(orc.brightness(gray) - gray.mean() / 255.0) < 1e-5
    assert orc.brightness(np.full((8, 8, 3), 255, np.uint8)) == pytest.approx(1.0, abs=1e-6)
