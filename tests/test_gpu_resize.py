"""GPU parity: cvResize (bridge.c:191) kernels vs the CPU oracle, through the C ABI. Bit-exact."""
import numpy as np
import pytest

import oracle_lib as orc
from conftest import noise_image, smooth_image

pytestmark = pytest.mark.gpu

MODES = [orc.INTER_NN, orc.INTER_LINEAR, orc.INTER_CUBIC, orc.INTER_AREA, orc.INTER_LANCZOS4]
NAMES = {0: "nn", 1: "linear", 2: "cubic", 3: "area", 4: "lanczos4"}


def gpu_resize(imp, arr, dw, dh, interp):
    im = imp.Image(arr)
    rc = im.cv_resize(dw, dh, interp)
    assert rc == 0, rc
    out = im.numpy()
    im.release()
    return out


@pytest.mark.parametrize("interp", MODES, ids=lambda m: NAMES[m])
@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("shape", [((48, 64), (17, 23)), ((29, 37), (13, 11)), ((100, 100), (50, 50)),
                                   ((90, 120), (30, 40)), ((61, 53), (61, 53)), ((8, 8), (1, 1)), ((5, 300), (3, 7))])
def test_downscale_bit_exact(gpu, interp, c, shape):
    (sh, sw), (dh, dw) = shape
    for arr in (noise_image(sh, sw, c, 1), smooth_image(sh, sw, c)):
        want = orc.cv_resize(arr, dw, dh, interp)
        got = gpu_resize(gpu, arr, dw, dh, interp)
        assert np.array_equal(got, want), "max diff %d" % np.abs(got.astype(int) - want.astype(int)).max()


@pytest.mark.parametrize("interp", [orc.INTER_NN, orc.INTER_LINEAR, orc.INTER_CUBIC, orc.INTER_LANCZOS4], ids=lambda m: NAMES[m])
@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("shape", [((17, 23), (48, 64)), ((13, 11), (29, 37)), ((1, 1), (9, 5)), ((2, 3), (40, 41)),
                                   ((30, 40), (20, 80))])
def test_upscale_and_mixed_bit_exact(gpu, interp, c, shape):
    (sh, sw), (dh, dw) = shape
    arr = noise_image(sh, sw, c, 2)
    want = orc.cv_resize(arr, dw, dh, interp)
    got = gpu_resize(gpu, arr, dw, dh, interp)
    assert np.array_equal(got, want)


def test_area_upscale_rejected(gpu):
    im = gpu.Image(noise_image(10, 10, 4, 3))
    assert im.cv_resize(20, 20, orc.INTER_AREA) == gpu.IMP_ERROR_INVALID_ARGS
    im.release()


@pytest.mark.parametrize("interp", [orc.INTER_CUBIC, orc.INTER_AREA], ids=lambda m: NAMES[m])
def test_headline_geometry_1080p_to_224(gpu, interp):
    """BASELINE cfg2 geometry on a few frames: 1920x1080 BGRA -> 224x224."""
    for seed in range(2):
        arr = noise_image(1080, 1920, 4, 10 + seed) if seed == 0 else smooth_image(1080, 1920, 4)
        want = orc.cv_resize(arr, 224, 224, interp)
        got = gpu_resize(gpu, arr, 224, 224, interp)
        assert np.array_equal(got, want)


def test_cfg4_geometry_4k_to_1080p_lanczos(gpu):
    arr = noise_image(2160, 3840, 4, 20)
    want = orc.cv_resize(arr, 1920, 1080, orc.INTER_LANCZOS4)
    got = gpu_resize(gpu, arr, 1920, 1080, orc.INTER_LANCZOS4)
    assert np.array_equal(got, want)


def test_resize_args_reference_rule(gpu):
    """Resize() picks AREA when shrinking, CUBIC when any axis grows, NN when simple (bridge.c:190)."""
    arr = noise_image(60, 80, 3, 5)
    for args, simple in [("40,30", 0), ("40", 0), ("0,30", 0), ("160,120,up", 0), ("160,20,up", 0), ("40,30", 1), ("500,500", 0)]:
        rc_o, want = orc.resize(arr, args, 2000, 2000, simple)
        im = gpu.Image(arr)
        rc = im.resize(args, gpu.Config(), simple)
        assert rc == rc_o == 0
        assert np.array_equal(im.numpy(), want), args
        im.release()


def test_batch_matches_single(gpu):
    import ctypes as C
    n, sh, sw, dh, dw = 5, 120, 160, 33, 47
    frames = [noise_image(sh, sw, 4, 30 + i) for i in range(n)]
    src = gpu.Image(np.concatenate(frames, axis=0))            # n frames stacked: stride = sh*sw*4
    dst = gpu.Image(np.zeros((n * dh, dw, 4), np.uint8))
    for interp in MODES:
        gpu.batch_cv_resize(src.device_ptr, sh * sw * 4, sw, sh, sw * 4, dst.device_ptr, dh * dw * 4, dw, dh, dw * 4,
                            4, n, interp)
        out = dst.numpy().reshape(n, dh, dw, 4)
        for i in range(n):
            assert np.array_equal(out[i], orc.cv_resize(frames[i], dw, dh, interp)), (interp, i)
    src.release(); dst.release()


@pytest.mark.parametrize("c", [3, 4])
@pytest.mark.parametrize("interp", [orc.INTER_LINEAR, orc.INTER_CUBIC, orc.INTER_LANCZOS4], ids=lambda m: NAMES[m])
@pytest.mark.parametrize("shape", [(16, 16), (122, 130), (240, 258), (2, 200), (300, 18), (64, 1000)])
def test_exact_2x_decimation_rolling_kernel(gpu, interp, shape, c):
    """Exact halves go through the register-rolling kernels (strips of 60 rows, 64 columns; BGRA and BGR forms): strip
    seams, partial last strips, partial column blocks and the clamped borders all sit inside these shapes."""
    sh, sw = shape
    for arr in (noise_image(sh, sw, c, 70), smooth_image(sh, sw, c)):
        want = orc.cv_resize(arr, sw // 2, sh // 2, interp)
        got = gpu_resize(gpu, arr, sw // 2, sh // 2, interp)
        assert np.array_equal(got, want), "max diff %d" % np.abs(got.astype(int) - want.astype(int)).max()


@pytest.mark.parametrize("interp", [orc.INTER_CUBIC, orc.INTER_LANCZOS4], ids=lambda m: NAMES[m])
@pytest.mark.parametrize("count", [1, 5, 8, 11])
@pytest.mark.parametrize("pad", [0, 32, 4], ids=["tight", "pitch+32", "pitch+4"])
def test_exact_2x_batch_dma_ring(gpu, interp, count, pad):
    """Exact halves of a resident batch: the LDS-DMA row-ring kernel (16-byte aligned pitches: pad 0 / 32) with its
    frame-per-XCD block order (batch sizes around the group of 8), a partial last column strip (260 = 4 * 64 + 4),
    a 5-row last row strip, and the register-rolling fallback when the pitch is only 4-byte aligned (pad 4)."""
    sh, sw = 250, 520
    dh, dw = sh // 2, sw // 2
    sstep = sw * 4 + pad
    frames = [noise_image(sh, sw, 4, 90 + i) for i in range(count)]
    packed = np.zeros((count, sh, sstep), np.uint8)
    for i, f in enumerate(frames):
        packed[i, :, :sw * 4] = f.reshape(sh, sw * 4)
    src = gpu.Image(packed.reshape(count * sh, sstep // 4, 4))
    dst = gpu.Image(np.zeros((count * dh, dw, 4), np.uint8))
    gpu.batch_cv_resize(src.device_ptr, sh * sstep, sw, sh, sstep, dst.device_ptr, dh * dw * 4, dw, dh, dw * 4, 4, count, interp)
    out = dst.numpy().reshape(count, dh, dw, 4)
    for i in range(count):
        assert np.array_equal(out[i], orc.cv_resize(frames[i], dw, dh, interp)), i
    src.release(); dst.release()


@pytest.mark.parametrize("interp", [orc.INTER_CUBIC, orc.INTER_LANCZOS4], ids=lambda m: NAMES[m])
@pytest.mark.parametrize("shape", [(4, 4), (2, 8), (12, 4), (14, 136), (126, 128), (130, 132), (480, 640), (64, 1028)])
def test_exact_2x_dma_ring_shapes(gpu, interp, shape):
    """Widths that are multiples of 4 take the DMA ring: windows narrower than the tap count, one-strip images, a
    strip boundary exactly at the image edge, and rows fewer than the ring depth."""
    sh, sw = shape
    for arr in (noise_image(sh, sw, 4, 75), smooth_image(sh, sw, 4)):
        want = orc.cv_resize(arr, sw // 2, sh // 2, interp)
        got = gpu_resize(gpu, arr, sw // 2, sh // 2, interp)
        assert np.array_equal(got, want), "max diff %d" % np.abs(got.astype(int) - want.astype(int)).max()


@pytest.mark.parametrize("c", [3, 4])
def test_area_row_groups_on_a_large_batch(gpu, c):
    """Big launches take the row-grouped AREA kernels (four destination rows per lane): 137 frames (not a multiple of
    the XCD group of 8), 75 destination rows (a partial last group of 3), fractional scales on both axes."""
    n, sh, sw, dh, dw = 137, 241, 322, 75, 100
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (n, sh, sw, c), dtype=np.uint8)
    src = gpu.Image(frames.reshape(n * sh, sw, c))
    dst = gpu.Image(np.zeros((n * dh, dw, c), np.uint8))
    sstep, dstep = src.step, dst.step
    gpu.batch_cv_resize(src.device_ptr, sh * sstep, sw, sh, sstep, dst.device_ptr, dh * dstep, dw, dh, dstep, c, n, orc.INTER_AREA)
    out = dst.numpy().reshape(n, dh, dw, c)
    for i in list(range(0, n, 9)) + [n - 1]:
        assert np.array_equal(out[i], orc.cv_resize(frames[i], dw, dh, orc.INTER_AREA)), i
    src.release(); dst.release()


@pytest.mark.parametrize("interp", [orc.INTER_CUBIC, orc.INTER_LANCZOS4], ids=lambda m: NAMES[m])
@pytest.mark.parametrize("shape", [(16, 16), (2, 32), (14, 144), (126, 128), (130, 272), (250, 528), (480, 640), (64, 1024)])
def test_exact_2x_bgr_dma_ring_shapes(gpu, interp, shape):
    """3-channel frames whose width is a multiple of 16 take the byte-granular DMA ring (k_resize_2x_dma3): one-strip
    images, a partial last column strip (264 = 4 * 64 + 8), strip seams every 60 rows, windows clipped at both borders."""
    sh, sw = shape
    for arr in (noise_image(sh, sw, 3, 77), smooth_image(sh, sw, 3)):
        want = orc.cv_resize(arr, sw // 2, sh // 2, interp)
        got = gpu_resize(gpu, arr, sw // 2, sh // 2, interp)
        assert np.array_equal(got, want), "max diff %d" % np.abs(got.astype(int) - want.astype(int)).max()


@pytest.mark.parametrize("interp", [orc.INTER_CUBIC, orc.INTER_LANCZOS4], ids=lambda m: NAMES[m])
@pytest.mark.parametrize("count", [1, 5, 9])
@pytest.mark.parametrize("pad", [0, 16, 4], ids=["tight", "pitch+16", "pitch+4"])
def test_exact_2x_bgr_batch(gpu, interp, count, pad):
    """Resident 3-channel batches: the DMA ring with 16-byte aligned pitches (frame-per-XCD order around the group of
    8), and the register-rolling fallback when the pitch is only 4-byte aligned."""
    sh, sw = 126, 528
    dh, dw = sh // 2, sw // 2
    sstep = sw * 3 + pad
    rng = np.random.default_rng(97)
    frames = rng.integers(0, 256, (count, sh, sw, 3), dtype=np.uint8)
    packed = np.zeros((count, sh, sstep), np.uint8)
    packed[:, :, :sw * 3] = frames.reshape(count, sh, sw * 3)
    src = gpu.Image(packed.reshape(count * sh, sstep // 4, 4))            # raw bytes; the batch call gives the real geometry
    dst = gpu.Image(np.zeros((count * dh, dw, 3), np.uint8))
    assert dst.step == dw * 3
    gpu.batch_cv_resize(src.device_ptr, sh * sstep, sw, sh, sstep, dst.device_ptr, dh * dw * 3, dw, dh, dw * 3, 3, count, interp)
    out = dst.numpy().reshape(count, dh, dw, 3)
    for i in range(count):
        assert np.array_equal(out[i], orc.cv_resize(frames[i], dw, dh, interp)), i
    src.release(); dst.release()


@pytest.mark.parametrize("c", [3, 4])
def test_dma_ring_results_do_not_depend_on_timing(gpu, c):
    """The ring kernels order their own LDS DMA against their own reads with hand-written waits; a mistake there shows
    as an occasional wrong row (profiles/r01_dma_stress.txt), so repeat the same resize and demand the same bytes."""
    arr = noise_image(480, 1024, c, 99)
    for interp in (orc.INTER_LANCZOS4, orc.INTER_CUBIC):
        want = orc.cv_resize(arr, 512, 240, interp)
        for rep in range(12):
            assert np.array_equal(gpu_resize(gpu, arr, 512, 240, interp), want), (NAMES[interp], rep)


@pytest.mark.parametrize("shape", [((270, 480), (1080, 1920)), ((100, 161), (333, 515)), ((64, 4), (200, 9)), ((50, 100), (200, 80)),
                                   ((37, 41), (37, 123)), ((90, 7), (91, 8)), ((33, 200), (1000, 230)), ((50, 100), (200, 128)),
                                   ((60, 40), (61, 256)), ((30, 70), (90, 64)), ((61, 100), (122, 200)), ((45, 64), (135, 192)),
                                   ((33, 64), (66, 64)), ((201, 64), (804, 128)), ((150, 32), (300, 128))])
@pytest.mark.parametrize("c", [3, 4])
def test_cubic_enlargement_kernel_bit_exact(gpu, shape, c):
    """The reference's only CUBIC dispatch (bridge.c:190: an axis grows): k_resize_up_cubic4 / _cubic3 -- wave-private strips,
    float H sums in a register ring, scalar row weights.  Full 480x270 -> 1080p, odd widths (a scalar tail in every row),
    4-pixel-wide sources, x shrinking while y grows, the edge columns the 2.4.9 x rule pins to src[0] / src[w-1], and exact
    2x / 3x / 4x heights (BGRA: the unrolled path whose footprint advances on a fixed schedule; several chunks per strip)."""
    (sh, sw), (dh, dw) = shape
    for arr in (noise_image(sh, sw, c, 21), smooth_image(sh, sw, c)):
        want = orc.cv_resize(arr, dw, dh, orc.INTER_CUBIC)
        got = gpu_resize(gpu, arr, dw, dh, orc.INTER_CUBIC)
        assert np.array_equal(got, want), "max diff %d" % np.abs(got.astype(int) - want.astype(int)).max()
        if dw > sw:     # OpenCV 2.4.9's x-edge rule: the outermost columns are the source's, resampled vertically only
            col = orc.cv_resize(arr[:, :1], 1, dh, orc.INTER_CUBIC)          # (a 1-pixel row takes the integer vertical form,
            assert np.abs(got[:, :1].astype(int) - col).max() <= 1          #  a wide one the SSE2 float form: within one LSB)


@pytest.mark.parametrize("c", [3, 4])
@pytest.mark.parametrize("shape", [(540, 960), (64, 64), (33, 45), (7, 3), (1, 1), (100, 2), (12, 1023), (300, 10), (9, 6)])
def test_area_2x2_streaming_kernel_bit_exact(gpu, c, shape):
    """(a+b+c+d+2)>>2: k_area2x2_c4 (BGRA, even widths: contiguous granules, short rows wrap inside a wave's run),
    k_area2x2_v4 / _v3 (four destination pixels per lane; widths that leave a partial quad)."""
    dh, dw = shape
    arr = noise_image(2 * dh, 2 * dw, c, 22)
    box = (arr[0::2, 0::2].astype(int) + arr[0::2, 1::2] + arr[1::2, 0::2] + arr[1::2, 1::2] + 2) >> 2
    got = gpu_resize(gpu, arr, dw, dh, orc.INTER_AREA)
    assert np.array_equal(got, box.astype(np.uint8)) and np.array_equal(got, orc.cv_resize(arr, dw, dh, orc.INTER_AREA))


def test_area_2x2_on_a_cropped_view_and_batch(gpu):
    """Crop folded into the box filter's source view (unaligned 16-byte loads) and a batch through the batch entry point."""
    arr = noise_image(200, 300, 4, 23)
    im = gpu.Image(arr)
    cfg = gpu.Config()
    rc, step = gpu.run_ops(im, cfg, crop="101px,66px,7px,5px", resize="50,33")       # (101, 66) -> clamp -> AREA general, not 2x
    assert rc == 0
    im.release()
    im = gpu.Image(arr)
    rc, step = gpu.run_ops(im, cfg, crop="100px,66px,7px,5px", resize="50,33")       # exact 2x of the 100 x 66 window at (7, 5)
    want = orc.resize(orc.crop(arr, "100px,66px,7px,5px")[1], "50,33")[1]
    assert rc == 0 and np.array_equal(im.numpy(), want)
    im.release()


@pytest.mark.parametrize("scale", [(3, 3), (4, 4), (4, 3), (3, 5), (5, 2), (6, 6), (7, 1), (8, 8), (8, 16), (2, 3), (9, 9), (16, 16)])
@pytest.mark.parametrize("dims", [(270, 480), (33, 45), (5, 3), (1, 1), (17, 130)])
@pytest.mark.parametrize("c", [4, 3])
def test_area_integer_scales_streaming_kernel_bit_exact(gpu, scale, dims, c):
    """resizeAreaFast_ at integer scales other than 2x2: k_area_boxc<4|8> (contiguous granules, rows shorter than a wave's
    run wrap inside it; 8 only), k_area_boxl<4,ISX> for 3..7 (any ISY with ISX*ISY <= 257) and the per-pixel fallback beyond;
    widths that leave a partial quad; saturate(cvRound(sum * (1.f / area)))."""
    isx, isy = scale
    dh, dw = dims
    if dh * isy * dw * isx > 6_000_000:
        pytest.skip("frame larger than this test needs")
    arr = noise_image(dh * isy, dw * isx, c, 24)
    want = orc.cv_resize(arr, dw, dh, orc.INTER_AREA)
    got = gpu_resize(gpu, arr, dw, dh, orc.INTER_AREA)
    assert np.array_equal(got, want), "max diff %d" % np.abs(got.astype(int) - want.astype(int)).max()
    if isx * isy > 1:
        s = arr.reshape(dh, isy, dw, isx, c).astype(np.int64).sum(axis=(1, 3))
        mean = np.rint((s.astype(np.float32) * np.float32(1.0 / (isx * isy))).astype(np.float64))      # float32 product, half-even
        assert np.abs(got.astype(int) - mean).max() <= 1


@pytest.mark.parametrize("c", [4, 3])
@pytest.mark.parametrize("geom", [((220, 300), (260, 190)), ((300, 520), (256, 150)), ((400, 900), (257, 115)), ((231, 333), (300, 200))])
def test_area_small_factors_four_columns_per_lane(gpu, c, geom):
    """k_resize_area_rows4: shrink factors below ~3.9 on a batch big enough to take it (a lane owns four adjacent
    destination columns, a wave 256): widths that leave a partial quad and a partial last strip, source widths that are
    not a multiple of 4 (the moved-back last granule), BGR byte windows; every frame against the oracle."""
    (sh, sw), (dw, dh) = geom
    n = 72
    rng = np.random.Generator(np.random.PCG64(0x1A4D8800 + sw + c))
    sstep, dstep = (sw * c + 3) & ~3, (dw * c + 3) & ~3
    frames = rng.integers(0, 256, size=(n, sh, sw, c), dtype=np.uint8)
    src = gpu.Image(frames.reshape(n * sh, sw, c))
    dst = gpu.Image(np.zeros((n * dh, dw, c), np.uint8))
    assert src.step == sstep and dst.step == dstep
    gpu.batch_cv_resize(src.device_ptr, sh * sstep, sw, sh, sstep, dst.device_ptr, dh * dstep, dw, dh, dstep, c, n, orc.INTER_AREA)
    out = dst.numpy().reshape(n, dh, dw, c)
    for i in range(0, n, 5):
        assert np.array_equal(out[i], orc.cv_resize(frames[i], dw, dh, orc.INTER_AREA)), (geom, c, i)
    src.release(); dst.release()


# ---- k_resize_strip2: the static schedule for footprint advances of period two (exact 2x enlargement, 1.5x reduction)
@pytest.mark.parametrize("c", [3, 4])
@pytest.mark.parametrize("interp", [orc.INTER_LINEAR, orc.INTER_LANCZOS4], ids=lambda m: NAMES[m])
@pytest.mark.parametrize("src,dst", [((96, 54), (192, 108)),        # 2x up: rows advance 1, 0, 1, 0
                                     ((333, 77), (666, 154)),       # 2x up, strips that end inside a 16-row block
                                     ((300, 150), (200, 100)),      # 1.5x down: 1, 2 / 2, 1
                                     ((999, 303), (666, 202)),      # 1.5x down, odd sizes
                                     ((64, 9), (128, 18)),          # shorter than one block
                                     ((150, 90), (300, 60))])       # x grows 2x, y shrinks 1.5x
def test_strip_static_schedule(gpu, interp, c, src, dst):
    (sw, sh), (dw, dh) = src, dst
    rng = np.random.default_rng(sw * 7 + dh + c)
    for count in (1, 3):
        frames = [rng.integers(0, 256, size=(sh, sw, c), dtype=np.uint8) for _ in range(count)]
        for f in frames:
            assert np.array_equal(gpu_resize(gpu, f, dw, dh, interp), orc.cv_resize(f, dw, dh, interp))


@pytest.mark.parametrize("c", [3, 4])
@pytest.mark.parametrize("src,dst", [((150, 90), (300, 60)), ((101, 303), (180, 202)), ((64, 1500), (100, 1000))])
def test_strip_static_schedule_cubic(gpu, c, src, dst):
    """CUBIC reaches the strips when x grows while y shrinks (bridge.c:190); y by 1.5 is the period-two pattern"""
    (sw, sh), (dw, dh) = src, dst
    rng = np.random.default_rng(sw + dh * 3 + c)
    f = rng.integers(0, 256, size=(sh, sw, c), dtype=np.uint8)
    assert np.array_equal(gpu_resize(gpu, f, dw, dh, orc.INTER_CUBIC), orc.cv_resize(f, dw, dh, orc.INTER_CUBIC))
