"""Host-side grammar of the product (imp_args.cpp, no GPU needed) vs the oracle's restatement of bridge.c / filters.c."""
import itertools

import pytest
from hypothesis import given, settings, strategies as st

import oracle_lib as orc
import ngx_http_imgproc_amd as imp

SIZES = [(640, 480), (1920, 1080), (37, 29), (1, 1), (500, 2), (3, 1000)]
CROP_ARGS = ["1,1", "16,9", "9,16", "4,3,l,t", "4,3,r,b", "4,3,c,c", "1,1,c", "320px,240px", "320px,240px,0px,0px",
             "320px,240px,r,b", "10px,10px,5px,7px", "0,0,320,240", "0,5", "5,0", "", ",", "abc", "10px,10", "10,10px",
             "1,1,x,y", "1,1,10,10", "1,1,10px,10px", "99999px,1px", "1px,99999px", "2,1,-5px,0px", "1,1,l", "3,2,,b",
             "1e3,1", "-4,3", "1,1,c,t,extra"]
GRAVITIES = [None, "l,t", "r,b", "c,c", "5px,5px", "r", "xx", "l,", ",t", "10px,b", "abc,def"]


@pytest.mark.parametrize("size", SIZES)
def test_crop_geometry_matches_oracle(size):
    w, h = size
    for args, g in itertools.product(CROP_ARGS, GRAVITIES):
        rc_o, geom_o = orc.crop_geometry(w, h, args, g)
        rc, geom = imp.crop_geometry(w, h, args, g)
        assert rc == rc_o, (args, g, rc, rc_o)
        if rc == 0:
            assert geom == geom_o, (args, g)


RESIZE_ARGS = ["224,224", "224", "0,224", "224,0", "0,0", "", "abc", "5000,10", "10,5000", "300,300,up", "300,up", "1,1",
               "1", "0,1", "100,100,down", "100,100,up,up", "-5,10", "2000,2001", "2001,2000", "7,3,up", "99999999999,1"]


@pytest.mark.parametrize("size", SIZES)
def test_resize_geometry_matches_oracle(size):
    w, h = size
    for args, simple, (mw, mh) in itertools.product(RESIZE_ARGS, (0, 1), ((2000, 2000), (0, 0), (100, 3000))):
        rc_o, g_o = orc.resize_geometry(w, h, args, mw, mh, simple)
        cfg = imp.Config(max_w=mw, max_h=mh)
        rc, g = imp.resize_geometry(w, h, args, cfg, simple)
        assert rc == rc_o, (args, simple, mw, mh, rc, rc_o)
        if rc == 0:
            assert g == g_o, (args, simple)


@settings(max_examples=300, deadline=None)
@given(w=st.integers(1, 5000), h=st.integers(1, 5000), a=st.integers(0, 40), b=st.integers(0, 40),
       gx=st.sampled_from(["l", "c", "r", "3px", "0px"]), gy=st.sampled_from(["t", "c", "b", "2px", "0px"]))
def test_crop_ratio_property(w, h, a, b, gx, gy):
    args = "%d,%d,%s,%s" % (a, b, gx, gy)
    rc_o, geom_o = orc.crop_geometry(w, h, args)
    rc, geom = imp.crop_geometry(w, h, args)
    assert (rc, geom if rc == 0 else None) == (rc_o, geom_o if rc_o == 0 else None)
    if rc == 0:
        x, y, cw, ch = geom
        assert 0 <= x and 0 <= y and x + cw <= w and y + ch <= h and cw >= 1 and ch >= 1


@settings(max_examples=300, deadline=None)
@given(w=st.integers(1, 4000), h=st.integers(1, 4000), tw=st.integers(0, 3000), th=st.integers(0, 3000), up=st.booleans())
def test_resize_property(w, h, tw, th, up):
    args = "%d,%d%s" % (tw, th, ",up" if up else "")
    assert imp.resize_geometry(w, h, args) == orc.resize_geometry(w, h, args)


FILTER_REQS = ["flip=10", "flip=00", "flip=2", "flip", "rotate=90", "rotate=91", "rotate=180x", "modulate=1,2,3", "modulate=1,2",
               "modulate=181,1,1", "modulate=-1,1,1", "modulate=0,0,0", "colorize=aabbcc", "colorize=aabbc", "colorize=aabbcc,1.5",
               "colorize=aabbcc,-0.1", "colorize=zzzzzz", "blur=1", "blur=-1", "blur=", "blur=abc", "gamma=2", "gamma=abc",
               "contrast=1", "contrast=0", "contrast=-1", "gradmap=aabbcc,ddeeff", "gradmap=aabbcc", "gradmap=aabbcc,ddeef",
               "gradmap=" + ",".join(["010203"] * 9), "vignette=0.5", "vignette=0.5,0.7", "gotham=x", "lomo=x", "kelvin=x",
               "rainbow=full", "rainbow=mid", "rainbow=pale", "rainbow=", "rainbow=x", "scanline=0.5", "scanline=1.5",
               "scanline=0.5,2", "scanline=0.5,0.5,0", "scanline=0.5,0.5,1,0", "cartoon=1", "=", "==", "flip=10=11", "nosuch=1"]


@pytest.mark.parametrize("allow", [0, 1])
def test_filter_argument_codes_match_oracle(allow):
    import numpy as np

    img = np.zeros((8, 8, 4), np.uint8)
    for req in FILTER_REQS:
        rc_o, _ = orc.filter(img, req, allow)
        assert imp.filter_check(req, allow) == rc_o, (req, allow)


def test_check_destructive_table():
    # filters.c:10-28: blur and vignette are destructive (prefix match, filters.c:35)
    assert imp.check_destructive("blur=2") == 1 and imp.check_destructive("vignette=1") == 1
    assert imp.check_destructive("gamma=2") == 0 and imp.check_destructive("blurry") == 1
    assert imp.check_destructive("nosuch") == 0


def test_batch_entry_points_refuse_geometry_the_kernels_cannot_index():
    """Round-3 hardening: the batch ABI takes caller geometry, and the kernels index pixels, row bytes and bytes inside
    a frame in 32 bits.  Every refusal is decided in 64-bit arithmetic before a device is even looked for."""
    import ctypes as C

    import ngx_http_imgproc_amd as imp

    lib = imp.lib
    INV = imp.IMP_ERROR_INVALID_ARGS
    buf = (C.c_ubyte * 64)()
    p = C.cast(buf, C.c_void_p)

    def cv(sw, sh, sstep, dw, dh, dstep, c=4, count=1, interp=imp.INTER_AREA, sstride=0, dstride=0):
        return lib.impgpu_batch_cv_resize(p, sstride, sw, sh, sstep, p, dstride, dw, dh, dstep, c, count, interp, None)

    assert cv(0, 10, 40, 4, 4, 16) == INV                                  # empty source
    assert cv(10, 10, 39, 4, 4, 16) == INV                                 # pitch shorter than a row
    assert cv(1 << 29, 4, 1 << 31, 4, 4, 16) == INV                        # w * c wraps an int
    assert cv(40000, 40000, 160000, 4, 4, 16) == INV                       # 1.6e9 pixels, 6.4 GB of rows
    assert cv(20000, 60000, 80000, 4, 4, 16) == INV                        # step * h > 4 GiB
    assert cv(100, 100, 400, 33000, 33000, 132000) == INV                  # destination too large
    assert cv(10, 10, 40, 4, 4, 16, count=70000) == INV
    assert cv(10, 10, 40, 4, 4, 16, count=2, sstride=-400) == INV
    assert cv(10, 10, 40, 4, 4, 16, interp=9) == INV
    assert cv(10, 10, 40, 4, 4, 16, c=2) == INV
    assert cv(10, 10, 40, 4, 4, 16) == imp.IMP_ERROR_DEVICE                # well-formed: only now is the device missed (CPU run)

    item = imp.ResizeItem(p, 1 << 29, 4, 1 << 31, p, 4, 4, 16)
    assert lib.impgpu_batch_resize_mixed(C.byref(item), 1, 4, 0, None) == INV
    item = imp.ResizeItem(p, 10, 10, 40, p, 40000, 40000, 160000)
    assert lib.impgpu_batch_resize_mixed(C.byref(item), 1, 4, 0, None) == INV
    item = imp.ResizeItem(p, 10, 10, 40, p, 4, 4, 16)
    assert lib.impgpu_batch_resize_mixed(C.byref(item), 1, 5, 0, None) == INV
    assert lib.impgpu_batch_resize_mixed(C.byref(item), 1, 4, 0, None) == imp.IMP_ERROR_DEVICE

    cfg = imp.Config()
    rrw = lib.impgpu_batch_resize_rotate_watermark
    assert rrw(p, 0, 1 << 29, 4, 1 << 31, p, 0, 16, 4, 4, 90, C.byref(cfg.c), 4, 1, None) == INV
    assert rrw(p, 0, 10, 10, 40, p, 0, 8, 4, 5, 90, C.byref(cfg.c), 4, 1, None) == INV           # turned frame is 5 wide: 8-byte rows do not hold it
    assert rrw(p, 0, 10, 10, 40, p, 0, 20, 4, 5, 90, C.byref(cfg.c), 4, 1, None) == imp.IMP_ERROR_DEVICE
    filt = (C.c_char_p * 1)(b"gamma=2")
    assert lib.impgpu_batch_filters(p, 0, 40000, 40000, 4, 160000, 1, filt, 1, 1, None) == INV
    assert lib.impgpu_batch_filters(p, 0, 10, 10, 4, 39, 1, filt, 1, 1, None) == INV
    assert lib.impgpu_batch_filters(p, 0, 10, 10, 4, 40, 1, filt, 1, 1, None) == imp.IMP_ERROR_DEVICE


def test_album_entry_points_check_their_arguments_before_the_device():
    import ctypes as C

    import ngx_http_imgproc_amd as imp

    lib = imp.lib
    INV = imp.IMP_ERROR_INVALID_ARGS
    buf = (C.c_ubyte * 4096)()
    ptrs = (C.c_void_p * 3)(*[C.addressof(buf)] * 3)
    h = C.c_void_p()
    assert lib.impgpu_album_upload(ptrs, 0, 8, 8, 4, None, C.byref(h)) == INV              # an album has frames
    assert lib.impgpu_album_upload(None, 3, 8, 8, 4, None, C.byref(h)) == INV
    assert lib.impgpu_album_upload(ptrs, 3, 8, 8, 4, None, None) == INV
    holes = (C.c_void_p * 3)(C.addressof(buf), None, C.addressof(buf))
    assert lib.impgpu_album_upload(holes, 3, 8, 8, 4, None, C.byref(h)) == INV             # a missing frame
    steps = (C.c_int * 3)(32, 31, 32)
    assert lib.impgpu_album_upload(ptrs, 3, 8, 8, 4, steps, C.byref(h)) == INV             # a pitch shorter than a row
    assert lib.impgpu_album_upload(ptrs, 3, 8, 8, 4, None, C.byref(h)) == imp.IMP_ERROR_DEVICE   # well-formed: no device here
    assert not h.value
    assert lib.impgpu_album_download(None, ptrs, None) == INV
    assert lib.impgpu_album_count(None) == 0


def test_jpeg_codec_entry_points_check_their_arguments_before_the_device():
    import ctypes as C

    import ngx_http_imgproc_amd as imp

    lib = imp.lib
    INV = imp.IMP_ERROR_INVALID_ARGS
    n = C.c_size_t()
    buf = (C.c_ubyte * 64)()
    assert lib.impgpu_image_encode_jpeg(None, 90, buf, 64, C.byref(n)) == INV
    assert lib.impgpu_batch_encode_jpeg(None, 3, 90, None, None, None, None) == INV
    assert lib.impgpu_batch_encode_jpeg(None, -1, 90, None, None, None, None) == INV
    assert lib.impgpu_batch_encode_jpeg(None, 0, 90, None, None, None, None) == imp.IMP_ERROR_DEVICE    # nothing to do, but no env either
    # the bound is a host computation: what a block can take in the file, every byte stuffed, plus the headers
    assert lib.impgpu_jpeg_encode_bound(224, 126, 3) == 1024 + 14 * 8 * 6 * 432
    assert lib.impgpu_jpeg_encode_bound(17, 9, 1) == 1024 + 3 * 2 * 432
    assert lib.impgpu_jpeg_encode_bound(0, 9, 3) == 0 and lib.impgpu_jpeg_encode_bound(9, 9, 2) == 0 and lib.impgpu_jpeg_encode_bound(70000, 9, 3) == 0
    h = C.c_void_p()
    assert lib.impgpu_image_decode_jpeg(None, 10, C.byref(h)) == INV
    codes = (C.c_int * 2)()
    imgs = (C.c_void_p * 2)()
    assert lib.impgpu_batch_decode_jpeg(None, None, 2, imgs, codes) == INV
