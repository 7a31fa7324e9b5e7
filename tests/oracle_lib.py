"""ctypes binding of oracle/liboracle.so -- the CPU checker (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(ORACLE_DIR, "liboracle.so")

INTER_NN, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 0, 1, 2, 3, 4
OK = 0
INVALID_ARGS = 50
NO_SUCH_FILTER = 52
TOO_BIG_TARGET = 54


class OrcRequest(C.Structure):
    _fields_ = [("buffer", C.c_void_p), ("crop", C.c_char_p), ("gravity", C.c_char_p), ("resize", C.c_char_p),
                ("quality", C.c_char_p), ("format", C.c_char_p), ("page", C.c_int), ("filters", C.c_char_p * 64),
                ("filter_count", C.c_int), ("mime", C.c_int), ("simple", C.c_int), ("need_flatten", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    P = C.c_void_p
    lib.orc_image_create.restype = P
    lib.orc_image_create.argtypes = [C.c_int] * 3
    lib.orc_image_from.restype = P
    lib.orc_image_from.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.orc_image_free.argtypes = [P]
    lib.orc_image_data.restype = C.c_void_p
    lib.orc_image_data.argtypes = [P]
    for f in ("width", "height", "channels", "step"):
        fn = getattr(lib, "orc_image_" + f)
        fn.restype = C.c_int
        fn.argtypes = [P]
    lib.orc_crop_geometry.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p] + [C.POINTER(C.c_int)] * 4
    lib.orc_crop.argtypes = [C.POINTER(P), C.c_char_p, C.c_char_p]
    lib.orc_resize_geometry.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_uint, C.c_uint, C.c_int] + [C.POINTER(C.c_int)] * 3
    lib.orc_resize.argtypes = [C.POINTER(P), C.c_char_p, C.c_uint, C.c_uint, C.c_int]
    lib.orc_cv_resize.argtypes = [P, P, C.c_int]
    lib.orc_set_cv_simd.argtypes = [C.c_int]
    lib.orc_cv_smooth_gaussian.argtypes = [P, C.c_double]
    lib.orc_gaussian_ksize.argtypes = [C.c_double]
    lib.orc_filter.argtypes = [C.POINTER(P), C.c_char_p, C.c_int]
    lib.orc_rgb2hsv.argtypes = [P]
    lib.orc_hsv2rgb.argtypes = [P]
    lib.orc_watermark.argtypes = [P, P, C.c_char, C.c_char, C.c_int, C.c_int, C.c_int]
    lib.orc_blend_with_paper.argtypes = [P]
    lib.orc_calc_perceived_brightness.restype = C.c_float
    lib.orc_calc_perceived_brightness.argtypes = [P]
    lib.orc_ascii.restype = C.c_long
    lib.orc_ascii.argtypes = [P, C.c_char_p, C.c_void_p]
    lib.orc_gray2bgr.argtypes = [C.POINTER(P)]
    lib.orc_ipl_to_fi.argtypes = [P, C.c_int, C.c_void_p, C.c_int]
    lib.orc_fi32_to_ipl.restype = P
    lib.orc_fi32_to_ipl.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.orc_parse_request.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.POINTER(OrcRequest))]
    lib.orc_request_free.argtypes = [C.POINTER(OrcRequest)]
    return lib


lib = _load()


def _b(s):
    return None if s is None else (s if isinstance(s, bytes) else s.encode())


class Img:
    """Owning handle on an orc_image; numpy in, numpy out (H x W x C uint8, tightly packed)."""

    def __init__(self, arr=None, handle=None):
        if handle is not None:
            self.h = C.c_void_p(handle)
            return
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        hh, ww, cc = arr.shape
        self.h = C.c_void_p(lib.orc_image_from(arr.ctypes.data, ww, hh, cc, ww * cc))
        if not self.h:
            raise ValueError("bad image shape %r" % (arr.shape,))

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_image_free(self.h)
            self.h = None

    @property
    def shape(self):
        return (lib.orc_image_height(self.h), lib.orc_image_width(self.h), lib.orc_image_channels(self.h))

    def numpy(self):
        hh, ww, cc = self.shape
        step = lib.orc_image_step(self.h)
        buf = (C.c_ubyte * (step * hh)).from_address(lib.orc_image_data(self.h))
        a = np.frombuffer(buf, dtype=np.uint8).reshape(hh, step)[:, : ww * cc].reshape(hh, ww, cc)
        return a.copy()


def crop_geometry(col, row, args, gravity=None):
    x, y, w, h = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib.orc_crop_geometry(col, row, _b(args), _b(gravity), x, y, w, h)
    return rc, (x.value, y.value, w.value, h.value)


def resize_geometry(col, row, args, max_w=2000, max_h=2000, simple=0):
    w, h, i = C.c_int(), C.c_int(), C.c_int()
    rc = lib.orc_resize_geometry(col, row, _b(args), max_w, max_h, simple, w, h, i)
    return rc, (w.value, h.value, i.value)


def crop(arr, args, gravity=None):
    im = Img(arr)
    rc = lib.orc_crop(C.byref(im.h), _b(args), _b(gravity))
    return rc, (im.numpy() if rc == 0 else None)


def resize(arr, args, max_w=2000, max_h=2000, simple=0):
    im = Img(arr)
    rc = lib.orc_resize(C.byref(im.h), _b(args), max_w, max_h, simple)
    return rc, (im.numpy() if rc == 0 else None)


def cv_resize(arr, dw, dh, interp):
    src = Img(arr)
    dst = Img(handle=lib.orc_image_create(dw, dh, src.shape[2]))
    rc = lib.orc_cv_resize(src.h, dst.h, interp)
    if rc:
        raise ValueError("orc_cv_resize rc=%d" % rc)
    return dst.numpy()


def filter(arr, request, allow_experiments=1):
    im = Img(arr)
    rc = lib.orc_filter(C.byref(im.h), _b(request), allow_experiments)
    return rc, (im.numpy() if rc == 0 else None)


def gaussian(arr, sigma):
    im = Img(arr)
    lib.orc_cv_smooth_gaussian(im.h, sigma)
    return im.numpy()


def watermark(arr, overlay, gx, gy, ox, oy, opacity):
    im, ov = Img(arr), Img(overlay)
    rc = lib.orc_watermark(im.h, ov.h, _b(gx), _b(gy), ox, oy, opacity)
    return rc, (im.numpy() if rc == 0 else None)


def blend_with_paper(arr):
    im = Img(arr)
    lib.orc_blend_with_paper(im.h)
    return im.numpy()


def brightness(arr):
    im = Img(arr)
    return float(lib.orc_calc_perceived_brightness(im.h))


def ascii_art(arr, args=""):
    im = Img(arr)
    hh, ww, _ = im.shape
    out = (C.c_ubyte * ((ww + 1) * hh))()
    n = lib.orc_ascii(im.h, _b(args), out)
    return bytes(out[:n])


def gray2bgr(arr):
    im = Img(arr)
    lib.orc_gray2bgr(C.byref(im.h))
    return im.numpy()


def rgb2hsv(arr):
    im = Img(arr)
    lib.orc_rgb2hsv(im.h)
    return im.numpy()


def hsv2rgb(arr):
    im = Img(arr)
    lib.orc_hsv2rgb(im.h)
    return im.numpy()


def parse_request(uri, exten="", max_filters=5):
    """-> (code, dict) per bridge.c:304-372 + :413-466."""
    r = C.POINTER(OrcRequest)()
    rc = lib.orc_parse_request(_b(uri), _b(exten), max_filters, C.byref(r))
    q = r.contents
    d = lambda v: None if v is None else v.decode()
    out = dict(crop=d(q.crop), gravity=d(q.gravity), resize=d(q.resize), quality=d(q.quality), format=d(q.format),
               page=q.page, filters=[q.filters[i].decode() for i in range(q.filter_count)], mime=q.mime,
               simple=q.simple, need_flatten=q.need_flatten)
    lib.orc_request_free(r)
    return rc, out


def ipl_to_fi(arr, bpp):
    """IplToFI32 / IplToFI24 -> (h, pitch) uint8 array in FreeImage layout."""
    im = Img(arr)
    hh, ww, _ = im.shape
    pitch = (ww * (bpp // 8) + 3) & ~3
    out = np.zeros((hh, pitch), np.uint8)
    assert lib.orc_ipl_to_fi(im.h, bpp, out.ctypes.data, pitch) == 0
    return out


class GifPage(C.Structure):
    _fields_ = [("indices", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("pitch", C.c_int),
                ("left", C.c_int), ("top", C.c_int), ("dispose", C.c_int), ("transparency_key", C.c_int),
                ("palette", C.c_void_p)]


def gif_compose(pages, destructive=False, page=-1):
    """orc_gif_compose over the same page dicts as ngx_http_imgproc_amd.gif_compose -> (code, [ndarray ...])."""
    keep = []
    arr = (GifPage * max(1, len(pages)))()
    for i, p in enumerate(pages):
        idx = np.ascontiguousarray(p["indices"], dtype=np.uint8)
        pal = np.ascontiguousarray(p["palette"], dtype=np.uint8)
        keep += [idx, pal]
        arr[i].indices = idx.ctypes.data
        arr[i].width = int(p["width"]); arr[i].height = idx.shape[0]; arr[i].pitch = idx.shape[1]
        arr[i].left = int(p.get("left", 0)); arr[i].top = int(p.get("top", 0))
        arr[i].dispose = int(p.get("dispose", 0)); arr[i].transparency_key = int(p.get("key", -1))
        arr[i].palette = pal.ctypes.data
    nout = 1 if page >= 0 else len(pages)
    outs = (C.c_void_p * max(1, nout))()
    lib.orc_gif_compose.restype = C.c_int
    lib.orc_gif_compose.argtypes = [C.POINTER(GifPage), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    rc = lib.orc_gif_compose(arr, len(pages), int(bool(destructive)), int(page), outs)
    if rc:
        return rc, []
    return 0, [Img(handle=outs[i]).numpy() for i in range(nout)]


def fi32_to_ipl(bits, w, h):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    return Img(handle=lib.orc_fi32_to_ipl(bits.ctypes.data, w, h, w * 4)).numpy()


# ---- JPEG decode (oracle/orc_jpeg.c; bridge.c:545-552's cvDecodeImage, pinned against Pillow's libjpeg-turbo)
UNSUPPORTED = 1
DECODE_FAILED = 3
lib.orc_jpeg_decode.argtypes = [C.c_char_p, C.c_long, C.POINTER(C.c_void_p)]
lib.orc_jpeg_info.argtypes = [C.c_char_p, C.c_long, C.POINTER(C.c_int)]
lib.orc_jpeg_coefficients.argtypes = [C.c_char_p, C.c_long, C.c_int, C.c_void_p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int)]


def jpeg_decode(blob):
    """-> (rc, H x W x {1,3} uint8 in B,G,R order or None)"""
    h = C.c_void_p()
    rc = lib.orc_jpeg_decode(blob, len(blob), C.byref(h))
    if rc:
        return rc, None
    return rc, Img(handle=h.value).numpy()


# ---- PNG decode (oracle/orc_png.c; cvDecodeImage for a PNG blob, pinned against Pillow's libpng)
lib.orc_png_decode.argtypes = [C.c_char_p, C.c_long, C.POINTER(C.c_void_p)]


def png_decode(blob):
    """-> (rc, H x W x {1,3,4} uint8 in gray / B,G,R / B,G,R,A order or None)"""
    h = C.c_void_p()
    rc = lib.orc_png_decode(blob, len(blob), C.byref(h))
    if rc:
        return rc, None
    return rc, Img(handle=h.value).numpy()


def jpeg_info(blob):
    info = (C.c_int * 8)()
    rc = lib.orc_jpeg_info(blob, len(blob), info)
    keys = ("width", "height", "components", "hs", "vs", "restart_interval", "mcux", "mcuy")
    return rc, (dict(zip(keys, list(info))) if rc == 0 else None)


def jpeg_coefficients(blob, ci):
    rc, info = jpeg_info(blob)
    if rc:
        return rc, None
    h, v = (info["hs"], info["vs"]) if ci == 0 else (1, 1)
    n = info["mcux"] * h * info["mcuy"] * v * 64
    out = np.zeros(n, dtype=np.int16)
    bw, bh = C.c_int(), C.c_int()
    rc = lib.orc_jpeg_coefficients(blob, len(blob), ci, out.ctypes.data, n, bw, bh)
    return rc, (out.reshape(bh.value, bw.value, 8, 8) if rc == 0 else None)


def jpeg_encode(arr, quality):
    """cvEncodeImage(".jpg", frame, {CV_IMWRITE_JPEG_QUALITY, quality}) -> (rc, file bytes or None); arr is H x W x {1,3,4} B,G,R(,A)."""
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    hh, ww, cc = a.shape
    lib.orc_jpeg_encode_bound.restype = C.c_long
    cap = lib.orc_jpeg_encode_bound(ww, hh, cc)
    out = (C.c_ubyte * cap)()
    n = C.c_long()
    lib.orc_jpeg_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long, C.POINTER(C.c_long)]
    rc = lib.orc_jpeg_encode(a.ctypes.data, ww, hh, cc, ww * cc, int(quality), out, cap, C.byref(n))
    return rc, (bytes(out[: n.value]) if rc == 0 else None)
