"""Python mirror of the reference's operator interface (bridge.h / filters.h) over libimpgpu.so.

`Image` stands for the IplImage* the reference hands from operator to operator; its methods
carry the reference's names and take the reference's argument strings.  Methods return the
IMP_* code (0 = IMP_OK) exactly like the C functions, so tests read like calls into bridge.c.
numpy arrays are H x W x C uint8 in B,G,R[,A] order.
"""
import ctypes as C

import numpy as np

from ._lib import lib, CConfig, CJob, ImpError

IMP_OK = 0
IMP_ERROR_UNSUPPORTED = 1
IMP_ERROR_MALLOC_FAILED = 2
IMP_ERROR_DECODE_FAILED = 3
IMP_ERROR_INVALID_ARGS = 50
IMP_ERROR_UPSCALE = 51
IMP_ERROR_NO_SUCH_FILTER = 52
IMP_ERROR_NO_SUCH_WATERMARK = 53
IMP_ERROR_TOO_BIG_TARGET = 54
IMP_ERROR_TOO_MUCH_FILTERS = 55
IMP_ERROR_DEVICE = 90
INTER_NN, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 0, 1, 2, 3, 4


def _b(s):
    return None if s is None else (s if isinstance(s, bytes) else str(s).encode())


def env_start(device=-1):
    """OnEnvStart (bridge.c:10). Raises when no GPU is usable."""
    rc = lib.impgpu_env_start(device)
    if rc:
        raise ImpError(rc, "impgpu_env_start")
    return lib.impgpu_env_device()


def env_destroy():
    lib.impgpu_env_destroy()


def sync():
    rc = lib.impgpu_sync()
    if rc:
        raise ImpError(rc, "impgpu_sync")


class Config:
    """The Config fields the operators read (required.h:108-118); defaults of module.c:168-187."""

    def __init__(self, max_w=2000, max_h=2000, max_filters=5, allow_experiments=False):
        self.c = CConfig()
        self.c.max_target_w = max_w
        self.c.max_target_h = max_h
        self.c.max_filters_count = max_filters
        self.c.allow_experiments = int(bool(allow_experiments))
        self.c.watermark_opacity = 100
        self.c.watermark_gravity_x = b"l"
        self.c.watermark_gravity_y = b"t"
        self.c.watermark_offset_x = 0
        self.c.watermark_offset_y = 0
        self.c.watermark = None

    def prepare_watermark(self, overlay, gravity_x="r", gravity_y="b", offset_x=0, offset_y=0, opacity=100):
        """PrepareWatermark (bridge.c:199-237) minus file read / decode: upload the decoded overlay once."""
        ov = np.ascontiguousarray(overlay, dtype=np.uint8)
        if ov.ndim == 2:
            ov = ov[:, :, None]
        h, w, c = ov.shape
        rc = lib.impgpu_prepare_watermark(C.byref(self.c), ov.ctypes.data, w, h, c, w * c)
        if rc:
            return rc
        self.c.watermark_gravity_x = _b(gravity_x)
        self.c.watermark_gravity_y = _b(gravity_y)
        self.c.watermark_offset_x = offset_x
        self.c.watermark_offset_y = offset_y
        self.c.watermark_opacity = opacity
        return IMP_OK

    def release(self):
        if self.c.watermark:
            h = C.c_void_p(self.c.watermark)
            lib.impgpu_image_release(C.byref(h))
            self.c.watermark = None


class Image:
    """Device-resident frame (impgpu_image*)."""

    def __init__(self, arr=None, handle=None):
        self.h = C.c_void_p()
        if handle is not None:
            self.h = C.c_void_p(handle)
            return
        a = np.ascontiguousarray(arr, dtype=np.uint8)
        if a.ndim == 2:
            a = a[:, :, None]
        hh, ww, cc = a.shape
        rc = lib.impgpu_image_upload(a.ctypes.data, ww, hh, cc, ww * cc, C.byref(self.h))
        if rc:
            raise ImpError(rc, "impgpu_image_upload")

    @classmethod
    def wrap(cls, device_ptr, width, height, channels, step):
        h = C.c_void_p()
        rc = lib.impgpu_image_wrap(C.c_void_p(device_ptr), width, height, channels, step, C.byref(h))
        if rc:
            raise ImpError(rc, "impgpu_image_wrap")
        return cls(handle=h.value)

    @classmethod
    def decode_jpeg(cls, blob):
        """cvDecodeImage(&rawencoded, -1) for a JPEG blob (bridge.c:545-552), on the device -> (code, Image or None)."""
        h = C.c_void_p()
        rc = lib.impgpu_image_decode_jpeg(bytes(blob), len(blob), C.byref(h))
        return rc, (cls(handle=h.value) if rc == 0 else None)

    @classmethod
    def decode_png(cls, blob):
        """cvDecodeImage(&rawencoded, -1) for a PNG blob (bridge.c:545-552): host inflate, filters undone on the device
        -> (code, Image or None)."""
        h = C.c_void_p()
        rc = lib.impgpu_image_decode_png(bytes(blob), len(blob), C.byref(h))
        return rc, (cls(handle=h.value) if rc == 0 else None)

    @classmethod
    def album(cls, frames):
        """The frames of one animation (same geometry) as ONE handle: every operator then runs once for all of them
        (impgpu_album_upload; Album, required.h:56-66)."""
        arrs = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
        arrs = [a[:, :, None] if a.ndim == 2 else a for a in arrs]
        hh, ww, cc = arrs[0].shape
        if any(a.shape != (hh, ww, cc) for a in arrs):
            raise ValueError("album frames differ in geometry")
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        h = C.c_void_p()
        rc = lib.impgpu_album_upload(ptrs, len(arrs), ww, hh, cc, None, C.byref(h))
        if rc:
            raise ImpError(rc, "impgpu_album_upload")
        return cls(handle=h.value)

    @property
    def count(self):
        return lib.impgpu_album_count(self.h)

    def frames(self):
        """Every frame of the album -> list of HxWxC arrays (impgpu_album_download: one copy, one wait)."""
        hh, ww, cc = self.shape
        outs = [np.empty((hh, ww, cc), dtype=np.uint8) for _ in range(self.count)]
        ptrs = (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
        rc = lib.impgpu_album_download(self.h, ptrs, None)
        if rc:
            raise ImpError(rc, "impgpu_album_download")
        return outs

    def encode_jpeg(self, quality=95):
        """cvEncodeImage(".jpg", frame, {CV_IMWRITE_JPEG_QUALITY, quality}) (bridge.c:704) on the device -> (code, file bytes or None)."""
        hh, ww, cc = self.shape
        cap = lib.impgpu_jpeg_encode_bound(ww, hh, cc)
        buf = np.empty(max(1, cap), dtype=np.uint8)
        n = C.c_size_t()
        rc = lib.impgpu_image_encode_jpeg(self.h, int(quality), buf.ctypes.data, cap, C.byref(n))
        return rc, (buf[: n.value].tobytes() if rc == 0 else None)

    def release(self):
        if self.h:
            lib.impgpu_image_release(C.byref(self.h))

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    @property
    def shape(self):
        return (lib.impgpu_image_height(self.h), lib.impgpu_image_width(self.h), lib.impgpu_image_channels(self.h))

    @property
    def device_ptr(self):
        return lib.impgpu_image_device_ptr(self.h)

    @property
    def step(self):
        """Device row pitch in bytes (cvCreateImage's 4-byte alignment rule)."""
        return lib.impgpu_image_step(self.h)

    def numpy(self):
        hh, ww, cc = self.shape
        out = np.empty((hh, ww, cc), dtype=np.uint8)
        rc = lib.impgpu_image_download(self.h, out.ctypes.data, ww * cc)
        if rc:
            raise ImpError(rc, "impgpu_image_download")
        return out

    def clone(self):
        h = C.c_void_p()
        rc = lib.impgpu_image_clone(self.h, C.byref(h))
        if rc:
            raise ImpError(rc, "impgpu_image_clone")
        return Image(handle=h.value)

    # ---- the reference's operators ----
    def crop(self, args, gravity=None):
        """Crop(IplImage**, args, gravity), bridge.c:18."""
        return lib.impgpu_crop(C.byref(self.h), _b(args), _b(gravity))

    def resize(self, args, config=None, simple=0):
        """Resize(IplImage**, args, config, simple), bridge.c:143."""
        cfg = config or Config()
        return lib.impgpu_resize(C.byref(self.h), _b(args), C.byref(cfg.c), simple)

    def cv_resize(self, width, height, interpolation):
        """cvResize(image, resized, filter), bridge.c:191."""
        return lib.impgpu_cv_resize(C.byref(self.h), width, height, interpolation)

    def filter(self, request, allow_experiments=1):
        """Filter(IplImage**, "name=args", allowExperiments), filters.c:43."""
        return lib.impgpu_filter(C.byref(self.h), _b(request), int(allow_experiments))

    def watermark(self, config):
        """Watermark(IplImage*, Config*), bridge.c:239."""
        return lib.impgpu_watermark(self.h, C.byref(config.c))

    def blend_with_paper(self):
        """BlendWithPaper(IplImage*), filters.c:666."""
        return lib.impgpu_blend_with_paper(self.h)

    def calc_perceived_brightness(self):
        """CalcPerceivedBrightness(IplImage*), filters.c:707."""
        out = C.c_float()
        rc = lib.impgpu_calc_perceived_brightness(self.h, C.byref(out))
        if rc:
            raise ImpError(rc, "impgpu_calc_perceived_brightness")
        return out.value

    def ascii(self, args=""):
        """ASCII(IplImage*, args, pool), filters.c:488."""
        hh, ww, _ = self.shape
        cap = (ww + 1) * hh
        buf = (C.c_ubyte * cap)()
        n = C.c_long()
        rc = lib.impgpu_ascii(self.h, _b(args), buf, cap, C.byref(n))
        if rc:
            raise ImpError(rc, "impgpu_ascii")
        return bytes(buf[: n.value])

    def gray2bgr(self):
        return lib.impgpu_gray2bgr(C.byref(self.h))

    def rgb2hsv(self):
        return lib.impgpu_rgb2hsv(self.h)

    def hsv2rgb(self):
        return lib.impgpu_hsv2rgb(self.h)


def run_ops(image, config, crop=None, gravity=None, resize=None, simple=0, filters=(), need_flatten=0):
    """RunJob's operator segment, bridge.c:574-656. Returns (code, step)."""
    job = CJob()
    job.crop = _b(crop)
    job.gravity = _b(gravity)
    job.resize = _b(resize)
    job.simple = simple
    arr = (C.c_char_p * max(1, len(filters)))(*[_b(f) for f in filters])
    job.filters = arr
    job.filter_count = len(filters)
    job.need_flatten = need_flatten
    step = C.c_int()
    rc = lib.impgpu_run_ops(C.byref(image.h), C.byref(job), C.byref(config.c), C.byref(step))
    return rc, step.value


def batch_decode_jpeg(blobs):
    """impgpu_batch_decode_jpeg -> [(code, Image or None)] in the order of `blobs`."""
    n = len(blobs)
    keep = [bytes(b) for b in blobs]
    arr = (C.c_char_p * n)(*keep)
    sizes = (C.c_size_t * n)(*[len(b) for b in keep])
    imgs = (C.c_void_p * n)()
    codes = (C.c_int * n)()
    rc = lib.impgpu_batch_decode_jpeg(arr, sizes, n, imgs, codes)
    if rc:
        raise ImpError(rc, "impgpu_batch_decode_jpeg")
    return [(codes[i], Image(handle=imgs[i]) if codes[i] == 0 else None) for i in range(n)]


def jpeg_unstuff(blob):
    """impgpu_jpeg_unstuff of libimpgpu_client.so (what a worker does on the way into its slot) -> (head, scan) or None when
    the file is to go as it is."""
    from .broker import clib

    blob = bytes(blob)
    out = np.empty(len(blob) + 2048, np.uint8)
    head, at, n, total = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    if not clib.impgpu_jpeg_unstuff(blob, len(blob), out.ctypes.data, out.size, C.byref(head), C.byref(at), C.byref(n), C.byref(total)):
        return None
    return out[:head.value].tobytes(), out[at.value:at.value + n.value].tobytes()


def batch_decode_jpeg_prepared(files, pinned=False):
    """impgpu_batch_decode_jpeg_prepared.  files: whole files (bytes) or (head, scan) pairs as jpeg_unstuff returns them;
    pinned: the scans are put into impgpu_host_alloc memory and handed over as `registered`."""
    from ._lib import CJpegPrepared

    n = len(files)
    arr = (CJpegPrepared * n)()
    keep, pins = [], []
    for i, f in enumerate(files):
        if isinstance(f, (bytes, bytearray)):
            b = np.frombuffer(bytes(f), np.uint8)
            keep.append(b)
            arr[i].head, arr[i].head_size, arr[i].scan, arr[i].scan_size, arr[i].registered = b.ctypes.data, b.size, None, 0, 0
            continue
        head, scan = f
        h = np.frombuffer(bytes(head), np.uint8)
        tail = 1024                                                # IMPGPU_JPEG_SCAN_TAIL
        if pinned:
            p = lib.impgpu_host_alloc(len(scan) + tail)
            if not p:
                raise ImpError(IMP_ERROR_DEVICE, "impgpu_host_alloc")
            pins.append(p)
            sc = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(len(scan) + tail,))
        else:
            sc = np.empty(len(scan) + tail, np.uint8)
        sc[:len(scan)] = np.frombuffer(bytes(scan), np.uint8)
        sc[len(scan):] = 0xFF
        keep += [h, sc]
        arr[i].head, arr[i].head_size, arr[i].scan, arr[i].scan_size, arr[i].registered = h.ctypes.data, h.size, sc.ctypes.data, len(scan), int(pinned)
    imgs = (C.c_void_p * n)()
    codes = (C.c_int * n)()
    try:
        rc = lib.impgpu_batch_decode_jpeg_prepared(arr, n, imgs, codes)
    finally:
        for p in pins:
            lib.impgpu_host_free(p)
    if rc:
        raise ImpError(rc, "impgpu_batch_decode_jpeg_prepared")
    return [(codes[i], Image(handle=imgs[i]) if codes[i] == 0 else None) for i in range(n)]


def jpeg_request_one_wait(blob, config, quality=86, **ops):
    """A request end to end with ONE wait (round 5): the file's decode is begun on the thread's own stream, its frame taken
    ahead of the verdict (impgpu_batch_decode_jpeg_pending), the operators (**ops as for run_ops) and the answer's encode are
    enqueued behind it, and only then the verdict and the file are waited for.  Returns (code, file bytes or None): a code
    other than 0 is the decode's -- the caller falls back exactly as after impgpu_image_decode_jpeg."""
    from ._lib import CJpegPrepared

    keep = np.frombuffer(bytes(blob), np.uint8)
    f = (CJpegPrepared * 1)()
    f[0].head, f[0].head_size, f[0].scan, f[0].scan_size, f[0].registered = keep.ctypes.data, keep.size, None, 0, 0
    batch = C.c_void_p()
    rc = lib.impgpu_batch_decode_jpeg_prepared_begin(f, 1, C.byref(batch))
    if rc:
        raise ImpError(rc, "impgpu_batch_decode_jpeg_prepared_begin")
    img = (C.c_void_p * 1)()
    code = (C.c_int * 1)()
    rc = lib.impgpu_batch_decode_jpeg_pending(batch, img)
    if rc or not img[0]:                                           # nothing to run ahead with: the two-wait form
        lib.impgpu_batch_decode_jpeg_finish(C.byref(batch), img, code)
        if code[0] or not img[0]:
            return code[0] or IMP_ERROR_DECODE_FAILED, None
        im = Image(handle=img[0])
        rc, _ = run_ops(im, config, **ops)
        out = im.encode_jpeg(quality) if rc == 0 else (rc, None)
        im.release()
        return out
    im = Image(handle=img[0])
    rc_ops, _ = run_ops(im, config, **ops)
    enc = C.c_void_p()
    rc_enc = IMP_ERROR_DEVICE
    if rc_ops == 0:
        hs = (C.c_void_p * 1)(im.h.value)
        rc_enc = lib.impgpu_batch_encode_jpeg_begin(hs, 1, int(quality), C.byref(enc))
    none = (C.c_void_p * 1)()
    rc = lib.impgpu_batch_decode_jpeg_finish(C.byref(batch), none, code)
    out = None
    ecode = (C.c_int * 1)()
    if rc_enc == 0:
        cap = lib.impgpu_jpeg_encode_bound(im.shape[1], im.shape[0], im.shape[2])
        buf = np.empty(cap, np.uint8)
        outs = (C.c_void_p * 1)(buf.ctypes.data)
        caps = (C.c_size_t * 1)(cap)
        lens = (C.c_size_t * 1)()
        rc2 = lib.impgpu_batch_encode_jpeg_finish(C.byref(enc), outs, caps, lens, ecode)
        if rc2 == 0 and ecode[0] == 0:
            out = buf[:lens[0]].tobytes()
    im.release()
    if rc or code[0]:
        return rc or code[0], None                                 # the frame did not hold the file's pixels: the answer is dropped
    if rc_ops:
        return rc_ops, None
    return (0, out) if out is not None else (rc_enc or ecode[0] or IMP_ERROR_DEVICE, None)


def batch_encode_jpeg(images, quality=95):
    """impgpu_batch_encode_jpeg -> [(code, file bytes or None)] in the order of `images`."""
    n = len(images)
    hs = (C.c_void_p * n)(*[im.h.value for im in images])
    caps = [lib.impgpu_jpeg_encode_bound(im.shape[1], im.shape[0], im.shape[2]) for im in images]
    bufs = [np.empty(max(1, c), dtype=np.uint8) for c in caps]
    outs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    ccaps = (C.c_size_t * n)(*caps)
    lens = (C.c_size_t * n)()
    codes = (C.c_int * n)()
    rc = lib.impgpu_batch_encode_jpeg(hs, n, int(quality), outs, ccaps, lens, codes)
    if rc:
        raise ImpError(rc, "impgpu_batch_encode_jpeg")
    return [(codes[i], bufs[i][: lens[i]].tobytes() if codes[i] == 0 else None) for i in range(n)]


def jpeg_info(blob):
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    rc = lib.impgpu_jpeg_info(bytes(blob), len(blob), w, h, c)
    return rc, (w.value, h.value, c.value)


def png_info(blob):
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    rc = lib.impgpu_png_info(bytes(blob), len(blob), w, h, c)
    return rc, (w.value, h.value, c.value)


def png_stage_times():
    """the calling thread's last decode_png: host microseconds (header, chunks + CRC + inflate, check + enqueue), scanline bytes"""
    t = (C.c_double * 4)()
    lib.impgpu_png_stage_times(t, 4)
    return list(t)


def crop_geometry(width, height, args, gravity=None):
    x, y, w, h = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib.impgpu_crop_geometry(width, height, _b(args), _b(gravity), x, y, w, h)
    return rc, (x.value, y.value, w.value, h.value)


def resize_geometry(width, height, args, config=None, simple=0):
    cfg = config or Config()
    w, h, i = C.c_int(), C.c_int(), C.c_int()
    rc = lib.impgpu_resize_geometry(width, height, _b(args), C.byref(cfg.c), simple, w, h, i)
    return rc, (w.value, h.value, i.value)


def filter_check(request, allow_experiments=1):
    return lib.impgpu_filter_check(_b(request), int(allow_experiments))


def check_destructive(request):
    return lib.impgpu_check_destructive(_b(request))


def batch_cv_resize(src_ptr, src_stride, sw, sh, sstep, dst_ptr, dst_stride, dw, dh, dstep, channels, count,
                    interpolation, stream=None):
    rc = lib.impgpu_batch_cv_resize(C.c_void_p(src_ptr), src_stride, sw, sh, sstep, C.c_void_p(dst_ptr), dst_stride,
                                    dw, dh, dstep, channels, count, interpolation, C.c_void_p(stream or 0))
    if rc:
        raise ImpError(rc, "impgpu_batch_cv_resize")


class ResizeItem(C.Structure):
    """impgpu_resize_item (include/impgpu.h)."""
    _fields_ = [("src", C.c_void_p), ("src_width", C.c_int), ("src_height", C.c_int), ("src_step", C.c_int),
                ("dst", C.c_void_p), ("dst_width", C.c_int), ("dst_height", C.c_int), ("dst_step", C.c_int)]


def batch_resize_mixed(items, channels, simple=False, stream=None):
    """items: [(src_ptr, sw, sh, sstep, dst_ptr, dw, dh, dstep)] of frames resident in HBM; returns the IMP_* code."""
    arr = (ResizeItem * len(items))(*[ResizeItem(*it) for it in items])
    return lib.impgpu_batch_resize_mixed(arr, len(items), channels, int(simple), C.c_void_p(stream or 0))


def batch_resize_rotate_watermark(src_ptr, src_stride, sw, sh, sstep, dst_ptr, dst_stride, dstep, rw, rh, rotate,
                                  config, channels, count, stream=None):
    rc = lib.impgpu_batch_resize_rotate_watermark(C.c_void_p(src_ptr), src_stride, sw, sh, sstep, C.c_void_p(dst_ptr),
                                                  dst_stride, dstep, rw, rh, rotate, C.byref(config.c), channels,
                                                  count, C.c_void_p(stream or 0))
    if rc:
        raise ImpError(rc, "impgpu_batch_resize_rotate_watermark")


def gif_compose(pages, destructive=False, page=-1, album=False):
    """LoadGIF's compositing loop (advancedio.c:204-247) on the device.  pages: dicts with `indices` (h x pitch uint8,
    FreeImage scanline order), `width`, `left`, `top`, `dispose`, `key`, `palette` (256 x 4 uint8, B,G,R,reserved).
    Returns (code, [Image ...]): every page, or just the requested one; album=True -> (code, one album Image)."""
    from ._lib import CGifPage
    keep = []
    arr = (CGifPage * max(1, len(pages)))()
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
    if album:
        rc = lib.impgpu_gif_compose_album(arr, len(pages), int(bool(destructive)), int(page), outs)
        return rc, (Image(handle=outs[0]) if rc == 0 else None)
    rc = lib.impgpu_gif_compose(arr, len(pages), int(bool(destructive)), int(page), outs)
    if rc:
        return rc, []
    return 0, [Image(handle=outs[i]) for i in range(nout)]


def batch_filters(ptr, stride, w, h, c, step, count, filters, allow_experiments=1, stream=None):
    """Pointwise filter-* requests on every frame of a resident batch, one fused launch. Returns the IMP_* code."""
    arr = (C.c_char_p * max(1, len(filters)))(*[_b(f) for f in filters])
    return lib.impgpu_batch_filters(C.c_void_p(ptr), stride, w, h, c, step, count, arr, len(filters), int(allow_experiments),
                                    C.c_void_p(stream or 0))


class Request:
    """RunJob's parsed request (bridge.c:304-372 + encoder choice :413-466)."""

    def __init__(self, uri, extension="", config=None):
        self.h = C.c_void_p()
        cfg = config or Config()
        self.code = lib.impgpu_parse_request(_b(uri), _b(extension), C.byref(cfg.c), C.byref(self.h))
        j = lib.impgpu_request_job(self.h).contents
        d = lambda v: None if v is None else v.decode()
        self.crop, self.gravity, self.resize = d(j.crop), d(j.gravity), d(j.resize)
        self.simple, self.need_flatten = j.simple, j.need_flatten
        self.filters = [j.filters[i].decode() for i in range(j.filter_count)]
        self.quality = d(lib.impgpu_request_quality(self.h))
        self.format = d(lib.impgpu_request_format(self.h))
        self.page = lib.impgpu_request_page(self.h)
        self.mime = lib.impgpu_request_mime(self.h)
        self.destructive = lib.impgpu_request_destructive(self.h)

    def run(self, image, config):
        """impgpu_run_ops with the parsed job. Returns (code, step)."""
        step = C.c_int()
        rc = lib.impgpu_run_ops(C.byref(image.h), lib.impgpu_request_job(self.h), C.byref(config.c), C.byref(step))
        return rc, step.value

    def __del__(self):
        try:
            if self.h:
                lib.impgpu_request_free(C.byref(self.h))
        except Exception:
            pass
