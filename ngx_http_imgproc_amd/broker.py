"""ctypes mirror of the WORKER side of include/impgpu_broker.h (glue/imp_gpu_client.c built as libimpgpu_client.so): no HIP
in this process -- used by the tests and tools; an nginx worker compiles the same C file into the module."""
import ctypes as C
import os

from ._lib import CConfig, CJob

HERE = os.path.dirname(os.path.abspath(__file__))
CLIENT_LIB_PATH = os.path.join(HERE, "libimpgpu_client.so")
BROKER_PATH = os.path.join(HERE, "impgpu_broker")

IN_FILE, IN_FRAME = 0, 1
OUT_JPEG, OUT_FRAME, OUT_INFO, OUT_ASCII = 0, 1, 2, 3
NOT_TAKEN = -1


class CRequest(C.Structure):
    _fields_ = [("in_kind", C.c_int), ("input", C.c_void_p), ("input_bytes", C.c_size_t),
                ("width", C.c_int), ("height", C.c_int), ("channels", C.c_int), ("step", C.c_int),
                ("job", C.POINTER(CJob)), ("config", C.POINTER(CConfig)), ("watermark_id", C.c_int),
                ("out_kind", C.c_int), ("quality", C.c_int), ("ascii_args", C.c_char_p)]


class CAnswer(C.Structure):
    _fields_ = [("code", C.c_int), ("step", C.c_int), ("data", C.c_void_p), ("bytes", C.c_size_t),
                ("width", C.c_int), ("height", C.c_int), ("channels", C.c_int), ("row_step", C.c_int),
                ("brightness", C.c_float), ("batch_size", C.c_int), ("broker_us", C.c_uint), ("error", C.c_char_p)]


if not os.path.exists(CLIENT_LIB_PATH):
    raise ImportError("%s is missing: python ngx_http_imgproc_amd/build.py" % CLIENT_LIB_PATH)
clib = C.CDLL(CLIENT_LIB_PATH)
clib.impgpu_client_attach.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
clib.impgpu_client_detach.argtypes = [C.POINTER(C.c_void_p)]
clib.impgpu_client_detach.restype = None
clib.impgpu_client_run.argtypes = [C.c_void_p, C.POINTER(CRequest), C.POINTER(CAnswer)]
clib.impgpu_client_prepare_watermark.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
clib.impgpu_client_last_error.restype = C.c_char_p
clib.impgpu_client_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
clib.impgpu_jpeg_unstuff.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t] + [C.POINTER(C.c_size_t)] * 4
clib.impgpu_client_input_buffer.argtypes = [C.c_void_p, C.c_size_t]
clib.impgpu_client_input_buffer.restype = C.c_void_p


class Client:
    """One worker's connection: one slot, one request in flight."""

    def __init__(self, name):
        self.h = C.c_void_p()
        rc = clib.impgpu_client_attach(name.encode(), C.byref(self.h))
        if rc != 0:
            raise RuntimeError("attach failed (%d): %s" % (rc, clib.impgpu_client_last_error().decode()))
        self._keep = []

    def close(self):
        if self.h:
            clib.impgpu_client_detach(C.byref(self.h))

    def prepare_watermark(self, overlay):
        import numpy as np

        ov = np.ascontiguousarray(overlay)
        self._keep.append(ov)                       # the client re-registers from this memory after a broker restart
        wid = C.c_int()
        rc = clib.impgpu_client_prepare_watermark(self.h, ov.ctypes.data, ov.shape[1], ov.shape[0], ov.shape[2], ov.strides[0], C.byref(wid))
        if rc != 0:
            raise RuntimeError("prepare_watermark failed (%d): %s" % (rc, clib.impgpu_client_last_error().decode()))
        return wid.value

    def run(self, blob=None, frame=None, crop=None, gravity=None, resize=None, filters=(), simple=0, need_flatten=0,
            config=None, watermark_id=0, out=OUT_JPEG, quality=86, ascii_args=None):
        """-> (transport rc, code, step, payload, answer).  payload: bytes (JPEG), ndarray (FRAME), or None."""
        import numpy as np

        r = CRequest()
        if blob is not None:
            buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
            r.in_kind, r.input, r.input_bytes = IN_FILE, C.cast(buf, C.c_void_p), len(blob)
        else:
            f = np.ascontiguousarray(frame)
            buf = f
            r.in_kind, r.input, r.input_bytes = IN_FRAME, f.ctypes.data, f.strides[0] * f.shape[0]
            r.width, r.height, r.channels, r.step = f.shape[1], f.shape[0], f.shape[2], f.strides[0]
        fl = [s.encode() for s in filters]
        arr = (C.c_char_p * max(len(fl), 1))(*fl)
        job = CJob(crop.encode() if crop else None, gravity.encode() if gravity else None, resize.encode() if resize else None,
                   simple, C.cast(arr, C.POINTER(C.c_char_p)), len(fl), need_flatten)
        cfg = config if config is not None else CConfig(2000, 2000, 5, 0, 0, b"l", b"t", 0, 0, None)
        r.job, r.config = C.pointer(job), C.pointer(cfg)
        r.watermark_id, r.out_kind, r.quality = watermark_id, out, quality
        r.ascii_args = ascii_args.encode() if ascii_args is not None else None
        a = CAnswer()
        rc = clib.impgpu_client_run(self.h, C.byref(r), C.byref(a))
        del buf
        if rc != 0 or a.code != 0:
            return rc, a.code, a.step, None, a
        if out in (OUT_JPEG, OUT_ASCII):
            return rc, 0, a.step, C.string_at(a.data, a.bytes), a
        if out == OUT_FRAME:
            rows = np.ctypeslib.as_array(C.cast(a.data, C.POINTER(C.c_ubyte)), shape=(a.height, a.row_step))
            return rc, 0, a.step, rows[:, :a.width * a.channels].reshape(a.height, a.width, a.channels).copy(), a
        return rc, 0, a.step, None, a

    def stats(self):
        s, b, e, p = C.c_ulonglong(), C.c_ulonglong(), C.c_uint(), C.c_uint()
        clib.impgpu_client_stats(self.h, C.byref(s), C.byref(b), C.byref(e), C.byref(p))
        return {"served": s.value, "batches": b.value, "epoch": e.value, "broker_pid": p.value}

    @staticmethod
    def last_error():
        return clib.impgpu_client_last_error().decode()
