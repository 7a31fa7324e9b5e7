"""ctypes signatures of libimpgpu.so (include/impgpu.h). Import fails loudly if the library is absent."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IMPGPU_LIB") or os.path.join(HERE, "libimpgpu.so")   # IMPGPU_LIB: A/B a differently built library


class ImpError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = "impgpu: %s failed with code %d" % (what, code)
        try:
            detail = lib.impgpu_last_error().decode()
            if code == 90 and detail:
                msg += " (%s)" % detail
        except Exception:
            pass
        super().__init__(msg)


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "ngx_http_imgproc_amd: %s is missing. Build it with `python -m ngx_http_imgproc_amd.build` "
        "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH
    )



def _share_torch_hip_runtime():
    """One HIP runtime per process.  libimpgpu.so needs `libamdhip64.so.7`; PyTorch-ROCm bundles its own copy under
    torch/lib, which its libraries ask for by another file name (`libamdhip64.so`), so the loader only recognises the two as
    the same library when torch's copy is the one already loaded.  With the system copy loaded first, `import torch`
    brings in a second runtime and finds no device.  So, when torch is installed and not imported yet, its copy is loaded
    here, before libimpgpu.so binds to it by soname -- whichever of the two a program imports first, both end up on the
    same runtime.  (A C caller -- nginx -- has no torch and links /opt/rocm's.)  IMPGPU_SYSTEM_HIP=1 skips this."""
    import sys

    if os.environ.get("IMPGPU_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):
        pass


_share_torch_hip_runtime()
lib = C.CDLL(LIB_PATH)


class CConfig(C.Structure):
    _fields_ = [
        ("max_target_w", C.c_uint),
        ("max_target_h", C.c_uint),
        ("max_filters_count", C.c_int),
        ("allow_experiments", C.c_int),
        ("watermark_opacity", C.c_int),
        ("watermark_gravity_x", C.c_char),
        ("watermark_gravity_y", C.c_char),
        ("watermark_offset_x", C.c_int),
        ("watermark_offset_y", C.c_int),
        ("watermark", C.c_void_p),
    ]


class CJob(C.Structure):
    _fields_ = [
        ("crop", C.c_char_p),
        ("gravity", C.c_char_p),
        ("resize", C.c_char_p),
        ("simple", C.c_int),
        ("filters", C.POINTER(C.c_char_p)),
        ("filter_count", C.c_int),
        ("need_flatten", C.c_int),
    ]


class CJpegPrepared(C.Structure):                      # impgpu_jpeg_prepared
    _fields_ = [("head", C.c_void_p), ("head_size", C.c_size_t), ("scan", C.c_void_p), ("scan_size", C.c_size_t), ("registered", C.c_int)]


class CGifPage(C.Structure):
    _fields_ = [
        ("indices", C.c_void_p),
        ("width", C.c_int), ("height", C.c_int), ("pitch", C.c_int),
        ("left", C.c_int), ("top", C.c_int),
        ("dispose", C.c_int),
        ("transparency_key", C.c_int),
        ("palette", C.c_void_p),
    ]


P = C.c_void_p
PP = C.POINTER(C.c_void_p)
IP = C.POINTER(C.c_int)

# every symbol include/impgpu.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "impgpu_env_start": (C.c_int, [C.c_int]),
    "impgpu_env_destroy": (None, []),
    "impgpu_env_device": (C.c_int, []),
    "impgpu_last_error": (C.c_char_p, []),
    "impgpu_sync": (C.c_int, []),
    "impgpu_env_stream": (P, []),
    "impgpu_fault_arm": (C.c_int, [C.c_int, C.c_long]),
    "impgpu_env_numa_node": (C.c_int, []),
    "impgpu_env_bind_thread": (C.c_int, []),
    "impgpu_image_upload": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, PP]),
    "impgpu_image_create": (C.c_int, [C.c_int, C.c_int, C.c_int, PP]),
    "impgpu_image_wrap": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, PP]),
    "impgpu_image_decode_jpeg": (C.c_int, [C.c_char_p, C.c_size_t, PP]),
    "impgpu_batch_decode_jpeg": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, PP, IP]),
    "impgpu_jpeg_info": (C.c_int, [C.c_char_p, C.c_size_t, IP, IP, IP]),
    "impgpu_batch_decode_jpeg_begin": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, PP]),
    "impgpu_batch_decode_jpeg_finish": (C.c_int, [PP, PP, IP]),
    "impgpu_batch_decode_jpeg_prepared": (C.c_int, [C.POINTER(CJpegPrepared), C.c_int, PP, IP]),
    "impgpu_batch_decode_jpeg_prepared_begin": (C.c_int, [C.POINTER(CJpegPrepared), C.c_int, PP]),
    "impgpu_batch_decode_jpeg_pending": (C.c_int, [P, PP]),
    "impgpu_batch_encode_jpeg_begin": (C.c_int, [PP, C.c_int, C.c_int, PP]),
    "impgpu_batch_encode_jpeg_finish": (C.c_int, [PP, PP, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), IP]),
    "impgpu_host_register": (C.c_int, [P, C.c_size_t]),
    "impgpu_host_unregister": (C.c_int, [P]),
    "impgpu_image_decode_png": (C.c_int, [C.c_char_p, C.c_size_t, PP]),
    "impgpu_png_info": (C.c_int, [C.c_char_p, C.c_size_t, IP, IP, IP]),
    "impgpu_png_stage_times": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
    "impgpu_png_scanlines": (C.c_int, [C.c_char_p, C.c_size_t, P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "impgpu_jpeg_coefficients": (C.c_int, [C.c_char_p, C.c_size_t, C.c_int, P, C.c_size_t, IP]),
    "impgpu_jpeg_sync_stats": (None, [IP]),
    "impgpu_jpeg_profile": (C.c_int, [C.c_int]),
    "impgpu_jpeg_counters": (C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]),
    "impgpu_jpeg_classify": (C.c_int, [C.c_char_p, C.c_size_t]),
    "impgpu_jpeg_stage_times": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
    "impgpu_host_alloc": (P, [C.c_size_t]),
    "impgpu_host_free": (None, [P]),
    "impgpu_image_upload_pinned": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, PP]),
    "impgpu_image_download_pinned": (C.c_int, [P, P, C.c_int]),
    "impgpu_image_upload_fi32": (C.c_int, [P, C.c_int, C.c_int, C.c_int, PP]),
    "impgpu_image_download_fi": (C.c_int, [P, C.c_int, P, C.c_int]),
    "impgpu_gif_compose": (C.c_int, [C.POINTER(CGifPage), C.c_int, C.c_int, C.c_int, PP]),
    "impgpu_gif_compose_album": (C.c_int, [C.POINTER(CGifPage), C.c_int, C.c_int, C.c_int, PP]),
    "impgpu_image_clone": (C.c_int, [P, PP]),
    "impgpu_image_download": (C.c_int, [P, P, C.c_int]),
    "impgpu_batch_download": (C.c_int, [PP, C.c_int, PP, IP]),
    "impgpu_image_encode_jpeg": (C.c_int, [P, C.c_int, P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "impgpu_batch_encode_jpeg": (C.c_int, [PP, C.c_int, C.c_int, PP, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), IP]),
    "impgpu_jpeg_encode_bound": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "impgpu_album_upload": (C.c_int, [PP, C.c_int, C.c_int, C.c_int, C.c_int, IP, PP]),
    "impgpu_album_download": (C.c_int, [P, PP, IP]),
    "impgpu_album_count": (C.c_int, [P]),
    "impgpu_image_width": (C.c_int, [P]),
    "impgpu_image_height": (C.c_int, [P]),
    "impgpu_image_channels": (C.c_int, [P]),
    "impgpu_image_step": (C.c_int, [P]),
    "impgpu_image_device_ptr": (P, [P]),
    "impgpu_image_release": (None, [PP]),
    "impgpu_crop": (C.c_int, [PP, C.c_char_p, C.c_char_p]),
    "impgpu_resize": (C.c_int, [PP, C.c_char_p, C.POINTER(CConfig), C.c_int]),
    "impgpu_cv_resize": (C.c_int, [PP, C.c_int, C.c_int, C.c_int]),
    "impgpu_filter": (C.c_int, [PP, C.c_char_p, C.c_int]),
    "impgpu_prepare_watermark": (C.c_int, [C.POINTER(CConfig), P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "impgpu_watermark": (C.c_int, [P, C.POINTER(CConfig)]),
    "impgpu_blend_with_paper": (C.c_int, [P]),
    "impgpu_calc_perceived_brightness": (C.c_int, [P, C.POINTER(C.c_float)]),
    "impgpu_ascii": (C.c_int, [P, C.c_char_p, P, C.c_long, C.POINTER(C.c_long)]),
    "impgpu_gray2bgr": (C.c_int, [PP]),
    "impgpu_rgb2hsv": (C.c_int, [P]),
    "impgpu_hsv2rgb": (C.c_int, [P]),
    "impgpu_run_ops": (C.c_int, [PP, C.POINTER(CJob), C.POINTER(CConfig), IP]),
    "impgpu_crop_geometry": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_char_p, IP, IP, IP, IP]),
    "impgpu_resize_geometry": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.POINTER(CConfig), C.c_int, IP, IP, IP]),
    "impgpu_filter_check": (C.c_int, [C.c_char_p, C.c_int]),
    "impgpu_check_destructive": (C.c_int, [C.c_char_p]),
    "impgpu_parse_request": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(CConfig), PP]),
    "impgpu_request_job": (C.POINTER(CJob), [P]),
    "impgpu_request_quality": (C.c_char_p, [P]),
    "impgpu_request_format": (C.c_char_p, [P]),
    "impgpu_request_page": (C.c_int, [P]),
    "impgpu_request_mime": (C.c_int, [P]),
    "impgpu_request_destructive": (C.c_int, [P]),
    "impgpu_request_free": (None, [PP]),
    "impgpu_batch_cv_resize": (C.c_int, [P, C.c_longlong, C.c_int, C.c_int, C.c_int, P, C.c_longlong, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int, P]),
    "impgpu_batch_resize_mixed": (C.c_int, [P, C.c_int, C.c_int, C.c_int, P]),
    "impgpu_batch_filters": (C.c_int, [P, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p),
                                       C.c_int, C.c_int, P]),
    "impgpu_batch_resize_rotate_watermark": (C.c_int, [P, C.c_longlong, C.c_int, C.c_int, C.c_int, P, C.c_longlong,
                                                       C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(CConfig),
                                                       C.c_int, C.c_int, P]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)   # AttributeError here = the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args
