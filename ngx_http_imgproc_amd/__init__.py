"""ngx_http_imgproc_amd -- host-side mirror of IMP's operator interface over libimpgpu.so.

The product is the C-ABI library (include/impgpu.h, csrc/); this package is the thin
Python binding the tests and bench.py drive it through.  Names, argument strings and
return codes follow the reference's operators (bridge.h, filters.h): Crop, Resize,
Filter, Watermark, BlendWithPaper, CalcPerceivedBrightness, ASCII, and RunJob's operator
segment.  There is no CPU fallback: importing fails loudly when the library is missing,
and every pixel-touching call fails with IMP_ERROR_DEVICE without a GPU.
"""
from ._lib import lib, LIB_PATH, ImpError  # noqa: F401
from .ops import (  # noqa: F401
    IMP_OK, IMP_ERROR_UNSUPPORTED, IMP_ERROR_MALLOC_FAILED, IMP_ERROR_DECODE_FAILED, IMP_ERROR_INVALID_ARGS, IMP_ERROR_NO_SUCH_FILTER, IMP_ERROR_TOO_BIG_TARGET,
    IMP_ERROR_TOO_MUCH_FILTERS, IMP_ERROR_DEVICE, IMP_ERROR_NO_SUCH_WATERMARK,
    INTER_NN, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4,
    Config, Image, env_start, env_destroy, sync,
    jpeg_info, png_info, png_stage_times, batch_decode_jpeg, batch_decode_jpeg_prepared, jpeg_unstuff, jpeg_request_one_wait, batch_encode_jpeg, crop_geometry, resize_geometry, filter_check, check_destructive,
    batch_cv_resize, batch_resize_mixed, ResizeItem, batch_resize_rotate_watermark, batch_filters, run_ops, Request, gif_compose,
)
