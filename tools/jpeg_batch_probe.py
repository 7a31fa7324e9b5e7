"""Where a batched JPEG request loop spends its time (one thread): decode call / per-request resize+download enqueue / sync / release."""
import ctypes as C
import os
import sys
import time

import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ngx_http_imgproc_amd as imp  # noqa: E402
from ngx_http_imgproc_amd.workloads import MIXED_RESIZE  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
imp.env_start(0)
lib = imp.lib
files = bench.jpeg_pool(64)
items = [files[i % 64] for i in range(B)]
cfg = imp.Config()
out_bytes = 224 * 224 * 4 * 4
hdst = lib.impgpu_host_alloc(out_bytes * B)
blobs = (C.c_char_p * B)(*[b for _, _, b in items])
sizes = (C.c_size_t * B)(*[len(b) for _, _, b in items])
for rep in range(6):
    imgs = (C.c_void_p * B)()
    codes = (C.c_int * B)()
    t0 = time.perf_counter()
    rc = lib.impgpu_batch_decode_jpeg(blobs, sizes, B, imgs, codes)
    t1 = time.perf_counter()
    assert rc == 0 and not any(codes), (rc, list(codes))
    for k in range(B):
        one = C.c_void_p(imgs[k])
        lib.impgpu_resize(C.byref(one), MIXED_RESIZE, C.byref(cfg.c), 0)
        ow = lib.impgpu_image_width(one)
        lib.impgpu_image_download_pinned(one, hdst + out_bytes * k, (ow * 3 + 3) & ~3)
        imgs[k] = one
    t2 = time.perf_counter()
    lib.impgpu_sync()
    t3 = time.perf_counter()
    for k in range(B):
        one = C.c_void_p(imgs[k])
        lib.impgpu_image_release(C.byref(one))
    t4 = time.perf_counter()
    print("batch %d: decode %.2f ms, resize+download enqueue %.2f ms, sync %.2f ms, release %.2f ms -> %.0f req/s" %
          (B, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, B / (t4 - t0)), flush=True)
imp.env_destroy()
