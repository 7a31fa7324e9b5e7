"""JPEG encode on the device against Pillow (libjpeg-turbo) on this host: one thumbnail, a batch of thumbnails, one 1080p frame."""
import io, sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ngx_http_imgproc_amd as gpu
from ngx_http_imgproc_amd.workloads import photo_like

gpu.env_start(0)


def timed(fn, reps):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def pillow(a, q):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(a[:, :, ::-1])).save(b, format="JPEG", quality=q, subsampling=2)
    return b.getvalue()


for (h, w, n) in ((126, 224, 64), (168, 224, 64), (224, 224, 64), (1080, 1920, 4)):
    frames = [photo_like(h, w, 100 + i) for i in range(n)]
    ims = [gpu.Image(f) for f in frames]
    one = timed(lambda: ims[0].encode_jpeg(90), 50)
    batch = timed(lambda: gpu.batch_encode_jpeg(ims, 90), 20)
    size = len(ims[0].encode_jpeg(90)[1])
    try:
        host = timed(lambda: pillow(frames[0], 90), 20)
        same = pillow(frames[0], 90) == ims[0].encode_jpeg(90)[1]
    except ImportError:
        host, same = float("nan"), None
    print("%4dx%-4d file %7d B | device: one call %8.1f us, batch of %2d %8.1f us = %7.1f us/frame (%.0f frames/s) | Pillow one core %8.1f us | same file: %s"
          % (w, h, size, one * 1e6, n, batch * 1e6, batch / n * 1e6, n / batch, host * 1e6, same))
