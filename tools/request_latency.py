"""One request at a time, as an nginx worker runs them (RunJob, bridge.c:302): JPEG file in -> resize=224,0 -> JPEG file out.
Device path: impgpu_image_decode_jpeg -> impgpu_run_ops -> impgpu_image_encode_jpeg (two waits, nothing else crosses the link
but the two files).  Host codecs for scale: libjpeg-turbo (Pillow) decode and encode of the same files on one core of the box --
the reference's cvResize in between is not timed (no OpenCV here), so the host column is a lower bound for the reference."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ngx_http_imgproc_amd as gpu
from ngx_http_imgproc_amd.workloads import photo_like
from PIL import Image

gpu.env_start(0)
cfg = gpu.Config()


def timed(fn, reps):
    for _ in range(5):
        fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    t.sort()
    return t[len(t) // 2] * 1e3, t[int(len(t) * 0.95)] * 1e3


for (w, h) in ((640, 480), (1280, 720), (1920, 1080), (3840, 2160)):
    b = io.BytesIO()
    Image.fromarray(photo_like(h, w, seed=2)).save(b, format="JPEG", quality=90, subsampling=2)
    blob = b.getvalue()

    def device():
        rc, im = gpu.Image.decode_jpeg(blob)
        assert rc == 0
        rc, _ = gpu.run_ops(im, cfg, resize="224,0")
        assert rc == 0
        rc, out = im.encode_jpeg(86)
        assert rc == 0
        im.release()
        return out

    def device_one_wait():                                       # the same request with everything enqueued behind the decode: one wait
        rc, out = gpu.jpeg_request_one_wait(blob, cfg, 86, resize="224,0")
        assert rc == 0
        return out

    def host():
        a = np.asarray(Image.open(io.BytesIO(blob)))
        small = a[:: max(1, h // 168), :: max(1, w // 224)][:168, :224]       # (a stand-in frame of the answer's size)
        o = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(small)).save(o, format="JPEG", quality=86, subsampling=2)
        return o.getvalue()

    def device_full():                                           # no resize: the answer has the request's size (a filter, a watermark)
        rc, im = gpu.Image.decode_jpeg(blob)
        assert rc == 0
        rc, _ = gpu.run_ops(im, cfg, filters=["gamma=1.2"])
        assert rc == 0
        rc, out = im.encode_jpeg(86)
        assert rc == 0
        im.release()
        return out

    def host_full():
        a = np.asarray(Image.open(io.BytesIO(blob)))
        o = io.BytesIO()
        Image.fromarray(a).save(o, format="JPEG", quality=86, subsampling=2)
        return o.getvalue()

    assert device_one_wait() == device()
    o50, o95 = timed(device_one_wait, 60)
    d50, d95 = timed(device, 60)
    h50, h95 = timed(host, 20)
    f50, f95 = timed(device_full, 40)
    g50, g95 = timed(host_full, 10)
    print("%4dx%-4d %8d B in | thumbnail answer, ONE wait: device median %6.3f ms, p95 %6.3f" % (w, h, len(blob), o50, o95))
    print("%4dx%-4d %8d B in | thumbnail answer: device median %6.2f ms, p95 %6.2f; host codecs alone (decode + encode, one core) %6.2f ms | full-size answer (filter-gamma): device %6.2f ms, p95 %6.2f; host codecs alone %6.2f ms"
          % (w, h, len(blob), d50, d95, h50, f50, f95, g50))
