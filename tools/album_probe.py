"""A 40-frame 480x270 BGRA animation through RunJob's operator segment: one album handle (each operator one launch)
against the reference's loop shape (one impgpu_run_ops per frame).  Wall time per request, uploads and downloads included."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ngx_http_imgproc_amd as gpu

gpu.env_start(0)
rng = np.random.default_rng(5)
n, h, w = 40, 270, 480
frames = [rng.integers(0, 256, (h, w, 4), dtype=np.uint8) for _ in range(n)]
cfg = gpu.Config(allow_experiments=True, max_filters=5)
kw = dict(crop="16,9", resize="240,0", filters=["gamma=1.3", "rotate=90"], need_flatten=0)


def per_frame():
    ims = [gpu.Image(f) for f in frames]
    for im in ims:
        assert gpu.run_ops(im, cfg, **kw)[0] == 0
    outs = [im.numpy() for im in ims]
    for im in ims:
        im.release()
    return outs


def album():
    al = gpu.Image.album(frames)
    assert gpu.run_ops(al, cfg, **kw)[0] == 0
    outs = al.frames()
    al.release()
    return outs


a, b = per_frame(), album()
assert all(np.array_equal(x, y) for x, y in zip(a, b))
for name, fn in (("per-frame loop", per_frame), ("album handle", album)):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        fn()
    dt = (time.perf_counter() - t0) / reps
    print("%-16s %8.2f ms per %d-frame request  (%.0f frames/s)" % (name, dt * 1e3, n, n / dt))
