"""A seeded sweep of JPEG files nobody chose -- sizes 8..1700, three kinds of content, six qualities, three samplings, optimised tables,
restart intervals by block count and by row, gray -- decoded every way the library offers (one at a time, one batch, batches of
seven, prepared by the caller: staged and out of pinned memory) against Pillow's pixels.  Run through gpurun; prints the failures."""
import io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from PIL import Image
from conftest import noise_image, smooth_image
from ngx_http_imgproc_amd.workloads import photo_like
import ngx_http_imgproc_amd as gpu
gpu.env_start(0)
rng = np.random.default_rng(20261005)
files, wants = [], []
for i in range(160):
    h = int(rng.integers(8, 1300)); w = int(rng.integers(8, 1700))
    kind = int(rng.integers(0, 4))
    a = photo_like(h, w, i) if kind < 2 else smooth_image(h, w, 3, seed=i) if kind == 2 else noise_image(h, w, 3, i)
    q = int(rng.choice([30, 50, 75, 85, 90, 95]))
    sub = str(rng.choice(["4:2:0", "4:2:2", "4:4:4"]))
    kw = dict(quality=q, subsampling=sub)
    r = int(rng.integers(0, 5))
    if r == 1: kw["optimize"] = True
    if r == 2: kw["restart_marker_blocks"] = int(rng.integers(1, 40))
    if r == 3: kw["restart_marker_rows"] = 1
    gray = rng.integers(0, 8) == 0
    b = io.BytesIO()
    try:
        (Image.fromarray(a[:, :, 0]) if gray else Image.fromarray(a)).save(b, "JPEG", **({k: v for k, v in kw.items() if k != "subsampling"} if gray else kw))
    except OSError:
        continue
    blob = b.getvalue(); files.append(blob)
    d = np.asarray(Image.open(io.BytesIO(blob)))
    wants.append(d[:, :, None] if d.ndim == 2 else d[:, :, ::-1])
bad = 0
def check(tag, res):
    global bad
    for k, ((code, im), want) in enumerate(zip(res, wants)):
        ok = code == 0 and np.array_equal(im.numpy(), want)
        if im is not None: im.release()
        if not ok:
            bad += 1; print("FAIL", tag, k, code, want.shape, len(files[k]), flush=True)
# one at a time, whole batch, batches of 7, prepared (staged and pinned)
check("lone", [gpu.batch_decode_jpeg([f])[0] for f in files])
check("batch", gpu.batch_decode_jpeg(files))
res = []
for i in range(0, len(files), 7): res += gpu.batch_decode_jpeg(files[i:i + 7])
check("sevens", res)
prep = [gpu.jpeg_unstuff(f) or f for f in files]
print("prepared:", sum(isinstance(p, tuple) for p in prep), "of", len(prep))
for pinned in (False, True):
    res = []
    for i in range(0, len(files), 5): res += gpu.batch_decode_jpeg_prepared(prep[i:i + 5], pinned=pinned)
    check("prepared pinned=%s" % pinned, res)
print("files", len(files), "failures", bad)
