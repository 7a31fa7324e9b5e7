"""Phase timing of k_jpeg_sync over a 64-file launch: IMPGPU_JPEG_TRACE=2 makes the library print every workgroup's clock at
its phase boundaries; this sums them up.  python tools/jpeg_batch_trace.py [files]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch  # noqa: F401
    sys.path.insert(0, ROOT)
    import bench
    import ngx_http_imgproc_amd as imp
    B = int(sys.argv[2])
    imp.env_start(0)
    lib = imp.lib
    files = bench.jpeg_pool(64)
    items = [files[i % 64] for i in range(B)]
    blobs = (C.c_char_p * B)(*[b for _, _, b in items])
    sizes = (C.c_size_t * B)(*[len(b) for _, _, b in items])
    for rep in range(4):
        if rep == 3:
            os.environ["IMPGPU_JPEG_TRACE"] = "2"
        imgs = (C.c_void_p * B)()
        codes = (C.c_int * B)()
        assert lib.impgpu_batch_decode_jpeg(blobs, sizes, B, imgs, codes) == 0
        assert not any(codes), list(codes)
        for k in range(B):
            one = C.c_void_p(imgs[k])
            lib.impgpu_image_release(C.byref(one))
    imp.env_destroy()
    sys.exit(0)

B = sys.argv[1] if len(sys.argv) > 1 else "64"
out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", B], capture_output=True, text=True)
rows = []
misc = []
ww = []
for line in out.stderr.splitlines():
    if line.startswith("ww "):
        ww.append([float(x) for x in line.split(":")[1].split()])
        continue
    if line.startswith("wg "):
        v = [float(x) for x in line.split(":")[1].split()]
        rows.append(v)
    elif line.startswith("jpeg"):
        misc.append(line)
if not rows:
    print(out.stderr[-2000:])
    sys.exit(1)
import numpy as np
a = np.array(rows)
names = ["start", "walks done", "candidates exchanged", "maps done", "scan + look-back done", "end"]
print("%d workgroups of k_jpeg_sync; microseconds since the first workgroup started" % len(a))
for i, nme in enumerate(names):
    col = a[:, i]
    print("  %-24s mean %8.1f  p50 %8.1f  p95 %8.1f  max %8.1f" % (nme, col.mean(), np.median(col), np.percentile(col, 95), col.max()))
d = np.diff(a, axis=1)
for i, nme in enumerate(["walks", "exchange wait", "maps + repair walks", "scan + look-back", "pick + chase + publish"]):
    col = d[:, i]
    print("  phase %-24s mean %8.1f  p50 %8.1f  p95 %8.1f  max %8.1f" % (nme, col.mean(), np.median(col), np.percentile(col, 95), col.max()))
rep = sum(int(l.split(" repair walks")[0].split()[-1]) for l in misc if "repair walks" in l)
ch = sum(int(l.split(" chunks chased")[0].split()[-1]) for l in misc if "chunks chased" in l)
nch = sum(int(l.split(" chunks of")[0].split()[-1]) for l in misc if "chunks of" in l)
sw = sum(int(l.split(" walks in k_jpeg_select")[0].split()[-1]) for l in misc if "walks in k_jpeg_select" in l)
print("  %d chunks, %d repair walks, %d chunks chased, %d walks left to k_jpeg_select" % (nch, rep, ch, sw))
if ww:
    w = np.array(ww)
    dur = w[:, 1] - w[:, 0]
    print("  k_jpeg_write: %d workgroups; start mean %.1f max %.1f; duration mean %.1f p50 %.1f p95 %.1f max %.1f; last end %.1f (microseconds since k_jpeg_select's first workgroup)" %
          (len(w), w[:, 0].mean() - w[:, 0].min(), w[:, 0].max() - w[:, 0].min(), dur.mean(), np.median(dur), np.percentile(dur, 95), dur.max(), w[:, 1].max() - w[:, 0].min()))
print(misc[-1])
