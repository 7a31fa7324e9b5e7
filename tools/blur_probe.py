#!/usr/bin/env python3
"""filter-blur=SIGMA on device-resident 1920x1080 frames, one frame per call (tools/blur_probe.py [sigma] [channels] [frames])."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ngx_http_imgproc_amd as imp

sigma = sys.argv[1] if len(sys.argv) > 1 else "8"
c = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = int(sys.argv[3]) if len(sys.argv) > 3 else 48
imp.env_start(0)
rng = np.random.Generator(np.random.PCG64(7))
base = imp.Image(rng.integers(0, 256, size=(1080, 1920, c), dtype=np.uint8))
imgs = [base.clone() for _ in range(n)]
w = base.clone(); w.filter("blur=" + sigma, 1); w.release(); imp.sync()
t0 = time.perf_counter()
for im in imgs:
    im.filter("blur=" + sigma, 1)
imp.sync()
dt = (time.perf_counter() - t0) / n
print("blur=%s c=%d: %.1f us/frame" % (sigma, c, dt * 1e6), flush=True)
for im in imgs:
    im.release()
imp.env_destroy()
