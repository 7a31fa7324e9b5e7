#!/usr/bin/env python3
"""Per-call latency of the geometric / blur / blend operators on one 1080p frame, 3 channels vs 4."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ngx_http_imgproc_amd as imp

imp.env_start(0)
rng = np.random.default_rng(1)
for c in (3, 4):
    arr = rng.integers(0, 256, (1080, 1920, c), dtype=np.uint8)
    for name in ("rotate=90", "rotate=180", "flip=10", "flip=01", "blur=1", "blur=2", "blur=8"):
        im = imp.Image(arr)
        for _ in range(3):
            assert im.filter(name) == 0
        imp.sync()
        t0 = time.perf_counter()
        reps = 40
        for _ in range(reps):
            im.filter(name)
        imp.sync()
        dt = (time.perf_counter() - t0) / reps
        print("c=%d %-12s %8.1f us/call" % (c, name, dt * 1e6), flush=True)
        im.release()
imp.env_destroy()
