#!/usr/bin/env python3
"""CalcPerceivedBrightness on a 1080p noise frame, a few calls, for a kernel trace of the terms + replay kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ngx_http_imgproc_amd as imp
imp.env_start(0)
rng = np.random.default_rng(3)
im = imp.Image(rng.integers(0, 256, (1080, 1920, 4), dtype=np.uint8))
for _ in range(3):
    im.calc_perceived_brightness()
t0 = time.perf_counter()
for _ in range(10):
    b = im.calc_perceived_brightness()
print("brightness %.4f  %.1f us/call" % (b, (time.perf_counter() - t0) / 10 * 1e6))
imp.env_destroy()
