#!/usr/bin/env python3
"""Where does the GPU resize differ from the oracle?  usage: diff_probe.py SW SH DW DH MODE(cubic|lanczos|linear|area) [CHANNELS=4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as orc
import ngx_http_imgproc_amd as imp
sw, sh, dw, dh = map(int, sys.argv[1:5])
mode = {"cubic": orc.INTER_CUBIC, "lanczos": orc.INTER_LANCZOS4, "linear": orc.INTER_LINEAR, "area": orc.INTER_AREA}[sys.argv[5]]
imp.env_start(0)
rng = np.random.default_rng(7)
cn = int(sys.argv[6]) if len(sys.argv) > 6 else 4
arr = rng.integers(0, 256, (sh, sw, cn), dtype=np.uint8)
want = orc.cv_resize(arr, dw, dh, mode)
im = imp.Image(arr)
im.cv_resize(dw, dh, mode)
got = im.numpy()
bad = np.argwhere((got.reshape(dh, dw, -1) != want.reshape(dh, dw, -1)).any(axis=2))
print("mismatching pixels:", len(bad), "of", dw * dh)
if len(bad):
    ys, xs = bad[:, 0], bad[:, 1]
    print("rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
    print("row histogram (mod 60):", np.bincount(ys % 60, minlength=60))
    print("col histogram (/64):", np.bincount(xs // 64))
    print("first:", bad[:8].tolist())
