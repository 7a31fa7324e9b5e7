#!/usr/bin/env python3
"""Bounded repeat of the exact-2x DMA-ring resizes against the oracle (one process): the ring's ordering is enforced by
hand-written waits, so a timing-dependent mistake would show as an occasional wrong row rather than a fault."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as orc
import ngx_http_imgproc_amd as imp
imp.env_start(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(5)
bad = 0
for c in (3, 4):
    for (sh, sw) in ((250, 528), (480, 640), (126, 1024), (1080, 1920)):
        arr = rng.integers(0, 256, (sh, sw, c), dtype=np.uint8)
        for mode, name in ((orc.INTER_LANCZOS4, "lanczos"), (orc.INTER_CUBIC, "cubic")):
            want = orc.cv_resize(arr, sw // 2, sh // 2, mode)
            n = 0
            for _ in range(reps):
                im = imp.Image(arr)
                im.cv_resize(sw // 2, sh // 2, mode)
                n += int(not np.array_equal(im.numpy(), want))
                im.release()
            bad += n
            print("c=%d %dx%d %-7s wrong %d / %d" % (c, sw, sh, name, n, reps), flush=True)
print("TOTAL WRONG", bad)
imp.env_destroy()
sys.exit(1 if bad else 0)
