import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import oracle_lib as orc
import ngx_http_imgproc_amd as imp
from conftest import noise_image
imp.env_start(0)
for (sh, sw), (dh, dw) in [((50, 100), (200, 128)), ((50, 64), (200, 64)), ((50, 100), (200, 64)), ((50, 32), (200, 64)), ((50, 16), (200, 64)),
                           ((50, 60), (200, 64)), ((50, 70), (200, 64)), ((50, 80), (200, 64)), ((200, 80), (200, 64)), ((100, 80), (200, 64)), ((270, 480), (1080, 1920))]:
    arr = noise_image(sh, sw, 4, 21)
    want = orc.cv_resize(arr, dw, dh, orc.INTER_CUBIC)
    im = imp.Image(arr); im.cv_resize(dw, dh, imp.INTER_CUBIC); got = im.numpy(); im.release()
    bad = (got != want).any(axis=2)
    print((sh, sw, dh, dw), "scale_x %.3f" % (sw / dw), "bad", int(bad.sum()), "bad rows", int(bad.any(axis=1).sum()), "of", dh)
imp.env_destroy()
