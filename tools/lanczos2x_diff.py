"""Where does the exact-2x LANCZOS4 kernel differ from the oracle?  (columns within a 64-column strip, channels, magnitudes)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as orc
import ngx_http_imgproc_amd as imp
from conftest import noise_image
imp.env_start(0)
arr = noise_image(256, 768, 4, 20)
want = orc.cv_resize(arr, 384, 128, orc.INTER_LANCZOS4)
im = imp.Image(arr); assert im.cv_resize(384, 128, imp.INTER_LANCZOS4) == 0
got = im.numpy()
d = got.astype(int) - want.astype(int)
bad = np.argwhere(d != 0)
print("differing values:", len(bad), "of", d.size)
if len(bad):
    cols = np.unique(bad[:, 1]); print("columns:", cols[:40], "... mod 64:", np.unique(cols % 64)[:64])
    print("channels:", np.unique(bad[:, 2]), "rows:", np.unique(bad[:, 0])[:20])
    print("max |d|:", np.abs(d).max(), "sample:", [(tuple(b), int(d[tuple(b)])) for b in bad[:8]])
    y, x, c = bad[0]
    print("got row segment", got[y, max(0, x-2):x+6, c], "want", want[y, max(0, x-2):x+6, c])
imp.env_destroy()
