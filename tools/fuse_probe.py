import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ngx_http_imgproc_amd as imp
imp.env_start(0)
rng = np.random.Generator(np.random.PCG64(7))
base = imp.Image(rng.integers(0, 256, size=(1080, 1920, 4), dtype=np.uint8))
ov = rng.integers(0, 256, size=(64, 256, 4), dtype=np.uint8)
cfg = imp.Config(allow_experiments=True)
cfg.prepare_watermark(ov, "r", "b", 16, 16, 60)
for name, kw in (("resize+rotate+wm (one launch)", dict(resize="224,0", filters=["rotate=90"])),
                 ("resize+gamma+rotate+wm (step by step)", dict(resize="224,0", filters=["gamma=1.0001", "rotate=90"]))):
    imgs = [base.clone() for _ in range(64)]
    w = base.clone(); imp.run_ops(w, cfg, **kw); w.release(); imp.sync()
    t0 = time.perf_counter()
    for im in imgs:
        imp.run_ops(im, cfg, **kw)
    imp.sync()
    print("%-40s %6.1f us/request" % (name, (time.perf_counter() - t0) / 64 * 1e6), flush=True)
    for im in imgs: im.release()
imp.env_destroy()
