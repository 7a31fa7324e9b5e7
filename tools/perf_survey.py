#!/usr/bin/env python3
"""Times every operator of the path on device-resident 1920x1080 BGRA frames (one frame per call, the way a
request uses them; launches are asynchronous, one sync at the end) and prints µs/frame and effective GB/s
(bytes the operator must read + write / time).  Not a benchmark of record -- a map of where the slow kernels are.

    python tools/perf_survey.py [--frames 64] [--json out.json]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import ngx_http_imgproc_amd as imp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    imp.env_start(0)
    rng = np.random.Generator(np.random.PCG64(7))
    W, H = 1920, 1080
    frame = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
    base = imp.Image(frame)
    ov = rng.integers(0, 256, size=(64, 256, 4), dtype=np.uint8)
    cfg = imp.Config(allow_experiments=True)
    cfg.prepare_watermark(ov, "r", "b", 16, 16, 60)
    px = W * H * 4
    results = []

    def bench(name, fn, bytes_moved, fresh=True):
        imgs = [base.clone() for _ in range(args.frames)] if fresh else None
        imp.sync()
        # warm (tables, pools)
        w = base.clone()
        fn(w)
        w.release()
        imp.sync()
        t0 = time.perf_counter()
        for i in range(args.frames):
            fn(imgs[i])
        imp.sync()
        dt = (time.perf_counter() - t0) / args.frames
        for im in imgs:
            im.release()
        results.append({"op": name, "us_per_frame": round(dt * 1e6, 1), "GBps": round(bytes_moved / dt / 1e9, 1)})
        print("%-34s %9.1f us/frame %8.1f GB/s" % (name, dt * 1e6, bytes_moved / dt / 1e9), flush=True)

    def filt(req):
        return lambda im: im.filter(req, 1)

    bench("crop=16,9 (copy)", lambda im: im.crop("16,10"), 2 * 1728 * 1080 * 4)
    bench("resize=224,224 (AREA)", lambda im: im.resize("224,224", cfg), px + 224 * 224 * 4)
    bench("cv_resize 224 CUBIC", lambda im: im.cv_resize(224, 224, imp.INTER_CUBIC), 896 * 896 * 4 + 224 * 224 * 4)
    bench("resize=960,540 (AREA 2x2)", lambda im: im.resize("960,540", cfg), px + px // 4)
    bench("filter-flip=10", filt("flip=10"), 2 * px)
    bench("filter-flip=01", filt("flip=01"), 2 * px)
    bench("filter-rotate=90", filt("rotate=90"), 2 * px)
    bench("filter-rotate=180", filt("rotate=180"), 2 * px)
    bench("filter-gamma=2.2 (LUT)", filt("gamma=2.2"), 2 * px)
    bench("filter-contrast=1.3 (LUT)", filt("contrast=1.3"), 2 * px)
    bench("filter-colorize (LUT)", filt("colorize=ff8000,0.3"), 2 * px)
    bench("filter-modulate (HSV)", filt("modulate=30,120,90"), 2 * px)
    bench("filter-gotham (HSV+LUT)", filt("gotham=1"), 2 * px)
    bench("filter-kelvin", filt("kelvin=1"), 2 * px)
    bench("filter-lomo", filt("lomo=1"), 2 * px)
    bench("filter-rainbow", filt("rainbow=mid"), 2 * px)
    bench("filter-scanline", filt("scanline=0.3,0.5,2,2"), 2 * px)
    bench("filter-gradmap", filt("gradmap=000000,ff8800,ffffff"), 2 * px)
    bench("filter-vignette (HSV+cos)", filt("vignette=0.6"), 2 * px)
    bench("filter-blur=1 (k7)", filt("blur=1"), 2 * px)
    bench("filter-blur=2 (k13)", filt("blur=2"), 2 * px)
    bench("filter-blur=8 (k49)", filt("blur=8"), 2 * px)
    bench("watermark 256x64", lambda im: im.watermark(cfg), 3 * 256 * 64 * 4)
    bench("blend_with_paper", lambda im: im.blend_with_paper(), 2 * px)
    for _ in range(3):                                 # (the first call loads the kernels)
        base.calc_perceived_brightness()
    t0 = time.perf_counter()
    for _ in range(10):
        b = base.calc_perceived_brightness()
    dt = (time.perf_counter() - t0) / 10
    results.append({"op": "calc_perceived_brightness (serial replay)", "us_per_frame": round(dt * 1e6, 1), "GBps": round(px / dt / 1e9, 3)})
    print("%-34s %9.1f us/frame   (value %.4f)" % ("brightness (exact replay)", dt * 1e6, b))
    chain = dict(crop="16,9", resize="640,360", filters=["gotham=1", "rotate=90"])
    bench("run_ops crop+resize+gotham+rot", lambda im: imp.run_ops(im, cfg, **chain), px)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(results, f, indent=1)
    base.release()
    cfg.release()
    imp.env_destroy()


if __name__ == "__main__":
    main()
