#!/usr/bin/env python3
"""Generates tests/golden/imp_golden.npz: inputs and expected outputs of the pixel path on small frames.

PROVENANCE: the reference (tommiv/ngx_http_imgproc) has no fixtures and cannot be built or run in this image,
so these vectors come from the CPU oracle under oracle/ (our restatement of the reference + OpenCV 2.4.9
semantics), not from the reference itself.  They freeze the oracle's behaviour: tests/test_golden.py checks the
oracle against them on CPU and the HIP path against them on the GPU, so a later edit to either side that changes
a single output byte is caught.  Re-run only on purpose:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as orc  # noqa: E402
from conftest import noise_image, smooth_image  # noqa: E402

RESIZE = [(48, 64, 17, 23), (37, 29, 60, 41), (64, 96, 32, 48), (90, 120, 21, 28)]   # sh, sw, dh, dw
FILTERS = ["flip=10", "flip=01", "flip=11", "rotate=90", "rotate=180", "rotate=270", "modulate=30,120,80",
           "modulate=180,0,250", "colorize=ff8000", "colorize=102030,0.25", "blur=0.8", "blur=2", "blur=5.5", "gamma=2.2",
           "gamma=0.45", "contrast=1.5", "contrast=0.3", "gradmap=000000,ffffff", "gradmap=ff0000,00ff00,0000ff",
           "gotham=1", "lomo=1", "kelvin=1", "rainbow=full", "rainbow=pale", "scanline=0.3,0.6,2,3"]
CROPS = [("16,9", None), ("1,1,r,b", None), ("30px,20px,5px,7px", None), ("1,2,c,c", "l,b")]


def main():
    data, manifest = {}, []

    def put(name, arr):
        data[name] = np.ascontiguousarray(arr)

    for c in (1, 3, 4):
        for i, (sh, sw, dh, dw) in enumerate(RESIZE):
            src = noise_image(sh, sw, c, 100 + i) if i % 2 == 0 else (smooth_image(sh, sw, c) if c > 1 else noise_image(sh, sw, 1, 7))
            put("resize_src_c%d_%d" % (c, i), src)
            for interp in range(5):
                if interp == orc.INTER_AREA and (dh > sh or dw > sw):
                    continue
                name = "resize_out_c%d_%d_m%d" % (c, i, interp)
                put(name, orc.cv_resize(src, dw, dh, interp))
                manifest.append({"kind": "cv_resize", "src": "resize_src_c%d_%d" % (c, i), "out": name, "dw": dw, "dh": dh,
                                 "interp": interp})
    for c in (3, 4):
        src = smooth_image(40, 56, c)
        put("filter_src_c%d" % c, src)
        for j, f in enumerate(FILTERS):
            rc, out = orc.filter(src, f)
            assert rc == 0, f
            name = "filter_out_c%d_%d" % (c, j)
            put(name, out)
            manifest.append({"kind": "filter", "src": "filter_src_c%d" % c, "out": name, "request": f})
        for j, (args, grav) in enumerate(CROPS):
            rc, out = orc.crop(src, args, grav)
            assert rc == 0, args
            name = "crop_out_c%d_%d" % (c, j)
            put(name, out)
            manifest.append({"kind": "crop", "src": "filter_src_c%d" % c, "out": name, "args": args, "gravity": grav})
    base = noise_image(50, 70, 4, 200)
    ov = noise_image(16, 24, 4, 201)
    ov[:, :, 3] = np.linspace(0, 255, 24).astype(np.uint8)[None, :]
    put("wm_base", base)
    put("wm_overlay", ov)
    for j, (gx, gy, ox, oy, op) in enumerate([("r", "b", 4, 4, 60), ("c", "c", 0, 0, 100), ("l", "t", -6, -3, 35)]):
        rc, out = orc.watermark(base, ov, gx, gy, ox, oy, op)
        assert rc == 0
        put("wm_out_%d" % j, out)
        manifest.append({"kind": "watermark", "src": "wm_base", "overlay": "wm_overlay", "out": "wm_out_%d" % j,
                         "pos": [gx, gy, ox, oy, op]})
    put("paper_out", orc.blend_with_paper(base))
    manifest.append({"kind": "paper", "src": "wm_base", "out": "paper_out"})
    manifest.append({"kind": "brightness", "src": "wm_base", "value": float(np.float32(orc.brightness(base)))})
    manifest.append({"kind": "brightness", "src": "filter_src_c3", "value": float(np.float32(orc.brightness(data["filter_src_c3"])))})
    np.savez_compressed(os.path.join(HERE, "imp_golden.npz"), **data)
    with open(os.path.join(HERE, "imp_golden.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "source": "oracle/liboracle.so (NOT the reference)",
                   "cases": manifest}, f, indent=0)
    print("wrote %d arrays, %d cases" % (len(data), len(manifest)))


if __name__ == "__main__":
    main()
