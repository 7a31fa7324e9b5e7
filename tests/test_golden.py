"""Committed fixtures (tests/golden/, generated from the oracle -- see make_golden.py for provenance):
the oracle must still reproduce them (CPU), and the HIP path must reproduce them through the C ABI (GPU)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as orc

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = np.load(os.path.join(HERE, "golden", "imp_golden.npz"))
CASES = json.load(open(os.path.join(HERE, "golden", "imp_golden.json")))["cases"]


def ids(c):
    return "%s-%s" % (c["kind"], c.get("out", c.get("src")))


@pytest.mark.parametrize("case", CASES, ids=ids)
def test_oracle_reproduces_golden(case):
    src = DATA[case["src"]]
    k = case["kind"]
    if k == "cv_resize":
        got = orc.cv_resize(src, case["dw"], case["dh"], case["interp"])
    elif k == "filter":
        rc, got = orc.filter(src, case["request"])
    elif k == "crop":
        rc, got = orc.crop(src, case["args"], case["gravity"])
    elif k == "watermark":
        rc, got = orc.watermark(src, DATA[case["overlay"]], *case["pos"])
    elif k == "paper":
        got = orc.blend_with_paper(src)
    elif k == "brightness":
        assert np.float32(orc.brightness(src)) == np.float32(case["value"])
        return
    assert np.array_equal(got, DATA[case["out"]])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=ids)
def test_gpu_reproduces_golden(gpu, case):
    src = DATA[case["src"]]
    k = case["kind"]
    im = gpu.Image(src)
    if k == "cv_resize":
        assert im.cv_resize(case["dw"], case["dh"], case["interp"]) == 0
    elif k == "filter":
        assert im.filter(case["request"]) == 0
    elif k == "crop":
        assert im.crop(case["args"], case["gravity"]) == 0
    elif k == "watermark":
        cfg = gpu.Config()
        assert cfg.prepare_watermark(DATA[case["overlay"]], *case["pos"]) == 0
        assert im.watermark(cfg) == 0
        cfg.release()
    elif k == "paper":
        assert im.blend_with_paper() == 0
    elif k == "brightness":
        assert np.float32(im.calc_perceived_brightness()) == np.float32(case["value"])
        im.release()
        return
    assert np.array_equal(im.numpy(), DATA[case["out"]])
    im.release()
