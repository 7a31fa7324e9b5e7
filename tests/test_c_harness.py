"""The boundary exercised from the reference's own language: tests/c/runjob_harness.c is plain C99 that includes only
<impgpu.h> and walks RunJob's path (bridge.c:302-724) -- env start, PrepareWatermark, request parse, upload, the operator
segment, then the json / text / encoder exits.  Everything it writes is compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as orc
from conftest import ROOT, noise_image, smooth_image

HARNESS = os.path.join(ROOT, "tests", "c", "_build", "runjob_harness")


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c")])
    return HARNESS


def run(tmp_path, frame, uri, ext, overlay=None, pos=None, env=None):
    h, w, c = frame.shape
    src, out = tmp_path / "frame.raw", tmp_path / "out.raw"
    frame.tofile(src)
    cmd = [build(), str(src), str(w), str(h), str(c), uri, ext, str(out)]
    if overlay is not None:
        ov = tmp_path / "overlay.raw"
        overlay.tofile(ov)
        oh, ow, oc = overlay.shape
        cmd += [str(ov), str(ow), str(oh), str(oc)] + [str(v) for v in pos]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, **(env or {})))
    fields = dict(kv.split("=") for kv in p.stdout.split()) if p.returncode == 0 else {}
    data = np.fromfile(out, dtype=np.uint8) if out.exists() else None
    return p, {k: int(v) for k, v in fields.items()}, data


def test_harness_is_c99_and_fails_loudly_without_a_device(tmp_path):
    """Compiles with -std=c99 -pedantic against include/impgpu.h alone; with no usable device the very first call
    (OnEnvStart) reports IMP_ERROR_DEVICE -- there is no CPU path behind the ABI."""
    import torch

    assert os.path.exists(build())
    env = {"HIP_VISIBLE_DEVICES": "-1", "ROCR_VISIBLE_DEVICES": "-1"} if torch.cuda.is_available() else {}
    p, fields, data = run(tmp_path, noise_image(8, 8, 3, 1), "/a.png?resize=4", "png", env=env)
    assert p.returncode == 3 and "impgpu_env_start" in p.stderr and data is None


@pytest.mark.gpu
@pytest.mark.parametrize("uri,ext,c", [
    ("/cat.jpg?crop=1,1,c,c&resize=120&filter-gotham=1", "jpg", 3),
    ("/a.png?resize=0,90&filter-rotate=90&filter-blur=1.5", "png", 4),
    ("/a.png?crop=16,9&filter-flip=10&format=jpg", "png", 4),          # 4-channel frame to a jpg encoder: flattened on white
    ("/a.jpg?resize=500,0,up&filter-modulate=30,150,100", "jpg", 3),     # enlargement: INTER_CUBIC
    ("/a.gif?resize=64&filter-gamma=1.7", "gif", 4),                    # GIF output: nearest-neighbour
])
def test_encoder_exit_matches_oracle(tmp_path, uri, ext, c):
    from test_gpu_chain import oracle_chain

    frame = smooth_image(240, 320, c)
    p, f, data = run(tmp_path, frame, uri, ext)
    assert p.returncode == 0, p.stderr
    rc_o, q = orc.parse_request(uri, ext, 5)
    rc_w, step_w, want = oracle_chain(frame, crop=q["crop"], gravity=q["gravity"], resize=q["resize"], simple=q["simple"],
                                      filters=q["filters"], flatten=q["need_flatten"])
    assert rc_o == 0 and f["code"] == rc_w == 0 and f["step"] == 8 and f["mime"] == q["mime"]
    assert (f["h"], f["w"], f["c"]) == want.shape
    assert np.array_equal(data.reshape(want.shape), want)


@pytest.mark.gpu
def test_watermark_json_and_text_exits_match_oracle(tmp_path):
    from test_gpu_chain import oracle_chain

    frame = noise_image(150, 200, 4, 31)
    overlay = noise_image(20, 30, 4, 32)
    overlay[:, :, 3] = np.linspace(0, 255, 30).astype(np.uint8)[None, :]
    pos = ("r", "b", 5, 7, 60)
    # configured watermark + encoder exit
    p, f, data = run(tmp_path, frame, "/a.png?resize=100", "png", overlay, pos)
    rc_w, step_w, want = oracle_chain(frame, resize="100", overlay=overlay, wm=pos)
    assert p.returncode == 0 and f["code"] == rc_w == 0 and np.array_equal(data.reshape(want.shape), want)
    # json exit: Info()'s integer percent (bridge.c:296) of the float-accumulator brightness
    p, f, data = run(tmp_path, frame, "/a.png?resize=100&format=json", "png")
    rc_w, step_w, want = oracle_chain(frame, resize="100")
    assert p.returncode == 0 and f["code"] == 0 and f["mime"] == -3
    assert f["brightness"] == int(round(orc.brightness(want) * 100))
    # text exit: ASCII() with quality= as the table name (bridge.c:670)
    for table in ("wide", "narrow"):
        p, f, data = run(tmp_path, frame, "/a.png?resize=60&format=text&quality=%s" % table, "png")
        rc_w, step_w, want = oracle_chain(frame, resize="60")
        text = orc.ascii_art(want, table)
        assert p.returncode == 0 and f["code"] == 0 and f["mime"] == -5 and f["bytes"] == len(text)
        assert bytes(data) == text


@pytest.mark.gpu
def test_error_codes_and_steps_reach_the_c_caller(tmp_path):
    frame = noise_image(60, 80, 3, 33)
    for uri, code, step in (("/a.jpg?crop=0,0,320,240", 50, 3), ("/a.jpg?resize=0,0", 50, 4), ("/a.jpg?filter-nope=1", 52, 5),
                            ("/a.jpg?resize=3000,10,up", 54, 4), ("/a.jpg", 50, 0), ("/a.ico?resize=5", 1, 0)):
        p, f, data = run(tmp_path, frame, uri, uri.split("?")[0].rsplit(".", 1)[1])
        assert p.returncode == 0 and (f["code"], f["step"]) == (code, step), uri


@pytest.mark.gpu
@pytest.mark.parametrize("fault,uri,step", [("2", "/a.png?resize=10", 2), ("3", "/a.png?crop=1,1", 3), ("4", "/a.png?crop=1,1&resize=10", 4),
                                            ("5", "/a.png?filter-gamma=2", 5), ("7", "/a.png?resize=10&format=json", 7),
                                            ("8", "/a.png?resize=10", 8)])
def test_injected_device_fault_reaches_the_c_caller_with_its_step(tmp_path, fault, uri, step):
    """impgpu_fault_arm(step) (asked for through the harness's HARNESS_FAULT): that stage behaves as if its HIP call had failed -> IMP_ERROR_DEVICE (90) and JobResult.Step,
    which BodyFilter turns into `Job failed at step %d with code %d` + HTTP 500 (module.c:305, :326-329)."""
    p, f, data = run(tmp_path, noise_image(40, 60, 4, 34), uri, "png", env={"HARNESS_FAULT": fault})
    assert p.returncode == 0 and (f["code"], f["step"]) == (90, step) and data is None


@pytest.mark.gpu
def test_a_worker_survives_an_injected_fault(tmp_path):
    """The n-th entry fails, the requests before and after it succeed on the same env: a failed request leaves the lane,
    its pool and the frame handle usable (the reference's `goto finalize` equivalent)."""
    import sys

    script = r'''
import ctypes as C, sys
import numpy as np
import torch  # first (see tests/conftest.py)
import ngx_http_imgproc_amd as imp
imp.env_start(0)
imp.lib.impgpu_fault_arm(4, 2)          # the second entry into IMP_STEP_RESIZE
cfg = imp.Config()
frame = np.random.default_rng(5).integers(0, 256, (50, 70, 3), dtype=np.uint8)
codes = []
for i in range(4):
    im = imp.Image(frame)
    rc, step = imp.run_ops(im, cfg, resize="20")
    codes.append((rc, step, imp.lib.impgpu_last_error().decode() if rc else "", im.shape))
    im.release()
imp.env_destroy()
print(repr(codes))
'''
    import os
    p = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env=dict(os.environ, IMPGPU_FAULT="4", PYTHONPATH=ROOT))      # (a stray variable of the old name arms nothing)
    assert p.returncode == 0, p.stderr[-2000:]
    codes = eval(p.stdout.strip().split("\n")[-1])
    assert [c[0] for c in codes] == [0, 90, 0, 0] and codes[1][1] == 4 and "injected fault" in codes[1][2]
    assert codes[0][3] == codes[2][3] == (14, 20, 3) and codes[1][3] == (50, 70, 3)      # the failed request's frame is intact


@pytest.mark.gpu
@pytest.mark.parametrize("c", [3, 4])
def test_mixed_batch_from_c(tmp_path, c):
    """tests/c/mixed_harness.c: impgpu_batch_resize_mixed called from C99 with frames of five different geometries (a
    general shrink, an exact 2x, an exact 4x, an enlargement, a wide shrink) -- the struct layout of impgpu_resize_item as a C
    compiler sees it, owned images' device pointers and pitches, NULL stream = the env stream; every output against the
    oracle under the interpolation Resize() picks (bridge.c:188-192)."""
    build()
    exe = os.path.join(ROOT, "tests", "c", "_build", "mixed_harness")
    geoms = [((90, 131), (57, 40)), ((64, 96), (48, 32)), ((80, 120), (30, 20)), ((33, 41), (90, 70)), ((70, 600), (123, 33))]
    cmd = [exe, str(c), "0", str(len(geoms))]
    frames = []
    for i, ((sh, sw), (dw, dh)) in enumerate(geoms):
        f = noise_image(sh, sw, c, 9100 + i)
        frames.append(f)
        src = tmp_path / ("f%d.raw" % i)
        f.tofile(src)
        cmd += [str(src), str(sw), str(sh), str(dw), str(dh), str(tmp_path / ("o%d.raw" % i))]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.strip() == "code=0", (p.stdout, p.stderr)
    for i, ((sh, sw), (dw, dh)) in enumerate(geoms):
        got = np.fromfile(tmp_path / ("o%d.raw" % i), dtype=np.uint8).reshape(dh, dw, c)
        interp = orc.INTER_CUBIC if (dw > sw or dh > sh) else orc.INTER_AREA
        assert np.array_equal(got, orc.cv_resize(frames[i], dw, dh, interp)), geoms[i]


@pytest.mark.gpu
@pytest.mark.parametrize("name,uri,ext", [
    ("c420_q90_dri4_95x51", "/a.jpg?resize=40", "jpg"),                               # the module's default quality, 86
    ("c444_q90_48x40", "/a.jpg?crop=1,1,c,c&filter-gamma=1.5&quality=70", "jpg"),
    ("gray_q90_57x43", "/a.jpg?resize=30,0&filter-rotate=90&quality=100", "jpg"),      # gray file: 3 channels after the operator segment
    ("c420_q92_opt_120x90", "/a.jpg?quality=0", "jpg"),
])
def test_jpeg_in_jpeg_out_from_c(tmp_path, name, uri, ext):
    """tests/c/jpeg_harness.c: decode on the device, the operator segment, encode on the device -- from C99; the file it
    writes is the oracle's (decode, chain, encode: each pinned or restated as DESIGN §2 says), byte for byte."""
    from test_gpu_chain import oracle_chain

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c")])
    src = os.path.join(ROOT, "tests", "golden", "jpeg", name + ".jpg")
    out = tmp_path / "out.jpg"
    p = subprocess.run([os.path.join(ROOT, "tests", "c", "_build", "jpeg_harness"), src, uri, ext, str(out)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, IMPGPU_JPEG_HUFF="device"))       # (small files: unset, their Huffman stage would run on the caller)
    assert p.returncode == 0, p.stderr
    f = {k: int(v) for k, v in (kv.split("=") for kv in p.stdout.split())}
    rc_o, q = orc.parse_request(uri, ext, 5)
    rc_d, frame = orc.jpeg_decode(open(src, "rb").read())
    rc_w, step_w, want = oracle_chain(frame, crop=q["crop"], gravity=q["gravity"], resize=q["resize"], simple=q["simple"],
                                      filters=q["filters"], flatten=q["need_flatten"])
    quality = int(q["quality"]) if q.get("quality") not in (None, "") else 86
    rc_e, file_want = orc.jpeg_encode(want, quality)
    assert rc_o == rc_d == rc_w == rc_e == 0 and f["code"] == 0 and f["step"] == 8
    assert (f["h"], f["w"], f["c"]) == want.shape
    assert out.read_bytes() == file_want


@pytest.mark.gpu
@pytest.mark.parametrize("ahead", [0, 1])
@pytest.mark.parametrize("quality", [0, 86])
def test_request_stream_from_c_threads(tmp_path, quality, ahead):
    """tests/c/stream_harness.c (what bench.py --stream --jpeg device --native starts): three C threads take JPEG files eight at
    a time through impgpu_batch_decode_jpeg -> impgpu_batch_resize_mixed -> impgpu_batch_encode_jpeg / impgpu_batch_download.
    The byte counts it reports are exact functions of the answers: they must be the oracle's.  ahead = 1: every thread begins
    its next batch (impgpu_batch_decode_jpeg_begin) before it finishes the current one (_finish)."""
    import json
    import struct

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c")])
    names = ["c420_q50_640x480", "c420_q90_67x45", "c444_q90_48x40", "c422_q85_49x37", "c420_q92_opt_120x90", "c420_q90_dri4_95x51", "c440_q80_dri2_patched_40x48"]
    blobs = [open(os.path.join(ROOT, "tests", "golden", "jpeg", n + ".jpg"), "rb").read() for n in names]
    pool = tmp_path / "pool.bin"
    with open(pool, "wb") as f:
        f.write(struct.pack("<I", len(blobs)))
        for b in blobs:
            f.write(struct.pack("<I", len(b)) + b)
    requests = 61
    per_file = []
    for b in blobs:
        rc, frame = orc.jpeg_decode(b)
        assert rc == 0
        rc, small = orc.resize(frame, "224,0")
        assert rc == 0
        if quality:
            rc, answer = orc.jpeg_encode(small, quality)
            assert rc == 0
            per_file.append(len(answer))
        else:
            per_file.append(((small.shape[1] * 3 + 3) & ~3) * small.shape[0])          # the rows as the device holds them (cvCreateImage's widthStep)
    p = subprocess.run([os.path.join(ROOT, "tests", "c", "_build", "stream_harness"), str(pool), str(requests), "3", "8", str(quality), "16", str(ahead)],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, IMPGPU_JPEG_HUFF="device"))
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["requests"] == requests and r["threads"] == 3 and r["batch"] == 8 and r["quality"] == quality and r["ahead"] == ahead
    assert r["file_bytes"] == sum(len(blobs[i % len(blobs)]) for i in range(requests))
    assert r["answer_bytes"] == sum(per_file[i % len(blobs)] for i in range(requests))


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["return", "exit", "busy"])
def test_a_worker_may_exit_with_its_env_alive(how):
    """A C worker that leaves without impgpu_env_destroy -- lanes of two threads, a frame never released, in `busy` even
    work still running on the device: impgpu_env_start's atexit hook returns everything before the HIP runtime's own exit
    handlers run.  The process must end by itself with status 0 (round 4: a child with a live env at exit once never came
    back; the library issued no HIP call at exit then)."""
    build()
    exe = os.path.join(ROOT, "tests", "c", "_build", "exit_harness")
    p = subprocess.run([exe, how], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr[-800:])
    assert p.stdout.strip().splitlines()[-1] == "leaving"
    assert "still busy at exit" not in p.stderr


@pytest.mark.gpu
def test_one_request_with_one_wait_from_c(tmp_path):
    """tests/c/latency_harness.c: the request an nginx worker runs -- JPEG in, resize=224,0, JPEG out -- in the library's two forms:
    two waits (verdict, answer) and ONE (the frame taken ahead of its verdict with impgpu_batch_decode_jpeg_pending, operators and
    the answer's encode enqueued behind the decode).  The harness compares the two files; here one of them is held to the oracle's."""
    import json

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c")])
    src = os.path.join(ROOT, "tests", "golden", "jpeg", "c420_q50_640x480.jpg")
    p = subprocess.run([os.path.join(ROOT, "tests", "c", "_build", "latency_harness"), src, "20"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, IMPGPU_JPEG_HUFF="device"))
    assert p.returncode == 0, p.stderr
    r = json.loads(p.stdout.strip().splitlines()[-1])
    print("\n" + json.dumps(r))
    rc, frame = orc.jpeg_decode(open(src, "rb").read())
    rc2, small = orc.resize(frame, "224,0")
    rc3, want = orc.jpeg_encode(small, 86)
    assert rc == rc2 == rc3 == 0
    assert r["same_answer"] and r["answer_bytes"] == len(want)
