"""The broker (include/impgpu_broker.h): N worker PROCESSES -- IMP's own model, docs/02 - Configuration.md:18 worker_processes,
RunJob synchronous (module.c:298) -- hand their one request at a time to the ONE process that owns the GPU, which runs
whatever is queued as a batch.  Every answer must be the bytes the oracle produces for that request alone: batching is
invisible to the worker."""
import ctypes as C
import io
import json
import os
import signal
import subprocess
import sys
import time

import numpy as np
import pytest

import oracle_lib as orc
from conftest import ROOT, noise_image, smooth_image
from test_gpu_chain import oracle_chain

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _photo(h, w, seed):
    from ngx_http_imgproc_amd.workloads import photo_like
    return photo_like(h, w, seed)[:, :, ::-1].copy()          # B,G,R


@pytest.fixture(scope="module")
def scaling():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c")])
    subprocess.check_call([sys.executable, os.path.join(ROOT, "ngx_http_imgproc_amd", "build.py")], stdout=subprocess.DEVNULL)
    import worker_scaling
    return worker_scaling


@pytest.fixture(params=["one batch at a time", "the next batch unpacked behind the answers"])
def broker(scaling, request):
    name = "/impgpu-test-%d" % os.getpid()
    p = scaling.start_broker(name, threads=2, gather_us=0, slots=16,
                             extra=["--slot-mb", "24", "--pipeline", "0" if request.param == "one batch at a time" else "1"])
    yield name, p
    err = scaling.stop_broker(p)
    assert p.returncode == 0, err[-800:]
    assert not os.path.exists("/dev/shm" + name)              # a clean stop leaves no segment behind


def _pool_with_answers(tmp_path, scaling):
    sizes = [(480, 640, 1), (720, 1280, 2), (1080, 1920, 3), (600, 800, 4), (1200, 1600, 5), (300, 256, 6), (2160, 3840, 7)]
    blobs, answers = [], []
    for h, w, seed in sizes:
        rc, blob = orc.jpeg_encode(_photo(h, w, seed), 90)
        assert rc == 0
        rc, frame = orc.jpeg_decode(blob)
        assert rc == 0
        rc, small = orc.resize(frame, "224,0")
        assert rc == 0
        rc, answer = orc.jpeg_encode(small, 86)
        assert rc == 0
        blobs.append(blob)
        answers.append(answer)
    pool, want = str(tmp_path / "pool.bin"), str(tmp_path / "answers.bin")
    scaling.write_pool(pool, blobs)
    scaling.write_pool(want, answers)
    return pool, want


def test_workers_through_the_broker_get_the_oracles_files(tmp_path, scaling, broker):
    """1, then 8 worker processes (tests/c/worker_harness.c, the client compiled in as in nginx), JPEG in -> resize=224,0 ->
    JPEG out: every answer -- rode alone or with seven others in one launch -- equals the file the oracle writes."""
    name, _ = broker
    pool, want = _pool_with_answers(tmp_path, scaling)
    one = scaling.run_point(pool, "broker", 1, 1.0, want, name)
    many = scaling.run_point(pool, "broker", 8, 1.5, want, name)
    print("\n" + json.dumps(one) + "\n" + json.dumps(many))
    for r in (one, many):
        assert r["checked"] and r["mismatches"] == 0 and r["requests"] > 0, r
    assert one["mean_batch"] == 1.0
    assert many["mean_batch"] > 1.5, many                      # requests of different workers did share launches
    assert many["requests_per_s"] > 1.5 * one["requests_per_s"], (one, many)


def test_sixteen_workers_for_a_while_every_answer_checked(tmp_path, scaling, broker):
    """A soak: sixteen workers over the broker's lanes, every answer compared with the oracle's file, for IMPGPU_SOAK_SECONDS
    (4 in the suite; profiles/r05_broker_soak.txt is a minute of it).  No mismatch, no refusal, no chain that timed out."""
    name, _ = broker
    pool, want = _pool_with_answers(tmp_path, scaling)
    seconds = float(os.environ.get("IMPGPU_SOAK_SECONDS", "4"))
    r = scaling.run_point(pool, "broker", 16, seconds, want, name)
    print("\n" + json.dumps(r))
    assert r["checked"] and r["mismatches"] == 0 and r["chain_timeouts"] == 0 and r["refused"] == 0, r
    assert r["requests"] > 1000 * seconds, r


def test_direct_workers_for_comparison(tmp_path, scaling):
    """The same worker linked against libimpgpu.so itself (a device context per worker): same files."""
    pool, want = _pool_with_answers(tmp_path, scaling)
    r = scaling.run_point(pool, "direct", 2, 1.0, want)
    print("\n" + json.dumps(r))
    assert r["checked"] and r["mismatches"] == 0 and r["chain_timeouts"] == 0 and r["refused"] == 0, r


def _client(name):
    from ngx_http_imgproc_amd import broker as B
    return B, B.Client(name)


def test_every_kind_of_request_and_answer(broker):
    """Operator chains the batch path does not take (crop, filters, watermark, flatten, gray) go through impgpu_run_ops in
    the broker; frames decoded on the host come in as pixels; answers as JPEG, as pixels for a host encoder, as Info."""
    name, _ = broker
    B, c = _client(name)
    from ngx_http_imgproc_amd._lib import CConfig

    bgr = _photo(480, 640, 11)
    rc, blob = orc.jpeg_encode(bgr, 90)
    rc, frame = orc.jpeg_decode(blob)
    # crop + resize + filters, pixels back
    exp = CConfig(2000, 2000, 5, 1, 0, b"l", b"t", 0, 0, None)               # AllowExperiments: gotham is one
    rc, code, step, got, a = c.run(blob=blob, crop="16,9", resize="320,0", filters=["gotham=1", "rotate=90"], config=exp, out=B.OUT_FRAME)
    rc_o, _, want = oracle_chain(frame, crop="16,9", resize="320,0", filters=["gotham=1", "rotate=90"])
    assert (rc, code, rc_o) == (0, 0, 0) and np.array_equal(got, want)
    # a BGRA frame decoded on the host (PNG fallback decoders): flatten for a JPEG encoder, answer as a JPEG file
    rgba = noise_image(200, 300, 4, 3)
    rc, code, step, got, a = c.run(frame=rgba, resize="150,100", need_flatten=1, out=B.OUT_JPEG, quality=77)
    rc_o, _, want = oracle_chain(rgba, resize="150,100", flatten=1)
    rc_e, want_file = orc.jpeg_encode(want, 77)
    assert (rc, code, rc_o, rc_e) == (0, 0, 0, 0) and got == want_file
    # gray file: gray -> BGR before the filters
    gray = smooth_image(120, 160, 1)
    rc, gblob = orc.jpeg_encode(gray, 85)
    rc, gframe = orc.jpeg_decode(gblob)
    rc, code, step, got, a = c.run(blob=gblob, resize="80,60", filters=["gamma=1.5"], out=B.OUT_FRAME)
    rc_o, _, want = oracle_chain(gframe, resize="80,60", filters=["gamma=1.5"])
    assert (rc, code, rc_o) == (0, 0, 0) and np.array_equal(got, want)
    # watermark of a location, registered once
    ov = noise_image(24, 40, 4, 9)
    wid = c.prepare_watermark(ov)
    cfg = CConfig(2000, 2000, 5, 0, 60, b"r", b"b", 4, 6, None)
    rc, code, step, got, a = c.run(blob=blob, resize="224,0", config=cfg, watermark_id=wid, out=B.OUT_FRAME)
    rc_o, _, want = oracle_chain(frame, resize="224,0", overlay=ov, wm=("r", "b", 4, 6, 60))
    assert (rc, code, rc_o) == (0, 0, 0) and np.array_equal(got, want)
    # Info exit
    rc, code, step, got, a = c.run(blob=blob, resize="100,0", out=B.OUT_INFO)
    rc_o, small = orc.resize(frame, "100,0")
    assert (rc, code) == (0, 0) and (a.width, a.height) == (small.shape[1], small.shape[0])
    assert abs(a.brightness - orc.brightness(small)) < 1e-6
    # the text exit
    rc, code, step, got, a = c.run(blob=blob, resize="60,0", out=B.OUT_ASCII, ascii_args="")
    rc_o, small = orc.resize(frame, "60,0")
    assert (rc, code) == (0, 0) and got == orc.ascii_art(small, "")
    # errors keep their code and step (bridge.c's JobResult)
    rc, code, step, got, a = c.run(blob=blob, resize="5000,0,up", out=B.OUT_JPEG)
    rc_o, _ = orc.resize(frame, "5000,0,up")
    assert rc == 0 and code == rc_o != 0 and step == 4
    rc, code, step, got, a = c.run(blob=blob, filters=["nosuch=1"], out=B.OUT_JPEG)
    assert rc == 0 and code == 52 and step == 5
    rc, code, step, got, a = c.run(blob=blob, filters=["gotham=1"], out=B.OUT_JPEG)       # experiments off: no such filter
    assert rc == 0 and code == 52 and step == 5
    c.close()


def test_files_the_device_does_not_decode_come_back_not_taken(broker):
    from PIL import Image

    name, _ = broker
    B, c = _client(name)
    b = io.BytesIO()
    Image.fromarray(_photo(64, 64, 1)[:, :, ::-1]).save(b, "JPEG", quality=90, progressive=True)
    rc, code, step, got, a = c.run(blob=b.getvalue(), resize="32,0")
    assert rc == 0 and code == B.NOT_TAKEN and step == 2      # the worker decodes on the host and comes back with pixels
    rc, code, step, got, a = c.run(blob=b"GIF89a" + bytes(64), resize="32,0")
    assert rc == 0 and code == B.NOT_TAKEN
    rc, code, step, got, a = c.run(blob=b"\xff\xd8\xff" + bytes(200), resize="32,0")     # damaged
    assert rc == 0 and code == B.NOT_TAKEN
    c.close()


def test_a_broker_that_dies_is_replaced_and_workers_carry_on(tmp_path, scaling):
    """--supervise: the parent never touches the GPU; the serving child is killed with SIGKILL between requests.  The
    worker's next request either fails fast with IMP_ERROR_DEVICE (never hangs) or is already served by the fresh child;
    within seconds requests succeed again -- the watermark included, which the client registers again by itself."""
    name = "/impgpu-test-sup-%d" % os.getpid()
    p = scaling.start_broker(name, threads=1, gather_us=0, slots=8, extra=["--slot-mb", "8", "--supervise"])
    try:
        B, c = _client(name)
        from ngx_http_imgproc_amd._lib import CConfig

        bgr = _photo(240, 320, 2)
        rc, blob = orc.jpeg_encode(bgr, 90)
        rc, frame = orc.jpeg_decode(blob)
        ov = noise_image(16, 16, 4, 1)
        wid = c.prepare_watermark(ov)
        cfg = CConfig(2000, 2000, 5, 0, 80, b"l", b"t", 1, 1, None)
        rc_o, _, want = oracle_chain(frame, resize="160,0", overlay=ov, wm=("l", "t", 1, 1, 80))
        rc, code, step, got, a = c.run(blob=blob, resize="160,0", config=cfg, watermark_id=wid, out=B.OUT_FRAME)
        assert (rc, code) == (0, 0) and np.array_equal(got, want)
        first = c.stats()
        assert first["broker_pid"] != p.pid                    # the child serves, not the supervisor
        os.kill(first["broker_pid"], signal.SIGKILL)
        t0 = time.time()
        ok = False
        failures = 0
        while time.time() - t0 < 120:
            t1 = time.time()
            rc, code, step, got, a = c.run(blob=blob, resize="160,0", config=cfg, watermark_id=wid, out=B.OUT_FRAME)
            assert time.time() - t1 < 15                       # a dead broker is noticed within ticks, not after the long timeout
            if rc == 0 and code == 0:
                ok = np.array_equal(got, want)
                break
            assert rc == 90, (rc, code, B.Client.last_error())
            failures += 1
            time.sleep(0.05)
        assert ok, (failures, B.Client.last_error())
        second = c.stats()
        assert second["broker_pid"] not in (first["broker_pid"], p.pid) and second["epoch"] != first["epoch"]
        c.close()
    finally:
        err = scaling.stop_broker(p)
    assert "starting a fresh one" in err


def test_a_worker_that_writes_its_slot_by_hand_gets_errors_not_memory(tmp_path, scaling, broker):
    """tests/c/rogue_worker.c: request records a correct client never writes -- a prepared file's scan unaligned, reaching past
    the bytes handed over, lying outside the slot, overlapped by its head, counts near 2^64, in_bytes past the slot; a head cut
    short of its scan -- each answered with IMP_ERROR_INVALID_ARGS at its step (validation, or the decoder's own check), and
    the same file handed over correctly right afterwards gets the oracle's answer: the broker is still there and still right."""
    name, p = broker
    blob = orc.jpeg_encode(_photo(600, 800, 21), 90)[1]
    src = tmp_path / "in.jpg"
    src.write_bytes(blob)
    r = subprocess.run([os.path.join(ROOT, "tests", "c", "_build", "rogue_worker"), name, str(src)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    got = {}
    for line in r.stdout.strip().splitlines():
        w = line.split()
        got[w[0]] = dict(rc=int(w[2]), code=int(w[4]), step=int(w[6]), bytes=int(w[8]))
    print("\n" + r.stdout)
    INVALID = 50                                                    # IMP_ERROR_INVALID_ARGS
    for case in ("unaligned", "overlong", "outside", "overlap", "toolong", "huge"):
        assert got[case]["rc"] == 0 and got[case]["code"] == INVALID and got[case]["bytes"] == 0, (case, got[case])
    assert got["shorthead"]["rc"] == 0 and got["shorthead"]["code"] != 0 and got["shorthead"]["bytes"] == 0
    rc, frame = orc.jpeg_decode(blob)
    rc2, small = orc.resize(frame, "224,0")
    rc3, want = orc.jpeg_encode(small, 86)
    assert rc == rc2 == rc3 == 0
    assert got["correct"] == dict(rc=0, code=0, step=got["correct"]["step"], bytes=len(want))
    assert p.poll() is None                                         # the broker did not go away
