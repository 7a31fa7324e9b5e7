"""CPU side of the broker (include/impgpu_broker.h): the client fails loudly and fast when nobody serves, the broker fails
loudly without a device (no CPU path behind it either), the records are one page each."""
import ctypes as C
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def built():
    subprocess.check_call([sys.executable, os.path.join(ROOT, "ngx_http_imgproc_amd", "build.py")], stdout=subprocess.DEVNULL)
    from ngx_http_imgproc_amd import broker as B
    return B


def test_client_library_exports_the_declared_symbols(built):
    hdr = open(os.path.join(ROOT, "include", "impgpu_broker.h")).read()
    import re

    names = set(re.findall(r"\b(impgpu_client_[a-z_]+)\s*\(", hdr))
    assert len(names) >= 7
    for n in names:
        assert hasattr(built.clib, n), n
    out = subprocess.run(["ldd", built.CLIENT_LIB_PATH], capture_output=True, text=True).stdout
    assert "amdhip" not in out and "stdc++" not in out          # workers: plain C, no HIP, no C++ runtime


def test_attach_without_a_broker_fails_at_once(built):
    with pytest.raises(RuntimeError) as e:
        built.Client("/impgpu-nobody-%d" % os.getpid())
    assert "no broker segment" in str(e.value)


def test_header_is_c99_and_records_are_pages(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include <impgpu_broker.h>\n#include <stdio.h>\nint main(void){printf("%zu %zu\\n", sizeof(impb_header), sizeof(impb_slot));return 0;}\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert subprocess.check_output([str(exe)], text=True).split() == ["4096", "4096"]


def test_broker_fails_loudly_without_a_device(built):
    import torch

    env = dict(os.environ)
    if torch.cuda.is_available():
        env.update({"HIP_VISIBLE_DEVICES": "-1", "ROCR_VISIBLE_DEVICES": "-1"})
    name = "/impgpu-nodev-%d" % os.getpid()
    p = subprocess.run([built.BROKER_PATH, "--name", name, "--slots", "2", "--slot-mb", "1"], capture_output=True, text=True, timeout=120, env=env)
    try:
        assert p.returncode == 4 and "impgpu_env_start" in p.stderr
        # nobody was ever told the segment is served
        with pytest.raises(RuntimeError) as e:
            built.Client(name)
        assert "no live broker" in str(e.value) or "no broker segment" in str(e.value)
    finally:
        try:
            os.unlink("/dev/shm" + name)
        except OSError:
            pass


def test_broker_rejects_bad_options(built):
    p = subprocess.run([built.BROKER_PATH, "--slots", "100000"], capture_output=True, text=True, timeout=30)
    assert p.returncode == 2
    p = subprocess.run([built.BROKER_PATH, "--name", "no-slash"], capture_output=True, text=True, timeout=30)
    assert p.returncode == 2


# ---- the worker side (glue/imp_gpu_client.c) against tests/c/mock_broker.c: the protocol without a GPU
@pytest.fixture()
def mock(built, tmp_path):
    """start(mode) -> (name, Popen) of a mock broker serving a fresh segment; everything started is stopped afterwards"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c"), os.path.join(ROOT, "tests", "c", "_build", "mock_broker"),
                           os.path.join(ROOT, "tests", "c", "_build", "client_asan")])
    started = []

    def start(mode="serve", slots=4, slot_kb=256, name=None):
        name = name or "/impgpu-mock-%d-%d" % (os.getpid(), len(started))
        p = subprocess.Popen([os.path.join(ROOT, "tests", "c", "_build", "mock_broker"), name, str(slots), str(slot_kb), mode],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert p.stdout.readline().strip() == "ready"
        started.append((name, p))
        return name, p

    yield start
    for name, p in started:
        if p.poll() is None:
            p.kill()
        p.wait()
        try:
            os.unlink("/dev/shm" + name)
        except OSError:
            pass


def test_round_trips_answers_and_operator_errors(built, mock):
    name, _ = mock()
    c = built.Client(name)
    rc, code, step, got, a = c.run(blob=b"0123456789" * 100, crop="1,1", resize="224,0", filters=["gamma=2", "blur=1"])
    assert (rc, code) == (0, 0) and got == (b"0123456789" * 100)[::-1]
    assert (a.width, a.channels) == (2, 3)              # the mock echoes the filter count and the crop string's length: the job crossed intact
    rc, code, step, got, a = c.run(blob=b"x" * 10, quality=1051)         # the mock answers "code 51 at step RESIZE"
    assert (rc, code, step) == (0, 51, 4)
    wid = c.prepare_watermark(__import__("numpy").zeros((4, 4, 4), "uint8"))
    rc, code, step, got, a = c.run(blob=b"abc", watermark_id=wid)
    assert (rc, code) == (0, 0) and a.height >= 1       # ... and the id the broker gave for the overlay
    # an input larger than a slot is refused on the worker's side, nothing is sent
    rc, code, step, got, a = c.run(blob=b"z" * (300 << 10))
    assert rc == 2
    st = c.stats()
    assert st["served"] == 4 and st["epoch"] == 1
    c.close()


def test_more_workers_than_slots_is_said_so(built, mock):
    name, _ = mock(slots=2)
    a, b = built.Client(name), built.Client(name)
    with pytest.raises(RuntimeError) as e:
        built.Client(name)
    assert "every slot" in str(e.value)
    a.close()
    c = built.Client(name)                              # a detached worker's slot is free again
    assert c.run(blob=b"ok")[3] == b"ko"
    b.close(); c.close()


def test_a_broker_that_is_killed_fails_the_request_within_ticks(built, mock, monkeypatch):
    """SIGKILL while the request is held: the worker notices at its next 50-ms tick (IMP_ERROR_DEVICE), long before the
    10-s timeout; later requests fail at once; a new broker under the same name is found by the next request."""
    import time
    name, p = mock("mute")
    c = built.Client(name)
    import threading
    out = {}

    def ask():
        t0 = time.time()
        out["r"] = c.run(blob=b"hello")
        out["dt"] = time.time() - t0
        out["why"] = built.Client.last_error()            # (the text is the calling thread's)
    t = threading.Thread(target=ask)
    t.start()
    time.sleep(0.3)
    p.kill(); p.wait()
    t.join(timeout=5)
    assert not t.is_alive() and out["r"][0] == 90 and out["dt"] < 2.0, out
    assert "went away" in out["why"]
    t0 = time.time()
    assert c.run(blob=b"again")[0] == 90 and time.time() - t0 < 0.5
    mock(name=name)                                      # a fresh broker, a fresh segment under the old name
    rc, code, step, got, a = c.run(blob=b"abc")
    assert (rc, code) == (0, 0) and got == b"cba"
    c.close()


def test_a_broker_that_never_answers_runs_into_the_timeout_and_the_slot_is_abandoned(built, mock, monkeypatch):
    monkeypatch.setenv("IMPGPU_BROKER_TIMEOUT_MS", "300")
    name, p = mock("mute", slots=2)
    c = built.Client(name)
    import time
    t0 = time.time()
    rc = c.run(blob=b"hello")[0]
    assert rc == 90 and 0.25 < time.time() - t0 < 2.0
    assert "did not answer in time" in built.Client.last_error()
    assert c.run(blob=b"second")[0] == 90                 # it claimed the other slot, which times out as well
    with pytest.raises(RuntimeError):
        built.Client(name)                                # both slots are held by requests the broker still owns
    c.close()


def test_client_under_the_sanitizers_and_across_fork(built, mock):
    """glue/imp_gpu_client.c built with AddressSanitizer + UBSan: 300 requests of varying size checked byte by byte, then three
    fork()ed copies of the attached process, each of which must claim a slot of its own and not touch the parent's."""
    name, _ = mock(slots=6)
    p = subprocess.run([os.path.join(ROOT, "tests", "c", "_build", "client_asan"), name, "300", "3"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.strip() == "ok 600", (p.returncode, p.stdout, p.stderr[-1500:])


# ---- impgpu_jpeg_unstuff: what a worker does to a JPEG on its way into the slot
def _py_unstuff(blob):
    """An independent restatement (ITU T.81 B.1.1.5 byte stuffing; the rules of csrc/imp_jpeg.cpp jpeg_prepare_scan for a
    file without restart intervals): (head, scan) or None."""
    b = bytes(blob)
    if b[:2] != b"\xff\xd8":
        return None
    at, frames, scan_begin = 2, 0, None
    while scan_begin is None:
        if at + 2 > len(b) or b[at] != 0xFF:
            return None
        while at < len(b) and b[at] == 0xFF:
            at += 1
        if at >= len(b):
            return None
        m = b[at]
        at += 1
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        if m == 0xD9 or at + 2 > len(b):
            return None
        ln = (b[at] << 8) | b[at + 1]
        if ln < 2 or ln > len(b) - at:
            return None
        if m in (0xC0, 0xC1):
            frames += 1
            if frames > 1:
                return None
        elif 0xC0 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            return None
        elif m == 0xDD and (ln != 4 or b[at + 2] or b[at + 3]):
            return None
        elif m == 0xDA:
            if not frames:
                return None
            scan_begin = at + ln
        at += ln
    if len(b) - scan_begin < 40960:
        return None
    out = bytearray()
    i = scan_begin
    while i < len(b):
        if b[i] != 0xFF:
            out.append(b[i])
            i += 1
            continue
        j = i + 1
        while j < len(b) and b[j] == 0xFF:
            j += 1
        if j >= len(b):
            break
        if b[j] == 0 and j == i + 1:
            out.append(0xFF)
            i = j + 1
        elif b[j] == 0 or 0xD0 <= b[j] <= 0xD7:
            return None
        else:
            break
    return (b[:scan_begin], bytes(out)) if out else None


def _photo_jpeg(w, h, quality=90, **kw):
    import io

    from PIL import Image
    from ngx_http_imgproc_amd.workloads import photo_like

    f = io.BytesIO()
    Image.fromarray(photo_like(h, w, seed=w + h)).save(f, "JPEG", quality=quality, **kw)
    return f.getvalue()


def test_unstuff_agrees_with_its_restatement(built):
    import ngx_http_imgproc_amd as imp

    files = [_photo_jpeg(640, 480), _photo_jpeg(1280, 720, 95, subsampling="4:4:4"), _photo_jpeg(800, 600, 85, optimize=True)]
    for f in files:
        assert f.count(b"\xff\x00") > 10                           # stuffed bytes do occur
        got, want = imp.jpeg_unstuff(f), _py_unstuff(f)
        assert want is not None and got == want
        head, scan = got
        assert f.startswith(head) and len(scan) < len(f) - len(head)
    # the files that go as they are: small scans, restart intervals, progressive, not a JPEG, damaged marker sequences
    small = _photo_jpeg(96, 64)
    assert imp.jpeg_unstuff(small) is None and _py_unstuff(small) is None
    assert imp.jpeg_unstuff(_photo_jpeg(640, 480, restart_marker_blocks=8)) is None
    assert imp.jpeg_unstuff(_photo_jpeg(640, 480, progressive=True)) is None
    assert imp.jpeg_unstuff(b"\x89PNG\r\n\x1a\n" + bytes(60000)) is None
    f = bytearray(files[0])
    head, scan = imp.jpeg_unstuff(bytes(f))
    at = len(head) + 5000
    for bad in (b"\xff\xd3", b"\xff\xff\x00"):                      # RSTn without an interval; FF FF 00
        g = bytes(f[:at]) + bad + bytes(f[at:])
        assert imp.jpeg_unstuff(g) is None and _py_unstuff(g) is None
    # fill bytes in front of the closing marker, a file cut off in its scan, a trailing FF: taken, the same bytes
    for g in (bytes(f[:-2]) + b"\xff\xff\xff\xd9", bytes(f[:len(f) - 3000]), bytes(f[:len(f) - 3000]) + b"\xff"):
        assert imp.jpeg_unstuff(g) == _py_unstuff(g) and imp.jpeg_unstuff(g) is not None
    # a marker other than EOI ends the scan where it stands
    g = bytes(f[:at]) + b"\xff\xda" + bytes(f[at:])
    assert len(imp.jpeg_unstuff(g)[1]) < len(scan) and imp.jpeg_unstuff(g) == _py_unstuff(g)


def test_unstuff_respects_its_capacity(built):
    f = _photo_jpeg(640, 480)
    out = (C.c_uint8 * (len(f) + 2048))()
    v = [C.c_size_t() for _ in range(4)]
    args = [C.byref(x) for x in v]
    assert built.clib.impgpu_jpeg_unstuff(f, len(f), out, len(f) + 2048, *args) == 1
    head, at, n, total = (x.value for x in v)
    assert at % 256 == 0 and at >= head and total == at + n + 1024 and bytes(out[at + n:total]) == b"\xff" * 1024
    assert built.clib.impgpu_jpeg_unstuff(f, len(f), out, len(f), *args) == 0          # no room for the tail: as it is


def test_unstuff_on_damaged_files_under_the_sanitizers(built, tmp_path):
    """What a worker runs over every request body before anything has looked at it: 6000 damaged copies of two files (cut
    short, bytes flipped in scan and headers, marker pairs and FF runs dropped in, segment lengths spoiled) and output buffers
    of exactly the capacity claimed, under AddressSanitizer / UBSan; every result of "taken" is checked against a byte-by-byte
    unstuffing (tests/c/unstuff_fuzz.c)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c"), os.path.join(ROOT, "tests", "c", "_build", "unstuff_fuzz")])
    exe = os.path.join(ROOT, "tests", "c", "_build", "unstuff_fuzz")
    for k, blob in enumerate((_photo_jpeg(640, 480), _photo_jpeg(512, 384, 95, subsampling="4:4:4", optimize=True))):
        path = tmp_path / ("f%d.jpg" % k)
        path.write_bytes(blob)
        r = subprocess.run([exe, str(path), "3000", str(k + 1)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout[-300:], r.stderr[-1500:])
        taken = int(r.stdout.split()[1])
        assert 300 < taken < 3000, r.stdout                       # both verdicts occur
