"""The host-side grammar under AddressSanitizer + UBSan (SURVEY 5 "race detection / sanitizers").

imp_args.cpp / imp_request.cpp / imp_tables.cpp parse attacker-controlled query strings before anything reaches the
GPU; the reference's own parser over-runs on some of them (RewindArgs without its separator, helpers.c:18-23; strtok on
NULL in Scanline, filters.c:408-409).  tests/c/fuzz_host.cpp links those three sources built with
`g++ -fsanitize=address,undefined -fno-sanitize-recover` and replays fuzzed cases; every answer must also equal what the
regular (unsanitized, hipcc-built) library returns through the C ABI, so nothing depends on undefined behaviour.
CPU only: sanitizers are not available on the GPU box.
"""
import ctypes as C
import os
import shutil
import subprocess

import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import ngx_http_imgproc_amd as imp
from conftest import ROOT

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ with libasan / libubsan")
DRIVER = os.path.join(ROOT, "tests", "c", "_build", "fuzz_host_asan")


def hx(s):
    if s is None:
        return "-"
    b = s if isinstance(s, bytes) else s.encode("utf-8", "surrogatepass")
    return b.hex() if b else "="


def drive(lines):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c"), DRIVER])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([DRIVER], input="\n".join(lines) + "\n", capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-3000:]
    out = p.stdout.strip().split("\n")
    assert len(out) == len(lines)
    return out


# grammar-shaped fragments so the fuzz spends its time near the parsers' decisions, plus raw text
TOKENS = ["", "0", "1", "16", "9", "320px", "-5px", "px", "l", "r", "c", "t", "b", "up", ",", ",,", "4294967296", "-1", "1e9", "0x10",
          "99999999999999999999", "ff8000", "zzzzzz", "0.5", "nan", "inf", "-inf", "full", "pale", "=", "&", "?", "%", "%3", "%zz", "\x7f", "é"]
frag = st.one_of(st.sampled_from(TOKENS), st.text(alphabet=st.characters(blacklist_characters="\x00\n\r", blacklist_categories=("Cs",)), max_size=6))
args_text = st.lists(frag, max_size=6).map(",".join)
FILTER_NAMES = ["flip", "rotate", "modulate", "colorize", "blur", "gamma", "contrast", "gradmap", "vignette", "gotham", "lomo",
                "kelvin", "rainbow", "scanline", "cartoon", "", "blurry", "FLIP"]
filter_req = st.one_of(st.builds(lambda n, a: n + "=" + a, st.sampled_from(FILTER_NAMES), args_text),
                       st.builds(lambda n, a, b: n + "=" + a + "=" + b, st.sampled_from(FILTER_NAMES), args_text, args_text),
                       st.sampled_from(FILTER_NAMES), args_text)
KEYS = ["crop", "gravity", "resize", "quality", "format", "page", "filter-", "filter", "cropx", "unknown", ""]
FORMATS = ["jpg", "png", "json", "text", "gif", "webp", "jp2", "bmp", "ico", "tiff", "JPG", "x.y.jpeg", ""]
param = st.one_of(st.builds(lambda k, v: k + "=" + v, st.sampled_from(KEYS), st.one_of(args_text, filter_req, st.sampled_from(FORMATS))),
                  st.builds(lambda k, v: k + v, st.sampled_from(KEYS), filter_req), st.sampled_from(KEYS))
uri = st.builds(lambda path, q, ps: path + q + "&".join(ps), st.sampled_from(["/a.jpg", "", "/a%20b.png", "?", "/x?y"]),
                st.sampled_from(["?", "", "??", "?&"]), st.lists(param, max_size=9))
dims = st.integers(min_value=1, max_value=5000)


@settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(st.lists(st.tuples(dims, dims, args_text, st.one_of(st.none(), args_text)), min_size=20, max_size=40))
def test_crop_grammar_under_sanitizers(cases):
    out = drive(["crop %d %d %s %s" % (w, h, hx(a), hx(g)) for w, h, a, g in cases])
    for (w, h, a, g), line in zip(cases, out):
        x, y, ow, oh = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rc = imp.lib.impgpu_crop_geometry(w, h, a.encode(), None if g is None else g.encode(), x, y, ow, oh)
        want = [rc] + ([x.value, y.value, ow.value, oh.value] if rc == 0 else [0, 0, 0, 0])
        assert [int(v) for v in line.split()] == want, (w, h, a, g)
        if rc == 0:
            assert 0 <= x.value and 0 <= y.value and x.value + ow.value <= w and y.value + oh.value <= h and ow.value > 0 and oh.value > 0


@settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(st.lists(st.tuples(dims, dims, args_text, st.sampled_from([0, 1, 2000, 100000]), st.sampled_from([0, 1, 2000]), st.booleans()),
                min_size=20, max_size=40))
def test_resize_grammar_under_sanitizers(cases):
    out = drive(["resize %d %d %s %d %d %d" % (w, h, hx(a), mw, mh, int(s)) for w, h, a, mw, mh, s in cases])
    for (w, h, a, mw, mh, s), line in zip(cases, out):
        cfg = imp.Config(max_w=mw, max_h=mh)
        ow, oh, ip = C.c_int(), C.c_int(), C.c_int()
        rc = imp.lib.impgpu_resize_geometry(w, h, a.encode(), C.byref(cfg.c), int(s), ow, oh, ip)
        want = [rc] + ([ow.value, oh.value, ip.value] if rc == 0 else [0, 0, 0])
        assert [int(v) for v in line.split()] == want, (w, h, a)


@settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(st.lists(st.tuples(filter_req, st.booleans()), min_size=20, max_size=40))
def test_filter_grammar_under_sanitizers(cases):
    out = drive(["filter %s %d 4 64 64" % (hx(r), int(al)) for r, al in cases] + ["destructive %s" % hx(r) for r, _ in cases])
    for i, (r, al) in enumerate(cases):
        assert int(out[i].split()[0]) == imp.lib.impgpu_filter_check(r.encode(), int(al)), r
        assert int(out[len(cases) + i]) == imp.lib.impgpu_check_destructive(r.encode()), r


@settings(max_examples=200, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(st.lists(st.tuples(uri, st.one_of(st.none(), st.sampled_from(FORMATS)), st.sampled_from([0, 1, 5, 9])), min_size=20, max_size=40))
def test_request_parser_under_sanitizers(cases):
    out = drive(["request %s %s %d" % (hx(u), hx(e), mf) for u, e, mf in cases])
    for (u, e, mf), line in zip(cases, out):
        r = imp.Request(u, e, imp.Config(max_filters=mf)) if e is not None else None
        if r is None:
            continue
        got = [int(v) for v in line.split()]
        assert got[0] == r.code, (u, e)
        if r.code == 0:
            assert got[1:7] == [r.mime, r.page, int(r.simple), int(r.need_flatten), len(r.filters), int(r.destructive)], (u, e)


def test_table_builders_under_sanitizers():
    """Every (source, destination) size pair up to 40 plus the benchmark geometries: run tables stay inside the row."""
    lines = []
    for s in range(1, 41):
        for d in range(1, 41):
            lines += ["taps %d %d %d 1" % (s, d, m) for m in (1, 2, 4)]
            if d <= s:
                lines.append("area %d %d" % (s, d))
    lines += ["area 1920 224", "area 1080 224", "area 3840 224", "taps 3840 1920 4 1", "taps 1920 224 2 1", "taps 480 1920 2 1"]
    lines += ["gauss %s" % v for v in ("0", "-1", "0.01", "0.1", "0.5", "2", "5.5", "25", "600", "1e9", "nan", "inf")]
    out = drive(lines)
    assert not any("out-of-range" in o for o in out)
    for line, o in zip(lines, out):
        if line.startswith("area"):
            s, d = map(int, line.split()[1:])
            assert abs(float(o.split()[0]) - d) < 1e-2 * d      # every run's weights sum to 1


def test_jpeg_front_under_sanitizers():
    """The marker parser, the table builder, the scan preparation, the sequential entropy decoder and the lane-by-lane
    model of the device's entropy stage on damaged files: truncated anywhere, bytes flipped anywhere (headers, tables,
    entropy-coded data), marker lengths inflated.  No over-read, no undefined shift, and the same verdict and coefficients
    as the regular library build."""
    import json

    import numpy as np

    gold = os.path.join(ROOT, "tests", "golden", "jpeg")
    names = ["c420_q90_dri4_95x51", "c444_q90_48x40", "gray_q60_dri3_40x24", "c422_q90_dri1_50x20", "c420_q92_opt_120x90", "c440_q90_patched_24x64", "c420_q90_3x2"]
    rng = np.random.Generator(np.random.PCG64(99))
    files = []
    for name in names:
        src = open(os.path.join(gold, name + ".jpg"), "rb").read()
        files.append(src)
        for cut in sorted(set([2, 3, 4, 5, 21, 160, 170, 200, len(src) // 2, len(src) - 3, len(src) - 1])):
            files.append(src[:cut])
        for _ in range(60):
            b = bytearray(src)
            for _ in range(int(rng.integers(1, 5))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            files.append(bytes(b))
        for _ in range(10):                                  # a marker segment that claims to be longer than the file
            b = bytearray(src)
            at = b.find(b"\xff\xc4")
            b[at + 2], b[at + 3] = int(rng.integers(0, 256)), int(rng.integers(0, 256))
            files.append(bytes(b))
    out = drive(["jpeg %s" % hx(f) for f in files])
    buf = np.zeros(3 * 4200000, dtype=np.int16)          # the driver refuses frames above 4 M pixels
    info = (C.c_int * 12)()
    accepted = 0
    for f, line in zip(files, out):
        rci, rc0, rc1, sum0, sum1 = [int(v) for v in line.split()]
        w, h, c = C.c_int(), C.c_int(), C.c_int()
        assert rci == imp.lib.impgpu_jpeg_info(f, len(f), w, h, c)
        if rci == 0 and w.value * h.value <= 4000000:
            assert rc0 == imp.lib.impgpu_jpeg_coefficients(f, len(f), 0, buf.ctypes.data, buf.size, info)
            assert rc1 == imp.lib.impgpu_jpeg_coefficients(f, len(f), 1, buf.ctypes.data, buf.size, info)
            assert (rc0 == 0) == (rc1 == 0)
            if rc0 == 0:
                assert sum0 == sum1
                accepted += 1
    assert accepted > 100


def test_png_front_under_sanitizers():
    """imp_png.cpp on damaged files: the chunk walk, the CRC checks, zlib's inflate into an exactly sized buffer, the filter-byte
    check -- truncated anywhere, bytes flipped anywhere (with the chunk CRCs repaired, so that the damage reaches the walk and
    the inflate instead of stopping at the first CRC), chunk lengths inflated.  No over-read of the file, no byte past the
    scanlines, and the same verdict and bytes as the regular library build."""
    import struct
    import zlib

    import numpy as np

    gold = os.path.join(ROOT, "tests", "golden", "png")
    names = ["f_rgb_130x70", "f_rgba_33x140", "f_gray_67x45", "f_rgb_3idat_50x40", "f_rgb_trns_gama_20x20", "pil_rgb_200x120", "f_rgb_1x1", "n_16bit", "n_palette", "d_short_stream", "d_filter7"]
    rng = np.random.Generator(np.random.PCG64(77))

    def repair(b):
        """recompute every chunk's CRC so that a flipped payload byte is seen by the code behind the CRC check"""
        b = bytearray(b)
        at = 8
        while at + 12 <= len(b):
            n = struct.unpack(">I", b[at:at + 4])[0]
            if at + 12 + n > len(b):
                break
            b[at + 8 + n:at + 12 + n] = struct.pack(">I", zlib.crc32(bytes(b[at + 4:at + 8 + n])) & 0xffffffff)
            at += 12 + n
        return bytes(b)

    files = []
    for name in names:
        src = open(os.path.join(gold, name + ".png"), "rb").read()
        files.append(src)
        for cut in sorted(set([0, 7, 8, 20, 32, 33, 40, 45, len(src) // 2, len(src) - 13, len(src) - 12, len(src) - 1])):
            files.append(src[:max(0, cut)])
        for k in range(50):
            b = bytearray(src)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(8, len(b)))] = int(rng.integers(0, 256))
            files.append(repair(b) if k % 2 else bytes(b))
        for _ in range(8):                                     # a chunk that claims to be longer (or shorter) than it is
            b = bytearray(src)
            at = b.find(b"IDAT") - 4
            b[at:at + 4] = struct.pack(">I", int(rng.integers(0, 1 << 31)) if rng.integers(0, 2) else int(rng.integers(0, 64)))
            files.append(bytes(b))
        for _ in range(6):                                     # IHDR rewritten (valid CRC): other sizes, depths, colour types over the same stream
            b = bytearray(src)
            b[16:29] = struct.pack(">IIBBBBB", int(rng.integers(0, 5000)), int(rng.integers(0, 300)), int(rng.choice([1, 8, 16])), int(rng.choice([0, 2, 3, 4, 6])), 0, 0, int(rng.integers(0, 2)))
            files.append(repair(b))
    out = drive(["png %s" % hx(f) for f in files])
    accepted = failed = refused = 0
    for f, line in zip(files, out):
        rci, rcs, need, total = [int(v) for v in line.split()]
        w, h, c = C.c_int(), C.c_int(), C.c_int()
        assert rci == imp.lib.impgpu_png_info(f, len(f), w, h, c)
        n = C.c_size_t()
        if rci == 0 and need <= 64 << 20:
            buf = np.zeros(max(1, need), dtype=np.uint8)
            assert rcs == imp.lib.impgpu_png_scanlines(f, len(f), buf.ctypes.data, need, C.byref(n)) and n.value == need
            if rcs == 0:
                acc = 0
                for v in buf[:need].tolist():
                    acc = (acc * 31 + v) & 0xffffffffffffffff
                assert acc == total
                assert need == (w.value * c.value + 1) * h.value
                # (what follows the image's last byte is not read -- libpng's rule -- and the Adler-32 trailer is not checked: raw deflate)
                assert zlib.decompressobj(-15).decompress(_png_idat(f)[2:], need) == buf[:need].tobytes()
                accepted += 1
            else:
                failed += 1
        else:
            refused += 1
    assert accepted >= 10 and failed > 100 and refused > 50, (accepted, failed, refused)


def _png_idat(blob):
    import struct

    at, out = 8, []
    while at + 12 <= len(blob):
        n, kind = struct.unpack(">I4s", blob[at:at + 8])
        if kind == b"IDAT":
            out.append(blob[at + 8:at + 8 + n])
        at += 12 + n
    return b"".join(out)
