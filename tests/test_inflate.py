"""csrc/imp_inflate.cpp (the PNG front's host inflate) against zlib, under AddressSanitizer / UBSan: tests/c/inflate_test.cpp
-- 1500 generated streams of every block type and content class, each decoded at the exact size, at fewer and at more bytes than
it holds, truncated at random places and with random bits flipped; zlib's verdict on the same bytes is the reference."""
import os
import subprocess

from conftest import ROOT


def test_inflate_exact_matches_zlib_under_sanitizers():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c"), os.path.join(ROOT, "tests", "c", "_build", "inflate_test_asan")])
    p = subprocess.run([os.path.join(ROOT, "tests", "c", "_build", "inflate_test_asan"), "1500"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = p.stdout.strip().splitlines()
    summary = [l for l in lines if l.endswith(" bad")][0]
    cases, bad = int(summary.split()[0]), int(summary.split()[2])
    assert cases > 15000 and bad == 0, summary


def test_png_front_with_the_zlib_switch_gives_the_same_scanlines(monkeypatch):
    """IMPGPU_PNG_INFLATE=zlib (round 5): an operator may put the audited library under the PNG front instead of
    csrc/imp_inflate.cpp.  Same filtered scanlines, same verdicts -- on the committed files, on truncated ones and on files
    with flipped bytes (a flipped byte inside the zlib header is the one place the two may differ: zlib checks its header's
    check bits, the one-shot inflate reads only the method; such a file is counted, not compared)."""
    import ctypes as C
    import glob
    import os
    import numpy as np
    from conftest import ROOT
    import ngx_http_imgproc_amd as imp

    def scan(blob):
        w, h, c = C.c_int(), C.c_int(), C.c_int()
        if imp.lib.impgpu_png_info(blob, len(blob), w, h, c) != 0:
            return None
        need = (w.value * c.value + 1) * h.value
        buf = np.zeros(max(need, 1), np.uint8)
        n = C.c_size_t()
        rc = imp.lib.impgpu_png_scanlines(blob, len(blob), buf.ctypes.data, need, C.byref(n))
        return rc, (buf[:need].tobytes() if rc == 0 else b"")

    rng = np.random.default_rng(5)
    files = []
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "png", "*.png"))):
        src = open(path, "rb").read()
        files.append(src)
        for cut in (len(src) // 2, len(src) - 20, len(src) - 13):
            files.append(src[:max(cut, 0)])
        for _ in range(4):
            b = bytearray(src)
            b[int(rng.integers(33, len(b)))] ^= 1 << int(rng.integers(0, 8))
            files.append(bytes(b))
    same = differ = 0
    for f in files:
        monkeypatch.delenv("IMPGPU_PNG_INFLATE", raising=False)
        own = scan(f)
        monkeypatch.setenv("IMPGPU_PNG_INFLATE", "zlib")
        lib = scan(f)
        if own is None:
            assert lib is None
            continue
        if own == lib:
            same += 1
        else:
            differ += 1
            assert own[0] == 0 or lib[0] == 0            # never two different sets of bytes: one of them refused
            assert not (own[0] == 0 and lib[0] == 0)
    assert same > 50 and differ <= 3, (same, differ)
