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
