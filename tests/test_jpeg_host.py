"""Host side of the JPEG front (csrc/imp_jpeg.cpp), no GPU: the marker parser, the scan preparation (FF00 unstuffing,
restart intervals cut onto chunk boundaries), the Huffman table builder, the sequential host decoder (A/B path) and --
most important -- the device's chunk-parallel entropy scheme executed lane by lane on the host with the very code the
kernel's lanes run (imp_jpeg_core.h).  All compared with the Pillow-pinned oracle's coefficients.
"""
import ctypes as C
import io
import json
import os

import numpy as np
import pytest

import oracle_lib as orc
from conftest import noise_image, smooth_image

import ngx_http_imgproc_amd as imp
from ngx_http_imgproc_amd._lib import lib

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))


def golden_blob(name):
    with open(os.path.join(GOLD, name + ".jpg"), "rb") as f:
        return f.read()


def product_coefficients(blob, how):
    out = np.zeros(4_000_000, dtype=np.int16)
    info = (C.c_int * 12)()
    rc = lib.impgpu_jpeg_coefficients(blob, len(blob), how, out.ctypes.data, out.size, info)
    return rc, out[: info[0]].copy(), list(info)


def oracle_coefficients(blob):
    rc, info = orc.jpeg_info(blob)
    assert rc == 0
    return np.concatenate([orc.jpeg_coefficients(blob, ci)[1].reshape(-1) for ci in range(info["components"])])


def encode(arr, **kw):
    Image = pytest.importorskip("PIL.Image")
    b = io.BytesIO()
    Image.fromarray(arr).save(b, "JPEG", **kw)
    return b.getvalue()


@pytest.mark.parametrize("case", MANIFEST["cases"], ids=[c["name"] for c in MANIFEST["cases"]])
@pytest.mark.parametrize("how", [0, 1], ids=["sequential", "chunk-parallel"])
def test_golden_files_decode_to_the_oracles_coefficients(case, how):
    blob = golden_blob(case["name"])
    rc, got, info = product_coefficients(blob, how)
    assert rc == 0 and info[1] == 0
    assert np.array_equal(got, oracle_coefficients(blob))
    w, h, c = case["shape"][1], case["shape"][0], case["shape"][2]
    assert imp.jpeg_info(blob) == (0, (w, h, c))


@pytest.mark.parametrize("sub", ["4:4:4", "4:2:2", "4:2:0"])
def test_chunk_parallel_scheme_on_larger_frames(sub):
    """Several workgroups' worth of chunks, with and without restart intervals.  On picture-like content nearly every chunk's
    true entry state is one of the states its walks arrive in (a miss costs a repair walk); white noise at quality 100 --
    blocks of a thousand bits that end without an end-of-block symbol -- is the scheme's worst case and only has to be right
    (the product keeps such files' Huffman stage on the calling thread: jpeg_entropy_on_device)."""
    for kind, q, rst in (("smooth", 90, None), ("noise", 75, None), ("smooth", 95, dict(restart_marker_rows=1)),
                         ("noise", 90, dict(restart_marker_blocks=3)), ("noise", 100, None)):
        arr = smooth_image(360, 500, 3) if kind == "smooth" else noise_image(360, 500, 3, 2)
        blob = encode(arr, quality=q, subsampling=sub, **(rst or {}))
        want = oracle_coefficients(blob)
        rc, got, info = product_coefficients(blob, 1)
        assert rc == 0 and info[1] == 0, (kind, q, rst, info)
        assert np.array_equal(got, want), (kind, q, rst)
        nchunks = len(blob) // 128
        st = (C.c_int * 8)()
        lib.impgpu_jpeg_sync_stats(st)
        assert st[1] == info[2] and st[0] >= nchunks // 2          # (chunks of up to 256 bytes)
        if kind == "smooth":
            assert st[1] <= max(4, st[0] // 10), "no self-synchronisation: %d misses for %d chunks" % (st[1], st[0])
        rc, got, _ = product_coefficients(blob, 0)
        assert rc == 0 and np.array_equal(got, want)


def test_gray_and_optimized_tables():
    g = smooth_image(150, 211, 3)[:, :, 0]
    for kw in (dict(quality=85), dict(quality=85, optimize=True), dict(quality=40, restart_marker_blocks=2)):
        blob = encode(g, **kw)
        for how in (0, 1):
            rc, got, info = product_coefficients(blob, how)
            assert rc == 0 and np.array_equal(got, oracle_coefficients(blob))


def test_refusals_match_the_oracle():
    arr = smooth_image(40, 40, 3)
    assert imp.jpeg_info(encode(arr, quality=90, progressive=True))[0] == imp.IMP_ERROR_UNSUPPORTED
    assert imp.jpeg_info(b"\x89PNG\r\n\x1a\n" + b"\0" * 64)[0] == imp.IMP_ERROR_UNSUPPORTED
    blob = golden_blob("c420_q90_dri4_95x51")
    for cut in (3, 20, 200, len(blob) // 2, len(blob) - 40):
        for how in (0, 1):
            rc, _, _ = product_coefficients(blob[:cut], how)
            assert rc in (imp.IMP_ERROR_UNSUPPORTED, imp.IMP_ERROR_DECODE_FAILED), (cut, how, rc)
        assert orc.jpeg_decode(blob[:cut])[0] in (orc.UNSUPPORTED, orc.DECODE_FAILED)


def test_damaged_files_get_the_same_verdict_from_every_decoder():
    """Bytes flipped anywhere in the file: the sequential decoder, the chunk-parallel scheme and the oracle agree on
    accept / refuse, and on every coefficient when they accept."""
    rng = np.random.Generator(np.random.PCG64(11))
    agree = 0
    for name in ("c420_q90_dri4_95x51", "c444_q90_48x40", "gray_q90_57x43", "c420_q30_noise_64x64", "c422_q85_49x37"):
        src = golden_blob(name)
        for _ in range(150):
            b = bytearray(src)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(2, len(b)))] = int(rng.integers(0, 256))
            b = bytes(b)
            rc_o, _ = orc.jpeg_decode(b)
            rc0, got0, _ = product_coefficients(b, 0)
            rc1, got1, _ = product_coefficients(b, 1)
            assert (rc0 == 0) == (rc1 == 0), (name, rc0, rc1)
            if rc_o in (0, orc.DECODE_FAILED) and rc0 in (0, imp.IMP_ERROR_DECODE_FAILED):
                assert (rc_o == 0) == (rc0 == 0), (name, rc_o, rc0)
            if rc0 == 0:
                assert np.array_equal(got0, got1)
                if rc_o == 0:
                    assert np.array_equal(got0, oracle_coefficients(b))
                    agree += 1
    assert agree > 50       # plenty of damaged-but-decodable files among them (a flipped bit inside a coefficient's value)


def test_classify_names_the_reason_of_a_refusal():
    """impgpu_jpeg_classify (round 5): 0 = the device takes the file, else why the cvDecodeImage fallback gets it."""
    Image = pytest.importorskip("PIL.Image")
    import io
    arr = smooth_image(40, 40, 3)
    cls = lambda b: imp.lib.impgpu_jpeg_classify(b, len(b))
    assert cls(encode(arr, quality=90)) == 0
    assert cls(encode(arr[:, :, 0], quality=90)) == 0                          # gray
    assert cls(encode(arr, quality=90, progressive=True)) == 1
    b = io.BytesIO()
    Image.fromarray(np.dstack([arr, arr[:, :, :1]]), "CMYK").save(b, "JPEG", quality=90)
    assert cls(b.getvalue()) == 4
    assert cls(b"\x89PNG\r\n\x1a\n" + b"\0" * 64) == 7                        # not a JPEG
    blob = encode(arr, quality=90)
    sof = blob.index(b"\xff\xc0")
    assert cls(blob[:sof + 4] + b"\x0c" + blob[sof + 5:]) == 3                 # 12-bit samples
    assert cls(blob[:sof + 1] + b"\xc9" + blob[sof + 2:]) == 2                 # arithmetic coding
    assert cls(blob[:sof + 7]) == 8                                            # cut inside the frame header
    sos = blob.index(b"\xff\xda")
    assert cls(blob[:sos + 4] + b"\x01" + blob[sos + 5:]) in (5, 8)            # a scan of one component of three
