"""RunJob's front half (bridge.c:304-372, :413-466): product parser vs oracle, then whole request lines through
the device chain.  Query strings are the examples of the reference's docs/03 - Usage.md plus grammar edge cases."""
import numpy as np
import pytest

import oracle_lib as orc
import ngx_http_imgproc_amd as imp
from conftest import noise_image, smooth_image

DOC_URIS = [   # docs/03 - Usage.md
    ("/img/cat.jpg?crop=1,1,c,c&resize=300&filter-gotham=1", "jpg"),
    ("/a.png?crop=16,9", "png"),
    ("/a.png?crop=400px,200px,46px,0px", "png"),
    ("/a.jpg?resize=200,0", "jpg"),
    ("/a.jpg?resize=0,200,up", "jpg"),
    ("/a.jpg?filter-flip=10&filter-rotate=90", "jpg"),
    ("/a.jpg?filter-modulate=30,150,100", "jpg"),
    ("/a.jpg?filter-colorize=ff0000,0.3&filter-blur=2.5", "jpg"),
    ("/a.jpg?filter-gamma=1.6&filter-contrast=1.3", "jpg"),
    ("/a.jpg?filter-gradmap=000000,ff8800,ffffff", "jpg"),
    ("/a.jpg?filter-vignette=0.6,0.9&filter-lomo=1", "jpg"),
    ("/a.jpg?filter-rainbow=pale&filter-scanline=0.3,0.5,2,2", "jpg"),
    ("/a.jpg?format=jpg&quality=70", "jpg"),
    ("/a.jpg?format=webp", "jpg"),
    ("/a.jpg?format=json", "jpg"),
    ("/a.jpg?format=text&quality=wide", "jpg"),
    ("/a.gif?crop=1,1,c,c&filter-gotham=1&page=10", "gif"),
    ("/a.gif?resize=64", "gif"),
]
EDGE_URIS = [
    ("/a.jpg", "jpg"), ("/a.jpg?", "jpg"), ("?crop=1,1", "jpg"), ("/a.jpg?crop", "jpg"), ("/a.jpg?filter=blur", "jpg"),
    ("/a.jpg?crop=1,1&crop=2,1", "jpg"), ("/a.jpg?cropper=3,2", "jpg"), ("/a.jpg?&&resize=10&&", "jpg"),
    ("/a.jpg?resize=10?crop=1,1", "jpg"), ("/a.jpg?unknown=1&resize=10", "jpg"), ("/a.jpg?crop%3D1%2C1", "jpg"),
    ("/a%20b.jpg?resize=10%2C20", "jpg"), ("/a.jpg?filter-a=1&filter-b=2&filter-c=3&filter-d=4&filter-e=5", "jpg"),
    ("/a.jpg?filter-a=1&filter-b=2&filter-c=3&filter-d=4&filter-e=5&filter-f=6", "jpg"),
    ("/a.jpg?format=bmp", "jpg"), ("/a.jpg?format=tiff&quality=lzw", "jpg"), ("/a.jpg?format=jpeg", "jpg"),
    ("/a.jpg?format=jp2", "jpg"), ("/a.jpg?format=ico", "jpg"), ("/a.jpg?format=nonsense", "jpg"), ("/a.xyz?resize=1", "xyz"),
    ("/a.PNG?resize=1", "PNG"), ("/a.jpg?gravity=r,b&crop=1,1", "jpg"), ("/a.jpg?page=3&page=x", "jpg"),
    ("/a.gif?format=png", "gif"), ("/a.gif?format=json", "gif"), ("/a.jpg?format=gif", "jpg"), ("/a.jpg?quality=101", "jpg"),
    ("/a.png?quality=10", "png"), ("/a.png?quality=9", "png"), ("/a.jpg?format=text", "jpg"),
]


@pytest.mark.parametrize("uri,ext", DOC_URIS + EDGE_URIS)
def test_parse_matches_oracle(uri, ext):
    rc_o, want = orc.parse_request(uri, ext, 5)
    r = imp.Request(uri, ext, imp.Config(max_filters=5))
    assert r.code == rc_o, (uri, r.code, rc_o)
    if rc_o == 0:
        got = dict(crop=r.crop, gravity=r.gravity, resize=r.resize, quality=r.quality, format=r.format, page=r.page,
                   filters=r.filters, mime=r.mime, simple=r.simple, need_flatten=r.need_flatten)
        assert got == want, uri


def test_parse_details():
    r = imp.Request("/x.jpg?crop=1,1&crop=2,1&filter-blur=2&resize=10", "jpg")
    assert (r.code, r.crop, r.resize, r.filters, r.destructive, r.need_flatten, r.mime) == (0, "2,1", "10", ["blur=2"], 1, 1, -1)
    r = imp.Request("/x.png?filter-gamma=2", "png")
    assert (r.code, r.destructive, r.need_flatten, r.simple, r.mime) == (0, 0, 0, 0, -2)
    r = imp.Request("/x.gif?resize=5", "gif")
    assert (r.code, r.simple, r.mime) == (0, 1, -4)
    assert imp.Request("/x.jpg", "jpg").code == 50 and imp.Request("/x.ico?resize=1", "ico").code == 1


def test_page_defaults_follow_the_encoder():
    """bridge.c:433-435 and :448-450, stated here without the oracle: an absent page= is page 0 for every one-frame
    encoder (jpg, png, text, every FreeImage format but GIF) and stays -1 (all pages) only for GIF output and json."""
    want = {("jpg", None): 0, ("png", None): 0, ("jpg", "text"): 0, ("jpg", "json"): -1, ("gif", None): -1,
            ("gif", "png"): 0, ("jpg", "gif"): -1, ("jpg", "webp"): 0, ("jpg", "bmp"): 0, ("gif", "json"): -1}
    for (ext, fmt), page in want.items():
        r = imp.Request("/a.%s?resize=10%s" % (ext, "&format=" + fmt if fmt else ""), ext)
        assert (r.code, r.page) == (0, page), (ext, fmt)
    assert imp.Request("/a.gif?page=3", "gif").page == 3 and imp.Request("/a.jpg?page=3", "jpg").page == 3


def test_quality_ranges_of_the_basic_encoders():
    """bridge.c:475-500: jpg quality 0..100, png compression 0..9, else IMP_ERROR_INVALID_ARGS; :511-519: j2k / jp2 / webp 0..512."""
    codes = {("jpg", "0"): 0, ("jpg", "100"): 0, ("jpg", "101"): 50, ("jpg", "-1"): 50, ("png", "9"): 0, ("png", "10"): 50,
             ("jpg", "abc"): 0, ("webp", "512"): 0, ("webp", "700"): 50, ("jp2", "-3"): 50, ("bmp", "700"): 0}
    for (fmt, q), code in codes.items():
        assert imp.Request("/a.jpg?format=%s&quality=%s" % (fmt, q), "jpg").code == code, (fmt, q)


@pytest.mark.gpu
@pytest.mark.parametrize("uri,ext", [u for u in DOC_URIS if "vignette" not in u[0]])
def test_request_line_end_to_end(gpu, uri, ext):
    """Literal request -> parse -> device chain, against the oracle running the same parsed request op by op."""
    from test_gpu_chain import oracle_chain

    arr = smooth_image(300, 400, 4 if ext != "jpg" else 3)
    rc_o, q = orc.parse_request(uri, ext, 5)
    cfg = gpu.Config(allow_experiments=True)
    r = gpu.Request(uri, ext, cfg)
    assert r.code == rc_o == 0
    rc_w, step_w, want = oracle_chain(arr, crop=q["crop"], gravity=q["gravity"], resize=q["resize"], simple=q["simple"],
                                      filters=q["filters"], flatten=q["need_flatten"])
    im = gpu.Image(arr)
    rc, step = r.run(im, cfg)
    assert rc == rc_w, (uri, rc, rc_w, step)
    if rc == 0:
        assert np.array_equal(im.numpy(), want), uri
    im.release()


@pytest.mark.gpu
def test_cfg1_jpeg_request_between_host_codecs(gpu):
    """BASELINE configs[0] with real codecs either side of the path, as RunJob has them (bridge.c:383-411 decode,
    :660-700 encode; advancedio.c stays on the host): a 640x480 JPEG is decoded on the host (Pillow's libjpeg standing
    in for FreeImage's), the decoded BGR frame goes through the literal request on the device, and the result is
    encoded back to JPEG with the request's quality.  The device chain is compared bit for bit with the oracle on the
    same decoded pixels; the codec round trip only has to keep the geometry and stay close to what was encoded."""
    Image = pytest.importorskip("PIL.Image")
    import io
    from test_gpu_chain import oracle_chain

    rgb = smooth_image(480, 640, 3, seed=3)[:, :, ::-1]
    blob = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(rgb)).save(blob, format="JPEG", quality=92)
    decoded = np.asarray(Image.open(io.BytesIO(blob.getvalue())).convert("RGB"))
    bgr = np.ascontiguousarray(decoded[:, :, ::-1])                       # what cvDecodeImage / FreeImage hand RunJob
    uri, ext = "/img.jpg?crop=320px,240px,0px,0px&resize=160,0&quality=85&filter-gamma=1.3", "jpg"
    cfg = gpu.Config(allow_experiments=True)
    r = gpu.Request(uri, ext, cfg)
    rc_o, q = orc.parse_request(uri, ext, 5)
    assert r.code == rc_o == 0 and r.quality == "85" and r.need_flatten == 1
    rc_w, _, want = oracle_chain(bgr, crop=q["crop"], gravity=q["gravity"], resize=q["resize"], simple=q["simple"],
                                 filters=q["filters"], flatten=q["need_flatten"])
    im = gpu.Image(bgr)
    rc, step = r.run(im, cfg)
    assert rc == rc_w == 0, (rc, step)
    out = im.numpy()
    im.release()
    assert out.shape == (120, 160, 3) and np.array_equal(out, want)
    enc = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(out[:, :, ::-1])).save(enc, format="JPEG", quality=int(r.quality))
    back = np.asarray(Image.open(io.BytesIO(enc.getvalue())).convert("RGB"))[:, :, ::-1]
    assert back.shape == out.shape and np.abs(back.astype(int) - out).mean() < 3.0
