"""LoadGIF's compositing loop (advancedio.c:204-247): oracle known-answer cases on CPU, GPU parity through the C ABI."""
import numpy as np
import pytest

import oracle_lib as orc


def palette(seed):
    rng = np.random.default_rng(seed)
    pal = rng.integers(0, 256, (256, 4), dtype=np.uint8)
    pal[:, 3] = 0
    return pal


def page(idx_top_down, left=0, top=0, dispose=0, key=-1, pal=None, pitch=None, pad_value=0):
    """A page from a top-down index picture: FreeImage stores scanlines bottom-up with a 4-byte aligned pitch."""
    a = np.asarray(idx_top_down, dtype=np.uint8)
    h, w = a.shape
    pitch = pitch or (w + 3) & ~3
    buf = np.full((h, pitch), pad_value, np.uint8)
    buf[:, :w] = a[::-1]
    return {"indices": buf, "width": w, "left": left, "top": top, "dispose": dispose, "key": key,
            "palette": palette(1) if pal is None else pal}


def expect(pal, idx, key):
    """BGRA canvas of palette colours for an index picture; -1 = transparent black."""
    idx = np.asarray(idx)
    out = np.zeros(idx.shape + (4,), np.uint8)
    ok = idx >= 0
    out[ok, :3] = pal[idx[ok], :3]
    out[..., 3] = np.where(idx == key, 0, 255)
    return out


# ------------------------------------------------------------------ oracle known answers (no GPU)
def test_single_page_is_a_palette_lookup_with_vertical_flip():
    pal = palette(3)
    idx = np.arange(12, dtype=np.uint8).reshape(3, 4) + 7
    rc, frames = orc.gif_compose([page(idx, pal=pal)])
    assert rc == 0 and len(frames) == 1
    assert np.array_equal(frames[0], expect(pal, idx.astype(int), -1))


def test_transparent_index_gets_alpha_zero_but_keeps_its_colour():
    pal = palette(4)
    idx = np.array([[5, 9], [9, 5]], np.uint8)
    rc, frames = orc.gif_compose([page(idx, key=9, pal=pal)])
    want = expect(pal, idx.astype(int), 9)
    assert rc == 0 and np.array_equal(frames[0], want)
    assert frames[0][0, 1, 3] == 0 and np.array_equal(frames[0][0, 1, :3], pal[9, :3])


def test_offset_page_outside_pixels_are_the_key_and_the_off_by_one_column_reads_past_the_row():
    """advancedio.c:213 tests `x > left + w`: column left + w is read from the byte after the row (pitch padding
    here, value 200), and row top + h is outside because its scanline index is -1."""
    pal = palette(5)
    first = np.zeros((4, 6), np.uint8)
    small = np.array([[1, 2], [3, 4]], np.uint8)
    rc, frames = orc.gif_compose([page(first, pal=pal), page(small, left=1, top=1, key=77, pal=pal, pad_value=200)])
    assert rc == 0
    want = np.full((4, 6), 77, int)
    want[1, 1:3] = [1, 2]
    want[2, 1:3] = [3, 4]
    want[1, 3] = 200                       # x == left + w: the padding byte of that scanline
    want[2, 3] = 200
    assert np.array_equal(frames[1], expect(pal, want, 77))


def test_destructive_album_carries_the_master_canvas():
    """DISPOSAL_LEAVE (1): a transparent pixel of page k > 0 shows what the master canvas holds (advancedio.c:232-238);
    DISPOSAL_BACKGROUND (2): a transparent pixel becomes index 0 and leaves the master alone (:222-228)."""
    pal = palette(6)
    p0 = np.array([[10, 11], [12, 13]], np.uint8)
    p1 = np.array([[99, 21], [22, 99]], np.uint8)          # 99 = key
    p2 = np.array([[99, 99], [31, 99]], np.uint8)
    pages = [page(p0, dispose=1, key=99, pal=pal), page(p1, dispose=1, key=99, pal=pal), page(p2, dispose=2, key=99, pal=pal)]
    rc, frames = orc.gif_compose(pages, destructive=True)
    assert rc == 0 and len(frames) == 3
    assert np.array_equal(frames[0], expect(pal, p0.astype(int), 99))
    assert np.array_equal(frames[1], expect(pal, np.array([[10, 21], [22, 13]]), 99))
    assert np.array_equal(frames[2], expect(pal, np.array([[0, 0], [31, 0]]), 99))
    # not destructive: transparent stays transparent
    rc, plain = orc.gif_compose(pages, destructive=False)
    assert np.array_equal(plain[1], expect(pal, p1.astype(int), 99))


def test_requested_page_stops_the_walk_and_is_returned_alone():
    pal = palette(7)
    pages = [page(np.full((2, 2), i + 1, np.uint8), dispose=1, key=0, pal=pal) for i in range(4)]
    rc, frames = orc.gif_compose(pages, destructive=True, page=2)
    assert rc == 0 and len(frames) == 1
    assert np.array_equal(frames[0], expect(pal, np.full((2, 2), 3), 0))
    # advancedio.c:114-116: a page past the last one is page 0, not an error
    rc, frames = orc.gif_compose(pages, page=4)
    assert rc == 0 and len(frames) == 1 and np.array_equal(frames[0], expect(pal, np.full((2, 2), 1), 0))
    assert orc.gif_compose(pages, page=-2)[0] == 50     # undefined in the reference (Frames[-2]); INVALID_ARGS here


def test_page_request_forces_the_destructive_walk():
    """advancedio.c:111-113: `if (page != -1) { ...; isdestructive = 1; }` -- whatever the filters said.  Page 1 is fully
    transparent with DISPOSAL_LEAVE, so it shows page 0 through the master canvas only when the walk is destructive."""
    pal = palette(3)
    p0 = np.array([[5, 6], [7, 8]], np.uint8)
    p1 = np.full((2, 2), 99, np.uint8)
    pages = [page(p0, dispose=1, key=99, pal=pal), page(p1, dispose=1, key=99, pal=pal)]
    rc, alone = orc.gif_compose(pages, destructive=False, page=1)
    assert rc == 0 and np.array_equal(alone[0], expect(pal, p0.astype(int), 99))
    rc, album = orc.gif_compose(pages, destructive=False)           # page == -1 keeps the caller's flag
    assert rc == 0 and (album[1][:, :, 3] == 0).all()


# ------------------------------------------------------------------ GPU parity
def random_album(seed, n, cw, ch):
    rng = np.random.default_rng(seed)
    pages = []
    for f in range(n):
        if f == 0:
            w, h, left, top = cw, ch, 0, 0
        else:
            w, h = int(rng.integers(1, cw + 3)), int(rng.integers(1, ch + 3))
            left, top = int(rng.integers(-2, cw)), int(rng.integers(-2, ch))
        key = int(rng.choice([-1, 0, 3, 255]))
        idx = rng.integers(0, 256, (h, w), dtype=np.uint8)
        idx[rng.random((h, w)) < 0.3] = key if key >= 0 else 3
        pages.append(page(idx, left=left, top=top, dispose=int(rng.integers(0, 4)), key=key, pal=palette(seed * 100 + f),
                          pad_value=int(rng.integers(0, 256))))
    return pages


@pytest.mark.gpu
@pytest.mark.parametrize("destructive", [False, True])
@pytest.mark.parametrize("seed,n,cw,ch", [(1, 1, 5, 4), (2, 3, 17, 9), (3, 6, 64, 48), (4, 12, 200, 150), (5, 2, 1, 1)])
def test_gif_compose_matches_oracle(gpu, destructive, seed, n, cw, ch):
    pages = random_album(seed, n, cw, ch)
    rc_o, want = orc.gif_compose(pages, destructive)
    rc, got = gpu.gif_compose(pages, destructive)
    assert rc == rc_o == 0 and len(got) == len(want) == n
    for f in range(n):
        assert np.array_equal(got[f].numpy(), want[f]), f
    for p in (0, n - 1, n // 2):
        rc_o, want1 = orc.gif_compose(pages, destructive, page=p)
        rc, got1 = gpu.gif_compose(pages, destructive, page=p)
        assert rc == rc_o == 0 and len(got1) == 1
        assert np.array_equal(got1[0].numpy(), want1[0]), p


@pytest.mark.gpu
def test_gif_compose_rejects_bad_arguments(gpu):
    pages = random_album(9, 2, 8, 8)
    assert gpu.gif_compose(pages, page=-2)[0] == gpu.IMP_ERROR_INVALID_ARGS
    assert gpu.gif_compose([], page=-1)[0] == gpu.IMP_ERROR_INVALID_ARGS


@pytest.mark.gpu
def test_gif_page_past_the_end_is_page_zero_and_always_destructive(gpu):
    pages = random_album(10, 3, 12, 9)
    for destructive in (False, True):
        rc, got = gpu.gif_compose(pages, destructive, page=7)       # advancedio.c:114-116 -> page 0
        rc_o, want = orc.gif_compose(pages, True, page=0)
        assert rc == rc_o == 0 and np.array_equal(got[0].numpy(), want[0])
        rc, got = gpu.gif_compose(pages, destructive, page=2)       # advancedio.c:113 -> destructive whatever the flag
        rc_o, want = orc.gif_compose(pages, True, page=2)
        assert rc == rc_o == 0 and np.array_equal(got[0].numpy(), want[0])


@pytest.mark.gpu
def test_gif_frames_feed_the_operator_chain(gpu):
    """A composed page is an ordinary 4-channel frame: resize it like RunJob does for every album frame."""
    pages = random_album(11, 3, 96, 64)
    rc, got = gpu.gif_compose(pages, True)
    rc_o, want = orc.gif_compose(pages, True)
    assert rc == rc_o == 0
    for im, w in zip(got, want):
        assert im.cv_resize(48, 32, orc.INTER_AREA) == 0
        assert np.array_equal(im.numpy(), orc.cv_resize(w, 48, 32, orc.INTER_AREA))


@pytest.mark.gpu
@pytest.mark.parametrize("destructive", [False, True])
def test_gif_compose_album_is_one_handle_for_the_operator_segment(gpu, destructive):
    """LoadGIF -> RunJob for an animation: the composed frames as ONE album handle, then crop / resize / filter for all
    frames at once (bridge.c:577-655 loops over album.Count for each)."""
    pages = random_album(12, 9, 120, 90)
    rc_o, want = orc.gif_compose(pages, destructive)
    rc, al = gpu.gif_compose(pages, destructive, album=True)
    assert rc == rc_o == 0 and al.count == 9 and al.shape == (90, 120, 4)
    for got, w in zip(al.frames(), want):
        assert np.array_equal(got, w)
    cfg = gpu.Config()
    rc, step = gpu.run_ops(al, cfg, crop="4,3", resize="60,0", simple=1, filters=["flip=10", "gamma=0.8"], need_flatten=1)
    assert rc == 0 and al.count == 9
    for got, w in zip(al.frames(), want):
        rc, cur = orc.crop(w, "4,3")
        rc, cur = orc.resize(cur, "60,0", 2000, 2000, 1)
        rc, cur = orc.filter(cur, "flip=10", 1)
        rc, cur = orc.filter(cur, "gamma=0.8", 1)
        assert np.array_equal(got, orc.blend_with_paper(cur))
    rc, one = gpu.gif_compose(pages, destructive, page=4, album=True)
    rc_o, want1 = orc.gif_compose(pages, destructive, page=4)
    assert rc == rc_o == 0 and one.count == 1 and np.array_equal(one.numpy(), want1[0])
    al.release(); one.release(); cfg.release()
