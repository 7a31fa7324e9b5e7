"""Third-party pins for the oracle: Pillow (its own C code, written by nobody here) run on the same frames.

The reference has no fixtures and cannot be built (parity unpinned), so wherever an operator of the
path has a textbook definition that an independent library implements, the oracle is checked against that library:
quarter turns and mirrors (filters.c:72-133 through cvTranspose / cvFlip) and integer-factor INTER_AREA (resizeAreaFast_:
the exact 2x2 mean with halves rounded up; other factors within one grey level -- Pillow rounds the exact quotient,
OpenCV the float product sum * (1.f / area)).  CPU only; skipped where Pillow is not installed.
"""
import numpy as np
import pytest

import oracle_lib as orc
from conftest import noise_image, smooth_image

Image = pytest.importorskip("PIL.Image")


def planes(arr, fn):
    """Apply a Pillow operation to every channel as its own 8-bit plane (Pillow's RGBA paths premultiply alpha)."""
    return np.stack([np.asarray(fn(Image.fromarray(np.ascontiguousarray(arr[:, :, k]), "L"))) for k in range(arr.shape[2])], axis=2)


@pytest.mark.parametrize("c", [3, 4])
@pytest.mark.parametrize("req,method", [("rotate=90", "ROTATE_270"), ("rotate=180", "ROTATE_180"), ("rotate=270", "ROTATE_90"),
                                        ("flip=10", "FLIP_LEFT_RIGHT"), ("flip=01", "FLIP_TOP_BOTTOM"), ("flip=11", "ROTATE_180")])
def test_turns_and_mirrors_equal_pillow_transpose(c, req, method):
    """filter-rotate=90 is a clockwise quarter turn (Pillow's ROTATE_270 counts counter-clockwise); flip=XY mirrors x
    when X is set and y when Y is set."""
    arr = noise_image(37, 52, c, 90)
    rc, got = orc.filter(arr, req)
    assert rc == 0
    want = planes(arr, lambda im: im.transpose(getattr(Image, method)))
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("c", [3, 4])
def test_area_2x2_equals_pillow_reduce(c):
    arr = noise_image(120, 200, c, 91)
    want = planes(arr, lambda im: im.reduce(2))
    assert np.array_equal(orc.cv_resize(arr, 100, 60, orc.INTER_AREA), want)
    arr = smooth_image(90, 130, c, 3)
    assert np.array_equal(orc.cv_resize(arr, 65, 45, orc.INTER_AREA), planes(arr, lambda im: im.reduce(2)))


@pytest.mark.parametrize("factor", [(3, 3), (4, 4), (2, 3), (8, 8), (5, 2), (7, 6)])
def test_area_integer_factors_within_one_of_pillow_reduce(factor):
    fx, fy = factor
    arr = noise_image(24 * fy, 31 * fx, 3, 92)
    want = planes(arr, lambda im: im.reduce((fx, fy))).astype(int)
    got = orc.cv_resize(arr, 31, 24, orc.INTER_AREA).astype(int)
    assert np.abs(got - want).max() <= 1
    exact = arr.reshape(24, fy, 31, fx, 3).astype(np.float64).mean(axis=(1, 3))
    assert np.abs(got - exact).max() <= 0.5 + 1e-3          # and it is the nearest grey level to the true mean
