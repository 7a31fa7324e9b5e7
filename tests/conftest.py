import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- FIRST: torch bundles its own HIP/HSA runtime; if libimpgpu.so's /opt/rocm copy is mapped before
#                              it, torch.cuda finds no device in this process (bench.py imports in the same order)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the library and the oracle are normally built by __graft_entry__.build(); build them here if the .so files
    # did not travel (hipcc cross-compiles without a GPU, gcc builds the oracle)
    import subprocess

    if not os.path.exists(os.path.join(ROOT, "ngx_http_imgproc_amd", "libimpgpu.so")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "ngx_http_imgproc_amd", "build.py")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])


@pytest.fixture(scope="session")
def gpu():
    """OnEnvStart once per test session; the library has no CPU fallback, so this fails loudly without a GPU."""
    import ngx_http_imgproc_amd as imp

    imp.env_start(0)
    yield imp
    imp.env_destroy()


def noise_image(h, w, c, seed):
    """i.i.d. uniform bytes (SURVEY 8d 'noise')."""
    rng = np.random.Generator(np.random.PCG64(0x1A4D0001 + seed))
    return rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)


def smooth_image(h, w, c, seed=0):
    """separable sinusoid + ramp, per-channel phase (SURVEY 8d 'smooth'); alpha uniform random."""
    y = np.arange(h, dtype=np.float64)[:, None]
    x = np.arange(w, dtype=np.float64)[None, :]
    chans = []
    for k in range(min(c, 3)):
        v = 127.5 + 80.0 * np.sin(2 * np.pi * x / 97.0 + 0.9 * k + seed) * np.cos(2 * np.pi * y / 61.0 + 0.4 * k) \
            + 40.0 * (x / max(w - 1, 1) - 0.5) + 20.0 * (y / max(h - 1, 1) - 0.5)
        chans.append(np.clip(np.rint(v), 0, 255).astype(np.uint8))
    if c == 4:
        rng = np.random.Generator(np.random.PCG64(0x1A4D0A00 + seed))
        chans.append(rng.integers(0, 256, size=(h, w), dtype=np.uint8))
    return np.stack(chans, axis=2)
