"""Synthetic request workloads shared by bench.py and the tests.

`mixed_sizes` is BASELINE configs[4]: a stream of independent requests whose source frames have a long side
log-uniform in [256, 3840], an aspect out of {1:1, 4:3, 3:2, 16:9} and a coin-flip orientation (seed 0x1A4D0005).
Every request is `resize=224,0` -- keep aspect, INTER_AREA, which is what the reference's Resize() dispatches for a
shrink (bridge.c:190) and what RunJob does frame by frame on whatever sizes arrive (bridge.c:588-604).
"""
import numpy as np

MIXED_SEED = 0x1A4D0005
MIXED_RESIZE = b"224,0"
ASPECTS = ((1, 1), (4, 3), (3, 2), (16, 9))


def mixed_sizes(n_requests, seed=MIXED_SEED):
    """[(width, height)] of the n source frames of the request stream."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sizes = []
    for _ in range(n_requests):
        long_side = int(round(np.exp(rng.uniform(np.log(256), np.log(3840)))))
        a, b = ASPECTS[rng.integers(0, 4)]
        short = max(1, int(round(long_side * b / a)))
        sizes.append((long_side, short) if rng.integers(0, 2) else (short, long_side))
    return sizes


def photo_like(h, w, seed=0):
    """H x W x 3 uint8 (R,G,B) with the statistics of a photograph rather than of a test card or of white noise: smooth
    shading, texture at an eighth and a half of the resolution, a little sensor noise -- about 2 bits per pixel as a
    quality-90 4:2:0 JPEG, what a camera file of that size weighs.  The content a JPEG request stream is measured on
    (bench.py --stream --jpeg); pixels alone never were content-dependent, entropy decoding is."""
    rng = np.random.Generator(np.random.PCG64(0x1A4D0100 + seed))
    y = np.arange(h, dtype=np.float32)[:, None]
    x = np.arange(w, dtype=np.float32)[None, :]

    def field(step, amp):
        gh, gw = h // step + 2, w // step + 2
        g = rng.standard_normal((gh, gw)).astype(np.float32)
        g = np.kron(g, np.ones((step, step), dtype=np.float32))[:h + step, :w + step]
        # two box passes turn the blocks into something smooth
        for _ in range(2):
            g = (g[:-1, :] + g[1:, :]) * 0.5
            g = (g[:, :-1] + g[:, 1:]) * 0.5
        return amp * g[:h, :w]

    luma = 120.0 + 50.0 * np.sin(x / 131.0 + seed) * np.cos(y / 97.0) + 25.0 * (x / max(w - 1, 1) - 0.5) + field(8, 28.0) + field(2, 9.0)
    luma = luma + rng.standard_normal((h, w)).astype(np.float32) * 2.5
    out = np.empty((h, w, 3), dtype=np.uint8)
    for k, (gain, shift) in enumerate(((1.0, 8.0), (0.95, 0.0), (0.85, -10.0))):
        tint = 18.0 * np.sin(x / 211.0 + 1.3 * k) * np.sin(y / 173.0 + 0.7 * k)
        out[:, :, k] = np.clip(np.rint(luma * gain + shift + tint), 0, 255).astype(np.uint8)
    return out
