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
