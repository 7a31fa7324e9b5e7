#!/usr/bin/env python3
"""Pointwise filters and byte movers on 3-channel (JPEG) vs 4-channel resident 1080p frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngx_http_imgproc_amd as imp

torch.cuda.set_device(0)
imp.env_start(0)
n = int(os.environ.get("PROBE_BATCH", "256"))
stream = torch.cuda.Stream()
for c in (3, 4):
    src = torch.randint(0, 256, (n, 1080, 1920, c), dtype=torch.uint8, device="cuda")
    for name, filt in (("gamma", ["gamma=2.2"]), ("gotham", ["gotham=1"]), ("modulate", ["modulate=50,120,90"])):
        def step():
            rc = imp.batch_filters(src.data_ptr(), 1080 * 1920 * c, 1920, 1080, c, 1920 * c, n, filt, 1, stream=stream.cuda_stream)
            assert rc == 0, rc
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10):
            step()
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("c=%d %-9s %8.3f ms / %d frames  %9.0f img/s  %6.0f GB/s r+w" % (c, name, ms, n, n / ms * 1e3, 2 * n * 1080 * 1920 * c / ms / 1e6), flush=True)
    del src
imp.env_destroy()
